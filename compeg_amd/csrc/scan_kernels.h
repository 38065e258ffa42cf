// Device-side scan preprocessing: launchers and the per-image descriptor.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace compeg {

// All pointers are device pointers.  `raw` must be readable 16 bytes before
// and after the segment (the input arenas are padded accordingly).
struct ScanDesc {
    const uint8_t *raw;      // entropy-coded segment incl. RSTn markers and FF 00 stuffing
    uint32_t len;
    uint32_t ntiles;         // ceil(len / 4096)
    uint32_t slots;          // next_power_of_two(expected intervals), >= 1
    uint32_t *tile_state;    // [ntiles] x 20 bytes of scratch (kScanTileStateBytes each)
    uint32_t *starts_out;    // [slots] the reference's start_positions
    uint8_t *words_out;      // preprocessed scan, (len + len/3 + 4) bytes
    uint32_t expected = 0;   // restart intervals the frame header announces (span_kernel only)
    uint32_t *result;        // [8]: intervals counted, kept bytes, output words, flags (bit 0: FF run too long; bit 1:
                             // a marker other than RSTn inside the segment -- it ends earlier than the caller said),
                             // widest 64-interval span in words (span_kernel), 3 unused
    // optional: where to drop the output word count and the number of start positions kept
    // (the nwords / nstarts fields of the image descriptor a following decode kernel reads)
    uint32_t *patch_nwords = nullptr;
    uint32_t *patch_nstarts = nullptr;
};

constexpr size_t kScanTileStateBytes = 20;
uint32_t scan_tiles(uint32_t len);
// dst (device) <- pinned_src (pinned host memory), both 16-byte aligned and readable / writable up to the
// next multiple of 16 bytes; the copy is a kernel on `stream`
hipError_t launch_pull(void *dst, const void *pinned_src, size_t bytes, hipStream_t stream);
// three such transfers in one launch (a transfer of 0 bytes is skipped)
hipError_t launch_pull3(void *const dst[3], const void *const pinned_src[3], const size_t bytes[3], hipStream_t stream);
// with_span: one more small kernel leaves in result[4] what max_wave_span() computes on the host (the decode
// kernels' LDS window is sized from it)
hipError_t launch_scan(const ScanDesc *descs, uint32_t images, uint32_t max_tiles, hipStream_t stream,
                       bool with_span = false);
constexpr size_t kScanResultBytes = 32;

} // namespace compeg
