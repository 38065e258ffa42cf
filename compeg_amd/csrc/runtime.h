// Device-side resource management and decode orchestration of libcompeg_hip:
// the counterpart of the reference's Gpu / Decoder / DecodeOp
// (src/lib.rs:64-574) and of its grow-only DynamicBuffer / DynamicTexture
// (src/dynamic.rs:11-79,166-257).  Bind groups have no equivalent: kernel
// arguments are plain device pointers.
#pragma once

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "device_types.h"
#include "front.h"
#include "kernels.h"
#include "scan.h"
#include "scan_kernels.h"

namespace compeg {

Status hip_status(hipError_t e, const char *what);
bool use_fused_pipeline();
bool use_pair_kernel(uint32_t max_intervals, uint32_t images);
bool use_coop_kernel(uint32_t max_intervals, uint32_t images, uint32_t restart_interval);
bool coop_preferred(const CoopPlan &cp, uint32_t max_intervals, uint32_t images, uint32_t restart_interval);
CoopSpans coop_spans_exact(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals, uint32_t restart_interval);
CoopSpans coop_spans_estimate(uint32_t span_of_64, uint32_t restart_interval, bool generous);
void coop_spans_max(CoopSpans &into, const CoopSpans &other);

// Grow-only device allocation; contents are not preserved across growth
// (every user rewrites the buffer in full before reading it).
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t capacity = 0;
    ~DeviceBuffer();
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    Status reserve(size_t bytes, bool *reallocated = nullptr);
    void *release();
};

struct PinnedBuffer {
    void *ptr = nullptr;
    size_t capacity = 0;
    ~PinnedBuffer();
    Status reserve(size_t bytes);
};

// Fills the kernel-facing descriptor from the reference-format metadata.
// Pointers are left for the caller to set.
void fill_desc(const ImageData &img, ImageDesc &d);

} // namespace compeg

struct compeg_gpu {
    std::atomic<int> refs{1};
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string name;
};

struct compeg_op {
    hipEvent_t done = nullptr;
    int device = 0;
    bool texture_changed = false;
};

struct compeg_decoder {
    compeg_gpu *gpu = nullptr;
    compeg::ScanBuffer scan;
    compeg::PinnedBuffer host_blob;   // ImageDesc + L1 + L2 for the next upload
    compeg::DeviceBuffer dev_blob, words, starts, ac, dc, out;
    // the cooperative kernel's walk tables (kernels.h) and the Huffman tables they were made from: frames of one
    // stream carry the same tables, the walk tables are made again only when those change
    compeg::DeviceBuffer walk_tables;
    std::vector<uint8_t> walk_key;
    // the walk + lane-per-MCU route's records (kernels_body.h): a stream word index and a state per MCU
    compeg::DeviceBuffer mcu_words, mcu_states;
    uint32_t out_w = 0, out_h = 0;
    size_t out_pitch = 0;
    uint32_t out_alloc_h = 0; // rows behind `out` (the extent rounded up to whole MCUs)
    hipEvent_t upload_done = nullptr; // host staging may be rewritten after this
    bool upload_pending = false;
    hipEvent_t decode_done = nullptr; // the device buffers may be rewritten after this (enqueue on another stream waits for it)
    bool decode_pending = false;
    hipStream_t last_stream = nullptr;
    std::string warning;
    int last_kernel = 0; // COMPEG_KERNEL_*: where the last enqueue went
    compeg_stage_times stage_times{0.0, 0.0, 0.0}; // host time of the last decode's stages (lib.rs:391-396,452-475,516-522)
    // what read_coefficients needs to rebuild the reference's buffer
    compeg::Metadata last_md{};
    bool have_last = false;
    compeg::HuffLdsPlan last_plan{};
    uint32_t last_span = 0;
    bool coefficients_valid = false; // ac/dc hold the last image's coefficients

    // false (default): the scan is preprocessed on the host like the reference;
    // true: the raw entropy-coded segment is uploaded and the scan kernels do it.
    bool device_preprocess = false;
    compeg::DeviceBuffer scan_arena;
    compeg::PinnedBuffer raw_stage;
    const void *dev_words = nullptr, *dev_starts = nullptr;

    compeg_decoder();
    ~compeg_decoder();
    // device preprocessing without a read-back in the middle (decode_blocking): the scan kernels'
    // result words land here, to be looked at by finish_deferred once the stream is done
    compeg::PinnedBuffer scan_result;
    const void *scan_result_dev = nullptr;
    const void *last_desc_dev = nullptr; // the last decode's image descriptor (dev_blob, or inside scan_arena)
    bool deferred_check = false;
    // writes the image's blob (descriptor + LUTs) at host_at as it will sit at dev_at, and says where
    // the scan kernels should drop nwords / nstarts
    using BlobWriter = std::function<compeg::Status(uint8_t *host_at, uint8_t *dev_at, uint32_t **patch_nwords,
                                                    uint32_t **patch_nstarts)>;
    uint32_t deferred_expected = 0;

    // may_defer: the caller will wait for the stream and then call finish_deferred(img)
    compeg::Status enqueue(const compeg::ImageData &img, hipStream_t stream, bool *changed, bool may_defer = false);
    compeg::Status finish_deferred(const compeg::ImageData &img, hipStream_t stream);
    compeg::Status check_scan_result(bool &fell_back);
    compeg::Status preprocess_on_device(const compeg::ImageData &img, hipStream_t stream, uint32_t &nwords,
                                        uint32_t &nstarts, uint32_t &span, bool &fell_back, size_t blob_bytes,
                                        const BlobWriter &before_submit);
};

struct compeg_batch {
    compeg_gpu *gpu = nullptr;
    size_t count = 0;
    std::vector<compeg::ImageDesc> descs; // host copy (device pointers inside)
    compeg::DeviceBuffer dev_descs, inputs, ac, dc, out;
    compeg::DeviceBuffer walk_tables; // the cooperative kernel's (kernels.h), made at upload when it may run
    compeg::DeviceBuffer unit_queue;  // the counter a uniform launch's resident waves draw their units from (kernels.h)
    hipEvent_t decode_done = nullptr; // behind the last decode (one on another stream waits for it)
    bool decode_recorded = false;
    compeg::Status make_walk_tables(hipStream_t stream, size_t n); // (behind the descriptors' upload; n images)
    compeg::PinnedBuffer stage; // host copy of the input arena (kept between uploads: pinning is slow)
    std::vector<size_t> out_offset;
    uint32_t max_intervals = 0, max_dus = 0, max_l2 = 0, max_span = 0;
    // some image is not 4:2:2 (extension): the whole batch takes the three-kernel pipeline
    bool generic_layout = false;
    uint32_t layout_h = 0, layout_v = 0; // luma sampling all images share (0: they differ)
    bool layout_pairs = false;           // ... and every restart interval holds two MCUs or more (8-pixel MCUs in pairs)
    bool one_mcu_intervals = false;      // every restart interval is one MCU
    uint32_t stream_mcu_words = 0;       // the batch's average MCU in stream words, rounded up (plan_stream)
    uint32_t min_restart_interval = 0;   // the smallest restart interval of the batch's images
    uint32_t max_restart_interval = 0;   // ... and the largest
    uint64_t total_waves = 0;            // units of 64 intervals over all images
    bool uniform = false; // same interval count and LUT bytes in every image (set by upload)
    // cooperative kernel: the restart interval all images share if every one of them qualifies (else 0), and the
    // largest word span of a wave's group of intervals
    uint32_t coop_r = 0;
    compeg::CoopSpans coop_spans{};
    // The walk + lane-per-MCU route (kernels_body.h): chosen at upload (make_walk_tables) for launches whose restart
    // intervals are too few, or too long, to fill the chip with a lane each; the walk's records (a stream word index and
    // a state per MCU) and the images' descriptors of MCUs
    bool mcu_route = false;
    compeg::DeviceBuffer mcu_words, mcu_states, mcu_descs;
    std::vector<compeg::ImageDesc> mcu_views; // host copy of mcu_descs
    uint32_t max_mcus = 0, mcu_span = 0;
    bool mcu_uniform = false;
    uint32_t max_out_w = 0, max_out_h = 0;
    uint64_t algorithmic_bytes = 0, pixels = 0;
    uint32_t chunk = 0; // images per launch pair, 0 = all
    // one event triple per decode since the last upload / timing reset:
    // [start, after huffman, end], recorded on the decode's own stream
    std::vector<hipEvent_t> events;
    std::vector<bool> has_stage_event; // per timed decode: the middle event was recorded (two-kernel routes)
    bool timing_on = true;             // compeg_batch_set_timing: off = a decode records no events at all
    size_t decodes_timed = 0;
    hipStream_t last_stream = nullptr;
    int last_kernel = 0; // COMPEG_KERNEL_*: where the first launch of the last decode went

    // 0: scans are preprocessed on the host at upload (the reference's data flow);
    // 1: raw scans are uploaded and preprocessed once by the scan kernels;
    // 2: like 1, and every decode() re-runs the scan kernels first, so that a
    //    timed step covers the whole path from raw entropy-coded bytes to RGBA.
    int preprocess_mode = 0;
    compeg::DeviceBuffer scan_descs, scan_arena;
    uint32_t max_tiles = 0;
    size_t host_fallbacks = 0; // images the scan kernels handed back to the host

    // host-fed use (upload_jpegs): the images parsed from the caller's bytes, kept until the next upload
    std::vector<std::unique_ptr<compeg::ImageData>> parsed;
    // the device's shared transfer streams (runtime.cpp: shared_copy_streams), not owned
    std::vector<hipStream_t> copy_streams;

    ~compeg_batch();
    compeg::Status upload(const compeg::ImageData *const *images, size_t n, int threads);
    // the same from JPEG bytes: ImageData::new for every image on the worker threads first (the bytes are
    // borrowed for the duration of the call)
    // begin_only: (device preprocessing from files) return as soon as the transfers are queued; finish_upload() completes it
    compeg::Status upload_jpegs(const uint8_t *const *jpegs, const size_t *lens, size_t n, int threads, unsigned flags, bool begin_only = false);
    using ImageSource = std::function<const compeg::ImageData *(size_t index, compeg::Status &status)>;
    compeg::Status upload_host(size_t n, int threads, const void *items, const ImageSource &image_of);
    void note_batch_properties(const compeg::ImageData *const *images, size_t n);
    struct FeedSource {
        const uint8_t *bytes; // what goes up for one image: its entropy-coded segment, or the whole file
        size_t len;
        uint32_t intervals;   // restart intervals its header announces
    };
    // defer_finish: return when every transfer is queued; finish_upload() then does the rest (pending_finish)
    compeg::Status upload_device_scan(size_t n, int threads, const FeedSource *src, const compeg::ImageData *const *given,
                                      const ImageSource &parse, const ImageSource &reparse, bool defer_finish = false);
    std::function<compeg::Status()> pending_finish; // the second step of an upload begun with upload_jpegs(..., begin_only)
    compeg::Status finish_upload();
    // the caller's (pointer, length) lists and the images parsed from them, for an upload whose second step comes later
    std::vector<const uint8_t *> feed_jpegs;
    std::vector<size_t> feed_lens;
    unsigned feed_flags = 0;
    std::vector<std::unique_ptr<compeg::ImageData>> feed_fresh;
    compeg::Status decode(hipStream_t stream);
};
