// Host front-end of libcompeg_hip: JPEG segment walk, validation of the
// supported subset, Huffman LUT construction and the uniform block.
// Mirrors the results of the reference's ImageData::new (src/lib.rs:597-824),
// JpegParser (src/file.rs:19-209), TableData/HuffmanTables
// (src/huffman.rs:33-119,247-271) and Metadata (src/metadata.rs:21-41).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "compeg_hip.h"
#include "device_types.h"

namespace compeg {

// Byte layout identical to the reference's #[repr(C)] Metadata (1112 bytes);
// qtables stay in zig-zag order exactly as read from DQT (lib.rs:693-699).
struct Component {
    uint32_t vsample, hsample, qtable, dchuff, achuff;
};
struct Metadata {
    uint32_t qtables[4][64];
    uint32_t restart_interval;
    Component components[3];
    uint32_t total_restart_intervals;
    uint32_t width_mcus;
    uint32_t max_hsample;
    uint32_t max_vsample;
    uint32_t dus_per_mcu;
    uint32_t retained_coefficients;
};
static_assert(sizeof(Metadata) == COMPEG_METADATA_BYTES, "Metadata layout");

// Internal parse flag (beside the public COMPEG_PARSE_*): do not search for the end of the entropy-coded segment
// (src/file.rs:163-201 walks every byte of it) when the file ends with an EOI marker -- take everything up to it.  For
// callers that have the segment checked where it is preprocessed anyway (batch uploads with device preprocessing).
constexpr unsigned kParseDeferScanEnd = 0x80000000u;

constexpr uint32_t kRetainedCoefficients = 32;       // metadata.rs:43
constexpr uint32_t kMaxRestartIntervals = 64u * 65535u; // lib.rs:298

struct Status {
    int code = COMPEG_OK;
    std::string message;
    bool ok() const { return code == COMPEG_OK; }
    static Status error(int c, std::string m) { return Status{c, std::move(m)}; }
};

// One Huffman table as a two-level LUT.  Entry = bits << 8 | value; an L1
// entry with bit 15 set delegates to l2[(entry & 0x7fff) + next 8 code bits].
struct HuffmanLut {
    uint16_t l1[256];
    std::vector<uint16_t> l2;
    // false when the code lengths do not describe a prefix code the reference
    // can tabulate (its builder asserts / indexes out of range there).
    bool build(const uint8_t counts[16], const uint8_t *symbols, size_t nsymbols);
    static HuffmanLut annex_k(int which); // 0 luma DC, 1 luma AC, 2 chroma DC, 3 chroma AC
};

struct ImageData {
    Metadata metadata;
    uint32_t width = 0, height = 0;
    uint16_t l1[4 * 256];
    std::vector<uint16_t> l2;
    // Decode-side acceleration table derived from l1/l2 (not part of the
    // reference's upload format): for each of the two AC tables, one entry per
    // 11-bit prefix in the format device_types.h describes.
    std::vector<uint16_t> ac_fast; // 2 x kFastEntries
    std::vector<uint16_t> dc_fast; // 2 x kDcFastEntries, for the two DC tables (device_types.h)
    unsigned flags = 0;            // COMPEG_PARSE_* this image was parsed with
    std::vector<uint8_t> owned; // Cow::Owned
    const uint8_t *jpeg = nullptr;
    size_t jpeg_len = 0;
    size_t scan_offset = 0, scan_len = 0;
    // parsed with kParseDeferScanEnd and the file ends with EOI: the segment was taken to run up to that EOI without
    // walking it (whoever preprocesses it has to see that no other marker lies inside: scan_kernels.hip, flag bit 1)
    bool scan_end_deferred = false;

    const uint8_t *scan_data() const { return jpeg + scan_offset; }
    uint32_t total_mcus() const
    {
        return metadata.total_restart_intervals * metadata.restart_interval;
    }
    uint32_t total_dus() const { return total_mcus() * metadata.dus_per_mcu; }

    // flags: COMPEG_PARSE_* (compeg_hip.h); 0 = exactly the reference's accept / reject behaviour
    static Status parse(const uint8_t *jpeg, size_t len, bool copy, ImageData **out, unsigned flags = 0);
};

} // namespace compeg
