// Plain-data types shared by the host runtime and the gfx950 kernels.
#pragma once

#include <cstdint>

namespace compeg {

constexpr int kWave = 64;              // CDNA wavefront
constexpr int kRetained = 32;          // zig-zag positions kept per data unit (metadata.rs:43)
constexpr int kMaxDusPerMcu = 6;
constexpr uint32_t kStripMcus = 32;    // MCUs per workgroup of the generic composite

// Direct AC tables (decode-side acceleration, derived from the reference LUTs;
// not part of the reference's upload format): one u16 per 11-bit code prefix,
//   bits 0..3   magnitude bits of the symbol
//   bits 4..8   code length + magnitude bits (<= 31)
//   bits 9..15  zig-zag positions to advance: run + 1; 17 for ZRL (quirk Q2);
//               64 for EOB (ends the data unit)
// and kFastEscape (size 0, advance 0, 15 magnitude bits: nothing a symbol can look like) when the code is longer
// than 11 bits.  An escape entry that is applied like a symbol changes nothing: no bits consumed, no position
// advanced -- the cooperative kernel's walk relies on that and looks at escapes once per step, late.
constexpr uint32_t kFastBits = 11;
constexpr uint32_t kFastEntries = 1u << kFastBits;
constexpr uint32_t kFastAdvEob = 64;
constexpr uint32_t kFastEscape = 15u;

// ref: an entry of the reference's LUTs, code length << 8 | symbol;
// zrl_advance: 17 like the reference (quirk Q2), 16 with COMPEG_PARSE_STANDARD_ENTROPY
constexpr uint32_t fast_entry(uint32_t ref, uint32_t zrl_advance)
{
    return (((ref & 0xffu) == 0u ? kFastAdvEob : ((ref & 0xffu) == 0xf0u ? zrl_advance : ((ref >> 4) & 15u) + 1u)) << 9) |
           ((((ref >> 8) & 31u) + (ref & 15u)) << 4) | (ref & 15u);
}

// Direct DC tables (cooperative kernel): one u16 per 9-bit code prefix in the same format -- bits 0..3 the
// category (magnitude bits), bits 4..8 code length + category, advance 1 -- or kFastEscape when the code is
// longer than 9 bits or the category larger than 15.  Stored behind the direct AC tables.
constexpr uint32_t kDcFastBits = 9;
constexpr uint32_t kDcFastEntries = 1u << kDcFastBits;

// The cooperative kernel's geometry (coop_body.h), a function of the restart interval alone -- host planning and
// the kernels compute it the same way.  A walk (a team of `waves` waves; or a lone wave) takes as many whole
// restart intervals as have 64 x waves data units together -- one interval if it alone is longer -- and decodes
// them in rounds of 64 data units; an interval may straddle rounds.
//   dpi       data units per interval (4:2:2: four per MCU)
//   ipw       intervals per walk
//   rounds    rounds of 64 data units that hold them
//   lpi       lanes of the walking wave that belong to one interval (lane 0 of them walks from its start, the
//             others speculate in fours: coop_body.h)
//   count     subsequences an interval is cut into: 1 + the speculative ones
//   list_cap  words of a walking lane's list of data-unit starts: 20 (64 lists = the bytes of a wave's 64 slots)
//             while a subsequence has at most 10 data units, more beyond (the lists then have an area of their own)
// Up to this many MCUs per interval one lane walks the whole interval through the walk tables (two symbols a step,
// no speculation, nothing to validate); longer intervals are cut into subsequences walked side by side.  Measured on
// one 960x720 frame (kernel time by HIP events, lane-per-interval / speculative): 5 MCUs 38 / 62 us, 10: 49 / 87,
// 16: 64 / 106, 30: 92 / 101, 60: 164 / 154, 120: 314 / 217 (round 3); with round 4's repairs of the speculative walks
// (kCoopEndSlack, coop_lane's interval ends) 30: 91 / 108, 40: 113 / 112, 60: - / 131, 128: - / 216, 240: - / 322.
#ifndef CG_COOP_LEAN_MAX
#define CG_COOP_LEAN_MAX 40
#endif
constexpr uint32_t kCoopLeanMaxRestart = CG_COOP_LEAN_MAX;
struct CoopShape {
    uint32_t dpi, ipw, rounds, lpi, count, list_cap;
};
constexpr uint32_t kCoopSlotListCap = 20;
constexpr CoopShape coop_shape(uint32_t restart_interval, uint32_t waves)
{
    CoopShape sh{};
    sh.dpi = 4u * restart_interval;
    const uint32_t room = uint32_t(kWave) * (waves ? waves : 1u);
    sh.ipw = (sh.dpi != 0u && sh.dpi <= room) ? room / sh.dpi : 1u;
    sh.rounds = (sh.ipw * sh.dpi + uint32_t(kWave) - 1u) / uint32_t(kWave);
    const uint32_t share = (sh.dpi + (waves ? waves : 1u) - 1u) / (waves ? waves : 1u); // (a lane per 64 / lanes-th of the walk's data units)
    const uint32_t fit = uint32_t(kWave) / sh.ipw;
    sh.lpi = share < fit ? (share ? share : 1u) : fit;
    sh.count = 1u + (sh.lpi - 1u) / 4u;
    const uint32_t per_sub = (sh.dpi + sh.count - 1u) / sh.count;
    // (a speculative lane's list that fills up ends its walk early; the lane-per-interval walks keep their entries in
    // the lanes' 20 words each: 4 (dpi + 1) <= 20 lpi)
    sh.list_cap = (restart_interval > kCoopLeanMaxRestart && per_sub > 10u) ? ((per_sub + 15u) & ~3u) : kCoopSlotListCap;
    return sh;
}

// LDS slot of one lane's data unit while it is being decoded: 32 int16 in
// zig-zag order + one dummy position (coefficients >= 32 are dropped there).
// 80 bytes per lane: 16-byte aligned, so that the slot is read back, cleared
// and -- as the quad exchange buffer of the composite -- written and read with
// 16-byte LDS accesses, and 20 dwords apart, which keeps the 16 lanes of each
// access phase on different banks.
constexpr int kDuSlotBytes = 80;

// The decoder's state at the start of an MCU, beside the index of the stream word it begins in (ImageDesc::mcu_word):
//   info bits 0..4   bit position inside that word
//        bits 5..10  the reference reader's `left` there (1..63: what quirk Q1 depends on; kernels_body.h, "Fast mode")
//        bit  11     kMcuDead: the reference's reader has run dry in an earlier data unit of the interval (quirk Q1):
//                    it reads zeros from here on, whatever the position
//   pred             the DC predictions (Y, Cb, Cr) up to here, i32 wrapping like the reference's
struct alignas(16) McuState {
    uint32_t info;
    int32_t pred[3];
};
constexpr uint32_t kMcuDead = 1u << 11;

// Everything the kernels need to know about one image.  Lives in device
// memory (one array entry per image of a batch); all pointers are device
// pointers.  Filled by the host from the reference-format Metadata block.
struct ImageDesc {
    // inputs, in the reference's upload format (lib.rs:397-407)
    const uint32_t *words;  // preprocessed scan, little-endian packed bytes
    const uint32_t *starts; // word index of every restart interval
    const uint16_t *l1;     // 4 x 256 entries
    const uint16_t *l2;     // l2_entries entries
    uint32_t nwords;
    uint32_t nstarts;
    uint32_t l2_entries;
    // two 2048-entry direct AC tables stored behind the L2 LUT (u16 index from l2)
    uint32_t fast_off;
    uint32_t fast_table[3]; // per component: 0 / 1, or 2 = none (out-of-range selector)
    // COMPEG_PARSE_STANDARD_ENTROPY: refill in front of DC codes, ZRL advances 16 (else 0: the reference)
    uint32_t standard_entropy;
    // cooperative kernel (coop_body.h): per component the direct DC table (0 / 1, 2 = none); whether the image
    // qualifies at all; and what a data unit decodes to once the reference's reader has underflown (quirk Q1:
    // it then reads zeros for the rest of the interval): [c][0] the DC difference, [c][1..31] the AC levels
    uint32_t dc_fast_table[3];
    uint32_t coop_ok;
    uint32_t zero_du_ok; // zero_du is known (4:2:2, direct tables for every component, no hostile category behind the all-zero prefix)
    int16_t zero_du[3][kRetained];
    // the walk tables (coop_body.h: kWalkWords words, made by launch_walk_tables from the direct
    // tables), or null: the cooperative kernel's walks then go symbol by symbol
    const uint32_t *walk;
    // geometry
    uint32_t total_intervals;
    uint32_t restart_interval; // MCUs per interval
    uint32_t dus_per_mcu;
    uint32_t width_mcus;
    uint32_t mcu_w, mcu_h; // pixels
    uint32_t total_dus;
    // per data unit inside an MCU: component index, 2 bits each
    uint32_t comp_of_du;
    // per component: L1 table index for DC / AC codes, DC quantiser
    uint32_t dc_table[3];
    uint32_t ac_table[3];
    uint32_t dc_quant[3];
    uint32_t hsample[3], vsample[3]; // sampling factors (composite stage)
    uint32_t du_base[3];             // first data unit of the component inside an MCU
    // quantisation tables as floats, zig-zag order, one row per component
    float quant[3][kRetained];
    // intermediates: quantised AC levels (slot 0 unused) and dequantised DC
    int16_t *ac;  // [total_dus][32]
    int32_t *dc;  // [total_dus]
    // output
    uint8_t *out; // RGBA8
    uint32_t out_w, out_h;
    // what lies behind the pointer: out_pitch / 4 pixels a row and this many rows -- the logical extent rounded up to
    // whole MCUs (16 x 16 pixels), so that an MCU the extent's edge cuts is stored whole, its outside into padding nobody
    // reads (0: as much as out_h)
    uint32_t out_alloc_h;
    uint32_t out_pitch; // bytes
    // launch bookkeeping for batched grids
    uint32_t first_huff_block; // block index of this image's first huffman block
    uint32_t first_idct_block;
    // The walk + lane-per-MCU route (kernels_body.h: walk_wave_422, decode_wave_fused_422<..., RECORDS>): the walk
    // kernel writes, for every MCU, the stream word its first data unit begins in (mcu_word) and the rest of the
    // decoder's state there (mcu_state); the decode kernel reads them through a second descriptor of the image in
    // which an "interval" is one MCU -- starts = mcu_word, total_intervals = the image's MCUs, restart_interval = 1.
    // mcu_ok: the image qualifies (4:2:2, direct tables for every component, no DC category above 15 in its tables:
    // the reader's state at an MCU's start is then a position, or "run dry" -- quirk Q1 -- and nothing else).
    uint32_t *mcu_word;
    McuState *mcu_state;
    uint32_t mcu_ok;
    uint32_t total_mcus;
};

// How the huffman kernel's dynamic LDS is carved (bytes).
struct HuffLdsPlan {
    uint32_t l2_entries_in_lds; // L2 entries staged (rest read from global)
    uint32_t window_words;      // per-wave scan window
    uint32_t waves_per_block;
    uint32_t total_bytes;
    uint32_t waves_that_fit;    // waves with such a window one CU's LDS holds
    bool window_cut;            // some wave's intervals are longer than the window (the rest: global reads)
};

} // namespace compeg
