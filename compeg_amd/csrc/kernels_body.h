// Per-lane bodies of the gfx950 kernels.
//
// The functions here are what one lane executes; kernels.hip wraps them in
// __global__ entry points.  Steps in which the lanes of a wave work together
// (the quad exchange of the composite, the record flush of the entropy kernel)
// are split into phases that every lane runs before any lane runs the next
// one, and GPU-only instructions (DPP, packed conversion, non-temporal stores,
// wave priority) sit behind __HIP_DEVICE_COMPILE__ next to a plain C++
// equivalent, so that tests/emul can compile the very same code with g++ under
// ASan/UBSan and drive it lane by lane, phase by phase (GPU sanitizers are not
// available on the target pool).
//
// Arithmetic contract (bit-exact with the reference's WGSL):
//   - bit reader: u32 arithmetic wraps, shift counts are taken modulo 32
//     (src/huffman.wgsl:35-79), including the reference's behaviour when the
//     buffer underflows at a DC code (quirk Q1) -- reproduced literally by the
//     exact path and arithmetically by the fast mode;
//   - ZRL advances 17 positions (Q2); only zig-zag positions < 32 are kept (Q3);
//     (Q1 and Q2 are switched off per image by COMPEG_PARSE_STANDARD_ENTROPY)
//   - IDCT: every operation is an individually rounded f32 operation in the
//     reference's order (src/dct.wgsl:73-201) -- this file must be compiled
//     with -ffp-contract=off;
//   - colour conversion: integer, arithmetic shifts (src/dct.wgsl:323-334).
#pragma once

#include <cstdint>
#include <cstring>

#include "device_types.h"

#if defined(__HIPCC__)
#define CG_DEV __device__ __forceinline__
#else
#define CG_DEV static inline
#endif

// Pointers that reach a kernel through the in-memory ImageDesc have no
// provable address space, and hipcc then emits flat_load / flat_store (slower,
// and they tie up the LDS wait counter as well).  CG_GLOBAL(T, p) states that
// p is a global-memory pointer; on the host it is a plain pointer.
#if defined(__HIP_DEVICE_COMPILE__)
#define CG_GLOBAL(T, p) (reinterpret_cast<__attribute__((address_space(1))) T *>(reinterpret_cast<uintptr_t>(p)))
// keeps a wave-uniform value in a scalar register instead of re-loading it
// from the descriptor inside hot loops
#define CG_PIN_SCALAR(x) asm volatile("" : "+s"(x))
#else
#define CG_GLOBAL(T, p) (p)
#define CG_PIN_SCALAR(x) (void)(x)
#endif

#ifndef CG_IDCT_PACKED
#define CG_IDCT_PACKED 1 // IDCT butterflies on pairs of floats (packed f32 instructions on the GPU)
#endif
#ifndef CG_PRIO_IDCT
#define CG_PRIO_IDCT 1 // wave priority (0..3) in the IDCT and, below, the composite; the entropy decode runs at 0
#endif
#ifndef CG_PRIO_ENTROPY
#define CG_PRIO_ENTROPY 0
#endif
#ifndef CG_PRIO_COMPOSITE
#define CG_PRIO_COMPOSITE 3
#endif
#ifndef CG_NT_STORES
#define CG_NT_STORES 1 // the output is written once and not read back by the kernel: non-temporal stores
#endif
// Knock-out arms of the kernel bodies (3 = AC loop twice, 4 = no IDCT, 5 = no colour arithmetic, 9 = no global stores,
// ...: what each phase costs inside the real mix) exist in lab builds only (lab.h: -DCOMPEG_LAB -DCG_EXP=n,
// tools/build_variant.sh, tools/ab_bench.sh); the shipped library is compiled with every arm off.
#if !defined(COMPEG_LAB)
#undef CG_EXP
#endif
#ifndef CG_EXP
#define CG_EXP 0
#endif

namespace compeg {

struct alignas(16) Vec4u {
    uint32_t x, y, z, w;
};

// 16 bytes of output pixels.  The output is written once and never read by the
// kernels: where a wave-wide store covers whole 64-byte segments (STREAM), a
// non-temporal store keeps it from displacing the L2's contents (7 % on
// 128-frame batches; `sc1` / `sc0 sc1` write-through variants were slower).
// Lone 16-byte pieces that the L2 has to merge with their neighbours (the
// paired kernel's per-lane stores) stay ordinary stores: non-temporal cost them 7 %.
template <bool STREAM>
CG_DEV void store_pixels(uint8_t *p, const Vec4u &v)
{
#if defined(COMPEG_LAB) && defined(CG_STORE_POLICY) && defined(__HIP_DEVICE_COMPILE__)
    // laboratory build: the cache-policy bits of the composite's stores by hand (sc0 / sc1 / nt), -DCG_STORE_POLICY="..."
    if (STREAM) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 val{v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off " CG_STORE_POLICY ::"v"(p), "v"(val) : "memory");
        return;
    }
#endif
#if CG_NT_STORES && defined(__HIP_DEVICE_COMPILE__)
    if (STREAM) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(
            u32x4{v.x, v.y, v.z, v.w},
            reinterpret_cast<__attribute__((address_space(1))) u32x4 *>(reinterpret_cast<uintptr_t>(p)));
        return;
    }
#endif
    *CG_GLOBAL(Vec4u, reinterpret_cast<Vec4u *>(p)) = v;
}

// ---------------------------------------------------------------------------
// Huffman / RLE decode, one lane per restart interval
// ---------------------------------------------------------------------------

// Views into the workgroup's LDS.
struct HuffShared {
    const uint16_t *l1;   // 5 x 256 entries; table 4 is all-zero (out-of-range selector)
    const uint16_t *l2;   // first l2_staged entries of the image's L2 LUT
    uint32_t l2_staged;
    const uint32_t *win;  // this wave's scan window, already byte-swapped to MSB-first
    uint32_t win_base;    // word index of win[0]
    uint32_t win_len;
    uint8_t *du_slots;    // this wave's 64 data-unit slots (kDuSlotBytes each)
};

CG_DEV uint32_t bswap32(uint32_t w)
{
    return (w << 24) | ((w & 0xff00u) << 8) | ((w >> 8) & 0xff00u) | (w >> 24);
}

struct BitReader {
    uint32_t cur, nxt, left, next_word;
};

// One 32-bit word of the interval's stream, MSB-first.  Words beyond the
// preprocessed data read as zero (the reference relies on WebGPU's robust
// buffer access there).
CG_DEV uint32_t fetch_word(const ImageDesc &d, const HuffShared &s, uint32_t idx)
{
    const uint32_t rel = idx - s.win_base;
    if (rel < s.win_len)
        return s.win[rel];
    return idx < d.nwords ? bswap32(CG_GLOBAL(const uint32_t, d.words)[idx]) : 0u;
}

CG_DEV void refill(BitReader &b, const ImageDesc &d, const HuffShared &s)
{
    if (b.left < 32u) {
        const uint32_t w = fetch_word(d, s, b.next_word);
        b.next_word += 1u;
        b.cur |= w >> (b.left & 31u);
        b.nxt = (w << 1) << ((31u - b.left) & 31u);
        b.left += 32u;
    }
}

CG_DEV void consume(BitReader &b, uint32_t n)
{
    b.cur = (b.cur << (n & 31u)) | ((b.nxt >> 1) >> ((31u - n) & 31u));
    b.nxt <<= (n & 31u);
    b.left -= n;
}

CG_DEV uint32_t peek(const BitReader &b, uint32_t n)
{
    return (b.cur >> 1) >> ((31u - n) & 31u);
}

CG_DEV int32_t huff_extend(int32_t v, uint32_t t)
{
    const int32_t vt = int32_t(1u << ((t - 1u) & 31u));
    const uint32_t ext = uint32_t(v) + (0xffffffffu << (t & 31u)) + 1u;
    return v < vt ? int32_t(ext) : v;
}

// The slot is written as int16 and read back in 16-byte pieces: that view must
// be allowed to alias (otherwise the compiler may reorder the two).
typedef uint32_t __attribute__((may_alias)) slot_word_t;
struct __attribute__((may_alias, aligned(16))) SlotVec {
    uint32_t x, y, z, w;
};

CG_DEV void zero_slot(uint8_t *slot)
{
    SlotVec *p = reinterpret_cast<SlotVec *>(slot);
#pragma unroll
    for (int i = 0; i < kRetained / 8; i++)
        p[i] = SlotVec{0u, 0u, 0u, 0u};
}

// Moves a finished data unit out of its slot (and clears the slot).
CG_DEV void take_slot(uint8_t *slot, uint32_t (&rec)[kRetained / 2])
{
    SlotVec *p = reinterpret_cast<SlotVec *>(slot);
#pragma unroll
    for (int i = 0; i < kRetained / 8; i++) {
        const SlotVec v = p[i];
        rec[4 * i + 0] = v.x;
        rec[4 * i + 1] = v.y;
        rec[4 * i + 2] = v.z;
        rec[4 * i + 3] = v.w;
    }
    zero_slot(slot);
}

constexpr uint32_t kL1Entries = 5 * 256; // 4 tables + the all-zero table
// How far one data unit can advance the reader: at most 63 symbols of at
// most 31 bits, one refill per symbol, plus the word kept in flight.
constexpr uint32_t kDuWordSlack = 66;

CG_DEV uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }
CG_DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
CG_DEV uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// Four consecutive dwords from a dword-aligned global address.
struct __attribute__((packed, aligned(4))) Dwords4 {
    uint32_t x, y, z, w;
};

// Workgroup prologue, thread `tid` of `nthreads`: copies the image's LUTs into
// LDS, 16 bytes per thread and load (both tables are 16-byte aligned in the
// upload blob and in LDS; tails shorter than 16 bytes go dword by dword), all of
// a thread's loads in flight before its first LDS write.
CG_DEV void copy_words_to_lds(uint32_t *dst, const uint32_t *src_, uint32_t n, uint32_t tid,
                              uint32_t nthreads)
{
    auto *src4 = CG_GLOBAL(const Dwords4, reinterpret_cast<const Dwords4 *>(src_));
    auto *src = CG_GLOBAL(const uint32_t, src_);
    const uint32_t n4 = n / 4u;
    uint32_t i = tid;
    for (; i + nthreads < n4; i += 2u * nthreads) {
        const Dwords4 a = src4[i], b = src4[i + nthreads];
        reinterpret_cast<SlotVec *>(dst)[i] = SlotVec{a.x, a.y, a.z, a.w};
        reinterpret_cast<SlotVec *>(dst)[i + nthreads] = SlotVec{b.x, b.y, b.z, b.w};
    }
    for (; i < n4; i += nthreads) {
        const Dwords4 a = src4[i];
        reinterpret_cast<SlotVec *>(dst)[i] = SlotVec{a.x, a.y, a.z, a.w};
    }
    for (uint32_t k = n4 * 4u + tid; k < n; k += nthreads)
        reinterpret_cast<slot_word_t *>(dst)[k] = src[k];
}

// extra: entries behind the direct AC tables that are wanted too (the cooperative kernel's direct DC tables)
CG_DEV void stage_luts(const ImageDesc &d, uint16_t *l1, uint16_t *l2, uint32_t l2_in_lds,
                       uint32_t tid, uint32_t nthreads, uint32_t extra = 0u)
{
    uint32_t *dst = reinterpret_cast<uint32_t *>(l1);
    copy_words_to_lds(dst, reinterpret_cast<const uint32_t *>(d.l1), 4 * 128, tid, nthreads);
    for (uint32_t i = tid; i < 128; i += nthreads)
        reinterpret_cast<slot_word_t *>(dst)[4 * 128 + i] = 0u;
    const uint32_t n2 = umin(l2_in_lds, d.fast_off + 2u * kFastEntries + extra);
    copy_words_to_lds(reinterpret_cast<uint32_t *>(l2), reinterpret_cast<const uint32_t *>(d.l2),
                      (n2 + 1) / 2, tid, nthreads);
}

// The scan window of the wave whose first interval is `wave_first`: the
// contiguous words of its 64 intervals plus the reader's two-word look-ahead,
// cut to the LDS budget.
// (in two steps, so that a wave can have the two loads of its next window in flight while it decodes)
CG_DEV void wave_window_fetch(const ImageDesc &d, uint32_t wave_first, uint32_t &raw_base, uint32_t &raw_end)
{
    raw_base = wave_first < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[wave_first] : 0u;
    const uint32_t after = wave_first + kWave;
    raw_end = (after < d.total_intervals && after < d.nstarts) ? CG_GLOBAL(const uint32_t, d.starts)[after] : d.nwords;
}

CG_DEV void wave_window_from(const ImageDesc &d, uint32_t raw_base, uint32_t raw_end, uint32_t window_words,
                             uint32_t &base, uint32_t &len)
{
    // the slack lets the last intervals of the wave pass the fast mode's in-window
    // test; positions past the end of the scan are staged as zeros
    // (+ 2: the reader keeps up to two words in hand, so at the start of the last data units of the wave's last
    // interval its position is that far beyond the interval's end -- still inside the window with these)
    const uint32_t end = umin(raw_end, d.nwords) + kDuWordSlack + 2u;
    base = umin(raw_base, d.nwords);
    len = end > base ? umin(end - base, window_words) : 0u;
}

CG_DEV void wave_window(const ImageDesc &d, uint32_t wave_first, uint32_t window_words,
                        uint32_t &base, uint32_t &len)
{
    uint32_t raw_base, raw_end;
    wave_window_fetch(d, wave_first, raw_base, raw_end);
    wave_window_from(d, raw_base, raw_end, window_words, base, len);
}

// Words idx .. idx+3 of the scan (zero past its end), MSB-first.
CG_DEV SlotVec scan_words4(const ImageDesc &d, uint32_t idx)
{
    Dwords4 w{0u, 0u, 0u, 0u};
    if (idx + 3u < d.nwords) {
        w = *CG_GLOBAL(const Dwords4, reinterpret_cast<const Dwords4 *>(d.words + idx));
    } else {
        auto *words = CG_GLOBAL(const uint32_t, d.words);
        w.x = idx < d.nwords ? words[idx] : 0u;
        w.y = idx + 1u < d.nwords ? words[idx + 1u] : 0u;
        w.z = idx + 2u < d.nwords ? words[idx + 2u] : 0u;
    }
    return SlotVec{bswap32(w.x), bswap32(w.y), bswap32(w.z), bswap32(w.w)};
}

// One wave stages its window: 16 bytes per lane and load, eight loads in
// flight per lane before the first LDS write -- a window of up to 2048 words
// (the benchmark frames need 1750) costs one memory round trip.  The window's
// LDS allocation is a multiple of 16 bytes, so the last vector may spill up to
// three words past `len` (they are never read).
// (in two halves, so that a kernel can put other loads between them)
CG_DEV void window_load_round(const ImageDesc &d, uint32_t base, uint32_t len, uint32_t v0, SlotVec (&w)[8])
{
    const uint32_t nvec = (len + 3u) / 4u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t v = v0 + uint32_t(j) * kWave;
        w[j] = v < nvec ? scan_words4(d, base + 4u * v) : SlotVec{0u, 0u, 0u, 0u};
    }
}

CG_DEV void window_store_round(uint32_t *win, uint32_t len, uint32_t v0, const SlotVec (&w)[8])
{
    const uint32_t nvec = (len + 3u) / 4u;
    SlotVec *dst = reinterpret_cast<SlotVec *>(win);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t v = v0 + uint32_t(j) * kWave;
        if (v < nvec)
            dst[v] = w[j];
    }
}

// first_round: the first round's loads have been issued by the caller already
CG_DEV void stage_window(const ImageDesc &d, uint32_t *win, uint32_t base, uint32_t len,
                         uint32_t lane)
{
    const uint32_t nvec = (len + 3u) / 4u;
    for (uint32_t v0 = lane; v0 < nvec; v0 += 8u * kWave) {
        SlotVec w[8];
        window_load_round(d, base, len, v0, w);
        window_store_round(win, len, v0, w);
    }
}

// The workgroup prologue of the decode kernels: the wave's window loads are
// issued first, the LUT copy runs under their latency.
CG_DEV void stage_luts_and_window(const ImageDesc &d, uint16_t *l1, uint16_t *l2, uint32_t l2_in_lds,
                                  uint32_t tid, uint32_t nthreads, uint32_t *win, uint32_t base, uint32_t len,
                                  uint32_t lane, uint32_t extra = 0u)
{
    SlotVec w[8];
    window_load_round(d, base, len, lane, w);
    stage_luts(d, l1, l2, l2_in_lds, tid, nthreads, extra);
    window_store_round(win, len, lane, w);
    const uint32_t nvec = (len + 3u) / 4u;
    for (uint32_t v0 = lane + 8u * kWave; v0 < nvec; v0 += 8u * kWave) {
        window_load_round(d, base, len, v0, w);
        window_store_round(win, len, v0, w);
    }
}

// Decodes restart interval `interval` of image d.  Quantised AC levels go
// through the lane's LDS slot and leave as one 64-byte record per data unit;
// the dequantised DC goes straight to d.dc.
CG_DEV void huff_decode_interval(const ImageDesc &d, const HuffShared &s, uint32_t interval,
                                 uint32_t lane)
{
    uint8_t *slot = s.du_slots + lane * kDuSlotBytes;
    int16_t *slot16 = reinterpret_cast<int16_t *>(slot);
    zero_slot(slot);

    BitReader b;
    b.next_word = interval < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[interval] : 0u;
    b.cur = b.nxt = b.left = 0u;
    refill(b, d, s);

    int32_t pred0 = 0, pred1 = 0, pred2 = 0;
    const uint32_t dpm = d.dus_per_mcu;
    const uint32_t du_count = d.restart_interval * dpm;
    uint32_t du_global = interval * du_count;
    uint32_t du_local = 0;
    uint32_t k = 0; // data unit inside the MCU
    uint32_t comp = d.comp_of_du & 3u;
    uint32_t dc_off = d.dc_table[0] * 256u, ac_off = d.ac_table[0] * 256u;
    if (comp == 1u) {
        dc_off = d.dc_table[1] * 256u;
        ac_off = d.ac_table[1] * 256u;
    } else if (comp == 2u) {
        dc_off = d.dc_table[2] * 256u;
        ac_off = d.ac_table[2] * 256u;
    }
    uint32_t pos = 0; // 0: the next symbol is the DC code; 1..63: AC position

    while (du_local < du_count) {
        const bool is_dc = pos == 0u;
        if (!is_dc || d.standard_entropy)
            refill(b, d, s); // the reference: no refill in front of a DC code (quirk Q1)

        // two-level LUT lookup on the next 16 bits (src/huffman.wgsl:85-113)
        const uint32_t code = b.cur >> 16;
        uint32_t e = s.l1[(is_dc ? dc_off : ac_off) + (code >> 8)];
        if (e & 0x8000u) {
            const uint32_t idx = (e & 0x7fffu) + (code & 0xffu);
            e = idx >= d.l2_entries ? 0u : (idx < s.l2_staged ? s.l2[idx] : CG_GLOBAL(const uint16_t, d.l2)[idx]);
        }
        consume(b, e >> 8);
        const uint32_t sym = e & 0xffu;

        // magnitude bits: category for DC, low nibble for AC (0 for EOB / ZRL)
        const uint32_t nbits = is_dc ? sym : (sym & 15u);
        const int32_t raw = int32_t(peek(b, nbits));
        consume(b, nbits);
        const int32_t val = huff_extend(raw, nbits);

        bool du_done = false;
        if (is_dc) {
            int32_t p = comp == 0u ? pred0 : (comp == 1u ? pred1 : pred2);
            p = int32_t(uint32_t(p) + uint32_t(val)); // dccat == 0 gives val == 0
            if (comp == 0u)
                pred0 = p;
            else if (comp == 1u)
                pred1 = p;
            else
                pred2 = p;
            const uint32_t q0 = comp == 0u ? d.dc_quant[0] : (comp == 1u ? d.dc_quant[1] : d.dc_quant[2]);
            CG_GLOBAL(int32_t, d.dc)[du_global] = int32_t(uint32_t(p) * q0);
            pos = 1u;
        } else if (sym == 0u) {
            du_done = true; // EOB
        } else {
            const uint32_t p = pos + (sym >> 4);
            const bool zrl = sym == 0xf0u;
            if (!zrl && p < uint32_t(kRetained))
                slot16[p] = int16_t(val);
            pos = p + ((zrl && !d.standard_entropy) ? 2u : 1u); // the reference: ZRL skips 17 in total (quirk Q2)
            du_done = pos >= 64u;
        }

        if (du_done) {
            // 64-byte record out, slot cleared for the next data unit
            uint32_t rec[kRetained / 2];
            take_slot(slot, rec);
            auto *dst = CG_GLOBAL(Vec4u, reinterpret_cast<Vec4u *>(d.ac + size_t(du_global) * kRetained));
#pragma unroll
            for (int i = 0; i < 4; i++)
                dst[i] = Vec4u{rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]};

            du_local++;
            du_global++;
            pos = 0u;
            k = k + 1u == dpm ? 0u : k + 1u;
            comp = (d.comp_of_du >> (2u * k)) & 3u;
            dc_off = (comp == 0u ? d.dc_table[0] : (comp == 1u ? d.dc_table[1] : d.dc_table[2])) * 256u;
            ac_off = (comp == 0u ? d.ac_table[0] : (comp == 1u ? d.ac_table[1] : d.ac_table[2])) * 256u;
        }
    }
}

// ---------------------------------------------------------------------------
// Inverse DCT of one data unit, entirely in one lane's registers
// ---------------------------------------------------------------------------

// f32 nearest to the reference's WGSL literals (src/dct.wgsl:24-27)
#define CG_S0 ((float)1.0)
#define CG_S1 ((float)1.387039845)
#define CG_S2 ((float)1.306562965)
#define CG_S3 ((float)1.175875602)
#define CG_S4 ((float)1.0)
#define CG_S5 ((float)0.785694958)
#define CG_S6 ((float)0.541196100)
#define CG_S7 ((float)0.275899379)
#define CG_C1414 ((float)1.414213562)
#define CG_C1847 ((float)1.847759065)
#define CG_C1082 ((float)1.082392200)
#define CG_C2613 ((float)2.613125930)

CG_DEV float aan_scale(int i)
{
    switch (i) {
    case 0: return CG_S0;
    case 1: return CG_S1;
    case 2: return CG_S2;
    case 3: return CG_S3;
    case 4: return CG_S4;
    case 5: return CG_S5;
    case 6: return CG_S6;
    default: return CG_S7;
    }
}

// natural (row-major) index -> zig-zag position (src/dct.wgsl:29-38)
CG_DEV int zigzag_of(int natural)
{
    constexpr int8_t t[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
                              3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                              10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
                              21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
    return t[natural];
}

// One 8-point AAN pass over v[0], v[stride], ... (libjpeg-turbo jidctflt
// structure as used by the reference).  `bias` is added to the DC input
// (128.5 in the row pass).
// T = float, or a pair of floats that the GPU handles with packed
// instructions (v_pk_add_f32 / v_pk_mul_f32: two independent IEEE operations,
// no contraction) -- two columns, or two rows, per instruction.
#if defined(__clang__)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#else
typedef float f32x2 __attribute__((vector_size(8)));
#endif
CG_DEV float aan_splat(float c, float) { return c; }
CG_DEV f32x2 aan_splat(float c, f32x2) { return f32x2{c, c}; }
template <class T> CG_DEV T aan_const(float c)
{
    return aan_splat(c, T{});
}

// The butterflies up to the last level: e[k] and o[k] with out[k] = e[k] + o[k], out[7 - k] = e[k] - o[k].
// The column pass works on inputs that carry the reference's factor 1/8 already (idct_data_unit folds it into
// the dequantisation constant: a power of two commutes with every rounding here); the row pass adds 128.5 to
// its first input.
template <class T, int STRIDE, bool ROW_PASS>
CG_DEV void aan_core(const T *v, T (&e)[4], T (&o)[4])
{
    T in0 = v[0 * STRIDE];
    const T in1 = v[1 * STRIDE], in2 = v[2 * STRIDE], in3 = v[3 * STRIDE];
    const T in4 = v[4 * STRIDE], in5 = v[5 * STRIDE], in6 = v[6 * STRIDE], in7 = v[7 * STRIDE];
    if (ROW_PASS)
        in0 = in0 + aan_const<T>(128.5f);
    // even part
    const T tmp10 = in0 + in4, tmp11 = in0 - in4;
    const T tmp13 = in2 + in6;
    const T tmp12 = (in2 - in6) * aan_const<T>(CG_C1414) - tmp13;
    e[0] = tmp10 + tmp13;
    e[3] = tmp10 - tmp13;
    e[1] = tmp11 + tmp12;
    e[2] = tmp11 - tmp12;
    // odd part
    const T z13 = in5 + in3, z10 = in5 - in3;
    const T z11 = in1 + in7, z12 = in1 - in7;
    const T o7 = z11 + z13;
    const T t11 = (z11 - z13) * aan_const<T>(CG_C1414);
    const T z5 = (z10 + z12) * aan_const<T>(CG_C1847);
    const T t10 = z5 - z12 * aan_const<T>(CG_C1082);
    const T t12 = z5 - z10 * aan_const<T>(CG_C2613);
    const T o6 = t12 - o7;
    const T o5 = t11 - o6;
    const T o4 = t10 - o5;
    o[0] = o7;
    o[1] = o6;
    o[2] = o5;
    o[3] = o4;
}

template <class T, int STRIDE, bool ROW_PASS>
CG_DEV void aan_1d(T *v)
{
    T e[4], o[4];
    aan_core<T, STRIDE, ROW_PASS>(v, e, o);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k * STRIDE] = e[k] + o[k];
        v[(7 - k) * STRIDE] = e[k] - o[k];
    }
}

// Column pass of two columns at once whose outputs come out transposed: for column HALF of the pair, {row k, row 7 - k}
// in one register pair.  The row pass then runs on the row pairs (0,7) (1,6) (2,5) (3,4) without a single move
// between the passes: a packed f32 instruction may take either half of each source for either half of its result
// (op_sel / op_sel_hi) and negate per half (neg_hi).
template <int HALF>
CG_DEV f32x2 aan_cross(f32x2 e, f32x2 o)
{
#if defined(__HIP_DEVICE_COMPILE__)
    f32x2 r;
    if (HALF == 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(e), "v"(o));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(r) : "v"(e), "v"(o));
    return r;
#else
    return f32x2{e[HALF] + o[HALF], e[HALF] - o[HALF]};
#endif
}

CG_DEV uint32_t sample_u8(float f)
{
    // clamp(f, 0, 255) = min(max(f, 0), 255), then truncation (dct.wgsl:174-197)
    float m = f > 0.0f ? f : 0.0f;
    m = m < 255.0f ? m : 255.0f;
    return uint32_t(m);
}

// clamp(f, 0, 255) then truncation for the 8 samples of each row, sample 0 in
// the low byte (src/dct.wgsl:174-197); o[2*r], o[2*r + 1]: row r.  On the GPU
// one v_cvt_pk_u8_f32 per sample does the saturation, the conversion and the
// byte insert; the instruction rounds with the current f32 rounding mode, so
// the mode is switched to round-toward-zero for exactly these instructions --
// inside one asm statement, so that nothing else can be scheduled between the
// two mode switches (which cost ~15 cycles each: rows go three at a time, as
// many as the operand limit of an asm statement allows).
CG_DEV void pack_rows3(const float *f0, const float *f1, const float *f2, uint32_t *o)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %7, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %8, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %9, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %11, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %12, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %13, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %15, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %16, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %17, 3, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %18, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %3, %19, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %20, 2, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %21, 3, %3\n\t"
                 "v_cvt_pk_u8_f32 %4, %22, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %4, %23, 1, %4\n\t"
                 "v_cvt_pk_u8_f32 %4, %24, 2, %4\n\t"
                 "v_cvt_pk_u8_f32 %4, %25, 3, %4\n\t"
                 "v_cvt_pk_u8_f32 %5, %26, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %5, %27, 1, %5\n\t"
                 "v_cvt_pk_u8_f32 %5, %28, 2, %5\n\t"
                 "v_cvt_pk_u8_f32 %5, %29, 3, %5\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
                 : "v"(f0[0]), "v"(f0[1]), "v"(f0[2]), "v"(f0[3]), "v"(f0[4]), "v"(f0[5]), "v"(f0[6]), "v"(f0[7]), "v"(f1[0]), "v"(f1[1]), "v"(f1[2]), "v"(f1[3]), "v"(f1[4]), "v"(f1[5]), "v"(f1[6]), "v"(f1[7]), "v"(f2[0]), "v"(f2[1]), "v"(f2[2]), "v"(f2[3]), "v"(f2[4]), "v"(f2[5]), "v"(f2[6]), "v"(f2[7]));
#else
    o[0] = sample_u8(f0[0]) | sample_u8(f0[1]) << 8 | sample_u8(f0[2]) << 16 | sample_u8(f0[3]) << 24;
    o[1] = sample_u8(f0[4]) | sample_u8(f0[5]) << 8 | sample_u8(f0[6]) << 16 | sample_u8(f0[7]) << 24;
    o[2] = sample_u8(f1[0]) | sample_u8(f1[1]) << 8 | sample_u8(f1[2]) << 16 | sample_u8(f1[3]) << 24;
    o[3] = sample_u8(f1[4]) | sample_u8(f1[5]) << 8 | sample_u8(f1[6]) << 16 | sample_u8(f1[7]) << 24;
    o[4] = sample_u8(f2[0]) | sample_u8(f2[1]) << 8 | sample_u8(f2[2]) << 16 | sample_u8(f2[3]) << 24;
    o[5] = sample_u8(f2[4]) | sample_u8(f2[5]) << 8 | sample_u8(f2[6]) << 16 | sample_u8(f2[7]) << 24;
#endif
}

CG_DEV void pack_rows2(const float *f0, const float *f1, uint32_t *o)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %7, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %11, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %15, 3, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %16, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %3, %17, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %18, 2, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %19, 3, %3\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                 : "v"(f0[0]), "v"(f0[1]), "v"(f0[2]), "v"(f0[3]), "v"(f0[4]), "v"(f0[5]), "v"(f0[6]), "v"(f0[7]), "v"(f1[0]), "v"(f1[1]), "v"(f1[2]), "v"(f1[3]), "v"(f1[4]), "v"(f1[5]), "v"(f1[6]), "v"(f1[7]));
#else
    o[0] = sample_u8(f0[0]) | sample_u8(f0[1]) << 8 | sample_u8(f0[2]) << 16 | sample_u8(f0[3]) << 24;
    o[1] = sample_u8(f0[4]) | sample_u8(f0[5]) << 8 | sample_u8(f0[6]) << 16 | sample_u8(f0[7]) << 24;
    o[2] = sample_u8(f1[0]) | sample_u8(f1[1]) << 8 | sample_u8(f1[2]) << 16 | sample_u8(f1[3]) << 24;
    o[3] = sample_u8(f1[4]) | sample_u8(f1[5]) << 8 | sample_u8(f1[6]) << 16 | sample_u8(f1[7]) << 24;
#endif
}

// quant: this component's 32 quantiser values as floats (zig-zag order).
// px[2*y], px[2*y+1]: the 8 samples of row y, sample 0 in the low byte --
// the reference's packed pixel format (dct.wgsl:187-201).
CG_DEV void idct_data_unit(const uint32_t (&ac)[kRetained / 2], int32_t dc, const float *quant,
                           uint32_t px[16])
{
    float v[64];
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int z = zigzag_of(r * 8 + c);
            // (the column pass's 1/8, src/dct.wgsl, rides on the constant: exact, a power of two)
            const float mul = (aan_scale(r) * aan_scale(c)) * 0.125f;
            float x = 0.0f;
            if (z == 0) {
                x = static_cast<float>(dc) * mul; // already dequantised with i32 wrap by the decoder
            } else if (z < kRetained) {
                // level * q is exact in f32 (|level| < 2^15, q < 2^8), so this
                // equals f32(i32(level * q)) of the reference
                // zig-zag position z sits in half (z & 1) of dword z / 2
                const int32_t lv = int32_t(int16_t(uint16_t(ac[z >> 1] >> ((z & 1) * 16))));
                const float level = static_cast<float>(lv);
#if CG_EXP == 13 // diagnostic build: no quantiser loads (wrong samples; what the scalar loads cost: nothing)
                x = (level * 3.0f) * mul;
#else
                x = (level * quant[z]) * mul;
#endif
            }
            v[r * 8 + c] = x;
        }
    }
#if CG_IDCT_PACKED
    // column pass on column pairs, row pass on the row pairs (k, 7 - k); element by element the
    // same operations in the same order as the scalar form below
    f32x2 cols[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int cp = 0; cp < 4; cp++)
            cols[r][cp] = f32x2{v[r * 8 + 2 * cp], v[r * 8 + 2 * cp + 1]};
    f32x2 rows[4][8];
#pragma unroll
    for (int cp = 0; cp < 4; cp++) {
        f32x2 e[4], o[4];
        aan_core<f32x2, 4, false>(&cols[0][cp], e, o);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            rows[k][2 * cp] = aan_cross<0>(e[k], o[k]);
            rows[k][2 * cp + 1] = aan_cross<1>(e[k], o[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        aan_1d<f32x2, 1, true>(&rows[k][0]);
    float f[8][8];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            f[k][c] = rows[k][c][0];
            f[7 - k][c] = rows[k][c][1];
        }
    pack_rows3(f[0], f[1], f[2], px + 0);
    pack_rows3(f[3], f[4], f[5], px + 6);
    pack_rows2(f[6], f[7], px + 12);
#else
#pragma unroll
    for (int c = 0; c < 8; c++)
        aan_1d<float, 8, false>(v + c);
#pragma unroll
    for (int r = 0; r < 8; r++)
        aan_1d<float, 1, true>(v + r * 8);
    pack_rows3(v + 0, v + 8, v + 16, px + 0);
    pack_rows3(v + 24, v + 32, v + 40, px + 6);
    pack_rows2(v + 48, v + 56, px + 12);
#endif
}

// ---------------------------------------------------------------------------
// Exact path: the reference reader, literally (fused / paired / entropy kernels fall back to it)
// ---------------------------------------------------------------------------

// Bit reader of the fused path.  Same state as the reference reader
// (src/huffman.wgsl:35-79) in a different shape: its two 32-bit words cur / nxt
// are the halves of one 64-bit register, so that
//   consume(n) : cur = cur << n | nxt >> (32 - n); nxt <<= n   is   buf <<= n
//   refill     : cur |= w >> left; nxt = (w << 1) << (31 - left)
//                is   hi(buf) |= hi(w << (32 - left)); lo(buf) = lo(w << (32 - left))
// (nxt is assigned, not merged: that matters only after a hostile DC category
// >= 32 made `left` lose track of the buffer).  `left` keeps the reference's
// wrapping u32 arithmetic, including its behaviour after an underflow at a DC
// code (quirk Q1: left wraps and no further word is ever loaded).  The next
// stream word is kept in flight: the read issued at one refill is consumed at
// the next, so its latency hides behind the symbols in between.
struct PrefetchReader {
    uint64_t buf;   // cur = high half, nxt = low half
    uint32_t left;
    uint32_t next_word;
    uint32_t pre;   // == word[next_word], MSB-first
};

CG_DEV uint32_t reader_cur(const PrefetchReader &r)
{
    return uint32_t(r.buf >> 32);
}

CG_DEV void reader_consume(PrefetchReader &r, uint32_t n)
{
    r.buf <<= (n & 31u); // the reference's shift counts are modulo 32
    r.left -= n;
}

CG_DEV uint32_t reader_peek(const PrefetchReader &r, uint32_t n)
{
    return (reader_cur(r) >> 1) >> ((31u - n) & 31u);
}

// FAST: the caller has checked that every word this data unit can touch lies
// inside the LDS window and that the whole L2 LUT is staged, so the loop body
// contains LDS accesses only.  Otherwise words outside the window (window cut
// short by the LDS budget, or a hostile start offset) and L2 entries beyond the
// staged part come from global memory; words past the end of the scan read as
// zero (the reference relies on robust buffer access there).
template <bool FAST>
CG_DEV uint32_t fetch_word_pf(const ImageDesc &d, const HuffShared &s, uint32_t idx)
{
    const uint32_t rel = idx - s.win_base;
    if (FAST)
        return s.win[rel];
    uint32_t w = s.win[rel < s.win_len ? rel : 0u];
    if (rel >= s.win_len)
        w = idx < d.nwords ? bswap32(CG_GLOBAL(const uint32_t, d.words)[idx]) : 0u;
    return w;
}

// The reference refill on the 64-bit register: cur |= w >> left, nxt := the
// bits of w that did not fit (callers pass need = left < 32).
CG_DEV uint64_t merge_word(uint64_t buf, uint32_t w, uint32_t left, bool need)
{
    const uint64_t ins = uint64_t(w) << ((32u - left) & 63u);
    const uint32_t hi = uint32_t(buf >> 32) | (need ? uint32_t(ins >> 32) : 0u);
    const uint32_t lo = need ? uint32_t(ins) : uint32_t(buf);
    return uint64_t(hi) << 32 | lo;
}

CG_DEV void reader_init(PrefetchReader &r, const ImageDesc &d, const HuffShared &s, uint32_t start)
{
    r.next_word = start;
    r.buf = 0u;
    r.left = 0u;
    r.pre = fetch_word_pf<false>(d, s, start);
}

// Branch-free form of the reference refill (src/huffman.wgsl:52-67): the word
// is merged only when `left < 32`, and the next word is (re)read every time --
// when nothing was consumed it is the same word again.
template <bool FAST>
CG_DEV void reader_refill(PrefetchReader &r, const ImageDesc &d, const HuffShared &s)
{
    const bool need = r.left < 32u;
    r.buf = merge_word(r.buf, r.pre, r.left, need);
    r.left += need ? 32u : 0u;
    r.next_word += need ? 1u : 0u;
    r.pre = fetch_word_pf<FAST>(d, s, r.next_word);
}

template <bool FAST>
CG_DEV uint32_t lut_lookup(const ImageDesc &d, const HuffShared &s, uint32_t table_off, uint32_t cur)
{
    const uint32_t code = cur >> 16;
    uint32_t e = s.l1[table_off + (code >> 8)];
    if (e & 0x8000u) {
        const uint32_t idx = (e & 0x7fffu) + (code & 0xffu);
        if (FAST) {
            e = idx < d.l2_entries ? s.l2[idx] : 0u;
        } else {
            uint32_t e2 = s.l2[idx < s.l2_staged ? idx : 0u];
            if (idx >= s.l2_staged)
                e2 = CG_GLOBAL(const uint16_t, d.l2)[idx < d.l2_entries ? idx : 0u];
            e = idx < d.l2_entries ? e2 : 0u;
        }
    }
    return e;
}

// Second half of a lookup whose L1 read was issued earlier: follows the
// delegate into the L2 LUT when the code is longer than 8 bits.
template <bool FAST>
CG_DEV uint32_t lut_resolve(const ImageDesc &d, const HuffShared &s, uint32_t e, uint32_t cur)
{
    if (e & 0x8000u) {
        const uint32_t idx = (e & 0x7fffu) + ((cur >> 16) & 0xffu);
        if (FAST) {
            e = idx < d.l2_entries ? s.l2[idx] : 0u;
        } else {
            uint32_t e2 = s.l2[idx < s.l2_staged ? idx : 0u];
            if (idx >= s.l2_staged)
                e2 = CG_GLOBAL(const uint16_t, d.l2)[idx < d.l2_entries ? idx : 0u];
            e = idx < d.l2_entries ? e2 : 0u;
        }
    }
    return e;
}

// DC difference of the next data unit.  No refill in front of it (quirk Q1);
// the two consume steps are kept separate so that a category >= 32 from a
// hostile table shifts exactly like the reference (counts modulo 32).
CG_DEV int32_t decode_dc_diff(PrefetchReader &r, const ImageDesc &d, const HuffShared &s,
                              uint32_t dc_off)
{
    if (d.standard_entropy)
        reader_refill<false>(r, d, s);
    const uint32_t e = lut_lookup<false>(d, s, dc_off, reader_cur(r));
    reader_consume(r, e >> 8);
    const uint32_t cat = e & 0xffu;
    const int32_t raw = int32_t(reader_peek(r, cat));
    reader_consume(r, cat);
    return huff_extend(raw, cat); // cat == 0 -> 0, as the reference special-cases
}

// AC coefficients of one data unit into the lane's (zeroed) LDS slot.
// Per symbol: code length <= 16 and magnitude bits <= 15, so both fields come
// out of `cur` alone and one combined consume (<= 31 bits) equals the
// reference's two.  EOB stores a 0 at the current position and ZRL a 0 at
// pos+15; both positions are still zero and are never revisited, so the
// stores need no special case; positions >= 32 (quirk Q3) land in the slot's
// padding.  ZRL advances 17 positions (quirk Q2).
// AC coefficients of one data unit, reference-exact reader (see fast_ac for
// the path normally taken).
// pos0: the zig-zag position to go on from (a data unit begun in fast mode: fast_ac<true>).
template <bool FAST>
CG_DEV void decode_ac_loop(PrefetchReader &r, const ImageDesc &d, const HuffShared &s,
                           uint32_t ac_off, int16_t *slot16, uint32_t pos0 = 1u)
{
    // Software-pipelined: the LUT lookup of the next symbol is issued as soon
    // as the bit position after the current one is known; magnitude
    // extraction, sign extension, the coefficient store and the position
    // update of the current symbol then run under that lookup's latency.
    // The look-ahead is speculative when the data unit ends here (EOB or
    // position 64): reader state is committed only if decoding continues, so
    // the DC code that follows sees exactly the reference's un-refilled
    // reader (quirk Q1).
    uint32_t pos = pos0;
    reader_refill<FAST>(r, d, s);
    uint32_t e = lut_lookup<FAST>(d, s, ac_off, reader_cur(r));
    bool done;
    do {
        const uint32_t len = e >> 8, sym = e & 0xffu, nb = sym & 15u;
        const uint32_t t = reader_cur(r) << (len & 31u); // magnitude bits at the top
        reader_consume(r, len + nb);

        // state the next symbol would start from; both LDS reads are issued here
        const bool need = r.left < 32u;
        const uint64_t buf_n = merge_word(r.buf, r.pre, r.left, need);
        const uint32_t nw_n = r.next_word + (need ? 1u : 0u);
        const uint32_t cur_n = uint32_t(buf_n >> 32);
        const uint32_t e1_n = s.l1[ac_off + (cur_n >> 24)];
        const uint32_t pre_n = fetch_word_pf<FAST>(d, s, nw_n);

        // the current symbol, under the latency of those reads.  Sign extension:
        // a magnitude whose first bit is 0 encodes raw - (2^nb - 1).
        const uint32_t raw = (t >> 1) >> (31u - nb);
        const uint32_t neg = ~uint32_t(int32_t(t) >> 31);      // all ones when that first bit is 0
        const uint32_t val = raw + (neg & ((0xffffffffu << nb) + 1u));
        const uint32_t p = pos + (sym >> 4);
        slot16[p < uint32_t(kRetained) ? p : uint32_t(kRetained)] = int16_t(val);
        pos = sym == 0u ? 64u : p + ((sym == 0xf0u && !d.standard_entropy) ? 2u : 1u);
        done = pos >= 64u;

        // commit the refill only if decoding continues, then rotate
        r.buf = done ? r.buf : buf_n;
        r.left += (need && !done) ? 32u : 0u;
        r.next_word = done ? r.next_word : nw_n;
        r.pre = done ? r.pre : pre_n;
        e = lut_resolve<FAST>(d, s, e1_n, cur_n);
    } while (!done);
}

// ---------------------------------------------------------------------------
// Fast mode
// ---------------------------------------------------------------------------
// The reference reader is what the results are defined by, but it is an
// expensive thing to step: a conditional word load in front of every AC
// symbol, none in front of a DC code (quirk Q1), two table levels.  Fast mode
// decodes the same symbols from a reader of its own and reconstructs the one
// piece of reference state that can influence a result:
//
//  * Its buffer holds true stream bits only and is topped up right after every
//    symbol (>= 32 valid bits before each AC symbol, like the reference, so an
//    AC symbol -- at most 31 bits -- never sees anything but stream bits in
//    either reader).
//  * The reference's `left` is congruent to minus the interval's consumed bit
//    count modulo 32 and lies in 32..63 after each of its refills, so its value
//    at the DC code that follows a data unit is
//        32 + ((left_fast + tot_last) & 31) - tot_last        (1..63)
//    where tot_last is the size of the data unit's last symbol.  The DC code
//    is then decoded from the buffer cut to that many bits (the reference
//    sees zeros behind them).  If the DC symbol needs more bits than that the
//    reference underflows; such a lane -- and any lane about to leave its LDS
//    window -- converts its state into the reference reader's and stays in
//    exact mode for the rest of the interval.
//  * AC symbols come from the direct tables (device_types.h): one LDS read
//    per symbol yields magnitude size, total size and position advance; codes
//    longer than 11 bits escape to the reference's two-level tables.
struct EntropyState {
    PrefetchReader r;     // fast mode: left = valid bits in buf, next_word unused
    const uint32_t *wptr; // fast mode: address (in the LDS window) of the next stream word; r.pre == *wptr
    uint32_t ref_left;    // fast mode: the reference reader's `left` at the coming DC code
    bool fast;
    int32_t pred0, pred1, pred2;
    // The streamed form of the window (decode_wave_fused_422_stream; nothing of it exists in the other kernels): the
    // lane reads its own column of the wave's rows -- row j, 64 words apart, holds the lane's stream word
    // r.next_word + j as it lies in memory (LSB-first: swapped when it is merged) --, wlimit is the first row that does
    // not hold one, and a lane that ran into it finishes its data unit with the reference reader and comes back to
    // fast mode when the rows are staged next (resume).
    const uint32_t *wlimit;
    bool resume;
};

// tests/emul counts how often the rare paths run (to prove the tests reach them)
#if defined(CG_EMUL_STATS)
struct EmulStats {
    unsigned long fast_dus, exact_dus, left_window, left_underflow, dc_cut, escapes;
    unsigned long symbols, lane_symbols, wave_steps, wave_step_symbols; // AC symbols; per-step maximum over a wave's lanes
};
inline EmulStats g_emul_stats{};
#define CG_COUNT(field) (++g_emul_stats.field)
#else
#define CG_COUNT(field) ((void)0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define CG_LDS(T, p) (reinterpret_cast<__attribute__((address_space(3))) T *>(reinterpret_cast<uintptr_t>(p)))
#else
#define CG_LDS(T, p) (p)
#endif

// LDS reads whose position in the instruction stream matters (the compiler
// otherwise sinks a load to its first use and the software pipelining is
// gone).  The value may only be used after lds_reads_done() on it.
CG_DEV uint32_t lds_read_u16_early(const uint16_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t v;
    asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(uint32_t(reinterpret_cast<uintptr_t>(p))) : "memory");
    return v;
#else
    return *p;
#endif
}

CG_DEV uint32_t lds_read_u32_early(const uint32_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(uint32_t(reinterpret_cast<uintptr_t>(p))) : "memory");
    return v;
#else
    return *p;
#endif
}

CG_DEV void lds_reads_done(uint32_t &a, uint32_t &b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
#else
    (void)a;
    (void)b;
#endif
}

CG_DEV void lds_reads_done3(uint32_t &a, uint32_t &b, uint32_t &c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c)::"memory");
#else
    (void)a;
    (void)b;
    (void)c;
#endif
}

CG_DEV bool fast_tables_usable(const ImageDesc &d, const HuffShared &s)
{
    return d.fast_table[0] < 2u && d.fast_table[1] < 2u && d.fast_table[2] < 2u &&
           s.l2_staged >= d.fast_off + 2u * kFastEntries;
}

CG_DEV void entropy_init(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t interval)
{
    reader_init(e.r, d, s, interval < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[interval] : 0u);
    reader_refill<false>(e.r, d, s); // left == 32: also a valid fast-mode state
    e.pred0 = e.pred1 = e.pred2 = 0;
    e.ref_left = 32u;
    const uint32_t rel = e.r.next_word - s.win_base;
    e.fast = fast_tables_usable(d, s) && rel < s.win_len;
    e.wptr = s.win + (e.fast ? rel : 0u);
}

// STREAM (decode_wave_fused_422_stream): index of the stream word e.wptr points at -- its row in the lane's column
// (s.win + lane; a column's rows are 64 words apart) behind the word of row 0.
CG_DEV uint32_t stream_word_index(const EntropyState &e, const HuffShared &s, uint32_t lane)
{
    return e.r.next_word + uint32_t(e.wptr - (s.win + lane)) / uint32_t(kWave);
}

// Fast -> exact: drop what was loaded ahead of the reference.
template <bool STREAM = false>
CG_DEV void leave_fast_mode(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t lane = 0u)
{
    PrefetchReader &r = e.r;
    if (d.standard_entropy)
        e.ref_left = r.left; // nothing to drop: the standard reader is topped up in front of DC codes too
    const uint32_t ahead = (r.left - e.ref_left) >> 5; // 0 or 1 word
    if (STREAM) {
        // (the rows hold the words as they lie in memory and a row behind the limit holds anything: both readers'
        // word in flight is read again, through the reference reader's fetch)
        r.next_word = stream_word_index(e, s, lane) - ahead;
        r.pre = fetch_word_pf<false>(d, s, r.next_word);
    } else {
        r.next_word = s.win_base + uint32_t(e.wptr - s.win) - ahead;
        if (ahead)
            r.pre = fetch_word_pf<false>(d, s, r.next_word);
    }
    r.buf &= ~uint64_t(0) << (64u - e.ref_left);
    r.left = e.ref_left;
    e.fast = false;
}

// STREAM, inside a data unit, behind its DC code (where the reference reader tops up in front of every symbol: from
// there on both readers hold the same bits): the lane has reached its last staged row.
CG_DEV void stream_leave_inside(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t lane)
{
    PrefetchReader &r = e.r;
    r.next_word = stream_word_index(e, s, lane);
    r.pre = fetch_word_pf<false>(d, s, r.next_word);
    e.fast = false;
    e.resume = true;
}

// Branch-free top-up: merges the stream word in flight when fewer than 32
// bits are left (0 <= left <= 63) and puts the next one in flight.
template <bool STREAM = false>
CG_DEV void fast_refill(EntropyState &e)
{
    PrefetchReader &r = e.r;
    const uint32_t t = r.left - 32u;
    const uint32_t f = t >> 31;                         // 1 when left < 32
    const uint32_t w = (STREAM ? bswap32(r.pre) : r.pre) & uint32_t(int32_t(t) >> 31);
    r.buf |= (uint64_t(w) << 32) >> (r.left & 63u);     // w == 0 when left >= 32
    r.left += f << 5;
    e.wptr += STREAM ? f * uint32_t(kWave) : f;
#if CG_EXP == 12
    r.pre = uint32_t(reinterpret_cast<uintptr_t>(e.wptr)) * 2654435761u;
#else
    r.pre = lds_read_u32_early(e.wptr); // due at the next lds_reads_done()
#endif
}

// nb-bit field that ends `tot` bits below the top of cur, sign-extended (0 for nb == 0)
CG_DEV int32_t signed_field(uint32_t cur, uint32_t tot, uint32_t nb)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sbfe(int32_t(cur), 32u - tot, nb); // v_bfe_i32 takes the offset modulo 32
#else
    return nb ? int32_t(cur << (tot - nb)) >> (32u - nb) : 0;
#endif
}

// DC difference in fast mode; false (nothing consumed) when the reference
// would run out of buffered bits here.
CG_DEV bool fast_dc(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t dc_off, int32_t &diff)
{
    PrefetchReader &r = e.r;
    if (d.standard_entropy)
        e.ref_left = r.left; // the reader was topped up behind the last symbol: >= 32 true bits
    const uint32_t seen = e.ref_left < 32u ? e.ref_left : 32u;
    if (e.ref_left < 32u)
        CG_COUNT(dc_cut);
    const uint32_t cur = reader_cur(r) & ~uint32_t(uint64_t(0xffffffffu) >> seen);
    const uint32_t ent = lut_lookup<true>(d, s, dc_off, cur);
    const uint32_t len = ent >> 8, cat = ent & 0xffu, n = len + cat;
    if (n > e.ref_left || cat > 16u)
        return false;
    const int32_t raw = int32_t(((cur << (len & 31u)) >> 1) >> (31u - cat));
    diff = huff_extend(raw, cat);
    r.buf <<= n;
    r.left -= n;
    return true;
}

#if defined(CG_AC_STAMPS) && defined(__HIPCC__)
// diagnostic build: [0] cycles inside the AC loop, [1] of those at the LDS wait, [2] iterations, [3] calls
__device__ unsigned long long g_ac_stamps[4];
#endif
#if defined(CG_AC_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define CG_AC_STAMP(x) x
#else
#define CG_AC_STAMP(x)
#endif

// Returns the zig-zag position of the coefficient decoded last: 63 or more at the data unit's end; STREAM: less when
// the lane's word in flight is no staged one (e.wptr at e.wlimit) -- whatever was consumed up to there was stream.
template <bool STREAM = false>
CG_DEV uint32_t fast_ac(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t ac_off,
                        uint32_t fast_base, int16_t *slot16)
{
    CG_AC_STAMP(const uint64_t ts0 = __builtin_readcyclecounter(); uint64_t tw = 0; uint32_t its = 0;)
    PrefetchReader &r = e.r;
    const uint16_t *tab = s.l2 + fast_base;
    if (STREAM && e.wptr >= e.wlimit)
        return 0u;
    fast_refill<STREAM>(e);
    uint32_t ent = lds_read_u16_early(tab + (reader_cur(r) >> (32u - kFastBits)));
    uint32_t at = 0u, tot = 0u; // at: zig-zag position of the coefficient decoded last
    // Software-pipelined like the exact loop, but nothing is speculative:
    // topping the buffer up early cannot change a result.  One LDS wait per
    // symbol: the lookup of the next symbol and the next stream word are
    // issued as early as the bit position allows and are both due at the top
    // of the next iteration; the coefficient store of a symbol is issued one
    // iteration late (ahead of those reads in the LDS queue), so that its
    // completion is never waited for.
    int16_t *pend_at = slot16 + kRetained; // the slot's padding
    int16_t pend_val = 0;
    while (at < 63u && (!STREAM || e.wptr < e.wlimit)) {
        CG_AC_STAMP(const uint64_t tw0 = __builtin_readcyclecounter();)
        lds_reads_done(ent, r.pre);
        CG_AC_STAMP(tw += __builtin_readcyclecounter() - tw0; its++;)
#if CG_EXP != 12
        *pend_at = pend_val;
#endif
#if CG_EXP != 7 // (7: diagnostic build without the escape test)
        if (__builtin_expect(ent == kFastEscape, 0))
#else
        if (false)
#endif
        {
            // code longer than 11 bits: this symbol through the reference's tables
            CG_COUNT(escapes);
            ent = fast_entry(lut_lookup<true>(d, s, ac_off, reader_cur(r)), d.standard_entropy ? 16u : 17u);
        }
#if CG_EXP >= 10 && CG_EXP <= 12 // diagnostic builds: every symbol is "5 bits, 2 of them magnitude, next position
                                  // + 5" (13 symbols per data unit in every lane); 11: without the table
                                  // read; 12: also without the stream-word read and the coefficient store
        ent = (5u << 9) | (5u << 4) | 2u;
#endif
        const uint32_t nb = ent & 15u, adv = ent >> 9;
        tot = (ent >> 4) & 31u;
        const uint32_t cur = reader_cur(r);
        r.buf <<= tot;
        r.left -= tot;
        fast_refill<STREAM>(e);
#if CG_EXP == 11 || CG_EXP == 12
        ent = reader_cur(r) >> 30;
#else
        ent = lds_read_u16_early(tab + (reader_cur(r) >> (32u - kFastBits)));
#endif
        // a magnitude whose first bit is 0 encodes field - (2^nb - 1)
        const int32_t sx = signed_field(cur, tot, nb);
        const uint32_t val = uint32_t(sx) + (((0xffffffffu << nb) ^ uint32_t(sx >> 31)) + 1u);
        at += adv;
        CG_COUNT(symbols);
        CG_COUNT(lane_symbols);
        // EOB / ZRL store a zero at a position that is zero anyway (or at the
        // slot's padding)
        pend_at = slot16 + umin(at, uint32_t(kRetained));
        pend_val = int16_t(val);
    }
    lds_reads_done(ent, r.pre);
    *pend_at = pend_val;
    CG_AC_STAMP(if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&g_ac_stamps[0], __builtin_readcyclecounter() - ts0);
        atomicAdd(&g_ac_stamps[1], tw);
        atomicAdd(&g_ac_stamps[2], its);
        atomicAdd(&g_ac_stamps[3], 1ull);
    })
    e.ref_left = 32u + ((r.left + tot) & 31u) - tot;
    return at;
}

// One data unit: DC difference + AC coefficients into `slot16` (zeroed by the
// consumer); returns the dequantised DC term.  comp is wave-uniform.
// STREAM: the wave's window in its streamed form (EntropyState); lane = the caller's.
template <bool STREAM = false>
CG_DEV int32_t entropy_data_unit(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t comp,
                                 int16_t *slot16, uint32_t lane = 0u)
{
    const uint32_t dc_off = d.dc_table[comp] * 256u, ac_off = d.ac_table[comp] * 256u;
    int32_t diff = 0;
    bool decoded = false;
    if (!STREAM && e.fast) {
        const uint32_t rel = uint32_t(e.wptr - s.win);
        if (s.win_len - rel < kDuWordSlack) { // rel < win_len holds in fast mode
            CG_COUNT(left_window);
            leave_fast_mode(e, d, s);
        }
    }
    if (STREAM && e.fast) {
        if (fast_dc(e, d, s, dc_off, diff)) {
            const uint32_t at = fast_ac<true>(e, d, s, ac_off, d.fast_off + d.fast_table[comp] * kFastEntries, slot16);
            CG_COUNT(fast_dus);
            if (at < 63u) {
                // the staged rows end inside this data unit: the reference reader takes the rest of it
                CG_COUNT(left_window);
                stream_leave_inside(e, d, s, lane);
                decode_ac_loop<false>(e.r, d, s, ac_off, slot16, at + 1u);
            }
            decoded = true;
        } else {
            CG_COUNT(left_underflow);
            leave_fast_mode<true>(e, d, s, lane);
        }
    } else if (e.fast) {
        if (fast_dc(e, d, s, dc_off, diff)) {
#if CG_EXP == 3 // diagnostic build: the AC loop twice (idempotent: same coefficients, same final state)
            {
                const EntropyState saved = e;
                fast_ac(e, d, s, ac_off, d.fast_off + d.fast_table[comp] * kFastEntries, slot16);
                e = saved;
            }
#endif
            fast_ac(e, d, s, ac_off, d.fast_off + d.fast_table[comp] * kFastEntries, slot16);
            decoded = true;
            CG_COUNT(fast_dus);
        } else {
            CG_COUNT(left_underflow);
            leave_fast_mode(e, d, s);
        }
    }
    if (!decoded) {
        CG_COUNT(exact_dus);
        diff = decode_dc_diff(e.r, d, s, dc_off);
        decode_ac_loop<false>(e.r, d, s, ac_off, slot16);
    }
    // (values, not addresses, are selected: the state has to stay in registers)
    const int32_t p0 = e.pred0, p1 = e.pred1, p2 = e.pred2;
    const int32_t p = int32_t(uint32_t(comp == 0u ? p0 : (comp == 1u ? p1 : p2)) + uint32_t(diff));
    e.pred0 = comp == 0u ? p : p0;
    e.pred1 = comp == 1u ? p : p1;
    e.pred2 = comp == 2u ? p : p2;
    return int32_t(uint32_t(p) * d.dc_quant[comp]);
}

// ---------------------------------------------------------------------------
// Composite: 4:2:2 chroma replication + YCbCr -> RGBA8
// ---------------------------------------------------------------------------

constexpr int kPxSlotWords = 17; // 16 words of samples + 1 pad (LDS banks)

CG_DEV uint32_t ycbcr_to_rgba(uint32_t y_, uint32_t cb_, uint32_t cr_)
{
    const int32_t y = int32_t(y_), cb = int32_t(cb_) - 128, cr = int32_t(cr_) - 128;
    int32_t r = y + ((45 * cr) >> 5);
    int32_t g = y - ((11 * cb + 23 * cr) >> 5);
    int32_t b = y + ((113 * cb) >> 6);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    return uint32_t(r) | uint32_t(g) << 8 | uint32_t(b) << 16 | 0xff000000u;
}

// Lane `j` of a wave composites pixel columns 4*(j%4) .. +3 of MCU
// `mcu = first_mcu + j/4` (4:2:2: MCU = 16x8 pixels, data units Y0 Y1 Cb Cr)
// for all 8 rows, so that one wave-wide 16-byte store covers 16 MCUs x 64 B
// of one pixel row.  px_slots: the wave's 64 sample slots in LDS, slot i =
// data unit (first_mcu*4 + i).
CG_DEV void composite_422(const ImageDesc &d, const uint32_t *px_slots, uint32_t first_mcu,
                          uint32_t total_mcus, uint32_t j)
{
    const uint32_t m = j >> 2, quarter = j & 3u;
    const uint32_t mcu = first_mcu + m;
    if (mcu >= total_mcus)
        return;
    const uint32_t mx = mcu % d.width_mcus, my = mcu / d.width_mcus;
    const uint32_t x0 = mx * 16u + quarter * 4u;
    if (x0 >= d.out_w)
        return;
    const uint32_t *ydu = px_slots + (m * 4u + (quarter >> 1)) * kPxSlotWords;
    const uint32_t *cbdu = px_slots + (m * 4u + 2u) * kPxSlotWords;
    const uint32_t *crdu = px_slots + (m * 4u + 3u) * kPxSlotWords;
    const uint32_t cshift = (quarter & 1u) * 16u; // chroma samples 2*quarter, 2*quarter+1
#pragma unroll
    for (uint32_t row = 0; row < 8; row++) {
        const uint32_t y = my * 8u + row;
        if (y >= d.out_h)
            break;
        const uint32_t yw = ydu[row * 2u + (quarter & 1u)];
        const uint32_t cbw = cbdu[row * 2u + (quarter >> 1)] >> cshift;
        const uint32_t crw = crdu[row * 2u + (quarter >> 1)] >> cshift;
        Vec4u o;
        o.x = ycbcr_to_rgba(yw & 0xffu, cbw & 0xffu, crw & 0xffu);
        o.y = ycbcr_to_rgba((yw >> 8) & 0xffu, cbw & 0xffu, crw & 0xffu);
        o.z = ycbcr_to_rgba((yw >> 16) & 0xffu, (cbw >> 8) & 0xffu, (crw >> 8) & 0xffu);
        o.w = ycbcr_to_rgba(yw >> 24, (cbw >> 8) & 0xffu, (crw >> 8) & 0xffu);
        uint8_t *p = d.out + size_t(y) * d.out_pitch + size_t(x0) * 4u;
        if (x0 + 3u < d.out_w && (d.out_pitch & 15u) == 0u) {
            store_pixels<true>(p, o);
        } else {
            auto *q = CG_GLOBAL(uint32_t, reinterpret_cast<uint32_t *>(p));
            q[0] = o.x;
            if (x0 + 1u < d.out_w)
                q[1] = o.y;
            if (x0 + 2u < d.out_w)
                q[2] = o.z;
            if (x0 + 3u < d.out_w)
                q[3] = o.w;
        }
    }
}


CG_DEV uint32_t ycbcr_to_rgba(uint32_t y_, uint32_t cb_, uint32_t cr_);

// Four pixels that share two chroma samples: yw = 4 luma bytes, cb2 / cr2 =
// the two chroma bytes in bits 0..15.  Integer BT.601 approximation of the
// reference (src/dct.wgsl:323-334); on the GPU it runs two 16-bit lanes per
// instruction (all intermediates fit 16 bits: |45*cr'| <= 5760,
// |11*cb'+23*cr'| <= 4352, |113*cb'| <= 14464).
CG_DEV Vec4u rgba_quad(uint32_t yw, uint32_t cb2, uint32_t cr2)
{
    Vec4u o;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s2 __attribute__((ext_vector_type(2)));
    // the two chroma bytes -> the two 16-bit lanes; the -128 bias is folded
    // into the constant term of each product (45 * 128 = 5760, ...)
    const s2 cb = __builtin_bit_cast(s2, __builtin_amdgcn_perm(0u, cb2, 0x0c010c00u));
    const s2 cr = __builtin_bit_cast(s2, __builtin_amdgcn_perm(0u, cr2, 0x0c010c00u));
    const s2 k45 = {45, 45}, k11 = {11, 11}, k23 = {23, 23}, k113 = {113, 113};
    const s2 b45 = {-5760, -5760}, b34 = {-4352, -4352}, b113 = {-14464, -14464};
    const s2 sh5 = {5, 5}, sh6 = {6, 6};
    const s2 rc = (cr * k45 + b45) >> sh5;
    const s2 gc = (cb * k11 + (cr * k23 + b34)) >> sh5;
    const s2 bc = (cb * k113 + b113) >> sh6;
    // luma lanes (p0, p2) and (p1, p3): lane 0 uses chroma sample 0, lane 1 sample 1
    const s2 ya = __builtin_bit_cast(s2, yw & 0x00ff00ffu);
    const s2 yb = __builtin_bit_cast(s2, (yw >> 8) & 0x00ff00ffu);
    uint32_t ra, ga, ba, rb, gb, bb; // saturated bytes of the two lanes in bits 0..15, upper half zero
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ra) : "v"(ya + rc));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ga) : "v"(ya - gc));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ba) : "v"(ya + bc));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(rb) : "v"(yb + rc));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(gb) : "v"(yb - gc));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(bb) : "v"(yb + bc));
    const uint32_t rga = ga << 16 | ra, rgb = gb << 16 | rb; // bytes R(p0) R(p2) G(p0) G(p2)
    // pixel = R, G, B, 255: selector bytes 0..3 pick from the second source, 4..7 from the first, 0x0d = 0xff
    o.x = __builtin_amdgcn_perm(ba, rga, 0x0d040200u);
    o.z = __builtin_amdgcn_perm(ba, rga, 0x0d050301u);
    o.y = __builtin_amdgcn_perm(bb, rgb, 0x0d040200u);
    o.w = __builtin_amdgcn_perm(bb, rgb, 0x0d050301u);
#else
    o.x = ycbcr_to_rgba(yw & 0xffu, cb2 & 0xffu, cr2 & 0xffu);
    o.y = ycbcr_to_rgba((yw >> 8) & 0xffu, cb2 & 0xffu, cr2 & 0xffu);
    o.z = ycbcr_to_rgba((yw >> 16) & 0xffu, (cb2 >> 8) & 0xffu, (cr2 >> 8) & 0xffu);
    o.w = ycbcr_to_rgba(yw >> 24, (cb2 >> 8) & 0xffu, (cr2 >> 8) & 0xffu);
#endif
    return o;
}

// Four pixels with a chroma sample each (cb4 / cr4: one byte per pixel) --
// the same arithmetic for layouts whose chroma is not shared pairwise.
CG_DEV Vec4u rgba_quad4(uint32_t yw, uint32_t cb4, uint32_t cr4)
{
    Vec4u o;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 k45 = {45, 45}, k11 = {11, 11}, k23 = {23, 23}, k113 = {113, 113};
    const s2 b45 = {-5760, -5760}, b34 = {-4352, -4352}, b113 = {-14464, -14464};
    const s2 sh5 = {5, 5}, sh6 = {6, 6};
    // lanes (p0, p2) and (p1, p3) of luma and chroma alike
    const s2 ya = __builtin_bit_cast(s2, yw & 0x00ff00ffu), yb = __builtin_bit_cast(s2, (yw >> 8) & 0x00ff00ffu);
    const s2 cba = __builtin_bit_cast(s2, cb4 & 0x00ff00ffu), cbb = __builtin_bit_cast(s2, (cb4 >> 8) & 0x00ff00ffu);
    const s2 cra = __builtin_bit_cast(s2, cr4 & 0x00ff00ffu), crb = __builtin_bit_cast(s2, (cr4 >> 8) & 0x00ff00ffu);
    uint32_t ra, ga, ba, rb, gb, bb;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ra) : "v"(ya + ((cra * k45 + b45) >> sh5)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ga) : "v"(ya - ((cba * k11 + (cra * k23 + b34)) >> sh5)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(ba) : "v"(ya + ((cba * k113 + b113) >> sh6)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(rb) : "v"(yb + ((crb * k45 + b45) >> sh5)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(gb) : "v"(yb - ((cbb * k11 + (crb * k23 + b34)) >> sh5)));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(bb) : "v"(yb + ((cbb * k113 + b113) >> sh6)));
    const uint32_t rga = ga << 16 | ra, rgb = gb << 16 | rb;
    o.x = __builtin_amdgcn_perm(ba, rga, 0x0d040200u);
    o.z = __builtin_amdgcn_perm(ba, rga, 0x0d050301u);
    o.y = __builtin_amdgcn_perm(bb, rgb, 0x0d040200u);
    o.w = __builtin_amdgcn_perm(bb, rgb, 0x0d050301u);
#else
    o.x = ycbcr_to_rgba(yw & 0xffu, cb4 & 0xffu, cr4 & 0xffu);
    o.y = ycbcr_to_rgba((yw >> 8) & 0xffu, (cb4 >> 8) & 0xffu, (cr4 >> 8) & 0xffu);
    o.z = ycbcr_to_rgba((yw >> 16) & 0xffu, (cb4 >> 16) & 0xffu, (cr4 >> 16) & 0xffu);
    o.w = ycbcr_to_rgba(yw >> 24, cb4 >> 24, cr4 >> 24);
#endif
    return o;
}

// px[k][2*row + half]: data unit k (Y0, Y1, Cb, Cr), 4 samples per word.
//
// The composite of an MCU (16x8 pixels, 8 rows of 64 bytes) goes out through
// the lane's quad: every lane converts its own MCU row by row, the four lanes
// of a quad exchange the row through LDS (each lane's coefficient slot is idle
// at this point and serves as the exchange buffer), and lane i of the quad
// stores piece i (16 bytes) of each of the quad's four rows.  A wave-wide
// store then writes 16 segments of 64 contiguous bytes instead of 64 separate
// 16-byte pieces; the vector memory path processes it in a quarter of the time.

// Row `row` of the lane's MCU into its slot (4 x 16 bytes).
CG_DEV void composite_row_to_slot(const uint32_t (&px)[4][16], uint32_t row, uint8_t *slot)
{
    SlotVec *p = reinterpret_cast<SlotVec *>(slot);
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
#if CG_EXP == 5 // diagnostic build: stores without the colour arithmetic
        p[q] = SlotVec{px[q >> 1][row * 2u + (q & 1u)], px[2][row * 2u + (q >> 1)], px[3][row * 2u + (q >> 1)], q};
#else
        const Vec4u o = rgba_quad(px[q >> 1][row * 2u + (q & 1u)], px[2][row * 2u + (q >> 1)] >> ((q & 1u) * 16u),
                                  px[3][row * 2u + (q >> 1)] >> ((q & 1u) * 16u));
        p[q] = SlotVec{o.x, o.y, o.z, o.w};
#endif
    }
}

// Lane `lane` stores its piece of row `row` of the four MCUs of its quad.
// bases[j]: top-left byte of the MCU of the quad's lane j; bit j of whole_mask:
// that MCU lies entirely inside the output and is stored this way.
CG_DEV void composite_row_from_quad(const ImageDesc &d, const uint8_t *wave_slots, uint32_t lane, uint32_t row,
                                    uint8_t *const (&bases)[4], uint32_t whole_mask)
{
    const uint32_t quad = lane & ~3u, piece = lane & 3u;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const SlotVec v = reinterpret_cast<const SlotVec *>(wave_slots + (quad + j) * kDuSlotBytes)[piece];
#if (CG_EXP == 9 || CG_EXP == 17 || CG_EXP == 18) && defined(__HIP_DEVICE_COMPILE__) // diagnostic build: everything but the global stores
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        if (v.x == 0x12345678u && v.y == 0x9abcdef0u && bases[j] == nullptr)
#else
        if (whole_mask >> j & 1u)
#endif
        {
            store_pixels<true>(bases[j] + size_t(row) * d.out_pitch + piece * 16u, Vec4u{v.x, v.y, v.z, v.w});
        }
    }
}

// The limits of four MCUs (byte j: target_limit of source j) as this lane needs them -- byte j: the rows of source j in
// which the lane's piece lies inside the output, none if the piece does not.
CG_DEV uint32_t cut_rows_for_piece(uint32_t limits, uint32_t piece)
{
    uint32_t rows = 0u;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t lim = (limits >> (8u * j)) & 0xffu;
        rows |= (piece < lim >> 5 ? lim & 31u : 0u) << (8u * j);
    }
    return rows;
}
// bit 8 j + 7: row `row` of source j is inside (bytes of at most 16 rows: no carry from byte to byte)
CG_DEV uint32_t cut_row_inside(uint32_t rows, uint32_t row) { return (rows + 0x01010101u * (127u - row)) & 0x80808080u; }

// The same for a restart interval of one MCU, where the wave's lanes hold 64 consecutive MCUs: store t of a row
// takes the MCUs of lanes 16 t .. 16 t + 15, lane l piece l & 3 of the MCU of lane 16 t + (l >> 2) -- 1 KB in one
// piece wherever those sixteen MCUs lie in one MCU row (64-byte segments 256 bytes apart are what every longer
// interval leaves the write path with).
// bases[t], bit t of whole_mask: of the MCU of lane 16 t + (lane >> 2).
CG_DEV void composite_row_from_wave(const ImageDesc &d, const uint8_t *wave_slots, uint32_t lane, uint32_t row,
                                    uint8_t *const (&bases)[4], uint32_t whole_mask)
{
    const uint32_t piece = lane & 3u, sub = lane >> 2;
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        const SlotVec v = reinterpret_cast<const SlotVec *>(wave_slots + (16u * t + sub) * kDuSlotBytes)[piece];
        if (whole_mask >> t & 1u)
            store_pixels<true>(bases[t] + size_t(row) * d.out_pitch + piece * 16u, Vec4u{v.x, v.y, v.z, v.w});
    }
}

// An MCU cut by the right / bottom edge of the output (stores outside it are
// dropped, like textureStore in the reference) or an unaligned pitch: the
// owning lane stores it pixel by pixel.
CG_DEV void composite_edge_mcu(const ImageDesc &d, const uint32_t (&px)[4][16], uint32_t mx, uint32_t my)
{
    const uint32_t x0 = mx * 16u, y0 = my * 8u;
    uint8_t *base = d.out + size_t(y0) * d.out_pitch + size_t(x0) * 4u;
#pragma unroll 1
    for (uint32_t row = 0; row < 8; row++) {
        if (y0 + row >= d.out_h)
            break;
#pragma unroll 1
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t w = row * 2u + (q & 1u), c = row * 2u + (q >> 1);
            // select the words without dynamic register indexing
            uint32_t yw = 0, cbw = 0, crw = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) {
                yw = (i == w) ? ((q >> 1) ? px[1][i] : px[0][i]) : yw;
                cbw = (i == c) ? px[2][i] : cbw;
                crw = (i == c) ? px[3][i] : crw;
            }
            const Vec4u o = rgba_quad(yw, cbw >> ((q & 1u) * 16u), crw >> ((q & 1u) * 16u));
            const uint32_t x = x0 + q * 4u;
            auto *p = CG_GLOBAL(uint32_t, reinterpret_cast<uint32_t *>(base + size_t(row) * d.out_pitch + q * 16u));
            if (x < d.out_w)
                p[0] = o.x;
            if (x + 1u < d.out_w)
                p[1] = o.y;
            if (x + 2u < d.out_w)
                p[2] = o.z;
            if (x + 3u < d.out_w)
                p[3] = o.w;
        }
    }
}

// ---------------------------------------------------------------------------
// The same work as two cooperating roles (decoder wave + transformer wave)
// ---------------------------------------------------------------------------

// Entropy decode of restart intervals into the coefficient records the IDCT
// kernels read (two-kernel pipeline, extension layouts): per data unit one
// 64-byte record of quantised AC levels (zig-zag order, position 0 unused) in
// d.ac and the dequantised DC term in d.dc.  Any sampling the front-end accepts.
//
// The records leave through the lane's quad, like the pixels of the fused
// kernel: lane i of a quad stores piece i (16 bytes) of the four records of
// its quad, straight out of the four LDS slots, so that a wave-wide store
// covers 16 x 64 contiguous bytes (a scattered one occupies the vector memory
// path for 64 cycles, and there are five per data unit otherwise).
// interval: this lane's (possibly past the image's last) restart interval.
CG_DEV void records_flush_quad(const ImageDesc &d, const uint8_t *wave_slots, uint32_t lane, uint32_t interval,
                               uint32_t du_in_interval, int32_t dc)
{
    const uint32_t quad = lane & ~3u, piece = lane & 3u;
    const uint32_t du_count = d.restart_interval * d.dus_per_mcu;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t other = interval - piece + j; // the interval of the quad's lane j
        if (other < d.total_intervals) {
            const SlotVec v = reinterpret_cast<const SlotVec *>(wave_slots + (quad + j) * kDuSlotBytes)[piece];
            auto *rec = CG_GLOBAL(Vec4u, reinterpret_cast<Vec4u *>(
                                             d.ac + (size_t(other) * du_count + du_in_interval) * kRetained));
            rec[piece] = Vec4u{v.x, v.y, v.z, v.w};
        }
    }
    if (interval < d.total_intervals)
        CG_GLOBAL(int32_t, d.dc)[size_t(interval) * du_count + du_in_interval] = dc;
}

// The lane's coefficient slot -> the data unit's 64 samples, in the slot (the
// extension layouts flush samples instead of coefficients: one pass over HBM less).
CG_DEV void slot_to_samples(const ImageDesc &d, uint32_t comp, uint8_t *slot, int32_t dc)
{
    uint32_t rec[kRetained / 2];
    take_slot(slot, rec);
    uint32_t px[16];
    idct_data_unit(rec, dc, d.quant[comp], px);
    SlotVec *p = reinterpret_cast<SlotVec *>(slot);
#pragma unroll
    for (int i = 0; i < 4; i++)
        p[i] = SlotVec{px[4 * i], px[4 * i + 1], px[4 * i + 2], px[4 * i + 3]};
}

#if defined(__HIPCC__)
// All 64 lanes of a wave call this together (tests/emul drives the same steps
// lane by lane); lanes past the last interval decode the last one again and
// only help their quad store.  SAMPLES: the records carry the IDCT's output
// instead of its input.
template <bool SAMPLES>
CG_DEV void entropy_wave_to_records(const ImageDesc &d, const HuffShared &s, uint32_t interval, uint32_t lane)
{
    uint8_t *slot = s.du_slots + lane * kDuSlotBytes;
    int16_t *slot16 = reinterpret_cast<int16_t *>(slot);
    zero_slot(slot);

    EntropyState e;
    entropy_init(e, d, s, interval < d.total_intervals ? interval : d.total_intervals - 1u);

    const uint32_t dpm = d.dus_per_mcu;
    const uint32_t du_count = d.restart_interval * dpm;
    uint32_t k = 0; // data unit inside the MCU (wave-uniform)
#pragma unroll 1
    for (uint32_t du = 0; du < du_count; du++) {
        const uint32_t comp = (d.comp_of_du >> (2u * k)) & 3u;
        const int32_t dc = entropy_data_unit(e, d, s, comp < 3u ? comp : 2u, slot16);
        if (SAMPLES)
            slot_to_samples(d, comp < 3u ? comp : 2u, slot, dc);
        records_flush_quad(d, s.du_slots, lane, interval, du, dc);
        zero_slot(slot);
        k = k + 1u == dpm ? 0u : k + 1u;
    }
}
#endif

// ---------------------------------------------------------------------------
// Layouts other than 4:2:2 (extension): three plain kernels
// ---------------------------------------------------------------------------
// entropy_samples_kernel decodes and transforms: every data unit's 64-byte
// record holds its 64 samples (what the reference's coefficient buffer holds
// after its dct pass, src/dct.wgsl:187-201), and a last pass converts four horizontally
// adjacent pixels per lane (the reference's finalize
// pass, src/dct.wgsl:257-321, continued to 16-row MCUs as orc_finalize_pass
// states it).

// Pixels x0 .. x0+3 (x0 a multiple of 4) of output row y.  Sampling factors
// are 1 or 2 and MCU sizes 8 or 16, so every division below is a shift.
// samples: the sample records, starting with MCU `mcu0` (the whole buffer with
// mcu0 = 0, or a strip of it that a workgroup has copied into LDS).
CG_DEV void composite_generic_4px(const ImageDesc &d, uint32_t x0, uint32_t y, const uint8_t *samples_, uint32_t mcu0)
{
    if (x0 >= d.out_w || y >= d.out_h)
        return;
    const uint32_t wsh = d.mcu_w == 16u ? 4u : 3u, hsh = d.mcu_h == 16u ? 4u : 3u; // log2 of the MCU size
    const uint32_t mcu_x = x0 >> wsh, mcu_y = y >> hsh;
    const uint32_t mcu = mcu_y * d.width_mcus + mcu_x;
    // right of the last MCU column, or behind the last decoded MCU (a truncated
    // last restart interval): nothing is stored there, the output keeps its zeros
    if (mcu_x >= d.width_mcus || mcu >= d.total_intervals * d.restart_interval)
        return;
    const uint32_t col = x0 & (d.mcu_w - 1u), row = y & (d.mcu_h - 1u);
    // the four pixels read one 8-sample row of one data unit per component (col is a
    // multiple of 4 and a data unit spans 8 or 16 pixel columns): one 8-byte load each
    struct alignas(8) Row8 {
        uint32_t lo, hi;
    };
    const Row8 *samples = reinterpret_cast<const Row8 *>(samples_);
    uint32_t val[3]; // the component's sample of each of the four pixels, one byte per pixel
#pragma unroll
    for (uint32_t c = 0; c < 3; c++) {
        const uint32_t hs = d.hsample[c], vs = d.vsample[c]; // 1 or 2
        const uint32_t xsh = (wsh - 3u) - (hs - 1u), ysh = (hsh - 3u) - (vs - 1u); // log2 of xscale, yscale
        const uint32_t yy = (row >> ysh) & 7u;
        const uint32_t du = d.du_base[c] + ((row * vs) >> hsh) * hs + ((col * hs) >> wsh);
        const Row8 r = samples[(size_t(mcu - mcu0) * d.dus_per_mcu + du) * 8u + yy];
        const uint64_t bits = uint64_t(r.hi) << 32 | r.lo;
        if (xsh == 0u) {
            val[c] = uint32_t(bits >> ((col & 7u) * 8u)); // four consecutive samples
        } else {
            // two samples, each under two pixels
            const uint32_t two = uint32_t(bits >> (((col >> 1) & 7u) * 8u));
            val[c] = (two & 0xffu) * 0x0101u | ((two >> 8) & 0xffu) * 0x01010000u;
        }
    }
    const Vec4u o = rgba_quad4(val[0], val[1], val[2]);
    uint8_t *p = d.out + size_t(y) * d.out_pitch + size_t(x0) * 4u;
    if (x0 + 3u < d.out_w && (d.out_pitch & 15u) == 0u) {
        store_pixels<true>(p, o);
        return;
    }
    auto *q = CG_GLOBAL(uint32_t, reinterpret_cast<uint32_t *>(p));
    q[0] = o.x;
    if (x0 + 1u < d.out_w)
        q[1] = o.y;
    if (x0 + 2u < d.out_w)
        q[2] = o.z;
    if (x0 + 3u < d.out_w)
        q[3] = o.w;
}

// Transformer role: the samples of the MCU being assembled and where it goes.
struct PixelState {
    uint32_t px[4][16]; // the MCU's four data units (Y0 Y1 Cb Cr), 4 samples per word
    uint32_t mx, my;
    bool active;        // false: a lane past the image's last interval; it only helps its quad store
};

CG_DEV void pixel_init(PixelState &t, const ImageDesc &d, uint32_t interval, bool active)
{
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int w = 0; w < 16; w++)
            t.px[k][w] = 0u;
    const uint32_t mcu = interval * d.restart_interval;
    t.mx = mcu % d.width_mcus;
    t.my = mcu / d.width_mcus;
    t.active = active;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define CG_PLACE_MARK(text) asm volatile(text)
#else
#define CG_PLACE_MARK(text) do { } while (0)
#endif

// One data unit: coefficients out of `slot` (cleared for reuse) and IDCT.
// k = 0..3: Y0 Y1 Cb Cr; at the end of the MCU px[k] holds data unit k.  One
// copy of the IDCT in the instruction stream: it leaves its 16 words in px[3],
// and the first three data units are moved to their place afterwards (16 moves
// behind a wave-uniform branch; a shift chain through all four arrays would
// move 48 words per data unit).
CG_DEV void pixel_transform(PixelState &t, const ImageDesc &d, uint32_t comp, uint32_t k, uint8_t *slot, int32_t dc)
{
    uint32_t rec[kRetained / 2];
    take_slot(slot, rec);
#if CG_EXP == 4 || CG_EXP == 17 // diagnostic build: no IDCT (same data flow); 17: nor any global store
#pragma unroll
    for (int w = 0; w < 16; w++)
        t.px[3][w] = rec[w] + uint32_t(dc);
#elif CG_EXP == 16 || CG_EXP == 18 // diagnostic build: no IDCT for the chroma data units (more than a sparse chroma transform
                                   // can save); 18: nor any global store
    if (k >= 2u) {
#pragma unroll
        for (int w = 0; w < 16; w++)
            t.px[3][w] = rec[w] + uint32_t(dc);
    } else {
        idct_data_unit(rec, dc, d.quant[comp], t.px[3]);
    }
#else
    idct_data_unit(rec, dc, d.quant[comp], t.px[3]);
#endif
    // (an empty asm statement of its own in every branch: without it the three blocks are merged into one
    // store through a computed index, and px leaves the register file)
    if (k == 0u) {
#pragma unroll
        for (int w = 0; w < 16; w++)
            t.px[0][w] = t.px[3][w];
        CG_PLACE_MARK("; data unit 0 in place");
    } else if (k == 1u) {
#pragma unroll
        for (int w = 0; w < 16; w++)
            t.px[1][w] = t.px[3][w];
        CG_PLACE_MARK("; data unit 1 in place");
    } else if (k == 2u) {
#pragma unroll
        for (int w = 0; w < 16; w++)
            t.px[2][w] = t.px[3][w];
        CG_PLACE_MARK("; data unit 2 in place");
    }
}

// Where the MCU the lane has just finished goes.
struct McuTarget {
    uint8_t *base; // top-left byte
    bool whole;    // entirely inside the output, 16-byte aligned rows: stored through the quad
};

// What of an MCU (group) the output's right or bottom edge cuts still goes through the quad: rows | 16-byte pieces of a row
// << 5 that lie inside (the cut at a piece's end, 16-byte aligned rows); 0: nothing -- an MCU outside, or the edge path's,
// pixel by pixel.  (1920x1080 in 16-row MCUs: the last MCU row is such; a wave that holds one MCU of the edge path's
// waits for it -- 256 x 1080p 4:2:0 1030 us against 786 for 1088 rows.)  Worked out where a wave has such an MCU only:
// the common case keeps its registers.  (The layouts' kernels; the 4:2:2 kernels of 168 registers have none to spare for
// it -- their MCUs are 8 rows high, what cuts them is rarer.)
CG_DEV uint32_t target_limit(const ImageDesc &d, bool active, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
    const uint32_t px_in = x0 < d.out_w ? umin(w, d.out_w - x0) : 0u, rows_in = y0 < d.out_h ? umin(h, d.out_h - y0) : 0u;
    return active && (d.out_pitch & 15u) == 0u && (px_in & 3u) == 0u && px_in && rows_in ? rows_in | (px_in >> 2) << 5 : 0u; // (rows <= 16, pieces <= 4: a byte)
}

CG_DEV McuTarget mcu_target(const PixelState &t, const ImageDesc &d)
{
#if CG_EXP == 14 // diagnostic build (DRI = 4 only; wrong picture, the same bytes): the MCUs of a wave's 64 intervals
                 // change places so that every store instruction of the composite writes 1 KB in one piece (lane
                 // 4 q + j at its MCU m goes where MCU 64 j + 16 m + q of the wave's 256 would): what the write
                 // path would do with ideal geometry
    const uint32_t mcu = t.my * d.width_mcus + t.mx, unit = mcu & ~255u, lane_ = (mcu >> 2) & 63u, m_ = mcu & 3u;
    const uint32_t moved = umin(unit + 64u * (lane_ & 3u) + 16u * m_ + (lane_ >> 2), d.total_intervals * d.restart_interval - 1u);
    const uint32_t x0 = (moved % d.width_mcus) * 16u, y0 = (moved / d.width_mcus) * 8u;
#else
    const uint32_t x0 = t.mx * 16u, y0 = t.my * 8u;
#endif
    McuTarget g;
    g.base = d.out + size_t(y0) * d.out_pitch + size_t(x0) * 4u;
    // (inside what is allocated: an MCU the output's edge cuts puts its outside into the rows' and the image's padding)
    g.whole = t.active && x0 + 16u <= umax(d.out_w, d.out_pitch / 4u) && y0 + 8u <= umax(d.out_h, d.out_alloc_h) && (d.out_pitch & 15u) == 0u &&
              x0 < d.out_w && y0 < d.out_h;
    return g;
}

CG_DEV void pixel_next_mcu(PixelState &t, const ImageDesc &d)
{
    t.mx++;
    if (t.mx == d.width_mcus) {
        t.mx = 0;
        t.my++;
    }
}

// true in every lane if `v` is in any (the host build runs lanes one after the other: tests/emul decides for the wave)
CG_DEV bool wave_any(bool v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(v) != 0u;
#else
    return v;
#endif
}

// ---- the window in its streamed form: restart intervals of any length ----
// A wave's window (above) holds its 64 intervals whole: 64 x DRI MCUs of stream, 17 KB with DRI = 10, beyond what
// the LDS has with DRI = 16 -- the waves per CU go, then the window itself.  Here a lane has `nrows` words of its
// own stream staged at any time, whatever the interval's length: row j of the wave's rows (64 words, one per lane)
// holds word j behind each lane's own position, staged (stream_stage_rows) behind the last data unit of an MCU -- when
// some lane is running short (stream_wants_rows).  The fast reader walks down its lane's column
// (EntropyState); a lane that reaches its last row inside a data unit finishes it, and the MCU, with the reference
// reader from global memory and is back in fast mode with the next rows.

// Row j of `rows` := word `first + j` of every lane's stream, j < nrows.
// (first <= d.nwords; the words behind an image's last are readable -- runtime.cpp pads its buffers -- and never used)
//
// Through registers: loads, then LDS writes.  The first form fetched the rows by LDS-DMA (global_load_lds_dword: no
// registers, landing under the IDCT) and was as fast, not faster (256 x 960x720 DRI = 10: 0.351 ms this way, 0.359
// that way); in dense streams the GPU fuzz found a few images in a thousand wrong with it, other ones on every run,
// and none with this -- with M0 held still per sixteen rows, the in-flight count under its six bits and a stand-alone
// stress of the staging clean, the cause was not found (profiles/r03/NOTES.md).
#if defined(COMPEG_LAB) && defined(CG_STREAM_LDSDMA) && defined(__HIP_DEVICE_COMPILE__)
// Laboratory builds only (-DCOMPEG_LAB -DCG_STREAM_LDSDMA; tools/repro_ldsdma.sh): the LDS-DMA staging of round 3, as it
// was when the GPU fuzz found a few images in a thousand wrong with it in dense streams (seed 9902, batch 235) -- kept so
// that the finding can be reproduced.  One global_load_lds_dword a row (per-lane source, the wave's 64 words to M0 +
// offset + 4 lane); sixteen rows to one M0, told apart by the instruction's offset, a full wait in front of every
// further sixteen; the count of vector memory operations in flight kept under its six bits.
#define CG_GLDS_ROW(J) \
    "global_load_lds_dword %[voff], %[base] offset:" #J "*256\n\t" \
    "s_cmp_eq_u32 %[n], " #J "+1\n\t" \
    "s_cbranch_scc1 9f\n\t" \
    "v_add_u32 %[voff], 0xffffff04, %[voff]\n\t"
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// four stream words of a lane in one load (they need not lie on a 16-byte boundary)
typedef uint32_t QuadWordsAligned __attribute__((ext_vector_type(4)));
typedef QuadWordsAligned QuadWords __attribute__((aligned(4)));
#endif

CG_DEV void stream_stage_rows(const ImageDesc &d, uint32_t *rows, uint32_t nrows, uint32_t first, uint32_t lane)
{
#if defined(COMPEG_LAB) && defined(CG_STREAM_LDSDMA) && defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    // CG_LDSDMA_TRY (a bit mask, tools/repro_ldsdma.sh): what was tried against the defect, one change a bit
#if defined(CG_LDSDMA_TRY) && (CG_LDSDMA_TRY & 4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (no store in flight beside the rows' loads)
#else
    asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
#endif
#if defined(CG_LDSDMA_TRY) && (CG_LDSDMA_TRY & 8)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (no LDS read in flight when the first row is issued)
#endif
#if defined(CG_LDSDMA_TRY) && (CG_LDSDMA_TRY & 2)
    uint32_t keep_m0;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0)::"memory"); // (M0 as the compiler left it, put back below)
#endif
    const uint8_t *base = reinterpret_cast<const uint8_t *>(d.words) - 4096;
    uint32_t m0 = uint32_t(reinterpret_cast<uintptr_t>(rows));
#pragma unroll 1
    for (uint32_t j = 0; j < nrows; j += 16u) {
        uint32_t voff = (first + j) * 4u + 4096u, n = umin(16u, nrows - j);
        if (j)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the loads that still need the old M0)
        asm volatile("s_mov_b32 m0, %[m0]\n\t"
                     "s_nop 0\n\t" //
                     CG_GLDS_ROW(0) CG_GLDS_ROW(1) CG_GLDS_ROW(2) CG_GLDS_ROW(3) CG_GLDS_ROW(4) CG_GLDS_ROW(5) CG_GLDS_ROW(6)
                         CG_GLDS_ROW(7) CG_GLDS_ROW(8) CG_GLDS_ROW(9) CG_GLDS_ROW(10) CG_GLDS_ROW(11) CG_GLDS_ROW(12)
                             CG_GLDS_ROW(13) CG_GLDS_ROW(14) CG_GLDS_ROW(15) "9:"
                     : [voff] "+v"(voff)
                     : [m0] "s"(m0), [base] "s"(base), [n] "s"(n)
                     : "memory", "scc");
        m0 += 16u * uint32_t(kWave) * 4u;
    }
#if defined(CG_LDSDMA_TRY) && (CG_LDSDMA_TRY & 2)
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep_m0) : "memory");
#endif
#elif defined(__HIP_DEVICE_COMPILE__) && !defined(CG_STAGE_DWORDS)
    // Four rows an instruction: a lane's words lie one behind the other in memory, every lane's in a line of its own -- the
    // texture path takes a line a cycle whatever is asked of it, 16 bytes of a line cost what 4 do.
    auto *words = CG_GLOBAL(const uint32_t, d.words);
    const uint32_t whole = nrows & ~3u;
#pragma unroll 4
    for (uint32_t j = 0; j < whole; j += 4u) {
        const QuadWords q = *reinterpret_cast<const __attribute__((address_space(1))) QuadWords *>(words + first + j);
        CG_LDS(uint32_t, rows)[(j + 0u) * uint32_t(kWave) + lane] = q.x;
        CG_LDS(uint32_t, rows)[(j + 1u) * uint32_t(kWave) + lane] = q.y;
        CG_LDS(uint32_t, rows)[(j + 2u) * uint32_t(kWave) + lane] = q.z;
        CG_LDS(uint32_t, rows)[(j + 3u) * uint32_t(kWave) + lane] = q.w;
    }
    for (uint32_t j = whole; j < nrows; j++)
        CG_LDS(uint32_t, rows)[j * uint32_t(kWave) + lane] = words[first + j];
#elif defined(__HIP_DEVICE_COMPILE__)
    auto *words = CG_GLOBAL(const uint32_t, d.words);
#pragma unroll 8
    for (uint32_t j = 0; j < nrows; j++)
        CG_LDS(uint32_t, rows)[j * uint32_t(kWave) + lane] = words[first + j];
#else
    for (uint32_t j = 0; j < nrows; j++)
        rows[j * uint32_t(kWave) + lane] = first + j < d.nwords ? d.words[first + j] : 0xfeedf00du;
#endif
}

// (the LDS-DMA form waits here for its rows; the place where a staging is complete)
CG_DEV void stream_rows_landed()
{
#if defined(COMPEG_LAB) && defined(CG_STREAM_LDSDMA) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(CG_LDSDMA_TRY) && (CG_LDSDMA_TRY & 1)
    asm volatile("s_sleep 8\n\ts_nop 7" ::: "memory"); // (some 500 cycles between the wait and the first read of a row)
#endif
#endif
}

// The lane's rows anew from the word it has in flight; lanes in fast mode go on in it, lanes that left it for want
// of rows come back (at a data unit's boundary the reference reader's state is a fast-mode state), lanes whose reader
// has underflowed (Q1) stay with the reference reader.
CG_DEV void stream_restage(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t nrows, uint32_t lane)
{
    const uint32_t word = e.fast ? stream_word_index(e, s, lane) : e.r.next_word;
    const uint32_t first = umin(word, d.nwords);
    // (a fast lane that stands at its limit -- its last data unit ended with the last staged word merged -- holds in
    // r.pre what lay behind the rows, not a stream word: that word comes from memory here, as the rows have it)
    if (e.fast && e.wptr >= e.wlimit) {
        CG_COUNT(left_window);
        e.r.pre = word < d.nwords ? CG_GLOBAL(const uint32_t, d.words)[word] : 0u;
    }
    stream_stage_rows(d, const_cast<uint32_t *>(s.win), nrows, first, lane);
    const bool back = !e.fast && e.resume && e.r.left < 64u; // (>= 2^31: the reference reader has underflowed since)
    if (back) {
        e.ref_left = e.r.left;
        e.r.pre = bswap32(e.r.pre); // (as it lies in memory, like the rows)
    }
    if (e.fast || back) {
        e.r.next_word = word; // fast mode: the word of row 0
        e.wptr = s.win + lane;
        e.wlimit = s.win + lane + umin(nrows, d.nwords - first) * uint32_t(kWave);
        e.fast = true;
    }
    e.resume = false;
}

// Would new rows help this lane?  It has fewer than `below` staged words in front of it and the scan goes on behind
// them, or it waits for rows to come back to fast mode.  (The wave stages when any lane says so: fewer, larger
// fetches -- a line of the scan is then fetched once or twice, not once per MCU.)
CG_DEV bool stream_wants_rows(const EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t lane, uint32_t below)
{
    if (!e.fast)
        return e.resume && e.r.left < 64u;
    const uint32_t left = uint32_t(e.wlimit - e.wptr) / uint32_t(kWave);
    const uint32_t end_word = e.r.next_word + uint32_t(e.wlimit - (s.win + lane)) / uint32_t(kWave);
    return left < below && end_word < d.nwords;
}

// The reference reader at the interval's start (two words from global memory), then the first rows and fast mode.
CG_DEV void stream_lane_init(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t nrows, uint32_t interval, uint32_t lane)
{
    reader_init(e.r, d, s, interval < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[interval] : 0u);
    reader_refill<false>(e.r, d, s); // left == 32: also a valid fast-mode state
    e.pred0 = e.pred1 = e.pred2 = 0;
    e.ref_left = 32u;
    e.fast = false;
    e.resume = fast_tables_usable(d, s);
    e.wptr = e.wlimit = s.win + lane;
    stream_restage(e, d, s, nrows, lane);
}

// ---- the walk + lane-per-MCU route: batches whose restart intervals are too few, or too long, for a lane each ----
// A lane per restart interval (the reference's parallelism) leaves the chip idle when a launch has few intervals -- 256
// frames of 960x720 with an interval per MCU row are 360 waves for 1024 SIMDs, each lane with 240 data units in front
// of it.  What is serial inside an interval is only where its MCUs begin and what the DC predictions are there.  So:
// a first kernel *walks* the intervals, a lane each, symbol sizes only (walk_body.h: walk_wave_422), and writes the
// decoder's state at every MCU's start (ImageDesc::mcu_word / mcu_state); a second one decodes with a lane per *MCU*
// (decode_wave_fused_422<..., RECORDS>: the kernel of the one-MCU intervals, started from those states), its rows
// leaving in pieces of 1 KB.  Both follow the reference's reader exactly -- the states are the reader's own --, so the
// pixels are the ones a lane per interval produces.

// The decoder at the start of MCU `mcu`, from the record the walk wrote (walk_body.h: walk_record; d: the image's
// descriptor of MCUs, see ImageDesc::mcu_word).
// The reference reader's state there -- `left` bits of stream in its buffer, zeros behind them, the next word in
// flight -- is also a fast-mode state (entropy_init).
CG_DEV void entropy_init_from_record(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t mcu)
{
    const uint32_t word = mcu < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[mcu] : 0u;
    const McuState st = CG_GLOBAL(const McuState, d.mcu_state)[mcu];
    const uint32_t bit = st.info & 31u, left = (st.info >> 5) & 63u;
    e.pred0 = st.pred[0];
    e.pred1 = st.pred[1];
    e.pred2 = st.pred[2];
    e.wlimit = nullptr;
    e.resume = false;
    if (st.info & kMcuDead) {
        // run dry (quirk Q1): zeros from here on, never topped up again
        e.r.buf = 0u;
        e.r.left = 0xffffff00u;
        e.r.next_word = word;
        e.r.pre = 0u;
        e.ref_left = 0u;
        e.fast = false;
        e.wptr = s.win;
        return;
    }
    // bits [p, p + left) of the stream: bit + left is 32 or 64
    const bool two = bit + left > 32u;
    const uint32_t w0 = fetch_word_pf<false>(d, s, word), w1 = two ? fetch_word_pf<false>(d, s, word + 1u) : 0u;
    e.r.buf = (uint64_t(w0) << 32 | w1) << bit;
    e.r.left = left;
    e.r.next_word = word + (two ? 2u : 1u);
    e.r.pre = fetch_word_pf<false>(d, s, e.r.next_word);
    e.ref_left = left;
    const uint32_t rel = e.r.next_word - s.win_base;
    e.fast = fast_tables_usable(d, s) && rel < s.win_len;
    e.wptr = s.win + (e.fast ? rel : 0u);
}

#if defined(__HIPCC__)
// value of lane j of the caller's quad
template <int J>
CG_DEV uint32_t quad_lane(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), J * 0x55, 0xf, 0xf, true));
#else
    return v; // host pass of hipcc: never executed
#endif
}

// value of lane 16 T + (lane >> 2) of the wave
template <int T>
CG_DEV uint32_t wave_lane(uint32_t v, uint32_t lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return uint32_t(__builtin_amdgcn_ds_bpermute(int((16u * T + (lane >> 2)) * 4u), int(v)));
#else
    (void)lane;
    return v; // host pass of hipcc: never executed
#endif
}

// The composite of the wave's 64 current MCUs; every lane of the wave calls
// this together (tests/emul drives the same steps lane by lane).
// wave_slots: the 64 slots the data units came through.
// QUADS = false: every lane stores its own MCU, 16 bytes at a time -- less
// latency (no LDS round trip per row) for launches that leave the vector
// memory path idle anyway (the paired kernel).
// WIDE: every interval of the launch is one MCU (composite_row_from_wave).
template <bool QUADS, bool WIDE = false>
CG_DEV void composite_mcus_422(PixelState &t, const ImageDesc &d, uint8_t *wave_slots, uint32_t lane)
{
    const McuTarget g = mcu_target(t, d);
    if (!QUADS) {
        if (g.whole) {
#pragma unroll
            for (uint32_t row = 0; row < 8; row++) {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++)
                    store_pixels<false>(g.base + size_t(row) * d.out_pitch + q * 16u,
                                 rgba_quad(t.px[q >> 1][row * 2u + (q & 1u)], t.px[2][row * 2u + (q >> 1)] >> ((q & 1u) * 16u),
                                           t.px[3][row * 2u + (q >> 1)] >> ((q & 1u) * 16u)));
            }
        } else if (t.active) {
            composite_edge_mcu(d, t.px, t.mx, t.my);
        }
        pixel_next_mcu(t, d);
        return;
    }
    const uint64_t addr = reinterpret_cast<uint64_t>(g.base);
    const uint32_t lo = uint32_t(addr), hi = uint32_t(addr >> 32), wh = g.whole ? 1u : 0u;
    uint8_t *slot = wave_slots + lane * kDuSlotBytes;
    if (WIDE) {
        // consecutive lanes, consecutive MCUs: rows leave in pieces of 1 KB (composite_row_from_wave)
        uint8_t *from[4] = {
            reinterpret_cast<uint8_t *>(uint64_t(wave_lane<0>(hi, lane)) << 32 | wave_lane<0>(lo, lane)),
            reinterpret_cast<uint8_t *>(uint64_t(wave_lane<1>(hi, lane)) << 32 | wave_lane<1>(lo, lane)),
            reinterpret_cast<uint8_t *>(uint64_t(wave_lane<2>(hi, lane)) << 32 | wave_lane<2>(lo, lane)),
            reinterpret_cast<uint8_t *>(uint64_t(wave_lane<3>(hi, lane)) << 32 | wave_lane<3>(lo, lane)),
        };
        const uint32_t mask = wave_lane<0>(wh, lane) | wave_lane<1>(wh, lane) << 1 | wave_lane<2>(wh, lane) << 2 | wave_lane<3>(wh, lane) << 3;
        if (__builtin_amdgcn_ballot_w64(mask != 0xfu) == 0u) {
#pragma unroll
            for (uint32_t row = 0; row < 8; row++) {
                composite_row_to_slot(t.px, row, slot);
                composite_row_from_wave(d, wave_slots, lane, row, from, 0xfu);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (uint32_t row = 0; row < 8; row++) {
                composite_row_to_slot(t.px, row, slot);
                composite_row_from_wave(d, wave_slots, lane, row, from, mask);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        zero_slot(slot);
        if (t.active && !g.whole)
            composite_edge_mcu(d, t.px, t.mx, t.my);
        pixel_next_mcu(t, d);
        return;
    }
    uint8_t *bases[4] = {
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<0>(hi)) << 32 | quad_lane<0>(lo)),
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<1>(hi)) << 32 | quad_lane<1>(lo)),
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<2>(hi)) << 32 | quad_lane<2>(lo)),
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<3>(hi)) << 32 | quad_lane<3>(lo)),
    };
    const uint32_t whole_mask = quad_lane<0>(wh) | quad_lane<1>(wh) << 1 | quad_lane<2>(wh) << 2 | quad_lane<3>(wh) << 3;
    // (rows stay apart in the schedule: interleaving them only costs registers)
#if CG_EXP == 15 // diagnostic build (wrong picture, the same bytes): an MCU at an even place stores nothing, the odd one
                 // behind it stores its rows twice -- to its neighbour's place and its own, back to back: what the
                 // write path does with the two halves of a 128-byte line arriving together (held MCUs)
    if (__builtin_amdgcn_ballot_w64(whole_mask != 0xfu) == 0u) {
        uint8_t *left[4] = {bases[0] - 64, bases[1] - 64, bases[2] - 64, bases[3] - 64};
#pragma unroll
        for (uint32_t row = 0; row < 8; row++) {
            composite_row_to_slot(t.px, row, slot);
            if (t.mx & 1u) {
                composite_row_from_quad(d, wave_slots, lane, row, left, 0xfu);
                composite_row_from_quad(d, wave_slots, lane, row, bases, 0xfu);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
#endif
    if (__builtin_amdgcn_ballot_w64(whole_mask != 0xfu) == 0u) {
        // the common case, all 64 MCUs inside the output: unconditional stores
#pragma unroll
        for (uint32_t row = 0; row < 8; row++) {
            composite_row_to_slot(t.px, row, slot);
            composite_row_from_quad(d, wave_slots, lane, row, bases, 0xfu);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (uint32_t row = 0; row < 8; row++) {
            composite_row_to_slot(t.px, row, slot);
            composite_row_from_quad(d, wave_slots, lane, row, bases, whole_mask);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    zero_slot(slot);
    if (t.active && !g.whole)
        composite_edge_mcu(d, t.px, t.mx, t.my);
    pixel_next_mcu(t, d);
}

// The whole path for 64 restart intervals of a 4:2:2 image, one per lane:
// entropy decode, IDCT and composite, data unit by data unit.  The
// per-data-unit loop in the entropy decoder is where lanes wait for each other
// (SIMT reconvergence), so that table selectors, quantisers, the IDCT and the
// quad exchange run with a full wave.  Lanes past the last interval stay for
// the exchange.
// ahead: what the wave does on the side for the unit it will decode next (see WindowAhead in kernels.hip); called
// with the number of the data unit that is about to start.
struct NothingAhead {
    CG_DEV void at(uint32_t, uint32_t) {}
    CG_DEV void decoded(uint32_t, uint32_t) {}
};

// RECORDS: d is an image's descriptor of MCUs (ImageDesc::mcu_word): every "interval" one MCU, begun from its record.
template <class AHEAD, bool WIDE = false, bool RECORDS = false>
CG_DEV void decode_wave_fused_422(const ImageDesc &d, const HuffShared &s, uint32_t interval, uint32_t lane, AHEAD &ahead)
{
    // A lane past the image's last interval decodes that last interval once
    // more (it lies in this wave's window) and simply never stores its own
    // MCUs: no divergent region around the main body.
    const bool active = interval < d.total_intervals;
    interval = active ? interval : d.total_intervals - 1u;
    uint8_t *slot = s.du_slots + lane * kDuSlotBytes;
    int16_t *slot16 = reinterpret_cast<int16_t *>(slot);
    zero_slot(slot);

    EntropyState e;
    PixelState t;
    if (RECORDS)
        entropy_init_from_record(e, d, s, interval);
    else
        entropy_init(e, d, s, interval);
    pixel_init(t, d, interval, active);
    __builtin_amdgcn_s_setprio(CG_PRIO_ENTROPY);

#if defined(CG_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define CG_STAMP(acc) do { const uint64_t now_ = __builtin_readcyclecounter(); acc += now_ - tprev; tprev = now_; } while (0)
    uint64_t tprev = __builtin_readcyclecounter(), t_dc = 0, t_ac = 0, t_idct = 0, t_comp = 0;
#else
#define CG_STAMP(acc) do { } while (0)
#endif
    // The four data units of an MCU pass through one loop body (one copy of
    // the IDCT in the instruction stream, bounded register pressure).
    const uint32_t du_total = d.restart_interval * 4u;
#pragma unroll 1
    for (uint32_t du = 0; du < du_total; du++) {
        const uint32_t k = du & 3u;
        const uint32_t comp = k < 2u ? 0u : k - 1u; // Y0 Y1 Cb Cr (wave-uniform)
        ahead.at(du, du_total);
        // Wave priority by phase: a wave in the dense, stall-free phases (IDCT,
        // composite) goes in front of waves in the entropy decode, whose
        // dependent chain leaves most issue slots unused anyway; the dense phases
        // finish sooner and their stores enter the memory system earlier.
        const int32_t dc = entropy_data_unit(e, d, s, comp, slot16);
        CG_STAMP(t_ac);
        ahead.decoded(du, du_total);
        __builtin_amdgcn_s_setprio(CG_PRIO_IDCT);
        pixel_transform(t, d, comp, k, slot, dc);
        CG_STAMP(t_idct);
        if (k == 3u) {
            __builtin_amdgcn_s_setprio(CG_PRIO_COMPOSITE);
            composite_mcus_422<true, WIDE>(t, d, s.du_slots, lane);
            CG_STAMP(t_comp);
        }
        __builtin_amdgcn_s_setprio(CG_PRIO_ENTROPY);
    }
#if defined(CG_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (lane == 0 && d.dc) { // diagnostic build only: per-wave phase cycles into the (otherwise unused) dc buffer
        uint64_t *o = reinterpret_cast<uint64_t *>(d.dc) + size_t(interval / 64u) * 4u;
        o[0] = t_dc;
        o[1] = t_ac;
        o[2] = t_idct;
        o[3] = t_comp;
    }
#endif
}

CG_DEV void decode_wave_fused_422(const ImageDesc &d, const HuffShared &s, uint32_t interval, uint32_t lane)
{
    NothingAhead none;
    decode_wave_fused_422(d, s, interval, lane, none);
}

// stage_after: bit k -- the rows may be staged anew behind data unit k of every MCU (Y0 Y1 Cb Cr); bit 3 always.
// stage_below: ... and are, when some lane has fewer words than that in front of it (stream_wants_rows).
CG_DEV void decode_wave_fused_422_stream(const ImageDesc &d, const HuffShared &s, uint32_t nrows, uint32_t stage_after, uint32_t stage_below,
                                         uint32_t interval, uint32_t lane)
{
    const bool active = interval < d.total_intervals;
    interval = active ? interval : d.total_intervals - 1u;
    uint8_t *slot = s.du_slots + lane * kDuSlotBytes;
    int16_t *slot16 = reinterpret_cast<int16_t *>(slot);
    zero_slot(slot);

    EntropyState e;
    PixelState t;
    stream_lane_init(e, d, s, nrows, interval, lane);
    pixel_init(t, d, interval, active);
    stream_rows_landed();
    __builtin_amdgcn_s_setprio(CG_PRIO_ENTROPY);

    const uint32_t du_total = d.restart_interval * 4u;
#pragma unroll 1
    for (uint32_t du = 0; du < du_total; du++) {
        const uint32_t k = du & 3u;
        const uint32_t comp = k < 2u ? 0u : k - 1u; // Y0 Y1 Cb Cr (wave-uniform)
        const int32_t dc = entropy_data_unit<true>(e, d, s, comp, slot16, lane);
        bool stage = ((stage_after | 8u) >> k & 1u) != 0u && du + 1u < du_total; // (wave-uniform)
        if (stage)
            stage = wave_any(stream_wants_rows(e, d, s, lane, stage_below));
        if (stage)
            stream_restage(e, d, s, nrows, lane); // (lands under this data unit's IDCT)
        __builtin_amdgcn_s_setprio(CG_PRIO_IDCT);
        pixel_transform(t, d, comp, k, slot, dc);
        // (the rows before the stores: whatever waits for a load waits for every store in front of it as well)
        if (stage)
            stream_rows_landed();
        if (k == 3u) {
            __builtin_amdgcn_s_setprio(CG_PRIO_COMPOSITE);
            composite_mcus_422<true>(t, d, s.du_slots, lane);
        }
        __builtin_amdgcn_s_setprio(CG_PRIO_ENTROPY);
    }
}

#endif // __HIPCC__

// ---------------------------------------------------------------------------
// The fused path for the extension layouts: 4:4:4, 4:4:0, 4:2:0
// ---------------------------------------------------------------------------
// The same single pass as decode_wave_fused_422 -- a lane per restart interval, coefficients through the lane's LDS
// slot, samples in registers, nothing but RGBA leaves the chip -- for luma sampled HS x VS (1 or 2 each) against
// chroma: HS * VS + 2 data units per MCU (luma blocks in raster order, Cb, Cr: the reference shader's own order,
// src/huffman.wgsl:148-155), MCUs of 8 HS x 8 VS pixels.  The composite is the reference's finalize pass
// (src/dct.wgsl:257-321; continued to 16-row MCUs as the oracle's orc_finalize_pass states it): the luma sample of
// its block, chroma nearest neighbour.  It replaces the two-kernel route (entropy_samples_kernel +
// composite_generic_kernel) and its round trip of sample records.
// MC: MCUs a lane holds before it composites them -- 1, or 2 for the layouts with 8-pixel MCUs (their rows are 32
// bytes; two neighbours of an interval make the 64-byte segments the write path takes twice as well; a pair never
// straddles two intervals: the last MCU of an odd one is composited alone).
template <int HS, int VS, int MC>
struct LayoutPixels {
    static constexpr int kDus = HS * VS + 2;
    uint32_t px[MC * kDus][16]; // data unit k of held MCU m at [m * kDus + k], 4 samples per word
    uint32_t mx, my;            // the first held MCU
    bool active;
};

// Four pixels of the lane's MCU group: row `row`, 4-pixel group `g` across the group's width (both unrolled: every
// sample word is a register).
template <int HS, int VS, int MC>
CG_DEV Vec4u layout_pixels4(const LayoutPixels<HS, VS, MC> &t, int row, int g)
{
    constexpr int kDus = HS * VS + 2;
    const int m = g / (2 * HS), gg = g % (2 * HS), o = m * kDus;
    const uint32_t yw = t.px[o + (row >> 3) * HS + (gg >> 1)][(row & 7) * 2 + (gg & 1)];
    const int cy = row / VS;
    if (HS == 2) // two chroma samples, each under two pixels
        return rgba_quad(yw, t.px[o + kDus - 2][cy * 2 + (gg >> 1)] >> ((gg & 1) * 16), t.px[o + kDus - 1][cy * 2 + (gg >> 1)] >> ((gg & 1) * 16));
    return rgba_quad4(yw, t.px[o + kDus - 2][cy * 2 + gg], t.px[o + kDus - 1][cy * 2 + gg]);
}

// Row `row` of the lane's MCU group into its slot: 2 HS MC pieces of 16 bytes.
template <int HS, int VS, int MC>
CG_DEV void layout_row_to_slot(const LayoutPixels<HS, VS, MC> &t, int row, uint8_t *slot)
{
    SlotVec *p = reinterpret_cast<SlotVec *>(slot);
#pragma unroll
    for (int g = 0; g < 2 * HS * MC; g++) {
        const Vec4u o = layout_pixels4<HS, VS, MC>(t, row, g);
        p[g] = SlotVec{o.x, o.y, o.z, o.w};
    }
}

// A 16-byte piece of a row of an MCU group.  Where the groups' rows are whole 64-byte segments the stores are
// non-temporal (store_pixels); where they are 32-byte halves of one -- single 8-pixel MCUs; pairs of them in an image
// with an odd number of MCUs a row, where every second MCU row's pairs lie across two segments -- ordinary stores
// (MERGE) have the L2 keep a half until the other has come: a non-temporal half goes to memory as a 32-byte write of
// its own (TCC_EA0_WRREQ of 256 x 1920x1088 4:4:4: 33.6 M, all of 64 bytes, in pairs, 1.1 ms; 67.2 M of 32 as single
// MCUs, 3.2 ms; 1912 across: 16.9 M of 64 + 38.7 M of 32, 1.7 ms -- profiles/r04/NOTES.md).  A template argument:
// chosen per store by a wave-uniform branch, the aligned case lost 7 %; per row of a group it is one branch for four.
template <bool MERGE>
CG_DEV void layout_store(uint8_t *p, const Vec4u &v)
{
    store_pixels<!MERGE>(p, v);
}

// The lane's share of row `row` of its quad's four MCU groups (composite_row_from_quad's counterpart): with 64-byte rows
// (PIECES = 4) lane i stores piece i of each of the four rows; with 32-byte rows (PIECES = 2) lanes 0, 1 store the two
// pieces of one group's row and lanes 2, 3 those of the next, twice -- a wave-wide store then writes whole rows of
// MCU groups instead of 64 separate 16-byte pieces.
template <int PIECES, bool MERGE>
CG_DEV void layout_row_from_quad(const ImageDesc &d, const uint8_t *wave_slots, uint32_t lane, uint32_t row,
                                 uint8_t *const (&bases)[4], uint32_t whole_mask)
{
    const uint32_t quad = lane & ~3u, liq = lane & 3u;
#pragma unroll
    for (uint32_t j = 0; j < uint32_t(PIECES); j++) {
        const uint32_t src = PIECES == 4 ? j : 2u * j + (liq >> 1), piece = PIECES == 4 ? liq : (liq & 1u);
        const SlotVec v = reinterpret_cast<const SlotVec *>(wave_slots + (quad + src) * kDuSlotBytes)[piece];
        uint8_t *base = src == 0u ? bases[0] : (src == 1u ? bases[1] : (src == 2u ? bases[2] : bases[3]));
        if (whole_mask >> src & 1u)
            layout_store<MERGE>(base + size_t(row) * d.out_pitch + piece * 16u, Vec4u{v.x, v.y, v.z, v.w});
    }
}

// ... of MCU groups the output's edge cuts: `rows` = cut_rows_for_piece of the quad's four groups for this lane's piece.
template <int PIECES>
CG_DEV void layout_row_from_quad_cut(const ImageDesc &d, const uint8_t *wave_slots, uint32_t lane, uint32_t row,
                                     uint8_t *const (&bases)[4], uint32_t rows)
{
    const uint32_t quad = lane & ~3u, liq = lane & 3u, inside = cut_row_inside(rows, row);
#pragma unroll
    for (uint32_t j = 0; j < uint32_t(PIECES); j++) {
        const uint32_t src = PIECES == 4 ? j : 2u * j + (liq >> 1), piece = PIECES == 4 ? liq : (liq & 1u);
        const SlotVec v = reinterpret_cast<const SlotVec *>(wave_slots + (quad + src) * kDuSlotBytes)[piece];
        uint8_t *base = src == 0u ? bases[0] : (src == 1u ? bases[1] : (src == 2u ? bases[2] : bases[3]));
        if (inside & (0x80u << (8u * src)))
            layout_store<true>(base + size_t(row) * d.out_pitch + piece * 16u, Vec4u{v.x, v.y, v.z, v.w}); // (cut groups are few: whatever their halves)
    }
}

// MCUs cut by the right / bottom edge of the output inside a 16-byte piece (stores outside it are dropped, like
// textureStore in the reference) or an unaligned pitch -- neither arises with outputs the runtime allocates (whole MCUs,
// 16 pixels each way) --: the owning lane stores them pixel by pixel, each held MCU at its own place (a pair's second MCU
// at the next MCU row's beginning where the first ends its row).
template <int HS, int VS, int MC>
CG_DEV void composite_layout_edge(const LayoutPixels<HS, VS, MC> &t, const ImageDesc &d, uint32_t held = uint32_t(MC))
{
    constexpr int kDus = HS * VS + 2;
    // (rolled loops, the words selected without dynamic register indexing: unrolled, the 64 conversions of a 16 x 16
    // MCU are hoisted over their stores' guards and the kernel spills)
#pragma unroll 1
    for (uint32_t m = 0; m < held; m++) {
        uint32_t mx = t.mx + m, my = t.my;
        if (mx >= d.width_mcus) {
            mx -= d.width_mcus;
            my++;
        }
        const uint32_t x0 = mx * (8u * HS), y0 = my * (8u * VS);
        uint8_t *base = d.out + size_t(y0) * d.out_pitch + size_t(x0) * 4u;
#pragma unroll 1
        for (uint32_t row = 0; row < uint32_t(8 * VS); row++) {
            if (y0 + row >= d.out_h)
                break;
#pragma unroll 1
            for (uint32_t g = 0; g < uint32_t(2 * HS); g++) {
                const uint32_t yb = m * uint32_t(kDus) + (row >> 3) * uint32_t(HS) + (g >> 1), yi = (row & 7u) * 2u + (g & 1u);
                const uint32_t cy = row / uint32_t(VS), ci = HS == 2 ? cy * 2u + (g >> 1) : cy * 2u + g;
                uint32_t yw = 0, cbw = 0, crw = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
#pragma unroll
                    for (uint32_t blk = 0; blk < uint32_t(MC * kDus); blk++) {
                        yw = (i == yi && blk == yb) ? t.px[blk][i] : yw;
                        cbw = (i == ci && blk == m * uint32_t(kDus) + uint32_t(kDus - 2)) ? t.px[blk][i] : cbw;
                        crw = (i == ci && blk == m * uint32_t(kDus) + uint32_t(kDus - 1)) ? t.px[blk][i] : crw;
                    }
                }
                const Vec4u o = HS == 2 ? rgba_quad(yw, cbw >> ((g & 1u) * 16u), crw >> ((g & 1u) * 16u)) : rgba_quad4(yw, cbw, crw);
                const uint32_t x = x0 + g * 4u;
                auto *q = CG_GLOBAL(uint32_t, reinterpret_cast<uint32_t *>(base + size_t(row) * d.out_pitch + size_t(g) * 16u));
                if (x < d.out_w)
                    q[0] = o.x;
                if (x + 1u < d.out_w)
                    q[1] = o.y;
                if (x + 2u < d.out_w)
                    q[2] = o.z;
                if (x + 3u < d.out_w)
                    q[3] = o.w;
            }
        }
    }
}

template <int HS, int VS, int MC>
CG_DEV void layout_init(LayoutPixels<HS, VS, MC> &t, const ImageDesc &d, uint32_t interval, bool active)
{
#pragma unroll
    for (int k = 0; k < MC * (HS * VS + 2); k++)
#pragma unroll
        for (int w = 0; w < 16; w++)
            t.px[k][w] = 0u;
    const uint32_t mcu0 = interval * d.restart_interval;
    t.mx = mcu0 % d.width_mcus;
    t.my = mcu0 / d.width_mcus;
    t.active = active;
}

// One data unit: coefficients out of `slot` (cleared for reuse) and IDCT; place: m * kDus + k, its held MCU and
// its place in it.  One copy of the IDCT in the instruction stream: it leaves its 16 words in the last block, the
// blocks in front of it are moved to their place (see pixel_transform).
template <int HS, int VS, int MC>
CG_DEV void layout_transform(LayoutPixels<HS, VS, MC> &t, const ImageDesc &d, uint32_t comp, uint32_t place, uint8_t *slot, int32_t dc)
{
    constexpr int kBlocks = MC * (HS * VS + 2);
    uint32_t rec[kRetained / 2];
    take_slot(slot, rec);
    idct_data_unit(rec, dc, d.quant[comp], t.px[kBlocks - 1]);
#pragma unroll
    for (int j = 0; j < kBlocks - 1; j++) {
        if (place == uint32_t(j)) {
#pragma unroll
            for (int w = 0; w < 16; w++)
                t.px[j][w] = t.px[kBlocks - 1][w];
            CG_PLACE_MARK("; data unit in place");
        }
    }
}

template <int HS, int VS, int MC>
CG_DEV McuTarget layout_target(const LayoutPixels<HS, VS, MC> &t, const ImageDesc &d)
{
    const uint32_t x0 = t.mx * (8u * HS), y0 = t.my * (8u * VS);
    McuTarget g;
    g.base = d.out + size_t(y0) * d.out_pitch + size_t(x0) * 4u;
    // (all held MCUs in one MCU row, inside the output)
    // (inside what is allocated: mcu_target)
    g.whole = t.active && t.mx + uint32_t(MC) <= d.width_mcus && x0 + 8u * HS * MC <= umax(d.out_w, d.out_pitch / 4u) &&
              y0 + 8u * VS <= umax(d.out_h, d.out_alloc_h) && (d.out_pitch & 15u) == 0u && x0 < d.out_w && y0 < d.out_h;
    return g;
}

template <int HS, int VS, int MC>
CG_DEV uint32_t layout_limit(const LayoutPixels<HS, VS, MC> &t, const ImageDesc &d)
{
    return target_limit(d, t.active && t.mx + uint32_t(MC) <= d.width_mcus, t.mx * (8u * HS), t.my * (8u * VS), 8u * HS * MC, 8u * VS);
}

// MCU pairs (MC = 2: 8-pixel MCUs, a 64-byte row of two halves of 32 bytes) in the cut branch of the composite: each
// half with its own limit and the second with its own place, so that a pair whose second MCU begins the next MCU row --
// one in every two MCU rows wherever a row holds an odd number of MCUs: 1080 pixels across, 360, 1912 -- goes through the
// quad like any other (it was the edge path's, pixel by pixel, its wave waiting; what such images cost beyond that is
// their pairs' place in the 64-byte segments: layout_store).
constexpr uint32_t kPairEdge = 1u << 16; // the group is the edge path's (cut inside a 16-byte piece, or an unaligned pitch)

// One 8-pixel MCU at (x0, y0): rows | pieces << 5 of it that the quad stores; 0: it lies outside; kPairEdge: see above.
template <int VS>
CG_DEV uint32_t pair_half_limit(const ImageDesc &d, uint32_t x0, uint32_t y0)
{
    if (x0 >= d.out_w || y0 >= d.out_h)
        return 0u;
    if ((d.out_pitch & 15u) == 0u && x0 + 8u <= umax(d.out_w, d.out_pitch / 4u) && y0 + 8u * VS <= umax(d.out_h, d.out_alloc_h))
        return uint32_t(8 * VS) | 2u << 5; // (inside what is allocated: mcu_target)
    const uint32_t lim = target_limit(d, true, x0, y0, 8u, 8u * VS);
    return lim ? lim : kPairEdge;
}

// bits 0-7: the limit of the pair's first MCU, 8-15: of its second (each rows | pieces << 5, pieces <= 2), or kPairEdge.
// alone: the pair's first MCU is all there is -- the last MCU of an odd restart interval.
template <int HS, int VS, int MC>
CG_DEV uint32_t pair_limits(const LayoutPixels<HS, VS, MC> &t, const ImageDesc &d, bool whole, bool alone = false)
{
    static_assert(HS == 1 && MC == 2, "pairs of 8-pixel MCUs");
    const uint32_t full = uint32_t(8 * VS) | 2u << 5;
    if (alone)
        return t.active ? pair_half_limit<VS>(d, t.mx * 8u, t.my * (8u * VS)) : 0u;
    if (whole)
        return full | full << 8;
    if (!t.active)
        return 0u;
    if (t.mx + 2u <= d.width_mcus) {
        const uint32_t lim = layout_limit<HS, VS, MC>(t, d);
        if (!lim)
            return kPairEdge;
        const uint32_t rows = lim & 31u, pieces = lim >> 5;
        return (rows | umin(pieces, 2u) << 5) | (pieces > 2u ? (rows | (pieces - 2u) << 5) << 8 : 0u);
    }
    const uint32_t a = pair_half_limit<VS>(d, t.mx * 8u, t.my * (8u * VS)), b = pair_half_limit<VS>(d, 0u, (t.my + 1u) * (8u * VS));
    return (a | b) & kPairEdge ? kPairEdge : a | b << 8;
}

// How far behind its place in the pair's row the second MCU lies (bytes; 0 in one MCU row with the first)
template <int HS, int VS, int MC>
CG_DEV uint32_t pair_second_offset(const LayoutPixels<HS, VS, MC> &t, const ImageDesc &d)
{
    return t.active && t.mx + 2u > d.width_mcus ? uint32_t(8 * VS) * d.out_pitch - t.mx * 32u - 32u : 0u;
}

// The receiving lane's side: `limits` = pair_limits of its quad's four groups; its piece's rows of each (cut_row_inside)
CG_DEV uint32_t pair_rows_for_lane(const uint32_t (&limits)[4], uint32_t liq)
{
    const uint32_t sh = (liq >> 1) * 8u;
    return cut_rows_for_piece(((limits[0] >> sh) & 0xffu) | ((limits[1] >> sh) & 0xffu) << 8 | ((limits[2] >> sh) & 0xffu) << 16 |
                                  ((limits[3] >> sh) & 0xffu) << 24,
                              liq & 1u);
}

// The lane's group is the edge path's: nothing of it went through the quad.
template <int HS, int VS, int MC>
CG_DEV bool layout_is_edge(const LayoutPixels<HS, VS, MC> &t, const ImageDesc &d, bool whole, bool alone = false)
{
    if (!t.active || (whole && !alone))
        return false;
    if constexpr (MC == 2)
        return pair_limits<HS, VS, MC>(t, d, false, alone) == kPairEdge;
    else
        return layout_limit<HS, VS, MC>(t, d) == 0u;
}

template <int HS, int VS, int MC>
CG_DEV void layout_next_group(LayoutPixels<HS, VS, MC> &t, const ImageDesc &d)
{
    t.mx += uint32_t(MC);
#pragma unroll
    for (int m = 0; m < MC; m++) { // (an image one MCU across: a pair is two MCU rows)
        if (t.mx >= d.width_mcus) {
            t.mx -= d.width_mcus;
            t.my++;
        }
    }
}

CG_DEV uint32_t layout_comp_of(uint32_t k, uint32_t luma_dus) { return k < luma_dus ? 0u : k - luma_dus + 1u; }

#if defined(__HIPCC__)
// The composite of the wave's 64 current MCU groups through the lanes' quads (composite_mcus_422's counterpart).
// alone (wave-uniform; pairs): the groups' first MCUs are all there is -- the last MCU of an odd restart interval.
template <int HS, int VS, int MC>
CG_DEV void composite_layout_mcus(LayoutPixels<HS, VS, MC> &t, const ImageDesc &d, uint8_t *wave_slots, uint32_t lane, bool alone = false)
{
    const McuTarget g = layout_target<HS, VS, MC>(t, d);
    const uint64_t addr = reinterpret_cast<uint64_t>(g.base);
    const uint32_t lo = uint32_t(addr), hi = uint32_t(addr >> 32), wh = g.whole ? 1u : 0u;
    uint8_t *bases[4] = {
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<0>(hi)) << 32 | quad_lane<0>(lo)),
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<1>(hi)) << 32 | quad_lane<1>(lo)),
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<2>(hi)) << 32 | quad_lane<2>(lo)),
        reinterpret_cast<uint8_t *>(uint64_t(quad_lane<3>(hi)) << 32 | quad_lane<3>(lo)),
    };
    const uint32_t whole_mask = quad_lane<0>(wh) | quad_lane<1>(wh) << 1 | quad_lane<2>(wh) << 2 | quad_lane<3>(wh) << 3;
    uint8_t *slot = wave_slots + lane * kDuSlotBytes;
    // (pairs, an odd number of MCUs a row or an interval: every second MCU row's, or every second interval's, lie across
    // two 64-byte segments -- layout_store; chosen row by row: two copies of the whole loop took the pairs' kernels beyond
    // their registers)
    const bool across = MC == 2 && ((d.width_mcus | d.restart_interval) & 1u) != 0u;
    if (!alone && __builtin_amdgcn_ballot_w64(whole_mask != 0xfu) == 0u) {
#pragma unroll
        for (int row = 0; row < 8 * VS; row++) {
            layout_row_to_slot<HS, VS, MC>(t, row, slot);
            if (across)
                layout_row_from_quad<2 * HS * MC, true>(d, wave_slots, lane, uint32_t(row), bases, 0xfu);
            else
                layout_row_from_quad<2 * HS * MC, 2 * HS * MC != 4>(d, wave_slots, lane, uint32_t(row), bases, 0xfu);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (MC == 2) {
        // (pairs: some group is cut, outside, or has its second MCU at the next MCU row's beginning)
        // (... or the groups are single MCUs: second halves without a limit store nothing)
        const uint32_t lim = pair_limits<HS, VS, MC>(t, d, g.whole, alone), off = pair_second_offset<HS, VS, MC>(t, d);
        const uint32_t lims[4] = {quad_lane<0>(lim), quad_lane<1>(lim), quad_lane<2>(lim), quad_lane<3>(lim)};
        const uint32_t limits = pair_rows_for_lane(lims, lane & 3u);
        // (the exchange by every lane, then masked: a select would let the compiler move the DPP read under the
        // selecting lanes' EXEC, where the quad's first two lanes -- the ones read -- are off and read as 0)
        const uint32_t second = 0u - ((lane >> 1) & 1u);
        uint8_t *moved[4] = {bases[0] + (quad_lane<0>(off) & second), bases[1] + (quad_lane<1>(off) & second),
                             bases[2] + (quad_lane<2>(off) & second), bases[3] + (quad_lane<3>(off) & second)};
#pragma unroll
        for (int row = 0; row < 8 * VS; row++) {
            layout_row_to_slot<HS, VS, MC>(t, row, slot);
            layout_row_from_quad_cut<2 * HS * MC>(d, wave_slots, lane, uint32_t(row), moved, limits);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // (some group of the wave is cut by the output's edge, or outside: what lies inside in whole pieces the same way)
        // (a whole group -- inside what is allocated, if not inside the extent -- stores all of itself)
        const uint32_t lim = g.whole ? uint32_t(8 * VS) | uint32_t(2 * HS * MC) << 5 : layout_limit<HS, VS, MC>(t, d);
        const uint32_t limits = cut_rows_for_piece(quad_lane<0>(lim) | quad_lane<1>(lim) << 8 | quad_lane<2>(lim) << 16 | quad_lane<3>(lim) << 24,
                                                   2 * HS * MC == 4 ? lane & 3u : lane & 1u);
#pragma unroll
        for (int row = 0; row < 8 * VS; row++) {
            layout_row_to_slot<HS, VS, MC>(t, row, slot);
            layout_row_from_quad_cut<2 * HS * MC>(d, wave_slots, lane, uint32_t(row), bases, limits);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    zero_slot(slot);
    if (layout_is_edge<HS, VS, MC>(t, d, g.whole, alone))
        composite_layout_edge<HS, VS, MC>(t, d, alone ? 1u : uint32_t(MC));
    layout_next_group<HS, VS, MC>(t, d);
}

// The whole path for 64 restart intervals, one per lane (decode_wave_fused_422's counterpart; tests/emul drives the
// same steps lane by lane).  Lanes past the image's last interval decode that last interval once more and never
// store an MCU of their own: they stay for their quad's exchange.  MC = 2: pairs by the MCU's place in its interval;
// the last MCU of an odd interval is composited alone (the second halves store nothing).
// STREAM: the window in its streamed form (decode_wave_fused_422_stream): nrows words of every lane's stream, staged
// anew behind an MCU's last data unit -- stage_after other than 8: behind every data unit -- when a lane has fewer than
// stage_below in front of it.
template <int HS, int VS, int MC, bool STREAM = false>
CG_DEV void decode_wave_fused_layout(const ImageDesc &d, const HuffShared &s, uint32_t interval, uint32_t lane, uint32_t nrows = 0u,
                                     uint32_t stage_after = 8u, uint32_t stage_below = 0u)
{
    constexpr uint32_t kDus = uint32_t(HS * VS + 2), kBlocks = kDus * uint32_t(MC);
    const bool active = interval < d.total_intervals;
    interval = active ? interval : d.total_intervals - 1u;
    uint8_t *slot = s.du_slots + lane * kDuSlotBytes;
    int16_t *slot16 = reinterpret_cast<int16_t *>(slot);
    zero_slot(slot);
    EntropyState e;
    if (STREAM)
        stream_lane_init(e, d, s, nrows, interval, lane);
    else
        entropy_init(e, d, s, interval);
    LayoutPixels<HS, VS, MC> t;
    layout_init<HS, VS, MC>(t, d, interval, active);
    if (STREAM)
        stream_rows_landed();
    __builtin_amdgcn_s_setprio(CG_PRIO_ENTROPY);
    const uint32_t du_total = d.restart_interval * kDus;
    uint32_t place = 0, k = 0; // the data unit's block among the held MCUs', its place inside its MCU (wave-uniform)
#pragma unroll 1
    for (uint32_t du = 0; du < du_total; du++) {
        const uint32_t comp = layout_comp_of(k, uint32_t(HS * VS));
        const int32_t dc = STREAM ? entropy_data_unit<true>(e, d, s, comp, slot16, lane) : entropy_data_unit(e, d, s, comp, slot16);
        bool stage = STREAM && (k == kDus - 1u || stage_after != 8u) && du + 1u < du_total; // (wave-uniform)
        if (stage)
            stage = wave_any(stream_wants_rows(e, d, s, lane, stage_below));
        if (stage)
            stream_restage(e, d, s, nrows, lane); // (lands under this data unit's IDCT)
        __builtin_amdgcn_s_setprio(CG_PRIO_IDCT);
        layout_transform<HS, VS, MC>(t, d, comp, place, slot, dc);
        if (stage)
            stream_rows_landed(); // (in front of the composite's stores)
        k = k == kDus - 1u ? 0u : k + 1u;
        // (pairs, an odd restart interval: its last MCU alone)
        const bool alone = MC == 2 && du + 1u == du_total && place != kBlocks - 1u;
        if (place == kBlocks - 1u || alone) {
            __builtin_amdgcn_s_setprio(CG_PRIO_COMPOSITE);
            composite_layout_mcus<HS, VS, MC>(t, d, s.du_slots, lane, alone);
            place = 0;
        } else {
            place++;
        }
        __builtin_amdgcn_s_setprio(CG_PRIO_ENTROPY);
    }
}
#endif // __HIPCC__

} // namespace compeg
