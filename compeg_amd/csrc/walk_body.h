// Per-lane bodies of walk_mcus_422_kernel, the first kernel of the walk + lane-per-MCU route (kernels_body.h, "the walk
// + lane-per-MCU route"): a lane per restart interval finds where the interval's MCUs begin and what the DC predictions
// are there, and writes that down for the second kernel (ImageDesc::mcu_word / mcu_state).
//
// A lone wave issues an instruction every four cycles at best, whatever its kind, and a launch that takes this route has
// about one wave of intervals per SIMD: the walk is bound by the instructions of its symbol loop.  So the loop is the
// cooperative kernel's hand-written one (coop_body.h: chase_run_lean) -- the walk tables, up to two symbols per 32-bit
// entry, 32 instructions a step, no values -- over the streamed form of the window (kernels_body.h: row j of a wave's
// rows holds word j behind every lane's own position), an MCU at a time: the lanes of a wave meet again at an MCU's end,
// not at every data unit's.  What the loop leaves out is done behind it with all lanes busy: the four DC differences
// of the MCU (walk_dc_pass: the direct DC tables at the data units' starts the loop has listed), quirk Q1's test at
// each of them, and whether the loop stayed inside the staged rows.  An MCU that fails any of this is decoded once
// more from its start by the reference's reader (walk_slow_mcu): entropy_data_unit, words from global memory.
#pragma once

#include "coop_body.h"

namespace compeg {

#if defined(CG_EMUL_STATS)
struct WalkStats {
    unsigned long mcus, lean_mcus, slow_mcus, slow_q1, slow_rows, slow_long_dc, dead_mcus, restages, long_codes, slow_cut;
};
inline WalkStats g_walk_stats{};
#define CG_WALK_COUNT(field) (++g_walk_stats.field)
#else
#define CG_WALK_COUNT(field) ((void)0)
#endif

// A lane's list: 16-byte entries {word A, state, this data unit's AC pair table, the next one's DC table}, one for where a
// chunk of `g` MCUs begins and one per data unit of the chunk.
constexpr uint32_t kWalkMaxChunk = 16;   // MCUs a lane walks before the wave looks at what it found
constexpr uint32_t walk_list_bytes(uint32_t g) { return 16u * (4u * g + 1u); }
constexpr uint32_t kWalkRowBytes = uint32_t(kWave) * 4u;
constexpr uint32_t kWalkHostRowBias = 0x10000u; // (host build: an entry's first word is a row index, -1 .. , plus this)

// Wave-uniform: the tables and what the data units of an MCU select of them.
struct WalkTabs {
    const uint32_t *walk;    // the walk tables (LDS, 32-byte aligned), or null: every MCU goes through walk_slow_mcu
    const uint16_t *dc_fast; // the two direct DC tables (LDS)
    uint32_t names[4][2];    // entry j = 1..4 of a list, words 2 and 3: data unit j - 1's AC pair table, data unit j's DC table (by name: coop_body.h)
    uint32_t first_dc_name;  // data unit 0's DC table
    uint32_t dcf[4];         // which direct DC table data unit k uses
    uint32_t dc_off[3], ac_off[3]; // per component: offset of its L1 tables (codes longer than a table's prefix)
    int32_t zero_diff[3];    // what a DC code decodes to once the reference's reader has run dry (ImageDesc::zero_du)
    uint32_t l1sel;          // which L1 table the codes of each eighth of the walk tables belong to (coop_body.h: chase_run_lean)
    uint32_t zrl;
    bool standard;           // COMPEG_PARSE_STANDARD_ENTROPY: the reader is topped up in front of DC codes too -- no quirk Q1
};

CG_DEV void walk_tabs(const ImageDesc &d, const HuffShared &s, const uint32_t *walk_lds, WalkTabs &w)
{
    CoopTables t;
    coop_tables(d, s, t);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t walk_base = uint32_t(reinterpret_cast<uintptr_t>(walk_lds));
#else
    const uint32_t walk_base = 0u;
#endif
    w.walk = t.walk_ok ? walk_lds : nullptr;
    w.dc_fast = t.dc_fast;
    for (uint32_t j = 1; j <= 4u; j++) {
        const uint32_t k = j - 1u;
        w.names[j - 1u][0] = walk_base + walk_pairs_name((t.walk_acsel >> (8u * k)) & 0xffu);
        w.names[j - 1u][1] = walk_base + walk_dc_name((t.walk_dcsel >> (8u * (j & 3u))) & 0xffu);
    }
    w.first_dc_name = walk_dc_name(t.walk_dcsel & 0xffu);
    for (uint32_t k = 0; k < 4u; k++)
        w.dcf[k] = d.dc_fast_table[comp_of_k(k)] & 1u;
    for (uint32_t c = 0; c < 3u; c++) {
        w.dc_off[c] = t.dc_off[c];
        w.ac_off[c] = t.ac_off[c];
        w.zero_diff[c] = d.zero_du[c][0];
    }
    const uint32_t l1dc0 = 2u * (t.walk_ids & 1u), l1dc1 = 2u * ((t.walk_ids >> 2) & 1u);
    w.l1sel = 0x1u | 0x1u << 4 | 0x3u << 8 | 0x3u << 12 | 0x1u << 16 | l1dc0 << 20 | 0x3u << 24 | l1dc1 << 28;
    w.zrl = t.zrl;
    w.standard = t.standard;
}

// A lane between two MCUs.
struct WalkLane {
    uint32_t wa;       // stream word A: its LDS byte address (GPU) / its row + kWalkHostRowBias (host); one row above the column for an aligned position at row 0
    uint32_t T;        // 32 - the bits of A that are consumed (0: the position is B's first bit); zig-zag state 0
    uint32_t ent;      // the walk-table entry at the position (the coming data unit's DC code and what follows it)
    uint32_t row0;     // index of the scan word that row 0 of the lane's column holds
    uint32_t valid;    // rows of the column that hold words of the scan (the rest: whatever lies behind it)
    uint32_t ref_left; // the reference reader's `left` at the position (quirk Q1)
    int32_t pred[3];
    // not walking: the reference's reader has run dry (quirk Q1: zeros from here on), or the lane's position has no
    // staged rows under it (after an MCU that went through walk_slow_mcu, or at the scan's end) -- at_word / at_bit say where
    bool dead, parked;
    uint32_t at_word, at_bit;
    bool active;        // false: a lane past the image's last interval
};

// Row of word A inside the lane's column (-1: above it).
CG_DEV int32_t walk_row_of(const HuffShared &s, uint32_t lane, uint32_t wa)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return int32_t(wa - uint32_t(reinterpret_cast<uintptr_t>(s.win + lane))) >> 8; // (rows are kWalkRowBytes apart)
#else
    (void)s;
    (void)lane;
    return int32_t(wa) - int32_t(kWalkHostRowBias);
#endif
}
CG_DEV uint32_t walk_wa_of(const HuffShared &s, uint32_t lane, int32_t row)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return uint32_t(reinterpret_cast<uintptr_t>(s.win + lane)) + uint32_t(row) * kWalkRowBytes;
#else
    (void)s;
    (void)lane;
    return uint32_t(int32_t(kWalkHostRowBias) + row);
#endif
}
// Position of a list entry / of the lane {wa, T}, in bits behind row 0's first (never negative: the row above the
// column goes with a shift of 0).
CG_DEV uint32_t walk_pos_of(const HuffShared &s, uint32_t lane, uint32_t wa, uint32_t T)
{
    return uint32_t(32 * walk_row_of(s, lane, wa) + 32 - int32_t(int16_t(T & 0xffffu)));
}
// The 32 stream bits at position p (rows hold the words most significant bit first).
CG_DEV uint32_t walk_bits_at(const HuffShared &s, uint32_t lane, uint32_t p)
{
    const uint32_t row = p >> 5;
    const uint32_t w0 = s.win[row * uint32_t(kWave) + lane], w1 = s.win[(row + 1u) * uint32_t(kWave) + lane];
    return uint32_t(((uint64_t(w0) << 32 | w1) << (p & 31u)) >> 32);
}
// Where the lane stands in the scan: word and bit.
CG_DEV void walk_where(const WalkLane &l, const HuffShared &s, uint32_t lane, uint32_t &word, uint32_t &bit)
{
    const uint32_t p = walk_pos_of(s, lane, l.wa, l.T);
    word = l.parked || l.dead ? l.at_word : l.row0 + (p >> 5);
    bit = l.parked || l.dead ? l.at_bit : p & 31u;
}

// The lane on its rows at bit `p` behind row 0's first, if three rows of scan lie under it there (the walk reads A B C
// at every step); else parked at that place.
CG_DEV void walk_place(WalkLane &l, const HuffShared &s, const WalkTabs &t, uint32_t lane, uint32_t p, bool lookup)
{
    const uint32_t bit = p & 31u;
    const int32_t row = int32_t(p >> 5) - (bit ? 0 : 1);
    l.at_word = l.row0 + (p >> 5);
    l.at_bit = bit;
    l.parked = row + 2 >= int32_t(l.valid) || t.walk == nullptr;
    if (l.parked)
        return;
    l.wa = walk_wa_of(s, lane, row);
    l.T = (32u - bit) & 31u;
    if (lookup)
        l.ent = walk_lookup(t.walk, t.first_dc_name, walk_bits_at(s, lane, p));
}

// The lane's rows anew: row j := scan word (position's word + j), most significant bit first; the lane goes on at the
// same position.  (The words behind an image's last are readable -- runtime.cpp pads its buffers -- and never used.)
CG_DEV void walk_restage(WalkLane &l, const ImageDesc &d, const HuffShared &s, const WalkTabs &t, uint32_t nrows, uint32_t lane)
{
    uint32_t word, bit;
    walk_where(l, s, lane, word, bit);
    const uint32_t first = umin(word, d.nwords);
    uint32_t *rows = const_cast<uint32_t *>(s.win);
#if defined(__HIP_DEVICE_COMPILE__)
    auto *words = CG_GLOBAL(const uint32_t, d.words);
    const uint32_t whole = nrows & ~3u; // (four rows a load: stream_stage_rows)
#pragma unroll 8
    for (uint32_t j = 0; j < whole; j += 4u) {
        const QuadWords q = *reinterpret_cast<const __attribute__((address_space(1))) QuadWords *>(words + first + j);
        CG_LDS(uint32_t, rows)[(j + 0u) * uint32_t(kWave) + lane] = bswap32(q.x);
        CG_LDS(uint32_t, rows)[(j + 1u) * uint32_t(kWave) + lane] = bswap32(q.y);
        CG_LDS(uint32_t, rows)[(j + 2u) * uint32_t(kWave) + lane] = bswap32(q.z);
        CG_LDS(uint32_t, rows)[(j + 3u) * uint32_t(kWave) + lane] = bswap32(q.w);
    }
    for (uint32_t j = whole; j < nrows; j++)
        CG_LDS(uint32_t, rows)[j * uint32_t(kWave) + lane] = bswap32(words[first + j]);
#else
    for (uint32_t j = 0; j < nrows; j++)
        rows[j * uint32_t(kWave) + lane] = first + j < d.nwords ? bswap32(d.words[first + j]) : 0xfeedf00du;
#endif
    l.row0 = first;
    l.valid = umin(nrows, d.nwords - first);
    if (l.dead)
        return;
    if (word > d.nwords) { // (behind the scan's end: the reference's reader goes on there, reading zeros)
        l.parked = true;
        l.at_word = word;
        l.at_bit = bit;
        return;
    }
    const bool was_parked = l.parked;
    walk_place(l, s, t, lane, bit, was_parked);
}

// The reference reader at bit `bit` of scan word `word` with `left` bits in its buffer (also a fast-mode state:
// kernels_body.h, entropy_init_from_record).
CG_DEV void entropy_state_at(EntropyState &e, const ImageDesc &d, const HuffShared &s, uint32_t word, uint32_t bit, uint32_t left)
{
    const bool two = bit + left > 32u;
    const uint32_t w0 = fetch_word_pf<false>(d, s, word), w1 = two ? fetch_word_pf<false>(d, s, word + 1u) : 0u;
    e.r.buf = (uint64_t(w0) << 32 | w1) << bit;
    e.r.left = left;
    e.r.next_word = word + (two ? 2u : 1u);
    e.r.pre = fetch_word_pf<false>(d, s, e.r.next_word);
    e.ref_left = left;
    e.fast = false;
    e.resume = false;
    e.wptr = e.wlimit = s.win;
}

// The lane at its interval's start (no rows yet: walk_restage puts it there).
CG_DEV void walk_lane_init(WalkLane &l, const ImageDesc &d, uint32_t interval, bool active)
{
    l.wa = l.T = l.ent = 0u;
    l.row0 = l.valid = 0u;
    l.ref_left = 32u;
    l.pred[0] = l.pred[1] = l.pred[2] = 0;
    l.dead = false;
    l.parked = true;
    l.at_word = interval < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[interval] : 0u;
    l.at_bit = 0u;
    l.active = active;
}

// The MCU's record, in front of its first data unit.
CG_DEV void walk_record(const WalkLane &l, const ImageDesc &d, const HuffShared &s, uint32_t lane, uint32_t mcu)
{
    uint32_t word, bit;
    walk_where(l, s, lane, word, bit);
    McuState st;
    st.info = l.dead ? kMcuDead : bit | (umin(l.ref_left, 63u) << 5);
    st.pred[0] = l.pred[0];
    st.pred[1] = l.pred[1];
    st.pred[2] = l.pred[2];
    CG_GLOBAL(uint32_t, d.mcu_word)[mcu] = word;
    CG_GLOBAL(McuState, d.mcu_state)[mcu] = st;
}

// The lists' table names (every lane's own copy: the loop reads them beside its entry) -- once per wave.
CG_DEV void walk_prepare_list(uint32_t *list, const WalkTabs &t, uint32_t g)
{
    // (entry j = 4 m + k + 1: data unit k's AC pair table, the DC table of the data unit behind it)
    for (uint32_t m = 0; m < g; m++) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            list[16u * m + 4u * (k + 1u) + 2u] = t.names[k][0];
            list[16u * m + 4u * (k + 1u) + 3u] = t.names[k][1];
        }
    }
    list[2] = 0u;
    list[3] = t.names[3][1]; // (data unit 0's DC table: a long DC code right at the chunk's start)
}

// `ndus` data units (whole MCUs) of the lanes with `go`: from {l.wa, l.T, l.ent} through the walk tables; entry j = 1..ndus
// of the lane's list := {word A, state} at the end of data unit j - 1.  Returns how many of them the lane completed (fewer
// than asked for: it met something the loop does not do -- a DC category above 15: ImageDesc::mcu_ok leaves none).
CG_DEV uint32_t walk_mcu_lean(WalkLane &l, const ImageDesc &d, const HuffShared &s, const WalkTabs &t, uint32_t *list, bool go, uint32_t ndus,
                              uint32_t lane)
{
    constexpr uint32_t kStMask = 0xffu << kWalkStShift, kKeep = kStMask | 31u;
    constexpr uint32_t kEndAbove = (64u << kWalkStShift) - 1u, kNearAbove = (kWalkNear << kWalkStShift) - 1u;
    uint32_t completed = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    const uint32_t walk_base = uint32_t(reinterpret_cast<uintptr_t>(t.walk));
    uint32_t alive = go ? 1u : 0u, bad_lane = 0u;
    const uint32_t l1sel = t.l1sel;
    const uint32_t lb = uint32_t(reinterpret_cast<uintptr_t>(list));
    uint32_t lpa = lb + 16u, wa = l.wa, T = l.T, ent = l.ent;
    const uint32_t lpmax = lb + 16u * (1u + ndus);
    if (__builtin_amdgcn_ballot_w64(alive != 0u) != 0u) {
        asm volatile(
            "s_mov_b64 s[74:75], exec\n\t"
            "v_cmp_ne_u32 vcc, 0, %[alive]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execz 4f\n\t"
            "v_mov_b32 v50, %[wa]\n\t"
            "v_mov_b32 v51, %[T]\n\t"
            "ds_read2st64_b32 v[42:43], v50 offset1:1\n\t"        // A, B (a row apart)
            "ds_read_b32 v44, v50 offset:512\n\t"                 // C
            "ds_read_b32 v56, v50 offset:768\n\t"                 // D: what C becomes when the position leaves A
            "ds_read_b64 v[48:49], %[lp] offset:8\n\t"            // this data unit's AC pair table, the next one's DC table
            "ds_read_b64 v[58:59], %[lp] offset:24\n\t"           // ... and the next data unit's two
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 v57, %[singles], v48\n"
            // A step waits for one LDS read only, the table entry's: the stream words and the table names it needs were
            // asked for a step ahead (D, the next entry's names) and move up through registers.  In flight at the top, in
            // issue order: the entry's write, the table read, D, the names -- the first two are waited for there, the
            // last two in the shadow of the next table read.
            "1:\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_pk_add_u16 v51, v51, %[ent]\n\t"                   // bits off the shift, advance onto the zig-zag state
            "v_alignbit_b32 v41, v42, v43, v51\n\t"               // (the shift is taken modulo 32)
            "v_alignbit_b32 v45, v43, v44, v51\n\t"
            "v_cmp_gt_i16 vcc, 0, v51\n\t"                        // the position has left A
            "v_cmp_lt_u32 s[72:73], %[endabove], v51\n\t"         // the data unit is complete: a DC code comes next
            "v_cmp_lt_u32 s[82:83], %[nearabove], v51\n\t"        // the next symbol could complete it: one at a time
            "v_cndmask_b32 v41, v41, v45, vcc\n\t"                // the next 32 stream bits
            "v_cndmask_b32_e64 v46, v48, v57, s[82:83]\n\t"       // (v57: the singles table beside v48's pairs)
            "v_cndmask_b32_e64 v46, v46, v49, s[72:73]\n\t"       // the table's name: address, shift in the low bits
            "v_bfe_u32 v45, v41, v46, 11\n\t"
            "v_and_b32 v52, 0xffffffe0, v46\n\t"
            "v_lshl_add_u32 v52, v45, 2, v52\n\t"
            "v_cmp_eq_u32 s[76:77], 0, %[ent]\n\t"                // lanes that met a long code
            "ds_write_b64 %[lp], v[50:51]\n\t"                    // (final when the data unit ends here)
            "ds_read_b32 %[ent], v52\n\t"
            // ---- under that read: move on in the stream, in the list; who goes on
            "v_cndmask_b32_e64 v40, 0, %[rowstep], vcc\n\t"
            "v_add_u32 v50, v50, v40\n\t"
            "v_cndmask_b32_e64 v40, %[keep], 31, s[72:73]\n\t"
            "v_and_b32 v51, v51, v40\n\t"
            "v_cndmask_b32_e64 v40, 0, 16, s[72:73]\n\t"
            "v_add_u32 %[lp], %[lp], v40\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"                            // D and the names of the step before are there
            "v_cndmask_b32 v42, v42, v43, vcc\n\t"                // A B C <- B C D
            "v_cndmask_b32 v43, v43, v44, vcc\n\t"
            "v_cndmask_b32 v44, v44, v56, vcc\n\t"
            "v_cndmask_b32_e64 v48, v48, v58, s[72:73]\n\t"       // the names <- the next entry's
            "v_cndmask_b32_e64 v49, v49, v59, s[72:73]\n\t"
            "ds_read_b32 v56, v50 offset:768\n\t"
            "ds_read_b64 v[58:59], %[lp] offset:24\n\t"
            "v_add_u32 v57, %[singles], v48\n\t"
            "v_cmp_ge_u32 s[78:79], %[lp], %[lpmax]\n\t"          // the MCU's four data units are complete
            "s_andn2_b64 exec, exec, s[78:79]\n\t"
            "s_and_b64 s[76:77], s[76:77], exec\n\t"              // (sets SCC: some walking lane met a long code)
            "s_cbranch_scc1 3f\n\t"
            "s_cbranch_execnz 1b\n\t"
            "s_branch 5f\n"
            "3:\n\t"
            // ---- (rare) lanes s[76:77] met a code longer than their table's prefix: the reference's two-level tables
            // decide, one symbol, and the entry is made here (coop_body.h: chase_run_lean)
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b64 s[86:87], exec\n\t"                        // the walking lanes
            "s_mov_b64 exec, s[76:77]\n\t"
            "v_add_u32 v40, -4, %[lp]\n\t"
            "ds_read_b32 v40, v40\n\t"
            "v_lshrrev_b32 v52, 21, v51\n\t"
            "v_cmp_eq_u32 vcc, 0, v52\n\t"
            "v_cndmask_b32_e64 v52, 0, 1, vcc\n\t"                // 1: a DC code
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cndmask_b32 v40, v46, v40, vcc\n\t"
            "v_subrev_u32 v40, %[walkbase], v40\n\t"
            "v_bfe_u32 v40, v40, 12, 3\n\t"                       // which eighth of the walk tables
            "v_lshlrev_b32 v40, 2, v40\n\t"
            "v_lshrrev_b32_e64 v45, v40, %[l1sel]\n\t"
            "v_and_b32 v45, 15, v45\n\t"                          // the L1 table of that walk table's codes
            "v_lshrrev_b32 v47, 24, v41\n\t"
            "v_lshl_add_u32 v45, v45, 8, v47\n\t"
            "v_lshl_add_u32 v45, v45, 1, %[l1base]\n\t"
            "ds_read_u16 v45, v45\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_cmp_lt_u32 vcc, 0x7fff, v45\n\t"                   // a delegate: the second level
            "s_and_saveexec_b64 s[88:89], vcc\n\t"
            "s_cbranch_execz 6f\n\t"
            "v_and_b32 v47, 0x7fff, v45\n\t"
            "v_bfe_u32 v40, v41, 16, 8\n\t"
            "v_add_u32 v47, v47, v40\n\t"
            "v_mov_b32 v45, 0\n\t"
            "v_cmp_gt_u32 vcc, %[l2n], v47\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "v_lshl_add_u32 v47, v47, 1, %[l2base]\n\t"
            "ds_read_u16 v45, v47\n\t"
            "s_waitcnt lgkmcnt(0)\n"
            "6:\n\t"
            "s_mov_b64 exec, s[76:77]\n\t"
            "v_lshrrev_b32 v40, 8, v45\n\t"                       // code length
            "v_and_b32 v47, 0xff, v45\n\t"                        // symbol
            "v_and_b32 v53, 15, v47\n\t"                          // AC: magnitude bits,
            "v_and_b32 v54, 31, v40\n\t"
            "v_add_u32 v54, v54, v53\n\t"                         // size,
            "v_lshrrev_b32 v55, 4, v47\n\t"
            "v_add_u32 v55, 1, v55\n\t"                           // advance: run + 1,
            "v_cmp_eq_u32 vcc, 0xf0, v47\n\t"
            "v_cndmask_b32_e64 v55, v55, %[zrl], vcc\n\t"         // ZRL,
            "v_cmp_eq_u32 vcc, 0, v47\n\t"
            "v_cndmask_b32_e64 v55, v55, 64, vcc\n\t"             // end-of-block
            "v_lshrrev_b32 v53, 5, v54\n\t"
            "v_or_b32 v55, v55, v53\n\t"
            "v_and_b32 v54, 31, v54\n\t"
            "v_add_u32 v53, v40, v47\n\t"                         // DC: size = length + category
            "v_cmp_lt_u32 s[88:89], 15, v47\n\t"
            "v_cmp_lt_u32 vcc, 31, v53\n\t"
            "s_or_b64 s[88:89], s[88:89], vcc\n\t"                // ... which only a hostile table makes that large
            "v_cmp_eq_u32 vcc, 1, v52\n\t"
            "v_cndmask_b32 v54, v54, v53, vcc\n\t"
            "v_cndmask_b32_e64 v55, v55, 1, vcc\n\t"
            "s_and_b64 s[88:89], s[88:89], vcc\n\t"
            "v_sub_u32 v53, 0, v54\n\t"
            "v_and_b32 v53, 0xffff, v53\n\t"
            "v_lshl_or_b32 v53, v54, 16, v53\n\t"
            "v_lshl_or_b32 %[ent], v55, 21, v53\n\t"
            "v_cndmask_b32_e64 %[bad], %[bad], 1, s[88:89]\n\t"
            "s_andn2_b64 s[86:87], s[86:87], s[88:89]\n\t"
            "s_mov_b64 exec, s[86:87]\n\t"                        // the walking lanes again, less those
            "s_cbranch_execnz 1b\n\t"
            "s_branch 5f\n"
            "4:\n"
            "5:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b64 exec, s[74:75]\n\t"
            "v_cmp_ne_u32 vcc, 0, %[alive]\n\t"
            "v_cndmask_b32 %[wa], %[wa], v50, vcc\n\t"            // (the others keep theirs)
            "v_cndmask_b32 %[T], %[T], v51, vcc\n\t"
            : [lp] "+v"(lpa), [ent] "+v"(ent), [T] "+v"(T), [wa] "+v"(wa), [bad] "+v"(bad_lane)
            : [alive] "v"(alive), [endabove] "s"(kEndAbove), [nearabove] "s"(kNearAbove), [singles] "v"(kWalkSinglesName), [keep] "v"(kKeep),
              [lpmax] "v"(lpmax), [walkbase] "s"(walk_base), [l1sel] "s"(l1sel), [rowstep] "v"(kWalkRowBytes),
              [l1base] "s"(uint32_t(reinterpret_cast<uintptr_t>(s.l1))), [l2base] "s"(uint32_t(reinterpret_cast<uintptr_t>(s.l2))),
              [l2n] "s"(d.l2_entries), [zrl] "v"(t.zrl)
            : "memory", "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
              "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s82",
              "s83", "s86", "s87", "s88", "s89");
    }
    if (go) {
        l.wa = wa;
        l.T = T;
        l.ent = ent;
        completed = bad_lane ? 0u : (lpa - lb) / 16u - 1u;
    }
#else
    if (go) {
        // (the same walk, a step at a time: coop_body.h, chase_run_lean's host branch -- rows instead of a window)
        int32_t row = walk_row_of(s, lane, l.wa); // of word A
        uint32_t T = l.T, ent = l.ent, j = 1u;
        bool bad = false;
        while (j < 1u + ndus) {
            if (row + 2 >= int32_t(l.valid)) // (the GPU walks on through whatever lies there; walk_chunk_pass's test sees where it has been)
                break;
            const uint32_t p_now = uint32_t(32 * row + 32 - int32_t(T & 31u));
            if (ent == 0u) {
                CG_WALK_COUNT(long_codes);
                const uint32_t st_ = T >> kWalkStShift, comp = comp_of_k((j - 1u) & 3u); // (a DC code: of the data unit the walk is about to begin)
                const uint32_t cur = walk_bits_at(s, lane, p_now);
                if (st_ == 0u) {
                    const uint32_t e2 = lut_lookup<true>(d, s, t.dc_off[comp], cur);
                    const uint32_t len = e2 >> 8, cat = e2 & 0xffu;
                    if (cat > 15u || len + cat > 31u) {
                        bad = true;
                        break;
                    }
                    ent = walk_pack(len + cat, len + cat, 1u);
                } else {
                    const uint32_t fe = fast_entry(lut_lookup<true>(d, s, t.ac_off[comp], cur), t.zrl);
                    ent = walk_pack((fe >> 4) & 31u, (fe >> 4) & 31u, fe >> 9);
                }
            }
            const uint32_t T1 = ((T + (ent & 0xffff0000u)) & 0xffff0000u) | ((T + ent) & 0xffffu);
            const int32_t sn = int32_t(int16_t(T1 & 0xffffu));
            const uint32_t p_next = uint32_t(32 * row + 32 - sn);
            const bool du_end = T1 > kEndAbove, near = T1 > kNearAbove;
            list[4u * j] = uint32_t(int32_t(kWalkHostRowBias) + row);
            list[4u * j + 1u] = T1;
            const uint32_t name = du_end ? list[4u * j + 3u] : list[4u * j + 2u] + (near ? kWalkSinglesName : 0u);
            ent = walk_lookup(t.walk, name, walk_bits_at(s, lane, p_next));
            row += sn < 0 ? 1 : 0;
            T = T1 & (du_end ? 31u : kKeep);
            j += du_end ? 1u : 0u;
        }
        l.wa = uint32_t(int32_t(kWalkHostRowBias) + row);
        l.T = T;
        l.ent = ent;
        completed = bad ? 0u : j - 1u;
        // (entries the walk did not reach: as the GPU's run over the rows' end leaves them -- beyond the staged rows)
        for (; j < 1u + ndus; j++) {
            list[4u * j] = uint32_t(int32_t(kWalkHostRowBias) + int32_t(l.valid));
            list[4u * j + 1u] = 0u;
        }
    }
#endif
    return completed;
}

// Behind the loop, MCU `m` of the chunk, for a lane that walked and has been fine up to here: did the walk stay inside
// the staged rows up to this MCU's end (it read three rows at every step, A B C, and its position only ever moved on:
// where the MCU ended says it all), the MCU's four DC differences from the data units' starts in the list, and the
// reference reader's `left` at each (quirk Q1).  k is wave-uniform, so are tables and components.  False: this MCU has
// to be decoded by walk_slow_mcu from where it began -- the lane's ref_left and predictions still stand there; true: they
// stand at the MCU's end.
CG_DEV bool walk_mcu_pass(WalkLane &l, const ImageDesc &d, const HuffShared &s, const WalkTabs &t, const uint32_t *list, uint32_t m, uint32_t lane)
{
    const uint32_t *at = list + 16u * m; // entry 4 m
    const uint32_t Te = at[4u * 4u + 1u];
    const int32_t end_row = walk_row_of(s, lane, at[4u * 4u]) + (int32_t(int16_t(Te & 0xffffu)) < 0 ? 1 : 0);
    if (end_row + 2 >= int32_t(l.valid)) {
        CG_WALK_COUNT(slow_rows);
        return false;
    }
    uint32_t ref_left = l.ref_left;
    int32_t pred[3] = {l.pred[0], l.pred[1], l.pred[2]};
    bool fine = true;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t comp = comp_of_k(k);
        const uint32_t p = walk_pos_of(s, lane, at[4u * k], at[4u * k + 1u]);
        if (k) {
            const uint32_t last = (at[4u * k + 1u] >> kWalkLastShift) & 31u;
            ref_left = 32u + ((last - p) & 31u) - last;
        }
        const uint32_t cur = walk_bits_at(s, lane, p);
        uint32_t side = t.dc_fast[t.dcf[k] * kDcFastEntries + (cur >> (32u - kDcFastBits))];
        if (__builtin_expect(side == kFastEscape, 0)) {
            // a DC code longer than the direct table's prefix
            const uint32_t e2 = lut_lookup<false>(d, s, t.dc_off[comp], cur);
            side = (1u << 9) | (((e2 >> 8) + (e2 & 15u)) << 4) | (e2 & 15u);
            CG_WALK_COUNT(slow_long_dc);
        }
        const uint32_t cat = side & 15u, n = (side >> 4) & 31u;
        fine = fine && (t.standard || n <= ref_left); // (more than the reference's reader has left: it runs dry here, quirk Q1)
        // The reference's reader is not topped up in front of a DC code: with fewer than 32 bits left it looks the code up
        // in those bits and zeros behind them (kernels_body.h: fast_dc), not in the stream's.  A code that fits the bits
        // it has reads the same either way, one that does not is caught above -- but bits that are no code at all (n = 0:
        // a corrupt scan, an interval read on into its padding) may be one once zeros stand behind the first few of
        // them (codes are handed out from the low end), of more bits than the reader has: the slow road finds out.
#if !(defined(COMPEG_LAB) && defined(CG_NO_CUT_CHECK)) // (laboratory builds may leave the test out: what it costs)
        if (__builtin_expect(n == 0u && ref_left < 32u && !t.standard, 0)) { // (32 bits or more left: the reader looks at the same bits)
            fine = false;
            CG_WALK_COUNT(slow_cut);
        }
#endif
        const int32_t sx = signed_field(cur, n, cat);
        const uint32_t diff = uint32_t(sx) + (((0xffffffffu << cat) ^ uint32_t(sx >> 31)) + 1u);
        pred[comp] = int32_t(uint32_t(pred[comp]) + diff);
    }
    if (!fine) {
        CG_WALK_COUNT(slow_q1);
        return false;
    }
    const uint32_t last = (Te >> kWalkLastShift) & 31u;
    const uint32_t pe = walk_pos_of(s, lane, at[4u * 4u], Te);
    l.ref_left = 32u + ((last - pe) & 31u) - last;
    l.pred[0] = pred[0];
    l.pred[1] = pred[1];
    l.pred[2] = pred[2];
    return true;
}

// The record of the MCU that begins at list entry `at` {word A, state} with the lane's ref_left and predictions.
CG_DEV void walk_record_at(const WalkLane &l, const ImageDesc &d, const HuffShared &s, uint32_t lane, const uint32_t *at, uint32_t mcu)
{
    const uint32_t p = walk_pos_of(s, lane, at[0], at[1]);
    McuState st;
    st.info = (p & 31u) | (umin(l.ref_left, 63u) << 5);
    st.pred[0] = l.pred[0];
    st.pred[1] = l.pred[1];
    st.pred[2] = l.pred[2];
    CG_GLOBAL(uint32_t, d.mcu_word)[mcu] = l.row0 + (p >> 5);
    CG_GLOBAL(McuState, d.mcu_state)[mcu] = st;
}

// The slow road's window: 64 words of lane `src`'s column from row `r0` on, side by side -- every lane of the wave copies
// one (row r0 + lane; `n`: how many of them hold words of the scan).
constexpr uint32_t kWalkSlowWords = uint32_t(kWave);
CG_DEV void walk_slow_window(uint32_t *window, const HuffShared &s, uint32_t src, uint32_t r0, uint32_t n, uint32_t lane)
{
    if (lane < n)
        window[lane] = s.win[(r0 + lane) * uint32_t(kWave) + src];
}

// The MCU once more from where it began (word, bit; the lane's ref_left and predictions: what its record says), with
// the reference's reader -- its words from `window` (the n scan words from window_word on, most significant bit first),
// from global memory behind them; dump: 80 bytes anybody may write to.  Behind it the lane stands at the MCU's end,
// on its rows or parked, or is dead.
CG_DEV void walk_slow_mcu(WalkLane &l, const ImageDesc &d, const HuffShared &s, const WalkTabs &t, uint32_t lane, uint32_t word, uint32_t bit,
                          const uint32_t *window, uint32_t window_word, uint32_t n, int16_t *dump)
{
    HuffShared g = s;
    g.win = window;
    g.win_base = window_word;
    g.win_len = n;
    EntropyState e;
    entropy_state_at(e, d, g, word, bit, l.ref_left);
    e.pred0 = l.pred[0];
    e.pred1 = l.pred[1];
    e.pred2 = l.pred[2];
#pragma unroll 1
    for (uint32_t k = 0; k < 4u; k++)
        entropy_data_unit(e, d, g, comp_of_k(k), dump);
    l.pred[0] = e.pred0;
    l.pred[1] = e.pred1;
    l.pred[2] = e.pred2;
    l.at_word = e.r.next_word;
    l.at_bit = 0u;
    if (e.r.left >= 64u) {
        l.dead = true; // (ImageDesc::mcu_ok: run dry, nothing else)
        return;
    }
    // the reader holds `left` bits in front of word next_word
    l.ref_left = e.r.left;
    const uint64_t p = 32ull * e.r.next_word - e.r.left;
    l.at_word = uint32_t(p >> 5);
    l.at_bit = uint32_t(p) & 31u;
    l.parked = true;
    if (l.at_word >= l.row0 && l.at_word - l.row0 < l.valid)
        walk_place(l, s, t, lane, 32u * (l.at_word - l.row0) + l.at_bit, true);
}

// A dead lane's MCU: every data unit decodes from zeros.
CG_DEV void walk_dead_mcu(WalkLane &l, const WalkTabs &t)
{
    l.pred[0] = int32_t(uint32_t(l.pred[0]) + 2u * uint32_t(t.zero_diff[0]));
    l.pred[1] = int32_t(uint32_t(l.pred[1]) + uint32_t(t.zero_diff[1]));
    l.pred[2] = int32_t(uint32_t(l.pred[2]) + uint32_t(t.zero_diff[2]));
}

// Would new rows help this lane?  It has fewer than `below` staged words in front of it, or it is parked where rows
// can be had (three words of the scan in front of it at least: what the walk reads at a step).
CG_DEV bool walk_wants_rows(const WalkLane &l, const ImageDesc &d, const HuffShared &s, uint32_t lane, uint32_t nrows, uint32_t below)
{
    if (!l.active || l.dead)
        return false;
    if (l.parked)
        return l.at_word + 3u <= d.nwords && !(l.at_word == l.row0 && l.valid == nrows);
    const int32_t row = walk_row_of(s, lane, l.wa);
    return l.valid == nrows && row + int32_t(below) >= int32_t(l.valid); // (valid < nrows: the scan ends inside the rows)
}

#if defined(__HIPCC__)
// The walk of 64 restart intervals, a lane each (tests/emul plays the same steps lane by lane), `chunk` MCUs at a time.
// lists: 64 x walk_list_bytes(chunk) of the wave's; slow_window: kWalkSlowWords words of the wave's; dump: 80 bytes anybody may write to.
CG_DEV void walk_wave_422(const ImageDesc &d, const HuffShared &s, const WalkTabs &t, uint32_t *lists, uint32_t *slow_window, int16_t *dump,
                          uint32_t nrows, uint32_t stage_below, uint32_t chunk, uint32_t interval, uint32_t lane)
{
    WalkLane l;
    const bool active = interval < d.total_intervals;
    interval = active ? interval : d.total_intervals - 1u;
    walk_lane_init(l, d, interval, active);
    uint32_t *list = lists + lane * (walk_list_bytes(chunk) / 4u);
    walk_prepare_list(list, t, chunk);
    walk_restage(l, d, s, t, nrows, lane);
    const uint32_t mcus = d.restart_interval, mcu0 = interval * d.restart_interval;
#pragma unroll 1
    for (uint32_t i = 0; i < mcus; i += chunk) {
        const uint32_t g = umin(chunk, mcus - i);
        const bool go = active && !l.dead && !l.parked;
        if (go) {
            list[0] = l.wa;
            list[1] = l.T;
        }
        const uint32_t walked = walk_mcu_lean(l, d, s, t, list, go, 4u * g, lane) / 4u; // whole MCUs
        // MCU by MCU (m wave-uniform): what the walk found; `good`: MCUs of the chunk that stand
        uint32_t good = 0u;
#pragma unroll 1
        for (uint32_t m = 0; m < g; m++) {
            if (go && good == m && m < walked) {
                // (the record first: it says where the MCU begins, whatever comes of it)
                walk_record_at(l, d, s, lane, list + 16u * m, mcu0 + i + m);
                if (walk_mcu_pass(l, d, s, t, list, m, lane))
                    good = m + 1u;
            }
        }
        // the rest of the chunk by the slow road: lanes that were not walking, or whose walk went wrong at MCU `good`
        // (rare: one lane after the other through the wave's one window)
        const bool slow = active && !(go && good == g);
        uint64_t todo = __builtin_amdgcn_ballot_w64(slow);
        while (todo) {
            const uint32_t src = uint32_t(__builtin_ctzll(todo));
            todo &= todo - 1u;
            const uint32_t src_good = uint32_t(__builtin_amdgcn_readlane(int(go ? good : 0u), int(src)));
#pragma unroll 1
            for (uint32_t m = src_good; m < g; m++) {
                // where src's MCU begins: on its rows (the window is filled from them), or not (global memory)
                uint32_t p0 = 0u;
                bool on_rows = false;
                if (lane == src) {
                    if (go && m == good) {
                        p0 = walk_pos_of(s, lane, list[16u * m], list[16u * m + 1u]);
                        on_rows = true;
                    } else if (!l.dead && !l.parked) {
                        p0 = walk_pos_of(s, lane, l.wa, l.T);
                        on_rows = true;
                    }
                }
                const uint32_t src_p0 = uint32_t(__builtin_amdgcn_readlane(int(p0), int(src)));
                const uint32_t src_on_rows = uint32_t(__builtin_amdgcn_readlane(int(on_rows ? 1u : 0u), int(src)));
                const uint32_t src_valid = uint32_t(__builtin_amdgcn_readlane(int(l.valid), int(src)));
                const uint32_t r0 = src_p0 >> 5, n = src_on_rows && src_valid > r0 ? umin(src_valid - r0, kWalkSlowWords) : 0u;
                walk_slow_window(slow_window, s, src, r0, n, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane == src) {
                    if (on_rows) {
                        l.at_word = l.row0 + (p0 >> 5);
                        l.at_bit = p0 & 31u;
                        l.parked = true; // (walk_slow_mcu puts the lane back on its rows, if there are rows where it ends)
                    }
                    walk_record(l, d, s, lane, mcu0 + i + m);
                    if (l.dead)
                        walk_dead_mcu(l, t);
                    else
                        walk_slow_mcu(l, d, s, t, lane, l.at_word, l.at_bit, slow_window, l.row0 + r0, n, dump);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (i + g < mcus && wave_any(walk_wants_rows(l, d, s, lane, nrows, stage_below)))
            walk_restage(l, d, s, t, nrows, lane);
    }
}
#endif

} // namespace compeg
