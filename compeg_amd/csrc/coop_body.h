// Per-lane bodies of the cooperative kernels: the lanes of a wave work inside the restart intervals.
//
// Two kernels use them (kernels.hip).  decode_coop_team_422_kernel, the default: a team of four waves takes 4 x 64
// data units' worth of intervals; one wave walks them (phase 1) -- for intervals of up to 16 data units a lane per
// interval, no speculation, through the walk tables and chase_run_lean below, and its entries go straight to the
// decoding lanes (coop_lean_emit; phase 2 is not needed) -- then each wave decodes 64 data units (phase 3); with
// DRI = 4 the decoding does not wait for the walk's end but follows it quarter by quarter (coop_decode_quarter_422).
// decode_coop_422_kernel, the first form (COMPEG_COOP_TEAM=0, and the walks of longer intervals in the team form):
// one wave does all three phases for 64 data units' worth of intervals, as described here.
//
// The reference decodes a restart interval with one thread (src/huffman.wgsl:118-204), and so do the other
// kernels here (one lane per interval).  A launch that has few intervals -- one 4K frame with DRI = 4 is
// 16 200 of them -- then leaves most of the chip idle while every lane walks ~170 dependent symbols.  Here a
// wave takes only as many intervals as have 64 data units together (4 with DRI = 4) and spends its lanes inside
// them:
//
//  1. Walk.  What is serial in an interval is only *where each data unit begins*: bit position, and the size of
//     the symbol in front of it (together they fix the reference reader's `left` at the DC code, quirk Q1).
//     Lane 0 of an interval walks the symbols from the interval's start (sizes and zig-zag advances only, no
//     values).  The other lanes start at word boundaries further into the interval ("subsequences") WITHOUT
//     knowing the decoder state there: each assumes it is inside the AC part of data unit h of an MCU, one
//     lane per h = 0..3, and walks on from that guess.  Huffman streams re-synchronise: after a few symbols a
//     wrong start lands on a true symbol boundary, and from the next end-of-block on it walks the true sequence
//     of data units (if its h was the right one).  Every lane lists the data-unit starts it passes.
//  2. Validate.  A lane walks a little beyond the end of its subsequence.  If one of the data-unit starts it
//     lists there equals -- same bit position, same size of the symbol in front, same index inside the MCU --
//     an entry of a lane of the next subsequence, the two walks are identical from that point on (the state is
//     the whole state), so the successor's list continues the predecessor's.  Following these links from lane
//     0, whose start state is known, yields the start state of every data unit of the interval.  A link that is
//     missing (no synchronisation within the overlap) is not an error: the last validated lane simply walks on
//     through the next subsequence, and validation runs again.  Nothing is ever assumed: a state is used only
//     if the chain from the interval's start proves it.
//  3. Decode.  One lane per DATA UNIT: the proven fast-mode decoder of the fused kernel (fast_dc / fast_ac,
//     exact path behind it) started at the data unit's state, coefficients into the lane's LDS slot, then DC
//     prediction (a sum over the interval's earlier data units of the component), IDCT and composite with
//     all 64 lanes busy.
//
// Quirk Q1 in this scheme: the walk follows the stream as if the reference's reader never ran dry.  The lane that
// decodes a data unit whose DC code does underflow the reference reader finds that out (fast_dc refuses); behind
// it the reference reads zeros for the rest of the interval, and those data units decode to a per-component
// constant that the host front-end computes with the exact-path reader (ImageDesc::zero_du) -- whatever their
// lanes decoded from the walk's states is dropped.  Anything else out of the ordinary (a walk that leaves the LDS
// window on a corrupt stream, DC categories above 15 from a hostile table) sends the interval to one lane
// running the interval-serial decoder the other kernels use.
#pragma once

#include "kernels_body.h"

namespace compeg {

#if defined(CG_EMUL_STATS)
struct CoopStats {
    unsigned long intervals, rounds, direct, continued, serial, dead, zero, chase_steps, wave_steps, true_steps, true_max, hist[16], link_tries, link_ok, link_full_ok, link_none;
    unsigned long dead_quarter[4], overrun; // quirk Q1's first dead data unit by quarter of its interval; walks that ended beyond their interval's words
};
inline CoopStats g_coop_stats{};
#define CG_COOP_COUNT(field, n) (g_coop_stats.field += (n))
#else
#define CG_COOP_COUNT(field, n) ((void)0)
#endif

constexpr uint32_t kCoopListCap = 20;      // state words per chasing lane (64 x 20 words = the slot area) ...
constexpr uint32_t kCoopMaxEntries = 19;   // ... of which the last is scratch (the entry being built); CoopGeom::list_cap / max_entries
                                           // are these, or more where an interval is longer than a team's 256 data units
#ifndef CG_COOP_MARGIN
#define CG_COOP_MARGIN 64
#endif
constexpr uint32_t kCoopMargin = CG_COOP_MARGIN; // bits a lane walks on beyond the end of its subsequence
constexpr uint32_t kCoopHead = 3;          // leading entries of a successor's list that a link may point at
constexpr uint32_t kCoopTail = 3;          // trailing entries of a lane's own list that may carry the link
constexpr uint32_t kCoopPosBits = 18;      // bit positions inside a walk's window
constexpr uint32_t kCoopPosMask = (1u << kCoopPosBits) - 1u;
constexpr uint32_t kCoopMaxWindow = (1u << (kCoopPosBits - 5u)) - 8u; // words (32 KB: what a team alone on its CU can have)
// Words of the stream a team's window holds behind its intervals' end, beside the reader's slack: the reference's reader
// stands up to three words beyond an interval's end when it begins that interval's last data units (quirk Q2, the words
// it has in hand), and a walk may not begin a data unit kDuWordSlack words in front of the window's end -- with two words
// of room (round 3) one frame in five with intervals of 60 to 128 MCUs had an interval whose walk stopped a data unit
// or two short of the end and went to the serial decoder: a millisecond for a 250 us frame (profiles/r04/NOTES.md).
constexpr uint32_t kCoopEndSlack = 16;
constexpr uint32_t kCoopMaxRestart = 256;  // MCUs per interval (1024 data units: 16 rounds of 64)
constexpr uint32_t kCoopMaxRounds = 16;    // rounds of 64 data units per walk
constexpr uint32_t kCoopQuantStride = 36;  // floats between the components' quantiser rows in LDS

// Data-unit start state: bits 0..17 bit position inside the walk's window (kCoopPosBits), 18..22 size of the symbol in
// front of it (with the position it fixes the reference reader's `left` at the DC code, quirk Q1).  The third part of
// the state, the data unit's index inside its MCU, is not stored: entry i of a list belongs to index
// (index of entry 0 + i) mod 4.  Two states are equal iff these 23 bits and that index are.
constexpr uint32_t kCoopStateMask = (32u << kCoopPosBits) - 1u;
constexpr uint32_t kCoopZero = 1u << 25;   // (decode phase) the data unit lies behind a dead one: zero-stream levels
constexpr uint32_t kCoopUnset = 1u << 27;  // (decode phase) no state: nothing to decode

constexpr uint32_t kStopAnomaly = 2u;      // the walk cannot go on (window exhausted, hostile table)

// Wave-uniform values of the image the lanes select from by component / data-unit index.
struct CoopTables {
    const uint16_t *ac_fast; // the two direct AC tables (LDS)
    const uint16_t *dc_fast; // the two direct DC tables (LDS)
    uint32_t acsel, dcsel;   // byte k: offset of the table data unit k of an MCU uses, in KiB from ac_fast
    uint32_t ac_off[3], dc_off[3]; // per component: offset of its L1 tables (escapes)
    uint32_t fast_base[3];
    uint32_t dc_quant[3];
    uint32_t zrl;            // 17 (the reference, quirk Q2) or 16
    bool standard;
    const uint32_t *walk;    // the walk tables (coop_walk_word), or null: kWalkWords words
    uint32_t walk_acsel, walk_dcsel; // byte k: which pairs / singles tables, which dc table data unit k of an MCU uses
    uint32_t walk_ids;       // bits 2i, 2i + 1: which direct DC table, which direct AC table dc[i] is made of
    bool walk_ok;            // the components use at most two different pairs of tables
};

// ---------------------------------------------------------------------------
// Walk tables: what a walk that only has to find where the data units begin looks up.  One 32-bit entry per
// prefix of the coming stream bits, covering up to TWO symbols (the second one's code has to lie inside the prefix
// as well; its magnitude bits need not):
//   bits  0..15  minus the bits consumed (two's complement: added to the reader's shift by the same packed add
//                that adds bits 21..28 to the zig-zag state)
//   bits 16..20  size of the last symbol consumed (what a data unit's start state records, quirk Q1)
//   bits 21..28  zig-zag advance, 64 for an end-of-block
//   0            the prefix does not determine a symbol (code longer than the prefix): the two-level tables decide
// Six tables in 8192 words (entry offsets below):
//   pairs[a]   11-bit prefixes of AC table a: a symbol and, where it fits, the one behind it.  The second symbol is
//              an AC symbol of the same data unit by assumption, so these tables are used only while the first one
//              cannot complete the data unit: zig-zag state below kWalkNear (its advance is at most 17, quirk Q2's
//              ZRL).  A data unit that ends without an end-of-block would otherwise have its successor's DC code
//              read as an AC symbol, and nothing in the loop could tell.
//   singles[a] 10-bit prefixes of AC table a, one symbol: for the states from kWalkNear on.
//   dc[i]      10-bit prefixes: a DC code and the first AC symbol (or end-of-block) behind it, for the (at most two)
//              pairs of DC and AC table that the components use.
// A table is named by one word: its byte address (LDS; offset from the tables' start on the host) with the shift
// that takes 32 stream bits to its index in the low five bits (tables are 32-byte aligned).
// ---------------------------------------------------------------------------
constexpr uint32_t kWalkWords = 8192;
constexpr uint32_t kWalkPairs0 = 0, kWalkPairs1 = 2048, kWalkSingles0 = 4096, kWalkDc0 = 5120, kWalkSingles1 = 6144, kWalkDc1 = 7168;
constexpr uint32_t kWalkSinglesBehindPairs = 4096; // words, for both AC tables
constexpr uint32_t kWalkNear = 47;                 // 46 + the largest single advance (17) = 63: the data unit goes on
constexpr uint32_t kWalkStShift = 21, kWalkLastShift = 16;

CG_DEV uint32_t walk_pack(uint32_t tot, uint32_t last, uint32_t adv)
{
    return (adv << kWalkStShift) | (last << kWalkLastShift) | ((0u - tot) & 0xffffu);
}

// Word i of the walk tables.  ac_fast / dc_fast: the direct tables (device_types.h); walk_ids: bits 2i, 2i + 1 name
// the direct DC table and the direct AC table that dc[i] is made of.
CG_DEV uint32_t coop_walk_word(const uint16_t *ac_fast, const uint16_t *dc_fast, uint32_t walk_ids, uint32_t i)
{
    // which table, how many prefix bits
    uint32_t bits = 10u, idx, ac_id, dc_id = 0u;
    bool dc = false, two = true;
    if (i < kWalkSingles0) {
        bits = 11u;
        ac_id = i >> 11;
        idx = i & 2047u;
    } else {
        const uint32_t q = (i - kWalkSingles0) >> 10; // singles0, dc0, singles1, dc1
        idx = i & 1023u;
        dc = (q & 1u) != 0u;
        two = dc;
        ac_id = q >> 1;
        if (dc) {
            const uint32_t ids = walk_ids >> (2u * (q >> 1));
            dc_id = ids & 1u;
            ac_id = (ids >> 1) & 1u;
        }
    }
    const uint32_t prefix = idx << (kFastBits - bits); // as an index of the 11-bit direct AC tables (zeros behind it)
    uint32_t tot1, adv1;
    if (dc) {
        const uint32_t e = dc_fast[dc_id * kDcFastEntries + (prefix >> (kFastBits - kDcFastBits))];
        tot1 = (e >> 4) & 31u;
        adv1 = 1u;
        if (e == kFastEscape || tot1 == 0u || tot1 - (e & 15u) > bits)
            return 0u;
    } else {
        const uint32_t e = ac_fast[ac_id * kFastEntries + prefix];
        tot1 = (e >> 4) & 31u;
        adv1 = e >> 9;
        if (e == kFastEscape || tot1 == 0u || tot1 - (e & 15u) > bits)
            return 0u;
    }
    if (two && adv1 != kFastAdvEob && tot1 < bits) {
        const uint32_t rest = (prefix << tot1) & (kFastEntries - 1u); // the prefix behind the first symbol, zeros behind it
        const uint32_t e = ac_fast[ac_id * kFastEntries + rest];
        const uint32_t tot2 = (e >> 4) & 31u, adv2 = e >> 9;
        if (e != kFastEscape && tot2 != 0u && tot2 - (e & 15u) <= bits - tot1 && tot1 + tot2 <= 31u)
            return walk_pack(tot1 + tot2, tot2, adv1 + adv2);
    }
    return walk_pack(tot1, tot1, adv1);
}

// The wave's share of LDS.
struct CoopShared {
    HuffShared h;        // tables, window, the 64 data-unit slots
    uint32_t *lists;     // 64 x list_cap state words: with 20 the same bytes as the slots (lists die before slots live)
    uint32_t *du_state;  // [64 x rounds] start state of every data unit of the walk
    uint32_t *lane_n;    // [64] chasing lanes: entries listed | stop reason << 8
    uint32_t *link;      // [64] chasing lanes: 1 | successor lane << 8 | its entry << 16 | own entry << 24, or 0
    uint32_t *seg;       // [64] this round's stretches of every interval's sequence (see coop_follow)
    uint32_t *verdict;   // [64] per interval: kVerdict* | lane << 8 | data units settled << 16
    uint32_t *nseg;      // [64] per interval: stretches in seg
    uint32_t *dead_from; // [64] per interval: first data unit whose DC code underflows the reference reader
    int32_t *diffs;      // [64] DC differences of the data units being decoded
    uint32_t *carry;     // [4 x rounds] per round: the DC sums (Y, Cb, Cr) of the interval that runs on into the next round
                         // (the bytes of `link`: the walk is over when the rounds begin)
    const float *quant;  // 3 rows of quantisers (workgroup-wide)
    // team form, quarters (coop_decode_quarter_422): the team's flag words; the four waves' diffs areas (wave m's at
    // team_diffs + m * team_diffs_stride); the team's number inside its workgroup
    uint32_t *flags;
    int32_t *team_diffs;
    uint32_t team_diffs_stride, team_in_wg;
};
// flag words of a team: [0] the walk: 1 = its lists are prepared, 2 = it is done; [1] quarters whose DC differences
// are final (bit q) -- rounds: how many of them are, in order (count); [2] quarters / rounds whose pixels are stored
// (count); [3] quarters that have read their start states (bit q)
constexpr uint32_t kTeamWalk = 0, kTeamDecoded = 1, kTeamStored = 2, kTeamStatesRead = 3;

CG_DEV void coop_bind_misc(CoopShared &cs, uint32_t *misc)
{
    cs.lane_n = misc;
    cs.link = misc + 64;
    cs.diffs = reinterpret_cast<int32_t *>(misc + 128);
    cs.seg = misc + 192;
    cs.verdict = misc + 256;
    cs.nseg = misc + 320;
    cs.dead_from = misc + 384;
    cs.du_state = misc + 448;
    cs.carry = misc + 64;
    cs.flags = nullptr;
}
// a walk's bookkeeping words: lane_n, link, diffs, seg, verdict, nseg, dead_from (64 each), then du_state (64 per round)
constexpr uint32_t coop_misc_words(uint32_t rounds) { return 7u * 64u + rounds * 64u; }

constexpr uint32_t kVerdictDone = 1u, kVerdictContinue = 2u, kVerdictSerial = 3u;

struct CoopGeom {
    uint32_t R, dpi, ipw;    // MCUs and data units per interval, intervals per walk (device_types.h: CoopShape)
    uint32_t rounds;         // the walk covers this many rounds of 64 data units
    uint32_t lpi;            // lanes that walk one interval
    uint32_t count;          // subsequences per interval: 1 + the speculative ones
    uint32_t list_cap, max_entries; // words per walking lane's list; entries it may hold (the last word is scratch)
    uint32_t first_interval; // of this walk
    uint32_t intervals;      // that exist (0..ipw)
    uint32_t dus;            // data units of the walk's existing intervals
};

// waves: 4 -- a team (kernels.hip): one wave walks the intervals of 4 x 64 data units, four decode them; 1 -- a lone
// wave does it all for 64 data units' worth of intervals, and every interval has as many lanes to walk it as it has
// data units (2: twice the intervals, half the lanes each).  An interval longer than that is alone in its walk and
// decoded in as many rounds as it needs.  spec_shift: the number of subsequences is divided by 2^spec_shift (0: as
// many as the lanes allow; large: none, lane 0 walks the whole interval).
CG_DEV void coop_geom(const ImageDesc &d, uint32_t walk_index, CoopGeom &g, uint32_t spec_shift = 0u, uint32_t waves = 1u)
{
    const CoopShape sh = coop_shape(d.restart_interval, waves >= 4u ? 4u : (waves >= 2u ? 2u : 1u));
    g.R = d.restart_interval; // 1 .. kCoopMaxRestart (ImageDesc::coop_ok)
    g.dpi = sh.dpi;
    g.ipw = sh.ipw;
    g.rounds = sh.rounds;
    g.lpi = sh.lpi;
    g.count = spec_shift < 31u && (sh.count >> spec_shift) ? sh.count >> spec_shift : 1u; // lane 0, then four lanes per speculative subsequence
    g.list_cap = sh.list_cap;
    g.max_entries = sh.list_cap - 1u;
    g.first_interval = walk_index * g.ipw;
    const uint32_t left = d.total_intervals > g.first_interval ? d.total_intervals - g.first_interval : 0u;
    g.intervals = left < g.ipw ? left : g.ipw;
    g.dus = g.intervals * g.dpi;
}

// Data unit n of a walk (round n / 64, lane n % 64): which of the walk's intervals, and which of its data units.
// (An interval may begin in one round and end in another.)
CG_DEV void coop_du_of(const CoopGeom &g, uint32_t n, uint32_t &il, uint32_t &tl)
{
    il = n / g.dpi;
    tl = n - il * g.dpi;
}

CG_DEV uint32_t sel3(uint32_t c, uint32_t a0, uint32_t a1, uint32_t a2)
{
    return c == 0u ? a0 : (c == 1u ? a1 : a2);
}

CG_DEV uint32_t comp_of_k(uint32_t k) { return k < 2u ? 0u : k - 1u; } // Y0 Y1 Cb Cr

CG_DEV void coop_tables(const ImageDesc &d, const HuffShared &s, CoopTables &t)
{
    t.ac_fast = s.l2 + d.fast_off;
    t.dc_fast = t.ac_fast + 2u * kFastEntries;
    t.acsel = t.dcsel = 0u;
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t c = comp_of_k(k);
        // direct AC tables: 4 KiB each; the direct DC tables, 1 KiB each, follow them
        t.acsel |= ((d.fast_table[c] & 1u) * (kFastEntries * 2u / 1024u)) << (8u * k);
        t.dcsel |= (2u * (kFastEntries * 2u / 1024u) + (d.dc_fast_table[c] & 1u) * (kDcFastEntries * 2u / 1024u)) << (8u * k);
    }
    for (uint32_t c = 0; c < 3u; c++) {
        t.ac_off[c] = d.ac_table[c] * 256u;
        t.dc_off[c] = d.dc_table[c] * 256u;
        t.fast_base[c] = d.fast_off + d.fast_table[c] * kFastEntries;
        t.dc_quant[c] = d.dc_quant[c];
    }
    t.standard = d.standard_entropy != 0u;
    t.zrl = t.standard ? 16u : 17u;
    t.walk = nullptr;
    t.walk_acsel = t.walk_dcsel = 0u;
    // the pairs (DC table, AC table) in use: component 0's is walk table 2, the first different one walk table 3
    const uint32_t pair0 = (d.dc_fast_table[0] & 1u) | ((d.fast_table[0] & 1u) << 1);
    uint32_t pair1 = pair0;
    t.walk_ok = true;
    for (uint32_t c = 1; c < 3u; c++) {
        const uint32_t pr = (d.dc_fast_table[c] & 1u) | ((d.fast_table[c] & 1u) << 1);
        if (pr != pair0 && pair1 == pair0)
            pair1 = pr;
        t.walk_ok = t.walk_ok && (pr == pair0 || pr == pair1);
    }
    t.walk_ids = pair0 | (pair1 << 2);
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t c = comp_of_k(k);
        const uint32_t pr = (d.dc_fast_table[c] & 1u) | ((d.fast_table[c] & 1u) << 1);
        t.walk_acsel |= (d.fast_table[c] & 1u) << (8u * k);
        t.walk_dcsel |= (pr == pair0 ? 0u : 1u) << (8u * k);
    }
}

// The window of the wave's intervals: their contiguous words plus the reader's slack.
// (in two steps, so that a kernel can put other loads between the two reads and their first use)
CG_DEV void coop_window_fetch(const ImageDesc &d, const CoopGeom &g, uint32_t &first_word, uint32_t &end_word)
{
    const uint32_t first = g.first_interval, after = first + g.ipw;
    first_word = first < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[first] : 0u;
    end_word = (after < d.total_intervals && after < d.nstarts) ? CG_GLOBAL(const uint32_t, d.starts)[after] : d.nwords;
}
CG_DEV void coop_window_from(const ImageDesc &d, uint32_t first_word, uint32_t end_word, uint32_t window_words,
                             uint32_t &base, uint32_t &len)
{
    // (+ 2: the reader keeps up to two words in hand, so at the start of the last data units of the wave's last
    // interval its position is that far beyond the interval's end -- still inside the window with these)
    const uint32_t end = umin(end_word, d.nwords) + kDuWordSlack + 2u + kCoopEndSlack;
    base = umin(first_word, d.nwords);
    len = end > base ? umin(end - base, window_words) : 0u;
}
CG_DEV void coop_window(const ImageDesc &d, const CoopGeom &g, uint32_t window_words, uint32_t &base, uint32_t &len)
{
    uint32_t first_word, end_word;
    coop_window_fetch(d, g, first_word, end_word);
    coop_window_from(d, first_word, end_word, window_words, base, len);
}

// First bit position at which a walk may not begin another data unit (it could leave the window).
CG_DEV uint32_t coop_hard_end(const HuffShared &s)
{
    return s.win_len > kDuWordSlack ? 32u * (s.win_len - kDuWordSlack) : 0u;
}

// ---------------------------------------------------------------------------
// 1. Chase
// ---------------------------------------------------------------------------

struct ChaseState {
    uint32_t p;        // bit position of the coming symbol, window-relative
    uint32_t s;        // 0: a DC code comes next; else 1 + zig-zag position of the coefficient decoded last
    uint32_t k8;       // 8 x (data units completed + the index inside its MCU the walk began in)
    uint32_t *lp;      // the list's next free slot
    uint32_t *lp_max;  // the walk ends when the list has grown to here
    uint32_t stop_p;   // ... or when a data unit is complete at or beyond this position
    uint32_t sub_end;  // bit position where this lane's own stretch ends: entries from here on may carry a link
    uint32_t next_sub; // the subsequence whose lanes may continue this lane's list (>= count: none)
    uint32_t k0;       // index (inside its MCU) of the data unit that entry 0 of the list starts
    uint32_t flags;    // kStopAnomaly
    bool used;         // this lane walks at all
    bool active;
    // chase_run_lean's results, until chase_lean_regular and coop_lean_emit have read them
    uint32_t lean_j0, lean_done; // entries [lean_j0, lean_done) of the list are 16-byte entries
    uint32_t lean_p;             // where the walk stands
    bool lean_walked;
#if defined(CG_COOP_STAMPS)
    uint64_t loop_cycles; // (diagnostic builds) inside the hand-written loop: cycles, steps, times entered
    uint32_t loop_steps, loop_entries;
    uint64_t init_cycles, tail_cycles; // before the loop is entered first, behind it
#endif
};

// The walk: a per-lane loop (lanes that are done drop out of it, the wave leaves it when the last one has).
// The reader is a bit position: the two stream words around it come from the window, the symbol's size
// and zig-zag advance from the direct table of the state the lane is in.  The state word of "the data unit
// that would begin behind this symbol" is written to the list's next free slot every time; it stays there when
// the symbol does end a data unit (no branch around the store).
//
// On the GPU the loop is hand-written: it is where a small launch spends most of its time, and the compiler's
// version of it carried twenty instructions of mask bookkeeping per symbol.  Lanes that are done leave through EXEC.  Codes longer than the direct tables' prefix are rare: when a lane meets one
// the block is left with the symbol unapplied (an escape entry applies as a no-op), the C++ below resolves it
// through the two-level tables, and the block is entered again at "apply".
CG_DEV void chase_run(ChaseState &c, const ImageDesc &d, const HuffShared &s, const CoopTables &t, unsigned long &steps)
{
    uint32_t p = c.p, st = c.s, k8 = c.k8;
    uint32_t *lp = c.lp;
    (void)steps;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CG_COOP_NO_ASM)
    uint32_t alive = c.active ? 1u : 0u, code = 0u;
    const uint32_t lpa0 = uint32_t(reinterpret_cast<uintptr_t>(lp)), k8_0 = k8;
    uint32_t lpa = lpa0;
    const uint32_t lpmax = uint32_t(reinterpret_cast<uintptr_t>(c.lp_max));
    const uint32_t win = uint32_t(reinterpret_cast<uintptr_t>(s.win)), tab = uint32_t(reinterpret_cast<uintptr_t>(t.ac_fast));
    const uint32_t acsel = t.acsel, dcsel = t.dcsel; // (vector copies: one scalar operand per instruction)
    const uint32_t dcsel_next = (dcsel >> 8) | (dcsel << 24); // byte k: the DC table of data unit k + 1
    // The reader: three stream words A B C in registers (A holds the position, of which 32 - sn bits are consumed,
    // 1..32 of them), reloaded from LDS every symbol -- but for the *next* symbol, so that no window read sits between
    // one symbol's table entry and the next one's table address.  A symbol is at most 31 bits: the next position lies
    // in {A B} or in {B C}, both are taken and one is kept.
    const uint32_t wi1 = (p + 31u) >> 5; // (the word in front of an aligned position is never looked at: sn = 0)
    uint32_t wa = win + 4u * wi1 - 4u, sn = 32u * wi1 - p;
    const uint32_t kc = k8 - 2u * lpa0; // 2 lp + kc = 8 x the data unit's index, modulo 32
    // the first symbol's entry, looked up here
    auto lookup = [&](uint32_t p_, uint32_t st_, uint32_t k8_) {
        const uint32_t *wp = s.win + (p_ >> 5);
        const uint32_t cur = uint32_t(((uint64_t(wp[0]) << 32 | wp[1]) << (p_ & 31u)) >> 32);
        const bool dc = st_ == 0u;
        const uint32_t kib = ((dc ? dcsel : acsel) >> (k8_ & 31u)) & 0xffu;
        return uint32_t((t.ac_fast + kib * 512u)[cur >> (dc ? 32u - kDcFastBits : 32u - kFastBits)]);
    };
    uint32_t ent = c.active ? lookup(p, st, k8) : 0u;
    if (__builtin_amdgcn_ballot_w64(alive != 0u) != 0u)
        for (;;) {
            // 33 vector instructions per symbol, 15 of them between a table entry's arrival and the next lookup's
            // issue; the rest (the list, the tables of the coming data unit, the reload, who is done) runs while that
            // lookup is in flight.  Masks go to the scalar side only at the end of the body, well behind the
            // compares that made them.  An escape entry (a code longer than the direct tables' prefix) applies as a
            // no-op (device_types.h): its lane runs through the body once, then everybody leaves, the C++ below
            // resolves the code through the two-level tables, and the block is entered again.
            asm volatile(
                "s_mov_b64 s[74:75], exec\n\t"
                "v_cmp_ne_u32 vcc, 0, %[alive]\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_cbranch_execz 4f\n\t"
                "ds_read2_b32 v[42:43], %[wa] offset1:1\n\t"          // A, B
                "ds_read_b32 v47, %[wa] offset:8\n\t"                 // C
                "v_lshl_add_u32 v44, %[lp], 1, %[kc]\n\t"
                "v_bfe_u32 v48, %[acsel], v44, 8\n\t"
                "v_bfe_u32 v49, %[dcseln], v44, 8\n\t"
                "v_lshl_add_u32 v48, v48, 10, %[tab]\n\t"             // the AC table of this data unit
                "v_lshl_add_u32 v49, v49, 10, %[tab]\n"                // the DC table of the next one
                "1:\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                // ---- from the entry to the next lookup
                "v_bfe_u32 v40, %[ent], 4, 5\n\t"                     // size
                "v_lshrrev_b32 v44, 9, %[ent]\n\t"                    // zig-zag advance
                "v_cmp_eq_u32 s[76:77], 15, %[ent]\n\t"               // kFastEscape: lanes that met a long code
                "v_sub_u32 %[sn], %[sn], v40\n\t"                     // (negative: the position has left A)
                "v_add_u32 %[st], %[st], v44\n\t"
                "v_alignbit_b32 v41, v42, v43, %[sn]\n\t"             // (the shift is taken modulo 32)
                "v_alignbit_b32 v45, v43, v47, %[sn]\n\t"
                "v_cmp_gt_i32 vcc, 0, %[sn]\n\t"
                "v_cmp_lt_u32 s[72:73], 63, %[st]\n\t"                // the data unit is complete: a DC code comes next
                "v_cndmask_b32 v41, v41, v45, vcc\n\t"                // the next 32 stream bits
                "v_cndmask_b32_e64 v46, v48, v49, s[72:73]\n\t"
                "v_cndmask_b32_e64 v45, 21, 23, s[72:73]\n\t"         // 32 - index bits
                "v_lshrrev_b32 v45, v45, v41\n\t"
                "v_lshl_add_u32 v46, v45, 1, v46\n\t"
                "ds_read_u16 %[ent], v46\n\t"
                // ---- under that read: the list, the coming data unit's tables, the reload, who goes on
                "v_add_u32 %[p], %[p], v40\n\t"
                "v_lshl_or_b32 v40, v40, %[posbits], %[p]\n\t"
                "ds_write_b32 %[lp], v40\n\t"                         // (stays when the data unit ends here)
                "v_cndmask_b32_e64 v44, 0, 4, s[72:73]\n\t"
                "v_add_u32 %[lp], %[lp], v44\n\t"
                "v_cndmask_b32_e64 %[st], %[st], 0, s[72:73]\n\t"
                "v_cndmask_b32_e64 v44, 0, 4, vcc\n\t"
                "v_add_u32 %[wa], %[wa], v44\n\t"
                "v_and_b32 %[sn], 31, %[sn]\n\t"
                "ds_read2_b32 v[42:43], %[wa] offset1:1\n\t"
                "ds_read_b32 v47, %[wa] offset:8\n\t"
                "v_lshl_add_u32 v44, %[lp], 1, %[kc]\n\t"
                "v_bfe_u32 v48, %[acsel], v44, 8\n\t"
                "v_bfe_u32 v49, %[dcseln], v44, 8\n\t"
                "v_lshl_add_u32 v48, v48, 10, %[tab]\n\t"
                "v_lshl_add_u32 v49, v49, 10, %[tab]\n\t"
                "v_cmp_ge_u32 s[78:79], %[p], %[stopp]\n\t"
                "v_cmp_ge_u32 s[80:81], %[lp], %[lpmax]\n\t"          // the list is full
                "s_or_b64 s[78:79], s[78:79], s[80:81]\n\t"
                "s_and_b64 s[72:73], s[72:73], s[78:79]\n\t"          // lanes whose walk ends here
                "s_andn2_b64 exec, exec, s[72:73]\n\t"
                "s_and_b64 s[76:77], s[76:77], exec\n\t"              // (sets SCC: some walking lane met a long code)
                "s_cbranch_scc1 3f\n\t"
                "s_cbranch_execnz 1b\n\t"
                "s_mov_b32 %[code], 0\n\t"
                "s_branch 5f\n"
                "3:\n\t" // ---- out: the long codes go through the two-level tables
                "s_mov_b32 %[code], 1\n\t"
                "s_branch 5f\n"
                "4:\n\t"
                "s_mov_b32 %[code], 0\n"
                "5:\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_mov_b64 s[72:73], exec\n\t"                        // lanes still walking
                "s_mov_b64 exec, s[74:75]\n\t"
                "v_cndmask_b32_e64 %[alive], 0, 1, s[72:73]\n\t"
                : [p] "+v"(p), [st] "+v"(st), [lp] "+v"(lpa), [ent] "+v"(ent), [sn] "+v"(sn), [wa] "+v"(wa),
                  [alive] "+v"(alive), [code] "=s"(code)
                : [tab] "s"(tab), [acsel] "v"(acsel), [dcseln] "v"(dcsel_next), [kc] "v"(kc),
                  [stopp] "v"(c.stop_p), [lpmax] "v"(lpmax), [posbits] "n"(kCoopPosBits)
                : "memory", "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",
                  "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81");
            if (code == 0u)
                break;
            if (alive != 0u && ent == kFastEscape) {
                // longer than the direct table's prefix: through the reference's two-level tables
                const uint32_t *wp = s.win + (p >> 5);
                const uint32_t cur = uint32_t(((uint64_t(wp[0]) << 32 | wp[1]) << (p & 31u)) >> 32);
                const uint32_t comp = comp_of_k(((k8_0 + 2u * (lpa - lpa0)) >> 3) & 3u);
                if (st == 0u) {
                    const uint32_t e2 = lut_lookup<true>(d, s, sel3(comp, t.dc_off[0], t.dc_off[1], t.dc_off[2]), cur);
                    const uint32_t len = e2 >> 8, cat = e2 & 0xffu;
                    const bool bad = cat > 15u || len + cat > 31u; // only a hostile table has such categories
                    ent = bad ? 0u : (1u << 9) | ((len + cat) << 4) | cat;
                    c.flags |= bad ? kStopAnomaly : 0u;
                    alive = bad ? 0u : alive;
                } else {
                    ent = fast_entry(lut_lookup<true>(d, s, sel3(comp, t.ac_off[0], t.ac_off[1], t.ac_off[2]), cur), t.zrl);
                }
            }
        }
    k8 = k8_0 + 2u * (lpa - lpa0);
    lp = c.lp + (lpa - uint32_t(reinterpret_cast<uintptr_t>(c.lp))) / 4u;
#else
    if (c.active)
        for (;;) {
#if !defined(__HIP_DEVICE_COMPILE__)
            steps++;
#endif
            const uint32_t *wp = s.win + (p >> 5);
            const uint32_t w0 = wp[0], w1 = wp[1];
            const uint32_t cur = uint32_t(((uint64_t(w0) << 32 | w1) << (p & 31u)) >> 32);
            const bool dc = st == 0u;
            // (the selectors repeat every four data units: the byte offset is taken modulo 32 on its own)
            const uint32_t kib = ((dc ? t.dcsel : t.acsel) >> (k8 & 31u)) & 0xffu;
            const uint16_t *tab = t.ac_fast + kib * 512u;
            uint32_t ent = tab[cur >> (dc ? 32u - kDcFastBits : 32u - kFastBits)];
            if (ent == kFastEscape) {
                // longer than the direct table's prefix: through the reference's two-level tables
                CG_COUNT(escapes);
                const uint32_t comp = comp_of_k((k8 >> 3) & 3u);
                if (dc) {
                    const uint32_t e2 = lut_lookup<true>(d, s, sel3(comp, t.dc_off[0], t.dc_off[1], t.dc_off[2]), cur);
                    const uint32_t len = e2 >> 8, cat = e2 & 0xffu;
                    if (cat > 15u || len + cat > 31u) { // only a hostile table has such categories
                        c.flags |= kStopAnomaly;
                        break;
                    }
                    ent = (1u << 9) | ((len + cat) << 4) | cat;
                } else {
                    ent = fast_entry(lut_lookup<true>(d, s, sel3(comp, t.ac_off[0], t.ac_off[1], t.ac_off[2]), cur), t.zrl);
                }
            }
            const uint32_t tot = (ent >> 4) & 31u;
            p += tot;
            const uint32_t s_new = st + (ent >> 9);
            const bool du_end = s_new >= 64u;
            *lp = p | (tot << kCoopPosBits);
            const uint32_t one = du_end ? 1u : 0u;
            lp += one;
            k8 += one << 3;
            st = du_end ? 0u : s_new;
            if (du_end && (p >= c.stop_p || lp >= c.lp_max))
                break;
        }
#endif
    c.p = p;
    c.s = st;
    c.k8 = k8;
    c.lp = lp;
    c.active = false;
}

constexpr uint32_t kLeanHostWordBias = 0x10000u; // (host build: an entry's first word is a word index, -1 .. 2047, plus this)
// The one-word names of the walk tables (see above), relative to the tables' start.
CG_DEV uint32_t walk_pairs_name(uint32_t ac_id) { return (ac_id ? kWalkPairs1 : kWalkPairs0) * 4u + 21u; }
CG_DEV uint32_t walk_dc_name(uint32_t i) { return (i ? kWalkDc1 : kWalkDc0) * 4u + 22u; }
constexpr uint32_t kWalkSinglesName = kWalkSinglesBehindPairs * 4u + 1u; // added to a pairs table's name: its singles table's
// The entry of table `name` for the 32 stream bits cur (host and first lookups; the loop does the same in three instructions).
CG_DEV uint32_t walk_lookup(const uint32_t *walk, uint32_t name, uint32_t cur)
{
    return walk[(name >> 5 << 3) + ((cur >> (name & 31u)) & 2047u)];
}

// The table names of the walks' list entries (see chase_run_lean), written by all 64 lanes together: entry j of
// every walking lane names the AC pairs table of data unit j - 1 and the DC table of data unit j.  (The walks of
// the team form all begin at data unit 0 of their interval.)
CG_DEV void coop_lean_prepare(const CoopShared &cs, const CoopTables &t, const CoopGeom &g, uint32_t lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t walk_base = uint32_t(reinterpret_cast<uintptr_t>(t.walk)); // (32-byte aligned)
#else
    const uint32_t walk_base = 0u;
#endif
    const uint32_t walkers = g.ipw < uint32_t(kWave) ? g.ipw : uint32_t(kWave), per = uint32_t(kWave) / walkers;
    const uint32_t il = lane % walkers;
    uint32_t *list = cs.lists + il * g.lpi * g.list_cap;
    // (entry 0 holds the walk's start state; its table names are there for a long DC code right at the start)
    for (uint32_t j = lane / walkers; j <= g.dpi && lane < per * walkers; j += per) {
        const uint32_t k = (j - 1u) & 3u;
        list[4u * j + 2u] = walk_base + walk_pairs_name((t.walk_acsel >> (8u * k)) & 0xffu);
        list[4u * j + 3u] = walk_base + walk_dc_name((t.walk_dcsel >> (8u * ((k + 1u) & 3u))) & 0xffu);
        if (j)
            list[4u * j] = 0u; // "not written yet" (the walk's first word of an entry is an LDS address, never 0)
    }
}

// The walk of a lane that starts at the beginning of a data unit and only has to find where the following ones
// begin (no speculation: the team form), through the walk tables: up to two symbols per step.
//
// While it runs, the lane's list is a row of 16-byte entries, one per data unit: {address of stream word A, state
// word} -- written after every step, final once the data unit is complete -- and {this data unit's AC pair table,
// the next one's DC table}, filled in beforehand.  The state word holds the reader's shift (bits 0..15, signed:
// 32 - the bits of A that are consumed, below zero when the position has moved on into B), the size of the last
// symbol (16..20) and the zig-zag state (21..28).  At the end the entries are rewritten as the state words
// everybody else reads (position | size << kCoopPosBits).
// Per step on the GPU: 32 instructions (a lone wave issues one every four cycles at best, whatever its kind); codes
// longer than a table's prefix are looked up in the two-level tables by a branch of the same block.
CG_DEV void chase_run_lean(ChaseState &c, const ImageDesc &d, const HuffShared &s, const CoopTables &t, uint32_t *list,
                           unsigned long &steps)
{
    (void)steps;
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    const uint64_t t_fn = __builtin_readcyclecounter();
#endif
    const uint32_t j0 = uint32_t(c.lp - list), jmax = uint32_t(c.lp_max - list), k0 = (c.k8 >> 3) & 3u;
    constexpr uint32_t kStMask = 0xffu << kWalkStShift, kKeep = kStMask | 31u;
    constexpr uint32_t kEndAbove = (64u << kWalkStShift) - 1u, kNearAbove = (kWalkNear << kWalkStShift) - 1u;
    uint32_t done = j0;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t walk_base = uint32_t(reinterpret_cast<uintptr_t>(t.walk)); // (32-byte aligned)
#else
    const uint32_t walk_base = 0u;
    (void)walk_base;
#endif
    // (the table names in the entries [j0, jmax] are in place: coop_lean_prepare)
    const uint32_t wi1 = (c.p + 31u) >> 5; // (the word in front of an aligned position is never looked at: shift 0)
    uint32_t T = 32u * wi1 - c.p;          // zig-zag state 0: a DC code comes next
    auto bits_at = [&](uint32_t p_) {
        const uint32_t *wp = s.win + (p_ >> 5);
        return uint32_t(((uint64_t(wp[0]) << 32 | wp[1]) << (p_ & 31u)) >> 32);
    };
    // a code longer than the prefix, through the reference's two-level tables: one symbol
    auto resolve = [&](uint32_t cur, uint32_t st_, uint32_t k, bool &bad) {
        const uint32_t comp = comp_of_k(k & 3u);
        bad = false;
        if (st_ == 0u) {
            const uint32_t e2 = lut_lookup<true>(d, s, sel3(comp, t.dc_off[0], t.dc_off[1], t.dc_off[2]), cur);
            const uint32_t len = e2 >> 8, cat = e2 & 0xffu;
            bad = cat > 15u || len + cat > 31u; // only a hostile table has such categories
            return walk_pack(len + cat, len + cat, 1u);
        }
        const uint32_t fe = fast_entry(lut_lookup<true>(d, s, sel3(comp, t.ac_off[0], t.ac_off[1], t.ac_off[2]), cur), t.zrl);
        return walk_pack((fe >> 4) & 31u, (fe >> 4) & 31u, fe >> 9);
    };
    uint32_t ent = c.active ? walk_lookup(t.walk, walk_dc_name((t.walk_dcsel >> (8u * k0)) & 0xffu), bits_at(c.p)) : 0u;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CG_COOP_NO_ASM)
    (void)resolve; // (the hand-written block has its own)
    uint32_t alive = c.active ? 1u : 0u, code = 0u, cur_out = 0u, bad_lane = 0u;
    // which L1 table the codes of each eighth of the walk tables belong to (pairs 0, pairs 1, singles 0, dc 0,
    // singles 1, dc 1; the AC table of direct table a is L1 table 2 a + 1, the DC table of direct table i is 2 i)
    const uint32_t dc0 = 2u * (t.walk_ids & 1u), dc1 = 2u * ((t.walk_ids >> 2) & 1u);
    const uint32_t l1sel = 0x1u | 0x1u << 4 | 0x3u << 8 | 0x3u << 12 | 0x1u << 16 | dc0 << 20 | 0x3u << 24 | dc1 << 28;
    const uint32_t lb = uint32_t(reinterpret_cast<uintptr_t>(list));
    const uint32_t win = uint32_t(reinterpret_cast<uintptr_t>(s.win));
    uint32_t lpa = lb + 16u * j0, wa = win + 4u * wi1 - 4u;
    const uint32_t lpmax = lb + 16u * jmax;
#if defined(CG_COOP_STAMPS)
    c.init_cycles += __builtin_readcyclecounter() - t_fn;
    uint32_t nsteps = 0;
#define CG_LEAN_STEP "s_add_u32 %[nsteps], %[nsteps], 1\n\t"
#define CG_LEAN_STEP_OP , [nsteps] "+s"(nsteps)
#else
#define CG_LEAN_STEP
#define CG_LEAN_STEP_OP
#endif
    if (__builtin_amdgcn_ballot_w64(alive != 0u) != 0u)
        for (;;) {
#if defined(CG_COOP_STAMPS)
            const uint64_t t_in = __builtin_readcyclecounter();
#endif
            asm volatile(
                "s_mov_b64 s[74:75], exec\n\t"
                "v_cmp_ne_u32 vcc, 0, %[alive]\n\t"
                "s_and_b64 exec, exec, vcc\n\t"
                "s_mov_b64 s[84:85], exec\n\t"                        // lanes that walk in this block
                "s_cbranch_execz 4f\n\t"
                "v_mov_b32 v50, %[wa]\n\t"
                "v_mov_b32 v51, %[T]\n\t"
                "ds_read2_b32 v[42:43], v50 offset1:1\n\t"            // A, B
                "ds_read_b32 v44, v50 offset:8\n\t"                   // C
                "ds_read_b64 v[48:49], %[lp] offset:8\n"               // this data unit's AC pair table, the next one's DC table
                "1:\n\t"
                CG_LEAN_STEP
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_pk_add_u16 v51, v51, %[ent]\n\t"                   // bits off the shift, advance onto the zig-zag state
                "ds_write_b64 %[lp], v[50:51]\n\t"                    // (final when the data unit ends here)
                "v_cmp_eq_u32 s[76:77], 0, %[ent]\n\t"                // lanes that met a long code
                "v_alignbit_b32 v41, v42, v43, v51\n\t"               // (the shift is taken modulo 32)
                "v_alignbit_b32 v45, v43, v44, v51\n\t"
                "v_cmp_gt_i16 vcc, 0, v51\n\t"                        // the position has left A
                "v_cmp_lt_u32 s[72:73], %[endabove], v51\n\t"         // the data unit is complete: a DC code comes next
                "v_cmp_lt_u32 s[82:83], %[nearabove], v51\n\t"        // the next symbol could complete it: one at a time
                "v_add_u32 v47, %[singles], v48\n\t"
                "v_cndmask_b32 v41, v41, v45, vcc\n\t"                // the next 32 stream bits
                "v_cndmask_b32_e64 v46, v48, v47, s[82:83]\n\t"
                "v_cndmask_b32_e64 v46, v46, v49, s[72:73]\n\t"       // the table's name: address, shift in the low bits
                "v_bfe_u32 v45, v41, v46, 11\n\t"                     // (a 10-bit table's shift is 22: the field ends at bit 31)
                "v_and_b32 v52, 0xffffffe0, v46\n\t"                  // (v46 keeps the name: the long-code branch asks which table)
                "v_lshl_add_u32 v52, v45, 2, v52\n\t"
                "ds_read_b32 %[ent], v52\n\t"
                // ---- under that read: move on in the stream, in the list; who goes on
                "v_cndmask_b32_e64 v40, 0, 4, vcc\n\t"
                "v_add_u32 v50, v50, v40\n\t"
                "v_cndmask_b32_e64 v40, %[keep], 31, s[72:73]\n\t"
                "v_and_b32 v51, v51, v40\n\t"
                "v_cndmask_b32_e64 v40, 0, 16, s[72:73]\n\t"
                "v_add_u32 %[lp], %[lp], v40\n\t"
                "ds_read2_b32 v[42:43], v50 offset1:1\n\t"
                "ds_read_b32 v44, v50 offset:8\n\t"
                "ds_read_b64 v[48:49], %[lp] offset:8\n\t"
                "v_cmp_ge_u32 s[78:79], %[lp], %[lpmax]\n\t"          // every data unit asked for is complete
                "s_andn2_b64 exec, exec, s[78:79]\n\t"
                "s_and_b64 s[76:77], s[76:77], exec\n\t"              // (sets SCC: some walking lane met a long code)
                "s_cbranch_scc1 3f\n\t"
                "s_cbranch_execnz 1b\n\t"
                "s_mov_b32 %[code], 0\n\t"
                "s_branch 5f\n"
                "3:\n\t"
                // ---- (rare) lanes s[76:77] met a code longer than their table's prefix: the reference's two-level
                // tables decide (lut_lookup<true>), one symbol, and the entry is made here (walk_pack of fast_entry's
                // fields; a DC category above 15 -- a hostile table -- ends that lane's walk)
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_mov_b64 s[86:87], exec\n\t"                        // the walking lanes
                "s_mov_b64 exec, s[76:77]\n\t"
                // which table the code belongs to: an AC code (zig-zag state above 0) to the one just looked at
                // (v46); a DC code -- the lookup of this pass took the new data unit's AC table for it -- to the one the
                // entry in front of the lane's current one names (coop_lean_prepare)
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
                "v_add_u32 v54, v54, v53\n\t"                         // size (fast_entry's sum: may pass 31 with a hostile table),
                "v_lshrrev_b32 v55, 4, v47\n\t"
                "v_add_u32 v55, 1, v55\n\t"                           // advance: run + 1,
                "v_cmp_eq_u32 vcc, 0xf0, v47\n\t"
                "v_cndmask_b32_e64 v55, v55, %[zrl], vcc\n\t"         // ZRL,
                "v_cmp_eq_u32 vcc, 0, v47\n\t"
                "v_cndmask_b32_e64 v55, v55, 64, vcc\n\t"             // end-of-block
                "v_lshrrev_b32 v53, 5, v54\n\t"
                "v_or_b32 v55, v55, v53\n\t"                          // (fast_entry's fields overlap when the size passes 31)
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
                "s_mov_b32 %[code], 0\n\t"
                "s_branch 5f\n"
                "4:\n\t"
                "s_mov_b32 %[code], 0\n"
                "5:\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_mov_b64 s[72:73], exec\n\t"                        // lanes still walking
                "s_mov_b64 exec, s[74:75]\n\t"
                "v_cndmask_b32_e64 %[alive], 0, 1, s[72:73]\n\t"
                "v_cndmask_b32_e64 %[wa], %[wa], v50, s[84:85]\n\t"   // (the others keep theirs)
                "v_cndmask_b32_e64 %[T], %[T], v51, s[84:85]\n\t"
                "v_mov_b32 %[cur], v41\n\t"                           // the stream bits at the position (lanes that walked)
                : [lp] "+v"(lpa), [ent] "+v"(ent), [T] "+v"(T), [wa] "+v"(wa), [alive] "+v"(alive), [code] "=s"(code),
                  [cur] "=v"(cur_out), [bad] "+v"(bad_lane) CG_LEAN_STEP_OP
                : [endabove] "s"(kEndAbove), [nearabove] "s"(kNearAbove), [singles] "v"(kWalkSinglesName), [keep] "v"(kKeep),
                  [lpmax] "v"(lpmax), [walkbase] "s"(walk_base), [l1sel] "s"(l1sel),
                  [l1base] "s"(uint32_t(reinterpret_cast<uintptr_t>(s.l1))), [l2base] "s"(uint32_t(reinterpret_cast<uintptr_t>(s.l2))),
                  [l2n] "s"(d.l2_entries), [zrl] "v"(t.zrl)
                : "memory", "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
                  "v52", "v53", "v54", "v55", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s82", "s83", "s84", "s85",
                  "s86", "s87", "s88", "s89");
#if defined(CG_COOP_STAMPS)
            c.loop_cycles += __builtin_readcyclecounter() - t_in;
            c.loop_entries++;
            c.loop_steps = nsteps;
#endif
            (void)cur_out;
            if (code == 0u)
                break;
        }
    c.flags |= bad_lane ? kStopAnomaly : 0u;
    done = (lpa - lb) / 16u;
    c.lean_p = 8u * (wa - win) + 32u - (T & 31u);
#else
    if (c.active) {
        uint32_t wa = wi1 - 1u, j = j0; // (word index of A; -1 for a walk that starts at position 0)
        while (j < jmax) {
            steps++;
            const uint32_t p_now = 32u * (wa + 1u) - (T & 31u);
            if (ent == 0u) {
                CG_COUNT(escapes);
                bool bad;
                ent = resolve(bits_at(p_now), T >> kWalkStShift, k0 + j - j0, bad);
                if (bad) {
                    c.flags |= kStopAnomaly;
                    break;
                }
            }
            // the packed add: two independent 16-bit sums
            const uint32_t T1 = ((T + (ent & 0xffff0000u)) & 0xffff0000u) | ((T + ent) & 0xffffu);
            const int32_t sn = int32_t(int16_t(T1 & 0xffffu));
            const uint32_t p_next = 32u * (wa + 1u) - uint32_t(sn);
            const bool du_end = T1 > kEndAbove, near = T1 > kNearAbove;
            list[4u * j] = wa + kLeanHostWordBias; // (never 0: "not written yet", see coop_lean_prepare)
            list[4u * j + 1u] = T1;
            const uint32_t name = du_end ? list[4u * j + 3u] : list[4u * j + 2u] + (near ? kWalkSinglesName : 0u);
            ent = walk_lookup(t.walk, name, bits_at(p_next));
            wa += sn < 0 ? 1u : 0u;
            T = T1 & (du_end ? 31u : kKeep);
            j += du_end ? 1u : 0u;
        }
        done = j;
        c.lean_p = 32u * (wa + 1u) - (T & 31u);
    }
#endif
    c.lean_walked = c.active;
    c.lean_j0 = j0;
    c.lean_done = done;
    c.active = false;
}

// Position and state word of a 16-byte entry {w, Tj} of chase_run_lean.
CG_DEV uint32_t lean_entry_pos(const HuffShared &s, uint32_t w, uint32_t Tj)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 8u * (w - uint32_t(reinterpret_cast<uintptr_t>(s.win))) + 32u - uint32_t(int32_t(int16_t(Tj & 0xffffu)));
#else
    (void)s;
    return 32u * (w - kLeanHostWordBias + 1u) - uint32_t(int32_t(int16_t(Tj & 0xffffu)));
#endif
}
CG_DEV uint32_t lean_entry_state(uint32_t pos, uint32_t Tj) { return (pos & kCoopPosMask) | (((Tj >> kWalkLastShift) & 31u) << kCoopPosBits); }

// The walk went as it has to for its entries to be used: every data unit found, all of them starting inside the part
// of the window where a data unit may begin.  (Beyond the interval's own end is fine: the reference decodes as many
// data units as the interval should have, from whatever follows -- quirk Q2's drift makes some intervals run over.)
CG_DEV bool chase_lean_regular(const ChaseState &c, const HuffShared &s, uint32_t *list)
{
    if (c.flags & kStopAnomaly)
        return false;
    return !c.lean_walked || (c.lean_done == uint32_t(c.lp_max - list) && c.lean_p < coop_hard_end(s));
}

// Word offset (inside the interval) at which subsequence j of `count` begins.
CG_DEV uint32_t coop_sub_start(uint32_t len_words, uint32_t j, const CoopGeom &g)
{
    return j >= g.count ? len_words : (len_words * j) / g.count; // (len_words <= kCoopMaxWindow, j < 16)
}

// The first subsequence behind j that begins further into the interval than j does (count: none).
CG_DEV uint32_t coop_next_sub(uint32_t len_words, uint32_t j, const CoopGeom &g)
{
    const uint32_t here = coop_sub_start(len_words, j, g);
    uint32_t n = j + 1u;
    while (n < g.count && coop_sub_start(len_words, n, g) == here)
        n++;
    return n;
}

// Where a walk that covers the stretch up to subsequence `next_sub` ends.
CG_DEV void chase_set_end(ChaseState &c, const HuffShared &s, const CoopGeom &g, uint32_t start_rel, uint32_t len_words)
{
    c.sub_end = 32u * (start_rel + coop_sub_start(len_words, c.next_sub, g));
    // a lane with a successor walks a little further, so that their lists overlap; the last one stops at the end
    const uint32_t want = c.next_sub < g.count ? c.sub_end + kCoopMargin : c.sub_end;
    const uint32_t stop = umin(want, coop_hard_end(s));
    c.stop_p = stop ? stop : 1u; // (never 0: the hand-written loop tests "end position >= stop_p" with 0 for "no end here")
}

// What lane `tl` of an interval does in the first round.  Lane 0 walks from the interval's start; lanes
// 1 + 4 (j - 1) + h, j = 1..count-1, walk subsequence j assuming data unit h.
// start_rel / len_words: the interval's first word inside the window, and its length.
CG_DEV void chase_assign(ChaseState &c, const HuffShared &s, const CoopGeom &g, uint32_t tl, uint32_t start_rel,
                         uint32_t len_words, bool exists, uint32_t *list)
{
    uint32_t j = 0u, h = 0u;
    bool used = exists && tl == 0u;
#if defined(CG_COOP_STAMPS)
    c.loop_cycles = c.init_cycles = c.tail_cycles = 0;
    c.loop_steps = c.loop_entries = 0;
#endif
    if (tl >= 1u && tl <= 4u * (g.count - 1u)) {
        j = 1u + (tl - 1u) / 4u;
        h = (tl - 1u) & 3u;
        used = exists;
    }
    const uint32_t begin = coop_sub_start(len_words, j, g);
    // subsequences that begin where the one before them begins (a tiny interval) are not walked twice
    if (j > 0u && begin == coop_sub_start(len_words, j - 1u, g))
        used = false;
    // a start outside the window (corrupt start positions) is nobody's to walk: the interval goes serial
    const bool inside = start_rel + begin + kDuWordSlack <= s.win_len;
    c.p = inside && used ? 32u * (start_rel + begin) : 0u;
    c.k8 = 8u * h;
    c.lp = list;
    c.lp_max = list;
    c.flags = used && !inside ? kStopAnomaly : 0u;
    c.next_sub = g.count;
    c.sub_end = c.stop_p = 0u;
    c.used = used && inside;
    c.active = false;
    c.s = 1u;
    c.k0 = 0u;
    if (!c.used)
        return;
    c.next_sub = coop_next_sub(len_words, j, g);
    chase_set_end(c, s, g, start_rel, len_words);
    c.active = true;
    if (j == 0u) {
        // lane 0 stands at the start of data unit 0, and needs the starts of the interval's data units only
        c.s = 0u;
        list[0] = c.p;
        c.lp = list + 1;
        c.lp_max = list + umin(g.max_entries, g.dpi);
        c.active = c.lp < c.lp_max;
    } else {
        // inside the AC part of data unit h, by assumption: its first entry starts data unit h + 1
        c.k0 = (h + 1u) & 3u;
        c.lp_max = list + g.max_entries;
    }
}

// ---------------------------------------------------------------------------
// 2. Validate
// ---------------------------------------------------------------------------

// Lane index (inside the interval) of the walk of subsequence j under assumption h
CG_DEV uint32_t coop_spec_lane(uint32_t j, uint32_t h) { return 1u + 4u * (j - 1u) + h; }

CG_DEV void coop_publish(const ChaseState &c, const CoopShared &cs, const CoopGeom &g, uint32_t lane)
{
    const uint32_t anomaly = c.flags | ((c.used && c.p >= coop_hard_end(cs.h)) ? kStopAnomaly : 0u);
    const uint32_t n = uint32_t(c.lp - (cs.lists + lane * g.list_cap));
#if defined(CG_EMUL_STATS)
    if (getenv("EMUL_COOP_FAILS") && anomaly)
        fprintf(stderr, "anomaly lane %u: flags %u used %d p %u hard end %u (window %u words) stop_p %u sub_end %u entries %u of %u\n", lane, c.flags, int(c.used), c.p,
                coop_hard_end(cs.h), cs.h.win_len, c.stop_p, c.sub_end, n, uint32_t(c.lp_max - (cs.lists + lane * g.list_cap)));
#endif
    cs.lane_n[lane] = n | (anomaly << 8) | (c.k0 << 16);
    cs.link[lane] = 0u;
}

// Every chasing lane looks among its last entries, at or beyond the end of its own stretch, for one that a lane
// of the next subsequence lists too (among its first entries).  lane0: first lane of the interval.
CG_DEV void coop_find_link(const ChaseState &c, const CoopShared &cs, const CoopGeom &g, uint32_t lane, uint32_t lane0)
{
    const uint32_t *mine = cs.lists + lane * g.list_cap;
    const uint32_t n = uint32_t(c.lp - mine);
    if (n == 0u || c.next_sub >= g.count)
        return;
    const uint32_t first = n > kCoopTail ? n - kCoopTail : 0u;
    uint32_t own[kCoopTail];
    bool own_ok[kCoopTail];
#pragma unroll
    for (uint32_t a = 0; a < kCoopTail; a++) {
        own[a] = mine[first + a] & kCoopStateMask; // (first + a <= n <= max_entries: inside the list)
        own_ok[a] = first + a < n && (own[a] & kCoopPosMask) >= c.sub_end;
    }
    uint32_t best = 0u;
#pragma unroll
    for (uint32_t h = 0; h < 4u; h++) {
        const uint32_t other = lane0 + coop_spec_lane(c.next_sub, h);
        const uint32_t info = cs.lane_n[other];
        const uint32_t n_other = umin(info & 0xffu, kCoopHead), k_other = (info >> 16) & 3u;
        const uint32_t *theirs = cs.lists + other * g.list_cap;
#pragma unroll
        for (uint32_t q = 0; q < kCoopHead; q++) {
            const uint32_t e = theirs[q] & kCoopStateMask;
#pragma unroll
            for (uint32_t a = 0; a < kCoopTail; a++) {
                // same position, same symbol in front, same index inside the MCU
                const bool hit = q < n_other && own_ok[a] && own[a] == e && ((c.k0 + first + a - k_other - q) & 3u) == 0u;
                // the earliest of the lane's own entries wins
                if (hit && (best == 0u || first + a < (best >> 24)))
                    best = 1u | (other << 8) | (q << 16) | ((first + a) << 24);
            }
        }
    }
    cs.link[lane] = best;
#if defined(CG_EMUL_STATS)
    if (lane == lane0 || true) {
        // (statistics only) would any of the lane's entries have matched any entry of a successor?
        bool full = false;
        for (uint32_t h = 0; h < 4u && !full; h++) {
            const uint32_t other = lane0 + coop_spec_lane(c.next_sub, h);
            const uint32_t info = cs.lane_n[other];
            const uint32_t n_o = info & 0xffu, k_o = (info >> 16) & 3u;
            for (uint32_t q = 0; q < n_o && !full; q++)
                for (uint32_t i = 0; i < n && !full; i++)
                    full = (mine[i] & kCoopStateMask) == (cs.lists[other * g.list_cap + q] & kCoopStateMask) &&
                           ((c.k0 + i - k_o - q) & 3u) == 0u && (mine[i] & kCoopPosMask) >= c.sub_end;
        }
        if (lane == lane0) {
            CG_COOP_COUNT(link_tries, 1);
            CG_COOP_COUNT(link_ok, best ? 1 : 0);
            CG_COOP_COUNT(link_full_ok, (!best && full) ? 1 : 0);
            CG_COOP_COUNT(link_none, (!best && !full) ? 1 : 0);
        }
    }
#endif
}

// First lane of every interval that is not settled yet: follows the links from the chain's current head
// (itself at first; later the lane that walked on) and notes, for this round, which entries of which lane
// continue the interval's sequence of data units -- one word per stretch: lane | first entry << 6 |
// entries << 13 | index of the first data unit << 20 -- and how the interval stands.
//   head: the lane the chain continues with; from: its first entry that has not been noted yet;
//   du: data units whose start has been noted so far
CG_DEV void coop_follow(const CoopShared &cs, const CoopGeom &g, uint32_t il, uint32_t head, uint32_t from, uint32_t du)
{
    uint32_t x = head, nseg = 0u;
    uint32_t *seg = cs.seg + il * g.count;
    uint32_t v = kVerdictSerial;
    for (uint32_t hops = 0; hops < g.count; hops++) {
        const uint32_t info = cs.lane_n[x], n = info & 0xffu, stop = info >> 8;
        const uint32_t l = cs.link[x];
        const bool linked = (l & 1u) && (l >> 24) >= from;
        const uint32_t upto = linked ? l >> 24 : (n > from ? n : from);
        seg[nseg++] = x | (from << 6) | ((upto - from) << 13) | (du << 20); // (entries of a list < 128, data units < 4096)
        du += upto - from;
        if (du >= g.dpi) {
            v = kVerdictDone;
            break;
        }
        if (!linked) {
#if defined(CG_EMUL_STATS)
            if (getenv("EMUL_COOP_FAILS")) {
                // (analysis only) the failed boundary: the chain lane's last entries and the successors' first ones
                fprintf(stderr, "fail il %u lane %u n %u du %u :", il, x, n, du);
                for (uint32_t i = n > 4 ? n - 4 : 0; i < n; i++)
                    fprintf(stderr, " %u/%u", cs.lists[x * g.list_cap + i] & kCoopPosMask, (cs.lists[x * g.list_cap + i] >> kCoopPosBits) & 31u);
                const uint32_t lane0 = x / g.lpi * g.lpi;
                const uint32_t j = x == lane0 ? 0u : 1u + (x - lane0 - 1u) / 4u;
                for (uint32_t h = 0; h < 4u && j + 1u < g.count; h++) {
                    const uint32_t o = lane0 + coop_spec_lane(j + 1u, h);
                    fprintf(stderr, " | h%u k0=%u n=%u:", h, (cs.lane_n[o] >> 16) & 3u, cs.lane_n[o] & 0xffu);
                    for (uint32_t q = 0; q < (cs.lane_n[o] & 0xffu) && q < 6u; q++)
                        fprintf(stderr, " %u/%u", cs.lists[o * g.list_cap + q] & kCoopPosMask, (cs.lists[o * g.list_cap + q] >> kCoopPosBits) & 31u);
                }
                fprintf(stderr, "\n");
            }
#endif
            // the chain ends with lane x: it walks on, unless it cannot
            v = ((stop & kStopAnomaly) || du == 0u) ? kVerdictSerial : (kVerdictContinue | (x << 8) | (du << 16));
#if defined(CG_EMUL_STATS)
            if (getenv("EMUL_COOP_FAILS") && v == kVerdictSerial)
                fprintf(stderr, "serial il %u: chain ends with lane %u, stop flags %u, data units noted %u of %u\n", il, x, stop, du, g.dpi);
#endif
            break;
        }
        x = (l >> 8) & 0xffu;
        from = (l >> 16) & 0xffu;
    }
    cs.nseg[il] = nseg;
    cs.verdict[il] = v;
}

// Every lane fetches the start state of its own data unit(s), if this round's stretches hold it: data unit tl of
// interval il of the wave, state word `slot`.
CG_DEV void coop_emit(const CoopShared &cs, const CoopGeom &g, uint32_t il, uint32_t tl, uint32_t slot)
{
    const uint32_t nseg = cs.nseg[il];
    const uint32_t *seg = cs.seg + il * g.count;
    for (uint32_t i = 0; i < nseg; i++) {
        const uint32_t sg = seg[i];
        const uint32_t x = sg & 63u, from = (sg >> 6) & 127u, cnt = (sg >> 13) & 127u, du0 = sg >> 20;
        if (tl >= du0 && tl - du0 < cnt)
            cs.du_state[slot] = cs.lists[x * g.list_cap + from + (tl - du0)] & kCoopStateMask;
    }
}

// After walks through the walk tables that all went regularly (chase_lean_regular): every lane fetches the start
// state of its data unit(s) from the walker's 16-byte entries -- no lists, no links to follow.
CG_DEV void coop_lean_emit(const CoopShared &cs, const CoopGeom &g, uint32_t lane)
{
    for (uint32_t pass = 0; pass < g.rounds; pass++) {
        uint32_t il, tl;
        coop_du_of(g, pass * uint32_t(kWave) + lane, il, tl);
        if (il >= g.intervals)
            continue;
        const uint32_t *list = cs.lists + il * g.lpi * g.list_cap;
        uint32_t state = list[0] & kCoopStateMask;
        if (tl != 0u) {
            const uint32_t w = list[4u * tl], Tj = list[4u * tl + 1u];
            state = lean_entry_state(lean_entry_pos(cs.h, w, Tj), Tj);
        }
        cs.du_state[pass * uint32_t(kWave) + lane] = state; // (of no consequence where the interval's verdict is "serial")
    }
}

// A lane whose interval's verdict names it walks on, through the next subsequence.  It stands at the state of
// its last entry (noted already): that entry becomes entry 0 of its new list, the ones it lists now follow.
// Every other lane rests.
CG_DEV void coop_continue(ChaseState &c, const CoopShared &cs, const CoopGeom &g, uint32_t il, uint32_t lane,
                          uint32_t start_rel, uint32_t len_words, uint32_t *list)
{
    const uint32_t v = cs.verdict[il];
    c.active = false;
    if ((v & 0xffu) != kVerdictContinue || ((v >> 8) & 0xffu) != lane || c.lp == list)
        return;
    const uint32_t settled = v >> 16;
    const uint32_t n = uint32_t(c.lp - list);
    list[0] = list[n - 1u] & kCoopStateMask;
    c.k0 = (c.k0 + n - 1u) & 3u;
    c.lp = list + 1;
    if (c.next_sub < g.count)
        c.next_sub = coop_next_sub(len_words, c.next_sub, g);
    chase_set_end(c, cs.h, g, start_rel, len_words);
    c.lp_max = list + umin(g.max_entries, 1u + (g.dpi - settled));
    c.active = c.lp < c.lp_max && c.p < coop_hard_end(cs.h);
}

// ---------------------------------------------------------------------------
// 3. Decode: one lane per data unit
// ---------------------------------------------------------------------------

CG_DEV void lds_read_done(uint32_t &a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory");
#else
    (void)a;
#endif
}

CG_DEV void copy_zero_levels(const ImageDesc &d, uint32_t comp, int16_t *slot16)
{
    auto *z = CG_GLOBAL(const int16_t, &d.zero_du[0][0]) + comp * kRetained;
    for (int i = 1; i < kRetained; i++)
        slot16[i] = z[i];
}

CG_DEV int32_t zero_diff(const ImageDesc &d, uint32_t comp)
{
    return CG_GLOBAL(const int16_t, &d.zero_du[0][0])[comp * kRetained];
}

// The data unit whose start state is `state`, of component comp, into the lane's zeroed slot; returns its DC
// difference.  The decoder is the fused kernel's: fast mode, with the exact reader behind it.  underflow: the
// reference reader runs dry inside this data unit's DC code (quirk Q1) -- the difference then comes from what
// is left of its buffer and the AC levels from zeros, and every later data unit of the interval decodes from
// zeros too (the caller sees to that).
// hostile: what the reference's reader finds at the underflowing DC code is a category above 15 (only a hostile
// table has one): its shifts wrap modulo 32 and the buffer does not run empty -- the interval has to go through
// the serial decoder, which follows the reference's reader literally.
CG_DEV int32_t coop_decode_du(const ImageDesc &d, const HuffShared &s, const CoopTables &t, uint32_t state, uint32_t comp,
                              int16_t *slot16, bool &underflow, bool &hostile)
{
    underflow = hostile = false;
    if (state & kCoopUnset)
        return 0;
    const uint32_t p = state & kCoopPosMask, rel = p >> 5, sh = p & 31u, tl = (state >> kCoopPosBits) & 31u;
    EntropyState e;
    e.r.buf = uint64_t(s.win[rel] << sh) << 32;
    e.r.left = 32u - sh;
    e.r.next_word = 0u;
    e.wptr = s.win + rel + 1u;
    e.r.pre = *e.wptr;
    e.fast = true;
    e.ref_left = 32u + ((tl - p) & 31u) - tl; // the reference reader's `left` here (see fast_ac)
    e.pred0 = e.pred1 = e.pred2 = 0;
    fast_refill(e);
    lds_read_done(e.r.pre);
    const uint32_t dc_off = sel3(comp, t.dc_off[0], t.dc_off[1], t.dc_off[2]);
    const uint32_t ac_off = sel3(comp, t.ac_off[0], t.ac_off[1], t.ac_off[2]);
    int32_t diff = 0;
    if (fast_dc(e, d, s, dc_off, diff)) {
        fast_ac(e, d, s, ac_off, sel3(comp, t.fast_base[0], t.fast_base[1], t.fast_base[2]), slot16);
        CG_COUNT(fast_dus);
        return diff;
    }
    {
        // the code the reference sees: its buffer holds ref_left bits, zeros behind them (see fast_dc)
        const uint32_t seen = e.ref_left < 32u ? e.ref_left : 32u;
        const uint32_t cut = reader_cur(e.r) & ~uint32_t(uint64_t(0xffffffffu) >> seen);
        hostile = t.standard || (lut_lookup<true>(d, s, dc_off, cut) & 0xffu) > 15u;
    }
    if (hostile)
        return 0;
    CG_COUNT(left_underflow);
    CG_COUNT(exact_dus);
    underflow = true;
    leave_fast_mode(e, d, s);
    diff = decode_dc_diff(e.r, d, s, dc_off);
    copy_zero_levels(d, comp, slot16);
    return diff;
}

// The slot's coefficients without clearing it (the slots are not used again).
CG_DEV void read_slot(const uint8_t *slot, uint32_t (&rec)[kRetained / 2])
{
    const SlotVec *p = reinterpret_cast<const SlotVec *>(slot);
#pragma unroll
    for (int i = 0; i < kRetained / 8; i++) {
        const SlotVec v = p[i];
        rec[4 * i + 0] = v.x;
        rec[4 * i + 1] = v.y;
        rec[4 * i + 2] = v.z;
        rec[4 * i + 3] = v.w;
    }
}

// Lane j composites pixel columns 4 (j % 4) .. +3 of MCU first_mcu + j / 4, all 8 rows: a wave-wide 16-byte
// store covers 64 contiguous bytes per MCU (1 KB of a pixel row when the 16 MCUs lie side by side).
// px: the wave's 64 sample records, kPxSlotWords apart, record i = data unit i of the wave.
// stride: MCUs between the MCUs of neighbouring lane quads (1: side by side)
CG_DEV void coop_composite(const ImageDesc &d, const uint32_t *px, uint32_t first_mcu, uint32_t mcus, uint32_t j, uint32_t stride = 1u)
{
    const uint32_t m = j >> 2, q = j & 3u;
    if (m >= mcus)
        return;
    const uint32_t mcu = first_mcu + m * stride;
    const uint32_t mx = mcu % d.width_mcus, my = mcu / d.width_mcus;
    const uint32_t x0 = mx * 16u + q * 4u;
    if (x0 >= d.out_w)
        return;
    const uint32_t *ydu = px + (m * 4u + (q >> 1)) * kPxSlotWords;
    const uint32_t *cbdu = px + (m * 4u + 2u) * kPxSlotWords;
    const uint32_t *crdu = px + (m * 4u + 3u) * kPxSlotWords;
    const bool whole = x0 + 3u < d.out_w && (d.out_pitch & 15u) == 0u;
    // (all of the lane's samples first: the rows' conversions then do not wait for LDS one after the other)
    uint32_t ys[8], cbs[8], crs[8];
#pragma unroll
    for (uint32_t row = 0; row < 8; row++) {
        ys[row] = ydu[row * 2u + (q & 1u)];
        cbs[row] = cbdu[row * 2u + (q >> 1)];
        crs[row] = crdu[row * 2u + (q >> 1)];
    }
    const uint32_t rows = d.out_h - umin(d.out_h, my * 8u); // (rows of this MCU inside the output, if fewer than 8)
    uint8_t *p = d.out + size_t(my * 8u) * d.out_pitch + size_t(x0) * 4u;
#pragma unroll
    for (uint32_t row = 0; row < 8; row++) {
        const Vec4u o = rgba_quad(ys[row], cbs[row] >> ((q & 1u) * 16u), crs[row] >> ((q & 1u) * 16u));
        if (row < rows) {
            if (whole) {
                store_pixels<true>(p, o);
            } else {
                auto *w = CG_GLOBAL(uint32_t, reinterpret_cast<uint32_t *>(p));
                w[0] = o.x;
                if (x0 + 1u < d.out_w)
                    w[1] = o.y;
                if (x0 + 2u < d.out_w)
                    w[2] = o.z;
                if (x0 + 3u < d.out_w)
                    w[3] = o.w;
            }
        }
        p += d.out_pitch;
    }
}

// ---------------------------------------------------------------------------
// The wave: phases in which the lanes work side by side, LDS in between
// ---------------------------------------------------------------------------
// One body for the GPU and for tests/emul: LANES = 1 on the GPU, where the function runs in every lane of the
// wave with `my_lane` = its index and the "for every lane" loops below run once; LANES = 64 in the emulation,
// where one call plays the whole wave, lane after lane, phase by phase.

#if defined(__HIP_DEVICE_COMPILE__)
// lanes of a wave exchange data through LDS between two phases: the hardware keeps a wave's LDS operations in
// order, the compiler has to as well
#define CG_WAVE_SYNC()                                                                                 \
    do {                                                                                               \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                         \
        __builtin_amdgcn_wave_barrier();                                                               \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                         \
    } while (0)
#else
#define CG_WAVE_SYNC() do { } while (0)
#endif

struct CoopLane {
    uint32_t lane, il, tl, lane0; // index in the wave; interval of the wave; index inside it; the interval's first lane
    bool exists;                  // the interval exists (the image's last wave may have fewer)
    uint32_t start_rel, len_words; // the interval's first word inside the window, its length
};

CG_DEV void coop_lane(const ImageDesc &d, const HuffShared &s, const CoopGeom &g, uint32_t lane, CoopLane &L)
{
    L.lane = lane;
    L.il = lane / g.lpi;
    L.tl = lane - L.il * g.lpi;
    L.lane0 = L.il * g.lpi;
    L.exists = L.il < g.intervals; // (the lanes behind the last interval's, where 64 is no multiple of lpi: none)
    const uint32_t interval = g.first_interval + (L.exists ? L.il : 0u);
    const uint32_t ws = interval < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[interval] : 0u;
    // (the image's last interval too ends where the next one begins, if the scan goes on behind it -- an image whose MCUs
    // the restart interval does not divide: lib.rs:784 counts whole intervals, the scan's preprocessing notes every
    // marker; up to the scan's end it would be that much longer, its subsequences laid over data that is not its own,
    // three times the walk -- tools/coop_tail_stamps.py)
    const uint32_t nx = interval + 1u < d.nstarts ? CG_GLOBAL(const uint32_t, d.starts)[interval + 1u] : d.nwords;
    const uint32_t we = (interval + 1u < d.total_intervals || (nx > ws && nx <= d.nwords)) ? nx : d.nwords;
    L.start_rel = ws >= s.win_base ? umin(ws - s.win_base, 0x7fffffu) : 0x7fffffu; // (outside the window: no walk)
    L.len_words = we > ws ? umin(we - ws, kCoopMaxWindow) : 0u;
}

// diagnostic builds (-DCG_COOP_STAMPS): cycles per phase of every wave, parked in the (otherwise unused) dc buffer
struct CoopClock {
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    uint64_t tprev, stamp[8], wall0, extra[5];
#endif
};
CG_DEV void coop_clock_start(CoopClock &clk)
{
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    clk.tprev = __builtin_readcyclecounter();
    clk.wall0 = wall_clock64(); // (100 MHz: when this wave got going, and below when it was done, on the device's clock)
    for (int i = 0; i < 8; i++)
        clk.stamp[i] = 0;
    for (int i = 0; i < 5; i++)
        clk.extra[i] = 0;
#else
    (void)clk;
#endif
}
CG_DEV void coop_clock_store(const CoopClock &clk, const ImageDesc &d, uint32_t wave_index, uint32_t lane)
{
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (lane == 0u && d.dc) {
        uint64_t *o = reinterpret_cast<uint64_t *>(d.dc) + size_t(wave_index) * 16u;
        for (int i = 0; i < 8; i++)
            o[i] = clk.stamp[i];
        o[8] = clk.wall0;
        o[9] = wall_clock64();
        o[10] = clk.extra[0];
        o[11] = clk.extra[1];
        o[12] = clk.extra[2];
        o[13] = clk.extra[3];
        o[14] = clk.extra[4];
    }
#else
    (void)clk, (void)d, (void)wave_index, (void)lane;
#endif
}
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define CG_COOP_STAMP(i) do { const uint64_t now_ = __builtin_readcyclecounter(); clk.stamp[i] += now_ - clk.tprev; clk.tprev = now_; } while (0)
#else
#define CG_COOP_STAMP(i) do { } while (0)
#endif

template <int LANES>
CG_DEV bool coop_any(const bool (&flag)[LANES])
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(flag[0]) != 0u;
#else
    bool any = false;
    for (int i = 0; i < LANES; i++)
        any = any || flag[i];
    return any;
#endif
}

#define CG_EACH_LANE for (int li = 0; li < LANES; li++)

// Phases 1 and 2 for the intervals of one walk (CoopGeom): the start state of every data unit into cs.du_state,
// how every interval stands into cs.verdict.
template <int LANES>
CG_DEV void coop_walk_422(const ImageDesc &d, const CoopShared &cs, const CoopTables &t, const CoopGeom &g,
                          uint32_t my_lane, uint32_t wave_index, CoopClock &clk, const CoopLane *ready = nullptr,
                          bool quarters = false)
{
    const HuffShared &s = cs.h;
    CoopLane L[LANES];
    ChaseState c[LANES];
    bool active[LANES];
    (void)wave_index;
    (void)clk;
    CG_EACH_LANE
    {
        if (LANES == 1 && ready)
            L[li] = *ready; // (looked up while the tables were on their way)
        else
            coop_lane(d, s, g, LANES == 1 ? my_lane : uint32_t(li), L[li]);
        const uint32_t lane = L[li].lane;
        for (uint32_t pass = 0; pass < g.rounds; pass++)
            cs.du_state[pass * uint32_t(kWave) + lane] = kCoopUnset;
        cs.verdict[lane] = 0u;
        cs.nseg[lane] = 0u;
        cs.dead_from[lane] = 0xffffu;
        chase_assign(c[li], s, g, L[li].tl, L[li].start_rel, L[li].len_words, L[li].exists, cs.lists + lane * g.list_cap);
        active[li] = c[li].active;
    }
    if (LANES != 1)
        CG_COOP_COUNT(intervals, g.intervals);
    CG_COOP_STAMP(0);
    // nobody speculates: the walk tables' loop (chase_run_lean), every walking lane to the end of its interval (its
    // list of 16-byte entries has the room: 20 bytes per data unit of lane space for 16 per entry)
    const bool lean = t.walk != nullptr && t.walk_ok && g.count == 1u;
    if (lean) {
        CG_EACH_LANE
        {
            uint32_t *list = cs.lists + L[li].lane * g.list_cap;
            if (c[li].active)
                c[li].lp_max = list + g.dpi;
        }
        CG_EACH_LANE coop_lean_prepare(cs, t, g, L[li].lane);
        CG_WAVE_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
        if (quarters) {
            // the waves that decode the first three quarters of the intervals may look at the lists from now on
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((LANES == 1 ? my_lane : 0u) == 0u)
                __hip_atomic_store(cs.flags + kTeamWalk, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#endif
    }

    // ---- 1 + 2: walk, link, follow; lanes that have to walk on do so, until every interval is settled ----
    for (uint32_t round = 0; round <= g.dpi + 1u; round++) {
        // (the walk tables' round also when no lane walks -- every start outside a window cut short: it still has to
        // give the intervals their verdicts)
        if (!coop_any<LANES>(active) && !(lean && round == 0u))
            break;
        if (LANES != 1)
            CG_COOP_COUNT(rounds, 1);
        {
            unsigned long most = 0, most_true = 0;
            CG_EACH_LANE
            {
                unsigned long steps = 0;
                if (lean && round == 0u)
                    chase_run_lean(c[li], d, s, t, cs.lists + L[li].lane * g.list_cap, steps);
                else
                    chase_run(c[li], d, s, t, steps);
                CG_COOP_COUNT(chase_steps, steps);
                most = steps > most ? steps : most;
                if (L[li].tl == 0u) {
                    CG_COOP_COUNT(true_steps, steps);
                    most_true = steps > most_true ? steps : most_true;
                }
                if (steps)
                    CG_COOP_COUNT(hist[steps / 16 < 15 ? steps / 16 : 15], 1);
            }
            CG_COOP_COUNT(wave_steps, most);
            CG_COOP_COUNT(true_max, most_true);
#if defined(CG_EMUL_STATS)
            if (const char *path = getenv("EMUL_COOP_WAVE_STEPS")) { // (analysis only) steps of every wave and round
                static FILE *f = fopen(path, "w");
                if (f)
                    fprintf(f, "%u %u %lu\n", wave_index, round, most);
            }
#endif
            (void)most_true;
        }
        if (lean) {
            // An interval whose walk went regularly -- all data units found, inside the window, in front of the
            // interval's end: nearly all of them -- hands its entries straight to the decoding lanes; any other one
            // (corrupt stream, hostile table, window too small) goes to the serial decoder, which follows the reference
            // literally.
            CG_EACH_LANE
            {
                if (L[li].tl == 0u && L[li].exists) {
                    // (a maximum: with quarters, a decoding lane may have said "serial" already -- hostile table)
                    const uint32_t v = chase_lean_regular(c[li], s, cs.lists + L[li].lane * g.list_cap) ? kVerdictDone : kVerdictSerial;
#if defined(__HIP_DEVICE_COMPILE__)
                    atomicMax(&cs.verdict[L[li].il], v);
#else
                    cs.verdict[L[li].il] = cs.verdict[L[li].il] > v ? cs.verdict[L[li].il] : v;
#endif
                }
            }
            CG_WAVE_SYNC();
            if (!quarters) { // (with quarters every decoding lane reads its entry itself)
                CG_EACH_LANE coop_lean_emit(cs, g, L[li].lane);
                CG_WAVE_SYNC();
            }
            if (LANES != 1)
                CG_COOP_COUNT(direct, 1);
            CG_COOP_STAMP(1);
            return;
        }
        CG_COOP_STAMP(1);
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
        clk.extra[3] += c[0].init_cycles;
        clk.extra[4] += c[0].tail_cycles;
#endif
        CG_EACH_LANE coop_publish(c[li], cs, g, L[li].lane);
        CG_WAVE_SYNC();
        CG_EACH_LANE coop_find_link(c[li], cs, g, L[li].lane, L[li].lane0);
        CG_WAVE_SYNC();
        CG_EACH_LANE
        {
            if (L[li].tl != 0u || !L[li].exists)
                continue;
            const uint32_t v = cs.verdict[L[li].il];
            if (v == 0u)
                coop_follow(cs, g, L[li].il, L[li].lane, 0u, 0u);
            else if ((v & 0xffu) == kVerdictContinue)
                coop_follow(cs, g, L[li].il, (v >> 8) & 0xffu, 1u, v >> 16);
            else
                cs.nseg[L[li].il] = 0u;
        }
        CG_WAVE_SYNC();
        CG_EACH_LANE
        {
            for (uint32_t pass = 0; pass < g.rounds; pass++) {
                uint32_t il_d, tl_d;
                coop_du_of(g, pass * uint32_t(kWave) + L[li].lane, il_d, tl_d);
                if (il_d < g.intervals)
                    coop_emit(cs, g, il_d, tl_d, pass * uint32_t(kWave) + L[li].lane);
            }
        }
        CG_WAVE_SYNC();
        CG_EACH_LANE
        {
            if (L[li].exists)
                coop_continue(c[li], cs, g, L[li].il, L[li].lane, L[li].start_rel, L[li].len_words,
                              cs.lists + L[li].lane * g.list_cap);
            else
                c[li].active = false;
            active[li] = c[li].active;
            if (LANES != 1 && active[li])
                CG_COOP_COUNT(continued, 1);
        }
        CG_WAVE_SYNC();
        CG_COOP_STAMP(2);
    }
    // (an interval still waiting for a walk after the last round goes the same way as one given up on)
    CG_EACH_LANE
    {
        if (L[li].tl == 0u && L[li].exists && (cs.verdict[L[li].il] & 0xffu) != kVerdictDone) {
#if defined(CG_EMUL_STATS)
            if (getenv("EMUL_COOP_FAILS"))
                fprintf(stderr, "serial il %u: still waiting for a walk after the last round (verdict %#x)\n", L[li].il, cs.verdict[L[li].il]);
#endif
            cs.verdict[L[li].il] = kVerdictSerial;
        }
    }
    CG_WAVE_SYNC();
}

// Phase 3 for 64 data units of a walk: round `r` of them -- one lane per data unit, from its start state to pixels.
// Lane l takes data unit n = 64 r + l of the walk's intervals laid end to end; an interval may begin in an earlier
// round and end in a later one (any restart interval: 10 MCUs are 40 data units).  What crosses the rounds goes
// through LDS in the order of the rounds (cs.flags, team form: the rounds run in different waves at the same time; a
// lone wave runs them one after the other): quirk Q1's "dead from" of an interval, and the sums of DC differences
// that the interval's data units in later rounds continue (cs.carry).  An interval that needs the serial decoder is
// left out here (coop_serial_intervals_422 decodes it when every round's pixels are out).
// cs.h.du_slots / cs.diffs: this wave's; cs.du_state / verdict / dead_from / carry: the walk's.
template <int LANES>
CG_DEV void coop_decode_round_422(const ImageDesc &d, const CoopShared &cs, const CoopTables &t, const CoopGeom &g,
                                  uint32_t my_lane, uint32_t wave_index, uint32_t r, CoopClock &clk)
{
    const HuffShared &s = cs.h;
    struct { uint32_t lane, il, tl; bool exists; } L[LANES];
    (void)wave_index;
    (void)clk;
    if (r * uint32_t(kWave) >= g.dus)
        return;
    uint32_t state[LANES];
    int32_t dc[LANES];
    CG_EACH_LANE
    {
        L[li].lane = LANES == 1 ? my_lane : uint32_t(li);
        coop_du_of(g, r * uint32_t(kWave) + L[li].lane, L[li].il, L[li].tl);
        L[li].exists = L[li].il < g.intervals;
        uint32_t st = kCoopUnset;
        if (L[li].exists) {
            st = cs.du_state[r * uint32_t(kWave) + L[li].lane];
            // (an interval still waiting for a walk after the last round goes the same way as one given up on)
            if ((cs.verdict[L[li].il] & 0xffu) != kVerdictDone)
                st = kCoopUnset;
            else if (st & kCoopUnset) {
                st = kCoopUnset;
#if defined(__HIP_DEVICE_COMPILE__)
                atomicMax(&cs.verdict[L[li].il], kVerdictSerial);
#else
                cs.verdict[L[li].il] = kVerdictSerial;
#endif
            }
        }
        state[li] = st;
    }
    CG_WAVE_SYNC(); // the lists have been read for the last time: their bytes become the data units' slots
    CG_EACH_LANE zero_slot(s.du_slots + L[li].lane * kDuSlotBytes);
    CG_WAVE_SYNC();
    bool under[LANES];
    CG_EACH_LANE
    {
        const uint32_t lane = L[li].lane;
        bool hostile = false;
        const int32_t diff = coop_decode_du(d, s, t, state[li], comp_of_k(lane & 3u),
                                            reinterpret_cast<int16_t *>(s.du_slots + lane * kDuSlotBytes), under[li], hostile);
        cs.diffs[lane] = diff;
        if (hostile) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicMax(&cs.verdict[L[li].il], kVerdictSerial);
#else
            cs.verdict[L[li].il] = kVerdictSerial; // (every lane that says so says the same)
#endif
        }
        if (under[li]) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicMin(&cs.dead_from[L[li].il], L[li].tl);
#else
            cs.dead_from[L[li].il] = umin(cs.dead_from[L[li].il], L[li].tl);
#endif
        }
    }
    CG_WAVE_SYNC();
    CG_COOP_STAMP(3);
    // What a round needs of the rounds before it: for the interval its first data units continue, if any (`head`), the
    // DC sums up to the round's start (cs.carry), Q1's "dead from" and the verdict -- all of them round r - 1's to
    // publish (bit r - 1 of the team's flag word).  What it owes the round behind it: the sums at its own end; where
    // the interval of its last data unit begins inside the round (`tail_own`: always so up to 16 MCUs an interval) they
    // are known before any waiting.  So the rounds of short intervals do not queue up behind each other: everything
    // is worked out from what the wave itself has decoded, published, and only the head's carry-in is added behind
    // the wait -- repeated in full in the rare case that the wait brings a Q1 event of the head's interval.
    const uint32_t first_n = r * uint32_t(kWave), last_n = umin(first_n + uint32_t(kWave), g.dus) - 1u;
    uint32_t head_il, head_tl, tail_il, tail_tl;
    coop_du_of(g, first_n, head_il, head_tl);
    coop_du_of(g, last_n, tail_il, tail_tl);
    const bool head = head_tl != 0u, tail_own = tail_tl <= last_n - first_n;
    (void)head_il;
    (void)tail_il;
    bool fixed[LANES];
    uint32_t sum[LANES];
    for (int pass = 0; pass < 2; pass++) {
        const bool last_pass = pass == 1 || !head;
        if (pass == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (cs.flags) {
                while (!(__hip_atomic_load(cs.flags + kTeamDecoded, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & (1u << ((r - 1u) & 31u))))
                    __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
#endif
        }
        CG_EACH_LANE
        {
            fixed[li] = false;
            if (state[li] & kCoopUnset)
                continue;
            // an interval that turned out to need the serial decoder after all: what its lanes decoded is dropped
            if ((cs.verdict[L[li].il] & 0xffu) == kVerdictSerial) {
                state[li] = kCoopUnset;
                continue;
            }
            // quirk Q1: behind the first data unit whose DC code underflows the reference reader, the interval decodes
            // from zeros -- whatever those lanes have decoded from the walk's states is replaced (this round's lanes
            // or an earlier round's may have found it; "dead from" only ever falls)
            const uint32_t first_dead = cs.dead_from[L[li].il];
            if (LANES != 1 && last_pass && L[li].tl == first_dead)
                CG_COOP_COUNT(dead, 1);
            if (L[li].tl <= first_dead || (state[li] & kCoopZero))
                continue;
            const uint32_t comp = comp_of_k(L[li].lane & 3u);
            uint8_t *slot = s.du_slots + L[li].lane * kDuSlotBytes;
            zero_slot(slot);
            copy_zero_levels(d, comp, reinterpret_cast<int16_t *>(slot));
            cs.diffs[L[li].lane] = zero_diff(d, comp);
            state[li] |= kCoopZero;
            fixed[li] = true;
            if (LANES != 1)
                CG_COOP_COUNT(zero, 1);
        }
        CG_WAVE_SYNC();
        // DC prediction (src/huffman.wgsl:137,170): the sum of the differences of the component's data units up to this
        // one, inside the interval; i32 wrap like the reference.  The part of the interval in earlier rounds: cs.carry.
        if (pass == 0 || coop_any<LANES>(fixed)) {
            CG_EACH_LANE
            {
                sum[li] = 0u;
                if (state[li] & kCoopUnset)
                    continue;
                const uint32_t lane = L[li].lane, k = lane & 3u, base = lane & ~3u;
                const uint32_t before = L[li].tl - (lane & 3u); // the interval's data units in front of this lane's MCU
                const uint32_t from = before > base ? 0u : base - before; // (lane of the interval's first data unit in this round)
                uint32_t acc = 0u;
                const int32_t *df = cs.diffs;
                for (uint32_t l = from; l < base; l += 4u)
                    acc += k < 2u ? uint32_t(df[l]) + uint32_t(df[l + 1u]) : uint32_t(df[l + k]);
                acc += k == 1u ? uint32_t(df[base]) + uint32_t(df[base + 1u]) : uint32_t(df[base + k]);
                sum[li] = acc;
            }
        }
        if (last_pass && head) {
            // (the interval began in an earlier round)
            CG_EACH_LANE
            {
                const uint32_t lane = L[li].lane;
                if (!(state[li] & kCoopUnset) && L[li].tl - (lane & 3u) > (lane & ~3u))
                    sum[li] += cs.carry[4u * (r - 1u) + comp_of_k(lane & 3u)];
            }
        }
        if (tail_own ? pass == 0 : last_pass) {
            CG_EACH_LANE
            {
                // (the round's last MCU: lanes 61 .. 63 hold the sums of Y, Cb, Cr up to the round's end)
                if (!(state[li] & kCoopUnset) && L[li].lane >= uint32_t(kWave) - 3u)
                    cs.carry[4u * r + (L[li].lane - (uint32_t(kWave) - 3u))] = sum[li];
            }
            CG_WAVE_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
            if (cs.flags) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if ((LANES == 1 ? my_lane : 0u) == 0u)
                    __hip_atomic_fetch_or(cs.flags + kTeamDecoded, 1u << (r & 31u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#endif
        }
        if (last_pass)
            break;
    }
    CG_EACH_LANE
    {
        if (!(state[li] & kCoopUnset))
            dc[li] = int32_t(sum[li] * sel3(comp_of_k(L[li].lane & 3u), t.dc_quant[0], t.dc_quant[1], t.dc_quant[2]));
    }
    CG_COOP_STAMP(4);
    uint32_t px[LANES][16];
    CG_EACH_LANE
    {
        const uint32_t lane = L[li].lane;
        if (state[li] & kCoopUnset)
            continue;
        uint32_t rec[kRetained / 2];
        read_slot(s.du_slots + lane * kDuSlotBytes, rec);
        idct_data_unit(rec, dc[li], cs.quant + comp_of_k(lane & 3u) * kCoopQuantStride, px[li]);
    }
    CG_WAVE_SYNC(); // every slot has been read: the area now holds the samples, kPxSlotWords apart
    CG_COOP_STAMP(5);
    uint32_t *samples = reinterpret_cast<uint32_t *>(s.du_slots);
    CG_EACH_LANE
    {
        if (state[li] & kCoopUnset)
            continue;
#pragma unroll
        for (int w = 0; w < 16; w++)
            reinterpret_cast<slot_word_t *>(samples)[L[li].lane * kPxSlotWords + w] = px[li][w];
    }
    CG_WAVE_SYNC();
    CG_EACH_LANE
    {
        if (!(state[li] & kCoopUnset)) // (the four lanes of an MCU belong to one interval: all of them or none)
            coop_composite(d, samples, g.first_interval * g.R + r * 16u, umin(g.intervals * g.R - r * 16u, 16u), L[li].lane);
    }
    CG_WAVE_SYNC(); // (the samples have been read: the area serves the next round's data units)
    CG_COOP_STAMP(6);
#if defined(__HIP_DEVICE_COMPILE__)
    if (cs.flags) {
        // (the pixels are out: an interval that has to be decoded again may be written over them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((LANES == 1 ? my_lane : 0u) == 0u)
            __hip_atomic_fetch_add(cs.flags + kTeamStored, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#endif
}

// The intervals of the walk that the chase gave up on (corrupt stream, hostile table, a window planned too small),
// when every round's pixels are out: lane 0 decodes such an interval the way the other kernels do, data unit after
// data unit -- the reference's reader followed literally, DC terms dequantised --, 64 data units at a time into the
// wave's slots; the wave transforms and composites them.
template <int LANES>
CG_DEV void coop_serial_intervals_422(const ImageDesc &d, const CoopShared &cs, const CoopTables &t, const CoopGeom &g,
                                      uint32_t my_lane)
{
    const HuffShared &s = cs.h;
    (void)t;
    for (uint32_t il = 0; il < g.intervals; il++) {
        bool serial[LANES];
        CG_EACH_LANE serial[li] = (cs.verdict[il] & 0xffu) == kVerdictSerial;
        if (!coop_any<LANES>(serial))
            continue;
        if (LANES != 1)
            CG_COOP_COUNT(serial, 1);
        EntropyState e[LANES];
        CG_EACH_LANE
        {
            if ((LANES == 1 ? my_lane : uint32_t(li)) == 0u)
                entropy_init(e[li], d, s, g.first_interval + il);
        }
        for (uint32_t first = 0; first < g.dpi; first += uint32_t(kWave)) {
            const uint32_t n = umin(g.dpi - first, uint32_t(kWave));
            CG_EACH_LANE zero_slot(s.du_slots + (LANES == 1 ? my_lane : uint32_t(li)) * kDuSlotBytes);
            CG_WAVE_SYNC();
            CG_EACH_LANE
            {
                if ((LANES == 1 ? my_lane : uint32_t(li)) != 0u)
                    continue;
                for (uint32_t du = 0; du < n; du++)
                    cs.diffs[du] = entropy_data_unit(e[li], d, s, comp_of_k(du & 3u),
                                                     reinterpret_cast<int16_t *>(s.du_slots + du * kDuSlotBytes));
            }
            CG_WAVE_SYNC();
            uint32_t px[LANES][16];
            CG_EACH_LANE
            {
                const uint32_t lane = LANES == 1 ? my_lane : uint32_t(li);
                if (lane >= n)
                    continue;
                uint32_t rec[kRetained / 2];
                read_slot(s.du_slots + lane * kDuSlotBytes, rec);
                idct_data_unit(rec, cs.diffs[lane], cs.quant + comp_of_k(lane & 3u) * kCoopQuantStride, px[li]);
            }
            CG_WAVE_SYNC();
            uint32_t *samples = reinterpret_cast<uint32_t *>(s.du_slots);
            CG_EACH_LANE
            {
                const uint32_t lane = LANES == 1 ? my_lane : uint32_t(li);
                if (lane >= n)
                    continue;
#pragma unroll
                for (int w = 0; w < 16; w++)
                    reinterpret_cast<slot_word_t *>(samples)[lane * kPxSlotWords + w] = px[li][w];
            }
            CG_WAVE_SYNC();
            CG_EACH_LANE
            {
                const uint32_t lane = LANES == 1 ? my_lane : uint32_t(li);
                if (lane < n)
                    coop_composite(d, samples, (g.first_interval + il) * g.R + first / 4u, n / 4u, lane);
            }
            CG_WAVE_SYNC();
        }
    }
}

// Team form, intervals of 16 data units (DRI = 4): the decoding waves do not wait for the walk to end.  Wave q of
// four takes *quarter* q of every interval -- MCU q, data units 4 q .. 4 q + 3 of each of the team's 16 intervals: lane
// 4 il + k decodes data unit 4 q + k of interval il -- as soon as the walker has passed those data units: an entry
// of the walker's list is final once the entry behind it has been written to.  The walker itself takes quarter 3.
// What crosses the quarters goes through LDS, in the order of the quarters: the DC differences (a data unit's DC
// term sums those of its component's earlier data units) and quirk Q1's "dead from" (coop_decode_round_422); an
// interval that turns out to need the serial decoder -- known when the walk is over, or when a lane meets a
// hostile DC category -- is decoded again, all of it, by the walker's wave once every quarter's pixels are out.
// The decode phases that waited for the walk's end ran four waves to a SIMD; now three quarters of that work run
// under the walk, which leaves its SIMD's issue slots mostly idle, and the last quarter runs alone.
template <int LANES>
CG_DEV void coop_decode_quarter_422(const ImageDesc &d, const CoopShared &cs, const CoopTables &t, const CoopGeom &g,
                                    uint32_t my_lane, uint32_t wave_index, uint32_t q, CoopClock &clk)
{
    const HuffShared &s = cs.h;
    struct { uint32_t lane; } L[LANES];
    CG_EACH_LANE L[li].lane = LANES == 1 ? my_lane : uint32_t(li);
    (void)wave_index;
    (void)clk;
    if (g.intervals == 0u)
        return;
#define Q_il(li) (L[li].lane >> 2)
#define Q_tl(li) (4u * q + (L[li].lane & 3u))
#define Q_exists(li) (Q_il(li) < g.intervals)
#define Q_list(li) (cs.lists + Q_il(li) * g.lpi * g.list_cap)
#if defined(__HIP_DEVICE_COMPILE__)
    if (q < 3u) {
        // until the walker has passed this quarter's data units in every interval -- or is done (an interval whose
        // walk ended early never gets there: it goes to the serial decoder)
        for (;;) {
            const uint32_t walk = __hip_atomic_load(cs.flags + kTeamWalk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            bool ready = true;
            if (walk >= 1u && Q_exists(0))
                ready = __hip_atomic_load(Q_list(0) + 4u * (4u * q + 4u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
            if (walk >= 2u || (walk >= 1u && __builtin_amdgcn_ballot_w64(!ready) == 0u))
                break;
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#endif
    uint32_t state[LANES];
    int32_t dc[LANES];
    CG_EACH_LANE
    {
        uint32_t st = kCoopUnset;
        if (Q_exists(li)) {
            const uint32_t *list = Q_list(li);
            const uint32_t tl = Q_tl(li);
            // the data unit's start is entry tl: final if the entry behind it has been written to (the walker's own
            // quarter: if its walk was regular)
            const bool final_ = q == 3u ? (cs.verdict[Q_il(li)] & 0xffu) == kVerdictDone : list[4u * (tl + 1u)] != 0u;
            if (final_)
                st = tl == 0u ? list[0] & kCoopStateMask
                              : lean_entry_state(lean_entry_pos(s, list[4u * tl], list[4u * tl + 1u]), list[4u * tl + 1u]);
        }
        state[li] = st;
    }
    CG_WAVE_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
    // the lists lie in the walker's slot area: it clears that only when the others have read what they need
    if ((LANES == 1 ? my_lane : 0u) == 0u)
        __hip_atomic_fetch_or(cs.flags + kTeamStatesRead, 1u << q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (q == 3u)
        while ((__hip_atomic_load(cs.flags + kTeamStatesRead, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & 7u) != 7u)
            __builtin_amdgcn_s_sleep(2);
#endif
    CG_EACH_LANE zero_slot(s.du_slots + L[li].lane * kDuSlotBytes);
    CG_WAVE_SYNC();
    bool under[LANES];
    CG_EACH_LANE
    {
        const uint32_t lane = L[li].lane;
        bool hostile = false;
        const int32_t diff = coop_decode_du(d, s, t, state[li], comp_of_k(lane & 3u),
                                            reinterpret_cast<int16_t *>(s.du_slots + lane * kDuSlotBytes), under[li], hostile);
        cs.diffs[lane] = diff;
        if (hostile) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicMax(&cs.verdict[Q_il(li)], kVerdictSerial);
#else
            cs.verdict[Q_il(li)] = kVerdictSerial;
#endif
        }
        if (under[li]) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicMin(&cs.dead_from[Q_il(li)], Q_tl(li));
#else
            cs.dead_from[Q_il(li)] = umin(cs.dead_from[Q_il(li)], Q_tl(li));
#endif
        }
    }
    CG_WAVE_SYNC();
    CG_COOP_STAMP(3);
#if defined(__HIP_DEVICE_COMPILE__)
    // the earlier quarters' DC differences and "dead from" are final
    {
        const uint32_t before = (1u << q) - 1u;
        while ((__hip_atomic_load(cs.flags + kTeamDecoded, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & before) != before)
            __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#endif
    // quirk Q1: behind the first data unit whose DC code underflows the reference reader, the interval decodes from
    // zeros -- whatever those lanes have decoded from the walk's states is replaced (this quarter's lanes or an
    // earlier quarter's may have found it)
    CG_EACH_LANE
    {
        if (!Q_exists(li) || (state[li] & kCoopUnset))
            continue;
        const uint32_t first_dead = cs.dead_from[Q_il(li)];
        if (LANES != 1 && Q_tl(li) == first_dead) {
            CG_COOP_COUNT(dead, 1);
            CG_COOP_COUNT(dead_quarter[q & 3u], 1);
        }
        if (Q_tl(li) <= first_dead)
            continue;
        const uint32_t comp = comp_of_k(L[li].lane & 3u);
        uint8_t *slot = s.du_slots + L[li].lane * kDuSlotBytes;
        zero_slot(slot);
        copy_zero_levels(d, comp, reinterpret_cast<int16_t *>(slot));
        cs.diffs[L[li].lane] = zero_diff(d, comp);
        state[li] |= kCoopZero;
        if (LANES != 1)
            CG_COOP_COUNT(zero, 1);
    }
    CG_WAVE_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((LANES == 1 ? my_lane : 0u) == 0u)
        __hip_atomic_fetch_or(cs.flags + kTeamDecoded, 1u << q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    CG_COOP_STAMP(4);
    uint32_t px[LANES][16];
    CG_EACH_LANE
    {
        const uint32_t lane = L[li].lane;
        if (state[li] & kCoopUnset)
            continue;
        // DC prediction (src/huffman.wgsl:137,170): the sum of the differences of the component's data units up to this
        // one, inside the interval; i32 wrap like the reference.  Quarter p's differences are its wave's.
        const uint32_t k = lane & 3u, base = lane & ~3u;
        uint32_t sum = 0u;
        for (uint32_t p_ = 0; p_ <= q; p_++) {
            const int32_t *dq = cs.team_diffs + ((p_ + cs.team_in_wg) & 3u) * cs.team_diffs_stride + base;
            if (k < 2u)
                sum += (p_ < q || k == 1u) ? uint32_t(dq[0]) + uint32_t(dq[1]) : uint32_t(dq[0]);
            else
                sum += uint32_t(dq[k]);
        }
        dc[li] = int32_t(sum * sel3(comp_of_k(k), t.dc_quant[0], t.dc_quant[1], t.dc_quant[2]));
        uint32_t rec[kRetained / 2];
        read_slot(s.du_slots + lane * kDuSlotBytes, rec);
        idct_data_unit(rec, dc[li], cs.quant + comp_of_k(k) * kCoopQuantStride, px[li]);
    }
    CG_WAVE_SYNC(); // every slot has been read: the area now holds the samples, kPxSlotWords apart
    CG_COOP_STAMP(5);
    uint32_t *samples = reinterpret_cast<uint32_t *>(s.du_slots);
    CG_EACH_LANE
    {
        if (state[li] & kCoopUnset)
            continue;
#pragma unroll
        for (int w = 0; w < 16; w++)
            reinterpret_cast<slot_word_t *>(samples)[L[li].lane * kPxSlotWords + w] = px[li][w];
    }
    CG_WAVE_SYNC();
    CG_EACH_LANE
    {
        // lane quad il holds MCU q of interval il: g.R (= 4) MCUs apart
        if (!(state[li] & kCoopUnset))
            coop_composite(d, samples, g.first_interval * g.R + q, g.intervals, L[li].lane, g.R);
    }
    CG_WAVE_SYNC();
    CG_COOP_STAMP(6);
#if defined(__HIP_DEVICE_COMPILE__)
    // (the pixels are out: an interval that has to be decoded again may be written over them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((LANES == 1 ? my_lane : 0u) == 0u)
        __hip_atomic_fetch_add(cs.flags + kTeamStored, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
#undef Q_il
#undef Q_tl
#undef Q_exists
#undef Q_list
}

// Behind the quarters / the rounds, by the walker's wave: once the pixels of all `parts` of them are out, the intervals
// whose verdict is "serial" once more, all of their data units.
template <int LANES>
CG_DEV void coop_team_serial_422(const ImageDesc &d, const CoopShared &cs, const CoopTables &t, const CoopGeom &g,
                                 uint32_t my_lane, uint32_t parts)
{
#if defined(__HIP_DEVICE_COMPILE__)
    while (__hip_atomic_load(cs.flags + kTeamStored, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < parts)
        __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
    (void)parts;
#endif
    coop_serial_intervals_422<LANES>(d, cs, t, g, my_lane);
}

// One wave does it all: the walk, then its rounds of 64 data units one after the other.
template <int LANES>
CG_DEV void coop_wave_422(const ImageDesc &d, const CoopShared &cs, const CoopTables &t, const CoopGeom &g,
                          uint32_t my_lane, uint32_t wave_index)
{
    CoopClock clk;
    coop_clock_start(clk);
    coop_walk_422<LANES>(d, cs, t, g, my_lane, wave_index, clk);
    for (uint32_t r = 0; r < g.rounds; r++)
        coop_decode_round_422<LANES>(d, cs, t, g, my_lane, wave_index, r, clk);
    coop_serial_intervals_422<LANES>(d, cs, t, g, my_lane);
    coop_clock_store(clk, d, wave_index, my_lane);
}

#undef CG_EACH_LANE

} // namespace compeg
