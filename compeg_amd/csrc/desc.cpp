// Translation of the reference-format Metadata block into the kernel-facing
// descriptor (no HIP dependency: shared with the host-side kernel emulation
// used by the sanitizer tests).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_types.h"
#include "front.h"
#include "kernels_body.h" // (host compile of the lane bodies: the exact-path reader, for the zero-stream data units)
#include "coop_body.h"    // (kCoopMaxRestart)

namespace compeg {

void fill_desc(const ImageData &img, ImageDesc &d);
void fill_coop(const ImageData &img, ImageDesc &d);
uint32_t max_wave_span(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals, uint32_t group);

void fill_desc(const ImageData &img, ImageDesc &d)
{
    d.walk = nullptr; // (set by callers that keep walk tables for this image)
    const Metadata &md = img.metadata;
    memset(&d, 0, sizeof d);
    d.l2_entries = uint32_t(img.l2.size());
    d.fast_off = uint32_t((img.l2.size() + 1) & ~size_t(1));
    d.standard_entropy = (img.flags & COMPEG_PARSE_STANDARD_ENTROPY) ? 1u : 0u;
    d.total_intervals = md.total_restart_intervals;
    d.restart_interval = md.restart_interval;
    d.dus_per_mcu = md.dus_per_mcu;
    d.width_mcus = md.width_mcus;
    d.mcu_w = md.max_hsample * 8;
    d.mcu_h = md.max_vsample * 8;
    d.total_dus = img.total_dus();
    uint32_t k = 0;
    for (uint32_t c = 0; c < 3; c++) {
        const Component &cm = md.components[c];
        d.du_base[c] = k;
        for (uint32_t i = 0; i < cm.hsample * cm.vsample && k < kMaxDusPerMcu; i++, k++)
            d.comp_of_du |= c << (2 * k);
        // selectors past the four uploaded tables read as zero in the
        // reference (robust buffer access): route them to the all-zero table
        d.dc_table[c] = cm.dchuff < 4 ? cm.dchuff : 4;
        d.ac_table[c] = cm.achuff < 4 ? cm.achuff : 4;
        d.fast_table[c] = (cm.achuff == 1 || cm.achuff == 3) ? cm.achuff >> 1 : 2;
        d.dc_quant[c] = md.qtables[cm.qtable & 3][0];
        d.hsample[c] = cm.hsample;
        d.vsample[c] = cm.vsample;
        for (int z = 0; z < kRetained; z++)
            d.quant[c][z] = float(md.qtables[cm.qtable & 3][z]);
        d.dc_fast_table[c] = (cm.dchuff == 0 || cm.dchuff == 2) ? cm.dchuff >> 1 : 2;
    }
    fill_coop(img, d);
    // The walk + lane-per-MCU route (device_types.h: ImageDesc::mcu_ok): 4:2:2, the direct tables for every component,
    // and no DC category above 15 in the DC tables the components use -- every consume is then shorter than 32 bits,
    // so the reference reader's state at an MCU's start is a position and its `left`, or "run dry" (quirk Q1), and
    // nothing else.  (At most 2^18 MCUs an interval: a reader that has run dry keeps its wrapped `left` above 64 for
    // 2^32 bits.)
    d.total_mcus = img.total_mcus();
    const bool is422 = md.dus_per_mcu == 4 && md.components[0].hsample == 2 && md.components[0].vsample == 1 &&
                       md.components[1].hsample == 1 && md.components[1].vsample == 1 &&
                       md.components[2].hsample == 1 && md.components[2].vsample == 1;
    // ... the zero-stream data units known (fill_coop), and at most two different pairs of (DC, AC) table among the
    // components: what the walk tables hold (coop_body.h: coop_tables)
    bool ok = is422 && md.restart_interval != 0 && md.restart_interval <= (1u << 18) && d.zero_du_ok != 0;
    if (ok) {
        const uint32_t pair0 = (d.dc_fast_table[0] & 1u) | ((d.fast_table[0] & 1u) << 1);
        uint32_t pair1 = pair0;
        for (uint32_t c = 1; c < 3; c++) {
            const uint32_t pr = (d.dc_fast_table[c] & 1u) | ((d.fast_table[c] & 1u) << 1);
            if (pr != pair0 && pair1 == pair0)
                pair1 = pr;
            ok = ok && (pr == pair0 || pr == pair1);
        }
    }
    for (uint32_t c = 0; c < 3 && ok; c++) {
        ok = d.fast_table[c] < 2 && d.dc_fast_table[c] < 2;
        const uint16_t *l1 = img.l1 + size_t(d.dc_table[c] & 3u) * 256;
        for (uint32_t i = 0; i < 256 && ok; i++) {
            const uint16_t e = l1[i];
            if (!(e & 0x8000u)) {
                ok = (e & 0xffu) <= 15u;
                continue;
            }
            for (uint32_t j = 0; j < 256 && ok; j++) {
                const size_t idx = size_t(e & 0x7fffu) + j;
                ok = idx >= img.l2.size() || (img.l2[idx] & 0xffu) <= 15u;
            }
        }
    }
    d.mcu_ok = ok ? 1u : 0u;
}

// What the cooperative kernel needs to know beyond the tables (coop_body.h): whether the image qualifies, and
// the "zero-stream" data unit of every component.  Once the reference's reader has underflown at a DC code
// (quirk Q1) its buffer is all zeros and is never topped up again, so every later data unit of the interval
// decodes from zeros -- always the same levels.  They are computed here by the very reader the kernels' exact
// path uses, on a reader in that state, so that host and device cannot disagree.
void fill_coop(const ImageData &img, ImageDesc &d)
{
    const Metadata &md = img.metadata;
    d.coop_ok = 0;
    d.zero_du_ok = 0;
    memset(d.zero_du, 0, sizeof d.zero_du);
    const bool is422 = md.dus_per_mcu == 4 && md.components[0].hsample == 2 && md.components[0].vsample == 1 &&
                       md.components[1].hsample == 1 && md.components[1].vsample == 1 &&
                       md.components[2].hsample == 1 && md.components[2].vsample == 1;
    if (!is422 || md.restart_interval == 0)
        return;
    for (uint32_t c = 0; c < 3; c++)
        if (d.fast_table[c] >= 2 || d.dc_fast_table[c] >= 2)
            return;
    // the tables as a kernel sees them: five L1 tables (the last one all zero), the whole L2 LUT "staged"
    std::vector<uint16_t> l1(kL1Entries, 0);
    memcpy(l1.data(), img.l1, sizeof img.l1);
    const uint32_t none[4] = {0, 0, 0, 0};
    ImageDesc t = d;
    t.l2 = img.l2.data();
    t.words = nullptr;
    t.nwords = 0;
    HuffShared s{};
    s.l1 = l1.data();
    s.l2 = img.l2.empty() ? l1.data() : img.l2.data();
    s.l2_staged = uint32_t(img.l2.size());
    s.win = none;
    s.win_base = 0;
    s.win_len = 4;
    for (uint32_t c = 0; c < 3; c++) {
        PrefetchReader r;
        r.buf = 0;
        r.left = 0x80000000u; // wrapped below zero: never topped up again
        r.next_word = 0;
        r.pre = 0;
        ImageDesc tq = t;
        tq.standard_entropy = 0;
        const int32_t diff = decode_dc_diff(r, tq, s, d.dc_table[c] * 256u);
        if (diff < -32767 || diff > 32767)
            return; // (a category above 15 for the all-zero prefix: hostile table)
        int16_t slot[kDuSlotBytes / 2] = {0};
        decode_ac_loop<false>(r, tq, s, d.ac_table[c] * 256u, slot);
        d.zero_du[c][0] = int16_t(diff);
        for (int z = 1; z < kRetained; z++)
            d.zero_du[c][z] = slot[z];
    }
    d.zero_du_ok = 1;
    // (the cooperative kernel: any restart interval of up to kCoopMaxRestart MCUs -- coop_shape, device_types.h)
    d.coop_ok = md.restart_interval <= kCoopMaxRestart ? 1u : 0u;
}


uint32_t max_wave_span(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals, uint32_t group)
{
    uint32_t best = 0;
    if (group == 0)
        group = kWave;
    for (size_t first = 0; first < intervals; first += group) {
        const uint64_t lo = first < nstarts ? starts[first] : 0;
        const size_t after = first + group;
        const uint64_t hi = (after < intervals && after < nstarts) ? starts[after] : nwords;
        if (hi > lo && hi - lo > best)
            best = uint32_t(hi - lo > 0xffffffffu ? 0xffffffffu : hi - lo);
    }
    return best;
}

} // namespace compeg
