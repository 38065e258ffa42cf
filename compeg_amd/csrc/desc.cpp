// Translation of the reference-format Metadata block into the kernel-facing
// descriptor (no HIP dependency: shared with the host-side kernel emulation
// used by the sanitizer tests).
#include <cstdlib>
#include <cstring>

#include "device_types.h"
#include "front.h"

namespace compeg {

void fill_desc(const ImageData &img, ImageDesc &d);
uint32_t max_wave_span(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals);

void fill_desc(const ImageData &img, ImageDesc &d)
{
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
    }
}


uint32_t max_wave_span(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals)
{
    uint32_t best = 0;
    for (size_t first = 0; first < intervals; first += kWave) {
        const uint64_t lo = first < nstarts ? starts[first] : 0;
        const size_t after = first + kWave;
        const uint64_t hi = (after < intervals && after < nstarts) ? starts[after] : nwords;
        if (hi > lo && hi - lo > best)
            best = uint32_t(hi - lo > 0xffffffffu ? 0xffffffffu : hi - lo);
    }
    return best;
}

} // namespace compeg
