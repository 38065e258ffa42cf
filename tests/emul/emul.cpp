// Host-side emulation of the gfx950 kernels for sanitizer runs (TEST ONLY).
//
// Compiles compeg_amd/csrc/kernels_body.h -- the code the GPU lanes execute --
// with g++ under ASan/UBSan and runs the lanes of every workgroup one after
// another, with a heap block standing in for LDS.  GPU AddressSanitizer is not
// available on the target pool, so this is where out-of-bounds LDS/global
// accesses and undefined shifts in the kernel bodies get caught.  It is not a
// fallback: nothing in compeg_amd loads this library.
#include <cstdint>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../compeg_amd/csrc/device_types.h"
#include "../../compeg_amd/csrc/front.h"
#include "../../compeg_amd/csrc/kernels_body.h"
#include "../../compeg_amd/csrc/coop_body.h"
#include "../../compeg_amd/csrc/walk_body.h"
#include "../../compeg_amd/csrc/scan.h"

namespace compeg {
void fill_desc(const ImageData &img, ImageDesc &d);
uint32_t max_wave_span(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals, uint32_t group);
}
using namespace compeg;

// EMUL_LAST_NZ=1: histogram (luma, chroma) of the highest zig-zag position filled per wave and data unit, on stderr
static unsigned long *last_nz_hist()
{
    static unsigned long hist[2 * compeg::kRetained];
    static const bool on = getenv("EMUL_LAST_NZ") != nullptr;
    return on ? hist : nullptr;
}

// EMUL_DUMP_SYMBOLS=path: one line per wave and data unit with the 64 lanes' AC symbol counts (analysis only)
static FILE *symbol_dump()
{
    static FILE *f = getenv("EMUL_DUMP_SYMBOLS") ? fopen(getenv("EMUL_DUMP_SYMBOLS"), "w") : nullptr;
    return f;
}

template <int H, int V, int M>
struct LayoutTag {
    static constexpr int hs = H, vs = V, mc = M;
};

// EMUL_PADDED=1: the output as the runtime allocates it -- rows of whole MCUs, 16 pixels each way (device_types.h:
// out_alloc_h); 0 / unset: a tight buffer
static uint32_t g_out_pitch = 0, g_out_alloc_h = 0;

extern "C" __attribute__((visibility("default")))
int emul_decode(const uint8_t *jpeg, size_t len, uint8_t *rgba, uint32_t tex_w, uint32_t tex_h,
                int16_t *ac_out, int32_t *dc_out, uint32_t waves_per_block, uint32_t window_words,
                uint32_t l2_in_lds, char *err, size_t errlen, int fused)
{
    ImageData *img = nullptr;
    // mode 4 = the extension pipeline: any luma sampling is accepted
    const unsigned parse_flags = ((fused == 4 || fused == 6) ? COMPEG_PARSE_ANY_LUMA_SAMPLING : 0u) |
                                 (getenv("EMUL_STANDARD") ? COMPEG_PARSE_STANDARD_ENTROPY : 0u);
    Status s = ImageData::parse(jpeg, len, false, &img, parse_flags);
    if (!s.ok()) {
        snprintf(err, errlen, "%s", s.message.c_str());
        return s.code;
    }
    ScanBuffer scan;
    Status pre = scan.process(img->scan_data(), img->scan_len, img->metadata.total_restart_intervals);
    if (!pre.ok() && pre.code != COMPEG_E_COUNT_MISMATCH) {
        delete img;
        return pre.code;
    }
    ImageDesc d;
    fill_desc(*img, d);
    // exact-size heap copies so that ASan sees every overrun
    std::vector<uint32_t> words(scan.words(), scan.words() + scan.nwords());
    std::vector<uint32_t> starts(scan.starts(), scan.starts() + scan.nstarts());
    std::vector<uint16_t> l1(img->l1, img->l1 + 1024), l2(img->l2);
    // global LUT blob as the runtime lays it out: L2, padded to a dword, then the direct AC tables
    if (l2.size() & 1)
        l2.push_back(0);
    l2.insert(l2.end(), img->ac_fast.begin(), img->ac_fast.end());
    l2.insert(l2.end(), img->dc_fast.begin(), img->dc_fast.end());
    std::vector<int16_t> ac(size_t(d.total_dus) * kRetained);
    std::vector<int32_t> dc(d.total_dus);
    d.words = words.data();
    d.starts = starts.data();
    d.nwords = uint32_t(words.size());
    d.nstarts = uint32_t(starts.size());
    d.l1 = l1.data();
    d.l2 = l2.data();
    d.ac = ac.data();
    d.dc = dc.data();
    d.out = rgba;
    d.out_w = tex_w;
    d.out_h = tex_h;
    d.out_alloc_h = g_out_alloc_h ? g_out_alloc_h : tex_h; // (a tight buffer: nothing behind the extent)
    d.out_pitch = g_out_pitch ? g_out_pitch : tex_w * 4;

    if (fused == 5) {
        // ---- decode_coop_422_kernel: coop_wave_422 plays a whole wave, lane after lane, phase by phase ----
        if (!d.coop_ok) {
            snprintf(err, errlen, "image does not qualify for the cooperative kernel");
            delete img;
            return -5;
        }
        l2_in_lds = (uint32_t(l2.size()) + 1u) & ~1u;
        // EMUL_COOP_PASSES: 4 = the team form's geometry (a walk covers 4 x 64 data units), 1 = a lone wave's
        const uint32_t waves = getenv("EMUL_COOP_PASSES") ? uint32_t(atoi(getenv("EMUL_COOP_PASSES"))) : 1u;
        const CoopShape shape = coop_shape(d.restart_interval, waves >= 4 ? 4u : (waves >= 2 ? 2u : 1u));
        const uint32_t ipw = shape.ipw;
        if (window_words == 0) { // as the runtime plans it
            window_words = max_wave_span(starts.data(), starts.size(), words.size(), d.total_intervals, ipw) + kDuWordSlack + 4u + kCoopEndSlack;
            window_words = std::min(std::max(window_words, 128u), kCoopMaxWindow);
        }
        window_words = (window_words + 3u) & ~3u;
        const uint32_t misc_words = coop_misc_words(shape.rounds);
        const uint32_t list_bytes = shape.list_cap > kCoopSlotListCap ? uint32_t(kWave) * shape.list_cap * 4u : 0u;
        const uint32_t lds_bytes = align16((kL1Entries + l2_in_lds) * 2u) + 3u * kCoopQuantStride * 4u +
                                   align16(window_words * 4u) + kWave * kDuSlotBytes + misc_words * 4u + list_bytes;
        const uint32_t nwaves = (d.total_intervals + ipw - 1) / ipw;
        for (uint32_t wave = 0; wave < nwaves; wave++) {
            uint8_t *smem = static_cast<uint8_t *>(aligned_alloc(16, align16(lds_bytes)));
            memset(smem, 0xa5, lds_bytes); // LDS is not zero-initialised
            uint16_t *sl1 = reinterpret_cast<uint16_t *>(smem);
            uint16_t *sl2 = sl1 + kL1Entries;
            float *quant = reinterpret_cast<float *>(smem + align16((kL1Entries + l2_in_lds) * 2u));
            uint32_t *win = reinterpret_cast<uint32_t *>(quant + 3u * kCoopQuantStride);
            uint8_t *slots = reinterpret_cast<uint8_t *>(win) + align16(window_words * 4u);
            uint32_t *misc = reinterpret_cast<uint32_t *>(slots + kWave * kDuSlotBytes);
            for (uint32_t tid = 0; tid < 128; tid++)
                stage_luts(d, sl1, sl2, l2_in_lds, tid, 128, 2u * kDcFastEntries);
            for (uint32_t t = 0; t < 3u * kRetained; t++)
                quant[(t / kRetained) * kCoopQuantStride + t % kRetained] = d.quant[t / kRetained][t % kRetained];
            CoopGeom g;
            const bool with_walk_tables = waves >= 4 && !(getenv("EMUL_COOP_LEAN") && atoi(getenv("EMUL_COOP_LEAN")) == 0);
            coop_geom(d, wave, g, getenv("EMUL_COOP_SPEC_SHIFT") ? uint32_t(atoi(getenv("EMUL_COOP_SPEC_SHIFT"))) : (with_walk_tables && d.restart_interval <= kCoopLeanMaxRestart ? 31u : 0u), waves);
            uint32_t wb = 0, wl = 0;
            coop_window(d, g, window_words, wb, wl);
            for (uint32_t i = 0; i < wl; i++)
                win[i] = wb + i < d.nwords ? bswap32(d.words[wb + i]) : 0u;
            CoopShared cs;
            cs.h = HuffShared{sl1, sl2, umin(l2_in_lds, d.fast_off + 2u * kFastEntries + 2u * kDcFastEntries), win, wb, wl, slots};
            // (lists longer than the bytes of the slots -- intervals of more than 64 MCUs -- have their own area)
            cs.lists = list_bytes ? misc + misc_words : reinterpret_cast<uint32_t *>(slots);
            coop_bind_misc(cs, misc);
            cs.quant = quant;
            CoopTables t;
            coop_tables(d, cs.h, t);
            // the team form (four rounds per walk) walks through the walk tables, like the GPU's
            std::vector<uint32_t> walk;
            if (with_walk_tables) {
                walk.resize(kWalkWords);
                for (uint32_t i = 0; i < walk.size(); i++)
                    walk[i] = coop_walk_word(t.ac_fast, t.dc_fast, t.walk_ids, i);
                t.walk = walk.data();
            }
            const bool quarters = with_walk_tables && t.walk_ok && g.dpi == 16u && g.count == 1u &&
                                  !(getenv("EMUL_COOP_QUARTERS") && atoi(getenv("EMUL_COOP_QUARTERS")) == 0);
            if (!quarters) {
                coop_wave_422<kWave>(d, cs, t, g, 0, wave);
            } else {
                // as the team kernel does it with DRI = 4: the walk (the waves of the team that wait for its entries
                // on the GPU run behind it here), then quarter q of every interval by "wave" q with its own slots and
                // DC differences, the walker's lists in the last wave's slots; then the serial intervals once more
                std::vector<uint8_t> areas(4u * (kWave * kDuSlotBytes) + 64u, 0xa5);
                std::vector<int32_t> diffs(4u * kWave, 0x5a5a5a5a);
                uint8_t *area0 = areas.data() + (16u - reinterpret_cast<uintptr_t>(areas.data()) % 16u) % 16u;
                uint32_t flag_words[4] = {0, 0, 0, 0};
                cs.flags = flag_words;
                cs.team_diffs = diffs.data();
                cs.team_diffs_stride = kWave;
                cs.team_in_wg = 0;
                cs.lists = reinterpret_cast<uint32_t *>(area0 + 3u * kWave * kDuSlotBytes);
                cs.h.du_slots = area0 + 3u * kWave * kDuSlotBytes;
                CoopClock clk;
                coop_walk_422<kWave>(d, cs, t, g, 0, wave, clk, nullptr, true);
                for (uint32_t q = 0; q < 4u; q++) {
                    cs.h.du_slots = area0 + q * kWave * kDuSlotBytes;
                    cs.diffs = diffs.data() + q * kWave;
                    coop_decode_quarter_422<kWave>(d, cs, t, g, 0, wave, q, clk);
                }
                coop_serial_intervals_422<kWave>(d, cs, t, g, 0);
            }
            free(smem);
        }
        delete img;
        return 0;
    }

    // ---- mode 8: the walk + lane-per-MCU route: walk_mcus_422_kernel (walk_wave_422_stream played lane by lane, MCU by
    // MCU), then decode_fused_422_mcu_rec_kernel -- the fused kernel over the image's descriptor of MCUs, every lane begun
    // from its MCU's record.  window_words: the walk's rows (0: 16); EMUL_WALK_TABLES=0: nobody takes the walk tables.
    bool from_records = false;
    std::vector<uint32_t> mcu_word;
    std::vector<McuState> mcu_state;
    if (fused == 8) {
        if (!d.mcu_ok) {
            snprintf(err, errlen, "image does not qualify for the walk + lane-per-MCU route");
            delete img;
            return -5;
        }
        mcu_word.assign(size_t(d.total_mcus) + kWave, 0xdeadbeefu);
        mcu_state.assign(d.total_mcus, McuState{0xffffffffu, {0x55555555, 0x55555555, 0x55555555}});
        d.mcu_word = mcu_word.data();
        d.mcu_state = mcu_state.data();
        const uint32_t nrows = window_words ? window_words : 16u;
        const uint32_t stage_below = getenv("EMUL_STREAM_BELOW") ? uint32_t(strtoul(getenv("EMUL_STREAM_BELOW"), nullptr, 0)) : nrows / 2u;
        const bool with_tables = !(getenv("EMUL_WALK_TABLES") && atoi(getenv("EMUL_WALK_TABLES")) == 0);
        const uint32_t l2n = (uint32_t(l2.size()) + 1u) & ~1u; // (everything staged: L2, direct AC and DC tables)
        const uint32_t walk_off = (align16((kL1Entries + l2n) * 2u) + 31u) & ~31u;
        constexpr uint32_t kSpareRows = 3; // (kernels.hip: kWalkSpareRows)
        const uint32_t chunk = getenv("EMUL_WALK_CHUNK") ? uint32_t(std::max(1, std::min(int(kWalkMaxChunk), atoi(getenv("EMUL_WALK_CHUNK"))))) : 1u;
        const uint32_t lds_bytes = walk_off + kWalkWords * 4u + 96u + uint32_t(kWave) * walk_list_bytes(chunk) + (nrows + kSpareRows) * kWalkRowBytes;
        for (uint32_t first = 0; first < d.total_intervals; first += kWave) {
            uint8_t *smem = static_cast<uint8_t *>(aligned_alloc(32, (lds_bytes + 31u) & ~31u));
            memset(smem, 0xa5, lds_bytes);
            uint16_t *sl1 = reinterpret_cast<uint16_t *>(smem);
            uint16_t *sl2 = sl1 + kL1Entries;
            uint32_t *walk = reinterpret_cast<uint32_t *>(smem + walk_off);
            int16_t *dump = reinterpret_cast<int16_t *>(smem + walk_off + kWalkWords * 4u);
            uint32_t *lists = reinterpret_cast<uint32_t *>(smem + walk_off + kWalkWords * 4u + 96u);
            uint32_t *win = lists + uint32_t(kWave) * walk_list_bytes(chunk) / 4u;
            for (uint32_t tid = 0; tid < 128; tid++)
                stage_luts(d, sl1, sl2, l2n, tid, 128, 2u * kDcFastEntries);
            HuffShared sh{sl1, sl2, umin(l2n, d.fast_off + 2u * kFastEntries + 2u * kDcFastEntries), win, 0u, 0u, nullptr};
            CoopTables ct;
            coop_tables(d, sh, ct);
            for (uint32_t i = 0; i < kWalkWords; i++)
                walk[i] = ct.walk_ok ? coop_walk_word(ct.ac_fast, ct.dc_fast, ct.walk_ids, i) : 0u;
            WalkTabs tabs;
            walk_tabs(d, sh, with_tables ? walk : nullptr, tabs);
            // walk_wave_422, lane by lane and step by step
            std::vector<WalkLane> ls(kWave);
            const uint32_t lanes = std::min<uint32_t>(kWave, d.total_intervals - first);
            const uint32_t lstride = walk_list_bytes(chunk) / 4u;
            for (uint32_t lane = 0; lane < lanes; lane++) {
                walk_lane_init(ls[lane], d, first + lane, true);
                walk_prepare_list(lists + lane * lstride, tabs, chunk);
            }
            for (uint32_t lane = 0; lane < lanes; lane++)
                walk_restage(ls[lane], d, sh, tabs, nrows, lane);
            std::vector<uint32_t> slow_window(kWalkSlowWords, 0xa5a5a5a5u);
            for (uint32_t i = 0; i < d.restart_interval; i += chunk) {
                const uint32_t g = umin(chunk, d.restart_interval - i);
                for (uint32_t lane = 0; lane < lanes; lane++) {
                    WalkLane &l = ls[lane];
                    uint32_t *list = lists + lane * lstride;
                    const uint32_t mcu0 = (first + lane) * d.restart_interval;
                    const bool go = !l.dead && !l.parked;
                    if (go) {
                        list[0] = l.wa;
                        list[1] = l.T;
                    }
                    const uint32_t walked = walk_mcu_lean(l, d, sh, tabs, list, go, 4u * g, lane) / 4u;
                    uint32_t good = 0u;
                    for (uint32_t m = 0; m < g; m++)
                        if (go && good == m && m < walked) {
                            walk_record_at(l, d, sh, lane, list + 16u * m, mcu0 + i + m);
                            if (walk_mcu_pass(l, d, sh, tabs, list, m, lane))
                                good = m + 1u;
                        }
                    g_walk_stats.mcus += g;
                    g_walk_stats.lean_mcus += go ? good : 0u;
                    // the rest of the chunk by the slow road, through the wave's window
                    for (uint32_t m = go ? good : 0u; m < g; m++) {
                        uint32_t p0 = 0u;
                        bool on_rows = false;
                        if (go && m == good) {
                            p0 = walk_pos_of(sh, lane, list[16u * m], list[16u * m + 1u]);
                            on_rows = true;
                        } else if (!l.dead && !l.parked) {
                            p0 = walk_pos_of(sh, lane, l.wa, l.T);
                            on_rows = true;
                        }
                        const uint32_t r0 = p0 >> 5, n = on_rows && l.valid > r0 ? umin(l.valid - r0, kWalkSlowWords) : 0u;
                        std::fill(slow_window.begin(), slow_window.end(), 0xa5a5a5a5u);
                        for (uint32_t copier = 0; copier < uint32_t(kWave); copier++)
                            walk_slow_window(slow_window.data(), sh, lane, r0, n, copier);
                        if (on_rows) {
                            l.at_word = l.row0 + (p0 >> 5);
                            l.at_bit = p0 & 31u;
                            l.parked = true;
                        }
                        walk_record(l, d, sh, lane, mcu0 + i + m);
                        if (l.dead) {
                            walk_dead_mcu(l, tabs);
                            g_walk_stats.dead_mcus++;
                        } else {
                            walk_slow_mcu(l, d, sh, tabs, lane, l.at_word, l.at_bit, slow_window.data(), l.row0 + r0, n, dump);
                            g_walk_stats.slow_mcus++;
                        }
                    }
                }
                bool any = false;
                for (uint32_t lane = 0; lane < lanes && i + g < d.restart_interval; lane++)
                    any = any || walk_wants_rows(ls[lane], d, sh, lane, nrows, stage_below);
                for (uint32_t lane = 0; lane < lanes && any; lane++)
                    walk_restage(ls[lane], d, sh, tabs, nrows, lane);
                g_walk_stats.restages += any ? 1 : 0;
            }
            free(smem);
        }
        fprintf(stderr, "walk mcus=%lu lean_mcus=%lu slow_mcus=%lu slow_q1=%lu slow_rows=%lu long_dc=%lu dead_mcus=%lu restages=%lu long_codes=%lu slow_cut=%lu\n", g_walk_stats.mcus,
                g_walk_stats.lean_mcus, g_walk_stats.slow_mcus, g_walk_stats.slow_q1, g_walk_stats.slow_rows, g_walk_stats.slow_long_dc, g_walk_stats.dead_mcus,
                g_walk_stats.restages, g_walk_stats.long_codes, g_walk_stats.slow_cut);
        // the image as the second kernel sees it: every "interval" one MCU (runtime.cpp: make_walk_tables)
        d.starts = mcu_word.data();
        d.nstarts = d.total_mcus;
        d.total_intervals = d.total_mcus;
        d.restart_interval = 1;
        from_records = true;
        fused = 1;
        window_words = getenv("EMUL_MCU_WINDOW") ? uint32_t(atoi(getenv("EMUL_MCU_WINDOW"))) : 512u;
    }

    if (fused == 1 || fused == 2 || fused == 7) {
        // (7: decode_fused_422_stream_kernel -- `window_words` rows of every lane's stream instead of the wave's window)
        const bool stream = fused == 7;
        const uint32_t nrows = window_words ? window_words : 16u;
        const uint32_t stage_after = getenv("EMUL_STREAM_STAGE") ? uint32_t(strtoul(getenv("EMUL_STREAM_STAGE"), nullptr, 0)) : 8u;
        const uint32_t stage_below = getenv("EMUL_STREAM_BELOW") ? uint32_t(strtoul(getenv("EMUL_STREAM_BELOW"), nullptr, 0)) : nrows;
        if (stream)
            window_words = nrows * kWave;
        // ---- decode_fused_422_kernel (one slot set) / decode_pair_422_kernel (two sets, decoder role and
        // transformer role): the 64 lanes of a wave advance data unit by data unit, and through the quad
        // exchange of the composite phase by phase, as the wave does on the GPU ----
        const uint32_t nsets = fused == 2 ? 2u : 1u;
        l2_in_lds &= ~1u;
        const uint32_t lds_bytes = align16((kL1Entries + l2_in_lds) * 2u) + align16(window_words * 4u) +
                                   nsets * kWave * kDuSlotBytes + nsets * kWave * 4u;
        for (uint32_t first = 0; first < d.total_intervals; first += kWave) {
            uint8_t *smem = static_cast<uint8_t *>(aligned_alloc(16, align16(lds_bytes)));
            memset(smem, 0xa5, lds_bytes); // LDS is not zero-initialised
            uint16_t *sl1 = reinterpret_cast<uint16_t *>(smem);
            uint16_t *sl2 = sl1 + kL1Entries;
            uint8_t *area = smem + align16((kL1Entries + l2_in_lds) * 2u);
            uint32_t *win = reinterpret_cast<uint32_t *>(area);
            uint8_t *slots = area + align16(window_words * 4u);
            int32_t *dcs = reinterpret_cast<int32_t *>(slots + nsets * kWave * kDuSlotBytes);
            for (uint32_t tid = 0; tid < 128; tid++)
                stage_luts(d, sl1, sl2, l2_in_lds, tid, 128);
            uint32_t wb = 0, wl = 0;
            if (!stream)
                wave_window(d, first, window_words, wb, wl);
            for (uint32_t i = 0; i < wl; i++)
                win[i] = wb + i < d.nwords ? bswap32(d.words[wb + i]) : 0u;
            for (uint32_t i = 0; i < nsets * kWave; i++)
                zero_slot(slots + i * kDuSlotBytes);
            HuffShared sh{sl1, sl2, umin(l2_in_lds, d.fast_off + 2u * kFastEntries), win, wb, wl, slots};
            std::vector<EntropyState> es(kWave);
            std::vector<PixelState> ps(kWave);
            std::vector<McuTarget> tg(kWave);
            for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                const bool active = first + lane < d.total_intervals;
                if (active && stream)
                    stream_lane_init(es[lane], d, sh, nrows, first + lane, lane);
                else if (active && from_records)
                    entropy_init_from_record(es[lane], d, sh, first + lane);
                else if (active)
                    entropy_init(es[lane], d, sh, first + lane);
                pixel_init(ps[lane], d, active ? first + lane : 0u, active);
            }
            const uint32_t du_total = d.restart_interval * 4u;
            for (uint32_t du = 0; du < du_total; du++) {
                const uint32_t k = du & 3u, comp = k < 2u ? 0u : k - 1u, set = nsets == 2 ? (du & 1u) : 0u;
                uint8_t *set_slots = slots + set * kWave * kDuSlotBytes;
                unsigned long step_max = 0;
                for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                    if (ps[lane].active) {
                        g_emul_stats.lane_symbols = 0;
                        int16_t *slot16 = reinterpret_cast<int16_t *>(set_slots + lane * kDuSlotBytes);
                        dcs[set * kWave + lane] = stream ? entropy_data_unit<true>(es[lane], d, sh, comp, slot16, lane)
                                                         : entropy_data_unit(es[lane], d, sh, comp, slot16);

                        step_max = std::max(step_max, g_emul_stats.lane_symbols);
                        if (FILE *dump = symbol_dump())
                            fprintf(dump, "%lu%c", g_emul_stats.lane_symbols, lane == uint32_t(kWave) - 1 ? '\n' : ' ');
                    }
                if (stream && ((stage_after | 8u) >> k & 1u) && du + 1u < du_total) {
                    // (the wave's decision, then every lane's rows)
                    bool any = false;
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                        any = any || (ps[lane].active && stream_wants_rows(es[lane], d, sh, lane, stage_below));
                    for (uint32_t lane = 0; lane < uint32_t(kWave) && any; lane++)
                        if (ps[lane].active)
                            stream_restage(es[lane], d, sh, nrows, lane);
                }
                g_emul_stats.wave_steps++;
                g_emul_stats.wave_step_symbols += step_max;
                if (last_nz_hist()) { // analysis only: the highest zig-zag position any lane of the wave has filled
                    int wave_last = 0;
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                        const int16_t *c = reinterpret_cast<const int16_t *>(set_slots + lane * kDuSlotBytes);
                        for (int z = kRetained - 1; z > wave_last; z--)
                            if (ps[lane].active && c[z] != 0) {
                                wave_last = z;
                                break;
                            }
                    }
                    last_nz_hist()[(k < 2u ? 0 : 1) * kRetained + wave_last]++;
                }
                for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                    if (ps[lane].active)
                        pixel_transform(ps[lane], d, comp, k, set_slots + lane * kDuSlotBytes, dcs[set * kWave + lane]);
                if (k != 3u)
                    continue;
                // composite_mcus_422, phase by phase
                for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                    tg[lane] = mcu_target(ps[lane], d);
                for (uint32_t row = 0; row < 8; row++) {
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                        composite_row_to_slot(ps[lane].px, row, set_slots + lane * kDuSlotBytes);
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                        const uint32_t quad = lane & ~3u;
                        uint8_t *bases[4];
                        uint32_t whole_mask = 0;
                        for (uint32_t j = 0; j < 4; j++) {
                            // (a restart interval of one MCU: the rows leave wave-wide, composite_row_from_wave)
                            const uint32_t from = d.restart_interval == 1u ? 16u * j + (lane >> 2) : quad + j;
                            bases[j] = tg[from].base;
                            whole_mask |= (tg[from].whole ? 1u : 0u) << j;
                        }
                        if (d.restart_interval == 1u)
                            composite_row_from_wave(d, set_slots, lane, row, bases, whole_mask);
                        else
                            composite_row_from_quad(d, set_slots, lane, row, bases, whole_mask);
                    }
                }
                for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                    zero_slot(set_slots + lane * kDuSlotBytes);
                    if (ps[lane].active && !tg[lane].whole)
                        composite_edge_mcu(d, ps[lane].px, ps[lane].mx, ps[lane].my);
                    pixel_next_mcu(ps[lane], d);
                }
            }
            free(smem);
        }
        delete img;
        return 0;
    }

    // ---- huffman_kernel ----
    l2_in_lds &= ~1u;
    // (fused == 6 with EMUL_STREAM_ROWS: the layout kernels' streamed form -- that many rows of every lane's stream)
    const uint32_t lay_rows = fused == 6 && getenv("EMUL_STREAM_ROWS") ? uint32_t(std::max(1, atoi(getenv("EMUL_STREAM_ROWS")))) : 0u;
    const uint32_t lay_stage = getenv("EMUL_STREAM_STAGE") ? uint32_t(strtoul(getenv("EMUL_STREAM_STAGE"), nullptr, 0)) : 8u;
    const uint32_t lay_below = getenv("EMUL_STREAM_BELOW") ? uint32_t(strtoul(getenv("EMUL_STREAM_BELOW"), nullptr, 0)) : lay_rows;
    if (lay_rows)
        window_words = lay_rows * kWave;
    const uint32_t threads = waves_per_block * kWave;
    const uint32_t wave_area = align16(window_words * 4u) + kWave * kDuSlotBytes;
    const uint32_t lds_bytes = align16((kL1Entries + l2_in_lds) * 2u) + waves_per_block * wave_area;
    for (uint32_t first = 0; first < d.total_intervals; first += threads) {
        uint8_t *smem = static_cast<uint8_t *>(aligned_alloc(16, align16(lds_bytes)));
        memset(smem, 0xa5, lds_bytes); // LDS is not zero-initialised
        uint16_t *sl1 = reinterpret_cast<uint16_t *>(smem);
        uint16_t *sl2 = sl1 + kL1Entries;
        uint8_t *wave_base = smem + align16((kL1Entries + l2_in_lds) * 2u);
        for (uint32_t tid = 0; tid < threads; tid++)
            stage_luts(d, sl1, sl2, l2_in_lds, tid, threads);
        for (uint32_t wave = 0; wave < waves_per_block; wave++) {
            uint32_t *win = reinterpret_cast<uint32_t *>(wave_base + wave * wave_area);
            uint8_t *slots = reinterpret_cast<uint8_t *>(win) + align16(window_words * 4u);
            const uint32_t wave_first = first + wave * kWave;
            if (wave_first >= d.total_intervals)
                continue;
            uint32_t wb = 0, wl = 0;
            if (!lay_rows)
                wave_window(d, wave_first, window_words, wb, wl);
            for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                stage_window(d, win, wb, wl, lane);
            HuffShared sh{sl1, sl2, umin(l2_in_lds, d.fast_off + 2u * kFastEntries), win, wb, wl, slots};
            if (fused == 6) {
                // decode_fused_444 / _440 / _420_kernel (decode_wave_fused_layout): the 64 lanes of a wave advance data
                // unit by data unit, and through the quad exchange of the composite phase by phase
                const uint32_t hs = img->metadata.components[0].hsample, vs = img->metadata.components[0].vsample;
                auto run = [&](auto tag) {
                    constexpr int HS = decltype(tag)::hs, VS = decltype(tag)::vs, MC = decltype(tag)::mc;
                    constexpr uint32_t kDus = uint32_t(HS * VS + 2), kBlocks = kDus * uint32_t(MC);
                    std::vector<EntropyState> es(kWave);
                    std::vector<LayoutPixels<HS, VS, MC>> ps(kWave);
                    std::vector<McuTarget> tg(kWave);
                    std::vector<int32_t> dcs(kWave);
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                        zero_slot(slots + lane * kDuSlotBytes);
                        const bool active = wave_first + lane < d.total_intervals;
                        const uint32_t iv = active ? wave_first + lane : d.total_intervals - 1u;
                        if (lay_rows)
                            stream_lane_init(es[lane], d, sh, lay_rows, iv, lane);
                        else
                            entropy_init(es[lane], d, sh, iv);
                        layout_init<HS, VS, MC>(ps[lane], d, iv, active);
                    }
                    const uint32_t du_total = d.restart_interval * kDus;
                    for (uint32_t du = 0, k = 0, place = 0; du < du_total; du++) {
                        const uint32_t comp = layout_comp_of(k, uint32_t(HS * VS));
                        for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                            int16_t *slot16 = reinterpret_cast<int16_t *>(slots + lane * kDuSlotBytes);
                            dcs[lane] = lay_rows ? entropy_data_unit<true>(es[lane], d, sh, comp, slot16, lane)
                                                 : entropy_data_unit(es[lane], d, sh, comp, slot16);
                        }
                        if (lay_rows && (k == kDus - 1u || lay_stage != 8u) && du + 1u < du_total) {
                            bool any = false; // (the wave's decision, then every lane's rows)
                            for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                                any = any || stream_wants_rows(es[lane], d, sh, lane, lay_below);
                            for (uint32_t lane = 0; lane < uint32_t(kWave) && any; lane++)
                                stream_restage(es[lane], d, sh, lay_rows, lane);
                        }
                        for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                            layout_transform<HS, VS, MC>(ps[lane], d, comp, place, slots + lane * kDuSlotBytes, dcs[lane]);
                        k = k == kDus - 1u ? 0u : k + 1u;
                        const bool alone = MC == 2 && du + 1u == du_total && place != kBlocks - 1u; // (decode_wave_fused_layout)
                        if (place != kBlocks - 1u && !alone) {
                            place++;
                            continue;
                        }
                        place = 0;
                        for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                            tg[lane] = layout_target<HS, VS, MC>(ps[lane], d);
                        for (int row = 0; row < 8 * VS; row++) {
                            for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                                layout_row_to_slot<HS, VS, MC>(ps[lane], row, slots + lane * kDuSlotBytes);
                            for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                                const uint32_t quad = lane & ~3u;
                                uint8_t *bases[4];
                                uint32_t limits = 0, pair_lims[4] = {0, 0, 0, 0};
                                for (uint32_t j = 0; j < 4; j++) {
                                    bases[j] = tg[quad + j].base;
                                    if constexpr (MC == 2) {
                                        pair_lims[j] = pair_limits<HS, VS, MC>(ps[quad + j], d, tg[quad + j].whole, alone);
                                    } else {
                                        limits |= (tg[quad + j].whole ? uint32_t(8 * VS) | uint32_t(2 * HS * MC) << 5 : layout_limit<HS, VS, MC>(ps[quad + j], d)) << (8u * j);
                                    }
                                }
                                bool all_whole = !alone; // (the kernel's ballot: every group of the wave whole)
                                for (uint32_t l2 = 0; l2 < uint32_t(kWave); l2++)
                                    all_whole = all_whole && tg[l2].whole;
                                if (all_whole) {
                                    layout_row_from_quad<2 * HS * MC, false>(d, slots, lane, uint32_t(row), bases, 0xfu);
                                } else if constexpr (MC == 2) {
                                    // (pairs: each half its own limit, the second its own place)
                                    for (uint32_t j = 0; j < 4 && (lane & 2u); j++)
                                        bases[j] += pair_second_offset<HS, VS, MC>(ps[quad + j], d);
                                    layout_row_from_quad_cut<2 * HS * MC>(d, slots, lane, uint32_t(row), bases, pair_rows_for_lane(pair_lims, lane & 3u));
                                } else {
                                    layout_row_from_quad_cut<2 * HS * MC>(d, slots, lane, uint32_t(row), bases,
                                                                          cut_rows_for_piece(limits, 2 * HS * MC == 4 ? lane & 3u : lane & 1u));
                                }
                            }
                        }
                        for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                            zero_slot(slots + lane * kDuSlotBytes);
                            if (layout_is_edge<HS, VS, MC>(ps[lane], d, tg[lane].whole, alone))
                                composite_layout_edge<HS, VS, MC>(ps[lane], d, alone ? 1u : uint32_t(MC));
                            layout_next_group<HS, VS, MC>(ps[lane], d);
                        }
                    }
                };
                // (as the runtime dispatches: pairs from two MCUs an interval on; EMUL_SINGLES: the single form for every odd interval)
                const bool pairs = hs == 1 && (getenv("EMUL_SINGLES") ? d.restart_interval % 2u == 0u : d.restart_interval >= 2u);
                if (hs == 1 && vs == 1 && pairs)
                    run(LayoutTag<1, 1, 2>{});
                else if (hs == 1 && vs == 1)
                    run(LayoutTag<1, 1, 1>{});
                else if (hs == 1 && vs == 2 && pairs)
                    run(LayoutTag<1, 2, 2>{});
                else if (hs == 1 && vs == 2)
                    run(LayoutTag<1, 2, 1>{});
                else if (hs == 2 && vs == 2)
                    run(LayoutTag<2, 2, 1>{});
                continue;
            }
            if (fused == 3 || fused == 4) {
                // entropy_wave_to_records, data unit by data unit and phase by phase
                std::vector<EntropyState> es(kWave);
                std::vector<int32_t> dcs(kWave);
                for (uint32_t lane = 0; lane < uint32_t(kWave); lane++) {
                    zero_slot(slots + lane * kDuSlotBytes);
                    const uint32_t iv = wave_first + lane;
                    entropy_init(es[lane], d, sh, iv < d.total_intervals ? iv : d.total_intervals - 1u);
                }
                const uint32_t du_count = d.restart_interval * d.dus_per_mcu;
                for (uint32_t du = 0, k = 0; du < du_count; du++, k = k + 1u == d.dus_per_mcu ? 0u : k + 1u) {
                    const uint32_t comp = (d.comp_of_du >> (2u * k)) & 3u;
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                        dcs[lane] = entropy_data_unit(es[lane], d, sh, comp < 3u ? comp : 2u,
                                                      reinterpret_cast<int16_t *>(slots + lane * kDuSlotBytes));
                    if (fused == 4) // entropy_samples_kernel: the IDCT runs here, the records carry samples
                        for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                            slot_to_samples(d, comp < 3u ? comp : 2u, slots + lane * kDuSlotBytes, dcs[lane]);
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                        records_flush_quad(d, slots, lane, wave_first + lane, du, dcs[lane]);
                    for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                        zero_slot(slots + lane * kDuSlotBytes);
                }
                continue;
            }
            for (uint32_t lane = 0; lane < uint32_t(kWave); lane++)
                if (wave_first + lane < d.total_intervals)
                    huff_decode_interval(d, sh, wave_first + lane, lane);
        }
        free(smem);
    }
    if (ac_out)
        memcpy(ac_out, ac.data(), ac.size() * 2);
    if (dc_out)
        memcpy(dc_out, dc.data(), dc.size() * 4);

    if (fused == 6) {
        delete img;
        return 0;
    }
    if (fused == 4) {
        // ---- composite_generic_kernel (extension layouts; the records hold samples) ----
        for (uint32_t y = 0; y < tex_h; y++)
            for (uint32_t x0 = 0; x0 < ((tex_w + 3u) & ~3u) + 8u; x0 += 4) // a few lanes past the row end, like the grid
                composite_generic_4px(d, x0, y, reinterpret_cast<const uint8_t *>(d.ac), 0u);
        delete img;
        return 0;
    }

    // ---- idct_composite_kernel ----
    const uint32_t total_mcus = d.dus_per_mcu ? d.total_dus / d.dus_per_mcu : 0;
    for (uint32_t first_du = 0; first_du < d.total_dus; first_du += 256) {
        std::vector<float> quant(3 * kRetained);
        for (uint32_t t = 0; t < 3 * kRetained; t++)
            quant[t] = d.quant[t / kRetained][t % kRetained];
        std::vector<uint32_t> px(4 * kWave * kPxSlotWords, 0xa5a5a5a5u);
        for (uint32_t tid = 0; tid < 256; tid++) {
            const uint32_t du = first_du + tid;
            if (du >= d.total_dus)
                continue;
            const uint32_t comp = (d.comp_of_du >> (2u * (du % d.dus_per_mcu))) & 3u;
            uint32_t rec[kRetained / 2];
            memcpy(rec, d.ac + size_t(du) * kRetained, sizeof rec);
            uint32_t out[16];
            idct_data_unit(rec, d.dc[du], quant.data() + comp * kRetained, out);
            memcpy(px.data() + size_t(tid / kWave) * kWave * kPxSlotWords + (tid % kWave) * kPxSlotWords,
                   out, sizeof out);
        }
        for (uint32_t tid = 0; tid < 256; tid++) {
            const uint32_t wave = tid / kWave, lane = tid % kWave;
            composite_422(d, px.data() + size_t(wave) * kWave * kPxSlotWords,
                          (first_du + wave * kWave) / 4u, total_mcus, lane);
        }
    }
    delete img;
    return 0;
}

// emul_runner in.jpg out.rgba out.ac out.dc waves_per_block window_words l2_in_lds [tex_w tex_h]
int main(int argc, char **argv)
{
    if (argc < 8) {
        fprintf(stderr, "usage: %s in.jpg out.rgba out.ac out.dc waves window_words l2_in_lds [tex_w tex_h]\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f)
        return 2;
    std::vector<uint8_t> jpeg;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0)
        jpeg.insert(jpeg.end(), buf, buf + n);
    fclose(f);
    // exact-size copy: reads past the end of the file are caught by ASan
    uint8_t *exact = static_cast<uint8_t *>(malloc(jpeg.size() ? jpeg.size() : 1));
    memcpy(exact, jpeg.data(), jpeg.size());

    ImageData *probe = nullptr;
    const int mode = getenv("EMUL_FUSED") ? atoi(getenv("EMUL_FUSED")) : 0;
    Status s = ImageData::parse(exact, jpeg.size(), false, &probe, (mode == 4 || mode == 6) ? COMPEG_PARSE_ANY_LUMA_SAMPLING : 0u);
    if (!s.ok()) {
        printf("error: %s\n", s.message.c_str());
        free(exact);
        return 1;
    }
    uint32_t tex_w = argc > 9 ? uint32_t(atoi(argv[8])) : probe->width;
    uint32_t tex_h = argc > 9 ? uint32_t(atoi(argv[9])) : probe->height;
    const uint32_t dus = probe->total_dus();
    delete probe;
    std::vector<uint8_t> rgba(size_t(tex_w) * tex_h * 4, 0);
    std::vector<uint8_t> padded;
    if (getenv("EMUL_PADDED") && atoi(getenv("EMUL_PADDED"))) {
        g_out_pitch = (tex_w + 15u) / 16u * 64u;
        g_out_alloc_h = (tex_h + 15u) / 16u * 16u;
        padded.assign(size_t(g_out_pitch) * g_out_alloc_h, 0);
    }
    std::vector<int16_t> ac(size_t(dus) * kRetained);
    std::vector<int32_t> dc(dus);
    char err[256] = "";
    int rc = emul_decode(exact, jpeg.size(), padded.empty() ? rgba.data() : padded.data(), tex_w, tex_h, ac.data(), dc.data(),
                         uint32_t(atoi(argv[5])), uint32_t(atoi(argv[6])), uint32_t(atoi(argv[7])),
                         err, sizeof err, mode);
    free(exact);
    if (rc != 0) {
        printf("error: %s\n", err);
        return 1;
    }
    auto dump = [](const char *path, const void *p, size_t bytes) {
        FILE *o = fopen(path, "wb");
        if (o) {
            if (bytes) // an image without a single complete restart interval has no data units
                fwrite(p, 1, bytes, o);
            fclose(o);
        }
    };
    for (uint32_t y = 0; !padded.empty() && y < tex_h; y++)
        memcpy(rgba.data() + size_t(y) * tex_w * 4, padded.data() + size_t(y) * g_out_pitch, size_t(tex_w) * 4);
    dump(argv[2], rgba.data(), rgba.size());
    dump(argv[3], ac.data(), ac.size() * 2);
    dump(argv[4], dc.data(), dc.size() * 4);
    printf("ok %u %u %u\n", tex_w, tex_h, dus);
    const EmulStats &st = g_emul_stats;
    const CoopStats &cst = g_coop_stats;
    if (cst.intervals)
        fprintf(stderr, "coop intervals=%lu rounds=%lu direct=%lu continued=%lu serial=%lu dead=%lu zero=%lu chase_steps=%lu wave_steps=%lu\n", cst.intervals,
                cst.rounds, cst.direct, cst.continued, cst.serial, cst.dead, cst.zero, cst.chase_steps, cst.wave_steps);
    if (cst.intervals && cst.dead)
        fprintf(stderr, "coopdead q0=%lu q1=%lu q2=%lu q3=%lu\n", cst.dead_quarter[0], cst.dead_quarter[1], cst.dead_quarter[2], cst.dead_quarter[3]);
    if (cst.intervals && getenv("EMUL_COOP_HIST")) {
        fprintf(stderr, "cooplinks (first lane of each interval) tries=%lu ok=%lu only_full_match=%lu none=%lu\n", cst.link_tries, cst.link_ok, cst.link_full_ok, cst.link_none);
        fprintf(stderr, "coophist true_steps=%lu true_max=%lu hist(16 steps per bin):", cst.true_steps, cst.true_max);
        for (int i = 0; i < 16; i++)
            fprintf(stderr, " %lu", cst.hist[i]);
        fprintf(stderr, "\n");
    }
    if (last_nz_hist())
        for (int c = 0; c < 2; c++) {
            fprintf(stderr, "last_nz %s:", c ? "chroma" : "luma");
            for (int z = 0; z < compeg::kRetained; z++)
                fprintf(stderr, " %lu", last_nz_hist()[c * compeg::kRetained + z]);
            fprintf(stderr, "\n");
        }
    fprintf(stderr, "stats fast_dus=%lu exact_dus=%lu left_window=%lu left_underflow=%lu dc_cut=%lu escapes=%lu symbols=%lu wave_steps=%lu wave_step_symbols=%lu\n",
            st.fast_dus, st.exact_dus, st.left_window, st.left_underflow, st.dc_cut, st.escapes, st.symbols, st.wave_steps,
            st.wave_step_symbols);
    return 0;
}
