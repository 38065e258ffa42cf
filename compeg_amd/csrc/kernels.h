// Host-callable launchers of the gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>

#include "device_types.h"

namespace compeg {

// Chooses workgroup shape and LDS carve-up for a decode of `images` images
// whose largest image has `max_intervals` restart intervals and `max_l2`
// L2 entries; max_wave_words (largest scan span of 64 consecutive intervals)
// sizes the per-wave scan window.
// wave_cap: at most that many waves per workgroup (0: the kernel family's own limit)
HuffLdsPlan plan_huffman(uint32_t max_intervals, uint32_t images, uint32_t max_l2,
                         uint32_t max_wave_words, bool fused, uint32_t wave_cap = 0);
// Largest word span covered by any group of 64 consecutive restart intervals.
uint32_t max_wave_span(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals,
                       uint32_t group = kWave);

// descs: device array of `images` descriptors.  Grid = (blocks for the
// largest image, images); blocks past an image's own extent exit at once.
hipError_t launch_huffman(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                          const HuffLdsPlan &plan, hipStream_t stream);
// Same outputs as launch_huffman (coefficient records + DC terms) from the fast-mode decoder.
hipError_t launch_entropy(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                          const HuffLdsPlan &plan, hipStream_t stream);
// uniform: all images have max_intervals intervals and byte-identical LUTs (workgroups may then span images)
// one_mcu_intervals: every image's restart interval is one MCU (the kernel whose rows leave wave-wide)
// from_records: descs are the images' descriptors of MCUs (ImageDesc::mcu_word), max_intervals the most MCUs of any
// queue: four bytes of device memory of the caller's (or null): the counter the resident waves of a uniform launch
// draw their units from (zeroed here, in stream order)
hipError_t launch_fused_422(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                            const HuffLdsPlan &plan, hipStream_t stream, bool uniform = false, bool one_mcu_intervals = false,
                            uint32_t *queue = nullptr, bool from_records = false);
// The batch kernel with the window in its streamed form (kernels_body.h: decode_wave_fused_422_stream), for launches
// whose whole-interval windows would leave a CU fewer than its twelve waves.
struct StreamPlan {
    uint32_t rows = 0, waves_per_block = 0, l2_entries_in_lds = 0, total_bytes = 0;
    uint32_t waves_per_image = 0; // != 0: the flat grid (uniform launches)
    uint32_t stage_after = 8;     // bit k: rows may be staged anew behind data unit k of an MCU ...
    uint32_t stage_below = 0;     // ... and are, when some lane has fewer staged words than this in front of it
    uint32_t cu_waves = 12;       // waves of the kernel a CU holds
};
// uniform: all images have max_intervals intervals and byte-identical LUTs (workgroups may then span images)
// cu_waves / group_waves: 0 (4:2:2: twelve waves a CU, in one workgroup), or what the layout's kernel holds
StreamPlan plan_stream(uint32_t max_intervals, uint32_t images, uint32_t max_l2, uint32_t mcu_words, bool uniform, uint32_t cu_waves = 0,
                       uint32_t group_waves = 0);
// (plan: plan_huffman's for the same launch)
bool stream_plan_preferred(const HuffLdsPlan &plan, uint32_t max_intervals, uint32_t images, uint32_t cu_waves = 0, uint32_t group_waves = 0);
// (hs, vs: the luma sampling all images of the launch share: 2x1, or an extension layout's -- paired kernels for 1x1 / 1x2)
hipError_t launch_fused_stream(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const StreamPlan &plan,
                                   hipStream_t stream, uint32_t hs = 2, uint32_t vs = 1, uint32_t *queue = nullptr);
// The first kernel of the walk + lane-per-MCU route (kernels_body.h): a lane per restart interval, entropy decode only,
// every MCU's record into ImageDesc::mcu_word / mcu_state; the second one is launch_fused_422(..., from_records) over the
// images' descriptors of MCUs.
// walk_tables: every image has ImageDesc::walk (launch_walk_tables): two symbols a step
struct WalkPlan {
    uint32_t rows = 0, stage_below = 0, waves_per_block = 0, l2_entries_in_lds = 0, total_bytes = 0;
    uint32_t chunk = 1;           // MCUs a lane walks between two looks at what it found (walk_body.h)
    uint32_t waves_per_image = 0; // != 0: the flat grid (uniform launches)
    bool walk_tables = false;
};
// restart_interval: the smallest of the launch's images
WalkPlan plan_walk(uint32_t max_intervals, uint32_t images, uint32_t max_l2, uint32_t mcu_words, uint32_t restart_interval, bool uniform, bool walk_tables);
hipError_t launch_walk_mcus(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const WalkPlan &plan, hipStream_t stream,
                            uint32_t *queue = nullptr);
// Extension layouts (luma hs x vs = 1x1, 1x2, 2x2), fused like the 4:2:2 kernel; plan with wave_cap = fused_layout_wave_cap.
// pairs: (8-pixel MCUs) every image of the launch has restart intervals of two MCUs or more -- a lane composites its MCUs two at a time (an odd interval's last alone)
uint32_t fused_layout_wave_cap(uint32_t hs, uint32_t vs, bool pairs);
hipError_t launch_fused_layout(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const HuffLdsPlan &plan,
                               uint32_t hs, uint32_t vs, bool pairs, hipStream_t stream);
// Latency variant: one decoder wave + one transformer wave per 64 intervals.
hipError_t launch_pair_422(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                           const HuffLdsPlan &plan, hipStream_t stream);
// Cooperative kernel (coop_body.h) for launches too small to fill the chip with a lane per restart interval.
struct CoopPlan {
    bool usable;
    uint32_t group_waves; // 4, 2 or 1: a team takes coop_shape(restart_interval, group_waves).ipw whole intervals
    uint32_t intervals_per_wave, waves_per_block, window_words, l2_entries_in_lds, total_bytes, total_waves; // (intervals_per_wave: per team)
    uint32_t fit_teams; // teams a CU's LDS holds at once with this window (1..4)
    uint32_t places;    // ... and the chip: teams resident at once
};
// The kernel's teams of four waves take the whole restart intervals of 4 x 64 data units -- of 2 x 64 or 64 where
// those do not fit the largest window (dense streams: bit positions inside a window are 16-bit).
// words[k]: largest word span of any group of coop_shape(restart_interval, 4 >> k).ipw consecutive intervals
// (max_wave_span with that group size), or an upper estimate of it
struct CoopSpans {
    uint32_t words[3];
};
CoopPlan plan_coop(uint32_t max_intervals, uint32_t images, uint32_t restart_interval, uint32_t max_l2,
                   const CoopSpans &spans);
hipError_t launch_coop_422(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const CoopPlan &plan,
                           hipStream_t stream);
// Fills ImageDesc::walk (kWalkTableBytes each) of every image that has one, from its direct tables.
constexpr size_t kWalkTableBytes = 4u * 2048u * 4u;
hipError_t launch_walk_tables(const ImageDesc *descs, uint32_t images, hipStream_t stream);
hipError_t launch_idct_composite(const ImageDesc *descs, uint32_t images, uint32_t max_dus,
                                 hipStream_t stream);

// Extension pipeline for layouts other than 4:2:2: entropy decode + IDCT into sample records (plan as for
// the fused kernel), then the composite of every output the descriptors name (max_w x max_h: largest output).
hipError_t launch_entropy_samples(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                                  const HuffLdsPlan &plan, hipStream_t stream);
hipError_t launch_generic_composite(const ImageDesc *descs, uint32_t images, uint32_t max_w, uint32_t max_h,
                                    hipStream_t stream);

#if defined(CG_AC_STAMPS)
// diagnostic build: AC-loop cycle counters (kernels_body.h)
hipError_t read_ac_stamps(unsigned long long out[4], bool reset);
#endif

} // namespace compeg
