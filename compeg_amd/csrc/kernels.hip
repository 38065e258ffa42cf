// gfx950 kernels of libcompeg_hip (bodies in kernels_body.h, design in DESIGN.md section 5).
//
//   decode_fused_422_kernel   the product path for launches that fill the chip: entropy
//                             decode, IDCT and composite of 64 restart intervals per wave,
//                             coefficients in LDS, samples in registers, nothing but the
//                             RGBA output written to HBM (replaces src/huffman.wgsl and both
//                             passes of src/dct.wgsl)
//   decode_pair_422_kernel    the same work as a decoder wave + a transformer wave per 64
//                             intervals, for launches that cannot fill the chip (one frame)
//   entropy_kernel            entropy stage alone -> coefficient records in HBM
//   idct_composite_kernel     IDCT + 4:2:2 composite from those records (two-kernel pipeline)
//   entropy_samples_kernel,   entropy decode + IDCT -> sample records, and the composite from
//   composite_generic_kernel  them, for the extension layouts (4:4:4, 4:4:0, 4:2:0)
//   huffman_kernel            a literal restatement of the reference's decode loop; debug
//                             coefficient read-back and cross-check of the fast path
//
// No MFMA: there is no dense contraction anywhere on this path; the kernels are bounded by
// instruction issue of the serial entropy decode and by HBM writes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "kernels_body.h"
#include "coop_body.h"
#include "walk_body.h"
#include "kernels.h"
#include "lab.h"

namespace compeg {

// What the planners size workgroups and grids for: the compute units and the LDS per compute unit of the device the
// calling thread has current (256 and 160 KB on an MI355X in its default mode; fewer CUs in a partitioned mode), asked
// once per device.  (A failed query -- nothing else would work either -- leaves the figures of the one architecture
// the library opens a device of: compeg_gpu_open refuses anything but gfx950.)
struct DeviceLimits {
    uint32_t cus, lds_bytes;
};
static DeviceLimits device_limits()
{
    static std::atomic<uint64_t> known[64]; // cus << 32 | lds_bytes
    int dev = 0;
    const bool indexed = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (indexed) {
        const uint64_t k = known[dev].load(std::memory_order_relaxed);
        if (k)
            return DeviceLimits{uint32_t(k >> 32), uint32_t(k)};
    }
    int cus = 0, lds = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds < 64 * 1024)
        lds = 160 * 1024;
    if (indexed)
        known[dev].store(uint64_t(cus) << 32 | uint32_t(lds), std::memory_order_relaxed);
    if (getenv("COMPEG_VERBOSE"))
        fprintf(stderr, "[compeg] device %d: %d compute units, %d bytes of LDS each\n", dev, cus, lds);
    return DeviceLimits{uint32_t(cus), uint32_t(lds)};
}

// LDS layout (dynamic, 16-byte aligned carve-outs):
//   [L1: 5*256 u16][L2: l2_in_lds u16][per wave: window_words u32 | 64 DU slots]
__global__ void __launch_bounds__(1024)
huffman_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ImageDesc &d = descs[blockIdx.y];
    const uint32_t first_interval = blockIdx.x * blockDim.x;
    if (first_interval >= d.total_intervals)
        return; // uniform for the whole block

    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    const uint32_t wave_area = align16(window_words * 4u) + kWave * kDuSlotBytes;
    uint8_t *wave_base = smem + align16((kL1Entries + l2_in_lds) * 2u);

    stage_luts(d, l1, l2, l2_in_lds, threadIdx.x, blockDim.x);

    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    uint32_t *win = reinterpret_cast<uint32_t *>(wave_base + wave * wave_area);
    uint8_t *slots = reinterpret_cast<uint8_t *>(win) + align16(window_words * 4u);

    const uint32_t wave_first = first_interval + wave * kWave;
    uint32_t win_base = 0, win_len = 0;
    if (wave_first < d.total_intervals) {
        wave_window(d, wave_first, window_words, win_base, win_len);
        stage_window(d, win, win_base, win_len, lane);
    }
    __syncthreads();

    const uint32_t interval = wave_first + lane;
    if (interval >= d.total_intervals)
        return;

    HuffShared s;
    s.l1 = l1;
    s.l2 = l2;
    s.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries);
    s.win = win;
    s.win_base = win_base;
    s.win_len = win_len;
    s.du_slots = slots;
    huff_decode_interval(d, s, interval, lane);
}

// Entropy stage of the two-kernel pipeline: the fast-mode decoder of the fused
// path, writing coefficient records instead of feeding the IDCT.  Few
// registers, so the workgroup is as large as the LDS allows.
template <bool SAMPLES>
__device__ __forceinline__ void entropy_kernel_body(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds,
                                                    uint32_t window_words)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ImageDesc &d = descs[blockIdx.y];
    const uint32_t first_interval = blockIdx.x * blockDim.x;
    if (first_interval >= d.total_intervals)
        return;

    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    const uint32_t wave_area = align16(window_words * 4u) + kWave * kDuSlotBytes;
    uint8_t *wave_base = smem + align16((kL1Entries + l2_in_lds) * 2u);

    stage_luts(d, l1, l2, l2_in_lds, threadIdx.x, blockDim.x);

    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    uint32_t *win = reinterpret_cast<uint32_t *>(wave_base + wave * wave_area);
    const uint32_t wave_first = first_interval + wave * kWave;
    uint32_t win_base = 0, win_len = 0;
    if (wave_first < d.total_intervals) {
        wave_window(d, wave_first, window_words, win_base, win_len);
        stage_window(d, win, win_base, win_len, lane);
    }
    __syncthreads();

    if (wave_first >= d.total_intervals)
        return; // the whole wave; lanes past the last interval of a partly used wave stay (quad stores)

    HuffShared s;
    s.l1 = l1;
    s.l2 = l2;
    s.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries);
    s.win = win;
    s.win_base = win_base;
    s.win_len = win_len;
    s.du_slots = reinterpret_cast<uint8_t *>(win) + align16(window_words * 4u);
    entropy_wave_to_records<SAMPLES>(d, s, wave_first + lane, lane);
}

__global__ void __launch_bounds__(1024)
entropy_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    entropy_kernel_body<false>(descs, l2_in_lds, window_words);
}

// Extension layouts: the same with the IDCT on top (the records carry samples).
// 768 threads: the IDCT's registers allow three waves per SIMD.
__global__ void __launch_bounds__(768)
entropy_samples_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    entropy_kernel_body<true>(descs, l2_in_lds, window_words);
}

// What a wave of the fused kernel does for its NEXT unit while it decodes the current one.  Vector memory
// operations complete in order, stores included, so whatever waits for a load also waits for every store issued
// before it: each step below sits where the wave's latest stores (the composite of an MCU) are an MCU old.
//   data unit 0        the two interval starts that bound the next window are asked for;
//   last MCU but one   one word of every 128-byte line of that window is touched (into the L2);
//   last data unit     decoded: the window is free -- the next one is staged in front of the last IDCT and composite.
//
// Which unit is the next: with a queue (a counter in global memory, zeroed in front of the launch) the one the wave
// draws -- asked for at data unit 0, looked at at data unit 1, the starts asked for there --, so that waves that
// get through their units faster take more of them and the launch's last units are shared out (256 1080p frames are
// 8160 units for 3072 waves: 2.66 each, i.e. three for all by turns); without one, every stride-th.
struct WindowAhead {
    // (an index, not a pointer: a pointer carried around the units' loop is no longer visibly derived from the
    // kernel's restrict argument, and every descriptor field would be read with vector loads)
    const ImageDesc *descs;
    uint32_t image;      // the next unit's image
    bool any;            // ... false: there is no next unit
    uint32_t first;      // ... and its first interval
    uint32_t window_words, lane;
    uint32_t *win;
    uint32_t raw_base, raw_end, base, len, sink;
    uint32_t *queue;     // null: `next` is known from the start (and with a queue: inside a group of units drawn together)
    uint32_t next, drawn, stride, units, waves_per_image, group;

    __device__ __forceinline__ void at(uint32_t du, uint32_t du_total)
    {
        if (queue) {
            if (du == 0u && lane == 0u)
                drawn = atomicAdd(queue, 1u);
            if (du == 1u) {
                // (the groups below stride: the waves' first)
                next = (stride + uint32_t(__builtin_amdgcn_readfirstlane(int(drawn)))) * group;
                any = next < units;
                if (any) {
                    image = uint32_t(__builtin_amdgcn_readfirstlane(int(next / waves_per_image)));
                    first = uint32_t(__builtin_amdgcn_readfirstlane(int((next % waves_per_image) * kWave)));
                }
            }
        }
        if (!any)
            return;
        const ImageDesc *dn = descs + image;
        if (du == (queue ? 1u : 0u))
            wave_window_fetch(*dn, first, raw_base, raw_end);
        if (du_total >= 8u && du == (du_total >= 10u ? du_total - 8u : 2u)) {
            uint32_t b, l;
            wave_window_from(*dn, raw_base, raw_end, window_words, b, l);
            const uint32_t have = dn->nwords > b ? umin(l, dn->nwords - b) : 0u;
            auto *words = CG_GLOBAL(const uint32_t, dn->words);
            for (uint32_t v = lane * 32u; v < have; v += kWave * 32u)
                sink ^= words[b + v];
        }
        if (du_total >= 8u && du == (du_total >= 10u ? du_total - 5u : 5u))
            asm volatile("" ::"v"(sink)); // (the touches' results: never looked at; the register is free from here)
    }

    __device__ __forceinline__ void decoded(uint32_t du, uint32_t du_total)
    {
        if (!any || du + 1u != du_total)
            return;
        const ImageDesc *dn = descs + image;
        uint32_t b, l;
        wave_window_from(*dn, raw_base, raw_end, window_words, b, l);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(b)));
        len = uint32_t(__builtin_amdgcn_readfirstlane(int(l)));
        stage_window(*dn, win, base, len, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

// Fused path: same prologue as huffman_kernel, then every lane runs the whole
// decode of its interval (entropy decode -> IDCT -> composite), so neither
// coefficients nor samples ever touch HBM.
//
// Two grid shapes.  waves_per_image == 0: grid (workgroups per image, images); the last workgroup of an image
// is filled only as far as the image's intervals go (one 4K frame: 21 workgroups of 12 waves and one of 2).
// waves_per_image != 0, for batches whose images all have that many waves and byte-identical LUTs (frames of
// one stream): a one-dimensional grid over the waves of all images, workgroups span image boundaries (every
// wave works from its own image's descriptor; the LUTs are staged from whichever images the workgroup's
// threads belong to -- the same bytes), and only the batch's last workgroup is short.
// (LAYOUT: a struct with the wave's body: Wave422 -- the reference's 4:2:2 -- or WaveLayout<HS, VS>, the extension layouts)
// (GROUP: units a wave draws from the queue at a time)
template <class LAYOUT, uint32_t GROUP>
__device__ __forceinline__ void fused_kernel_body(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words,
                                                  uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    constexpr uint32_t group = GROUP;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x / kWave))), lane = threadIdx.x % kWave;
    uint32_t image = blockIdx.y, wave_first = (blockIdx.x * (blockDim.x / kWave) + wave) * kWave;
    bool has_work = true;
    if (waves_per_image) {
        // (with a queue a wave's units come in groups of `group` consecutive ones -- short units, DRI = 1 --: its first
        // group is the one of its place in the grid)
        const uint32_t flat = (blockIdx.x * (blockDim.x / kWave) + wave) * (queue ? group : 1u); // wave-uniform
        image = flat / waves_per_image;
        wave_first = (flat % waves_per_image) * kWave;
        has_work = image < images;
        image = has_work ? image : images - 1u; // (it still helps staging the LUTs)
    }
    // (wave-uniform by construction; said explicitly so that the descriptor is read through scalar loads)
    image = uint32_t(__builtin_amdgcn_readfirstlane(int(image)));
    wave_first = uint32_t(__builtin_amdgcn_readfirstlane(int(wave_first)));
    if (!waves_per_image && blockIdx.x * blockDim.x >= descs[image].total_intervals)
        return; // the whole workgroup
    has_work = has_work && wave_first < descs[image].total_intervals;

    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    const uint32_t wave_area = align16(window_words * 4u) + kWave * kDuSlotBytes;
    uint8_t *wave_base = smem + align16((kL1Entries + l2_in_lds) * 2u);
    uint32_t *win = reinterpret_cast<uint32_t *>(wave_base + wave * wave_area);
    uint8_t *slots = reinterpret_cast<uint8_t *>(win) + align16(window_words * 4u);

    uint32_t win_base = 0, win_len = 0;
    if (has_work)
        wave_window(descs[image], wave_first, window_words, win_base, win_len);
    stage_luts_and_window(descs[image], l1, l2, l2_in_lds, threadIdx.x, blockDim.x, win, win_base, win_len, lane);
    __syncthreads();

    if (!has_work)
        return; // the whole wave; lanes past the last interval of a partly used wave stay (quad exchange)

    // A launch of more waves than the chip holds at once (flat grid only) is a grid of as many workgroups as
    // fit, whose waves go on to further units of 64 intervals on their own: no workgroup prologue, no barrier and
    // no CU waiting for the slowest wave of a workgroup between two units -- a wave stages its next window itself
    // (WindowAhead).
    const uint32_t stride = gridDim.x * (blockDim.x / kWave), units = waves_per_image * images;
    const uint32_t per_draw = waves_per_image && queue ? group : 1u;
    uint32_t flat = (blockIdx.x * (blockDim.x / kWave) + wave) * per_draw;
    for (;;) {
        const ImageDesc &d = descs[image];
        HuffShared s;
        s.l1 = l1;
        s.l2 = l2;
        s.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries);
        s.win = win;
        s.win_base = win_base;
        s.win_len = win_len;
        s.du_slots = slots;
        // the next unit: the one behind this inside a group, else drawn (WindowAhead) or, without a queue, every stride-th
        // (groups begin at multiples of their size: the place inside one is the unit's index modulo that)
        const bool draw = waves_per_image && queue && flat % per_draw == per_draw - 1u;
        flat += waves_per_image && queue ? 1u : stride; // (wave-uniform; drawn: overwritten below)
        WindowAhead ahead;
        ahead.descs = descs;
        ahead.queue = draw ? queue : nullptr;
        ahead.stride = stride;
        ahead.units = units;
        ahead.waves_per_image = waves_per_image;
        ahead.group = per_draw;
        ahead.next = ahead.drawn = 0u;
        ahead.any = waves_per_image && !draw && flat < units;
        ahead.image = ahead.first = 0u;
        if (ahead.any) {
            ahead.image = uint32_t(__builtin_amdgcn_readfirstlane(int(flat / waves_per_image)));
            ahead.first = uint32_t(__builtin_amdgcn_readfirstlane(int((flat % waves_per_image) * kWave)));
        }
        ahead.window_words = window_words;
        ahead.lane = lane;
        ahead.win = win;
        ahead.sink = 0u;
        LAYOUT::decode(d, s, wave_first + lane, lane, ahead);
        if (!ahead.any)
            break;
        if (draw)
            flat = ahead.next;
        image = ahead.image;
        wave_first = ahead.first;
        win_base = ahead.base;
        win_len = ahead.len;
    }
}

// units of one MCU a lane (DRI = 1) are drawn from the queue four at a time (launch_fused_422)
constexpr uint32_t kMcuQueueGroup = 4;

template <bool WIDE, bool RECORDS = false>
struct Wave422 {
    template <class AHEAD>
    static __device__ __forceinline__ void decode(const ImageDesc &d, const HuffShared &s, uint32_t interval, uint32_t lane, AHEAD &ahead)
    {
        decode_wave_fused_422<AHEAD, WIDE, RECORDS>(d, s, interval, lane, ahead);
    }
};
#if !defined(CG_FUSED_BOUNDS) || !defined(COMPEG_LAB)
#define CG_FUSED_BOUNDS 768 // (laboratory builds: 512 = two waves to a SIMD and their registers)
#endif
__global__ void __launch_bounds__(CG_FUSED_BOUNDS)
decode_fused_422_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words,
                        uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    fused_kernel_body<Wave422<false>, 1>(descs, l2_in_lds, window_words, waves_per_image, images, queue);
}
// The same for launches whose every restart interval is one MCU (BASELINE configs[4]: 8K, DRI = 1): consecutive lanes
// hold consecutive MCUs, and the rows of sixteen of them leave in one piece of 1 KB (composite_row_from_wave).  A
// kernel of its own: as a branch inside the one above the second exchange costs it two spilled registers.
__global__ void __launch_bounds__(768)
decode_fused_422_mcu_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words,
                            uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    fused_kernel_body<Wave422<true>, kMcuQueueGroup>(descs, l2_in_lds, window_words, waves_per_image, images, queue);
}
// The second kernel of the walk + lane-per-MCU route (kernels_body.h): the kernel above over the images' descriptors of
// MCUs -- every "interval" one MCU, begun from the record the walk wrote for it (walk_mcus_422_kernel below).
__global__ void __launch_bounds__(768)
decode_fused_422_mcu_rec_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words,
                                uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    fused_kernel_body<Wave422<true, true>, kMcuQueueGroup>(descs, l2_in_lds, window_words, waves_per_image, images, queue);
}
// Extension layouts: decode_wave_fused_layout behind the plain prologue -- grid (workgroups per image,
// images), no resident waves: inside the larger body above these kernels spill (4:2:0: 356 registers), alone they
// do not (134 / 152 / 186 VGPRs).  Luma 1x1 (4:4:4), 1x2 (4:4:0) ...
template <int HS, int VS, int MC>
__device__ __forceinline__ void fused_layout_kernel_body(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ImageDesc &d = descs[blockIdx.y];
    const uint32_t first_interval = blockIdx.x * blockDim.x;
    if (first_interval >= d.total_intervals)
        return;
    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    const uint32_t wave_area = align16(window_words * 4u) + kWave * kDuSlotBytes;
    uint8_t *wave_base = smem + align16((kL1Entries + l2_in_lds) * 2u);
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x / kWave))), lane = threadIdx.x % kWave;
    uint32_t *win = reinterpret_cast<uint32_t *>(wave_base + wave * wave_area);
    const uint32_t wave_first = first_interval + wave * kWave;
    uint32_t win_base = 0, win_len = 0;
    if (wave_first < d.total_intervals)
        wave_window(d, wave_first, window_words, win_base, win_len);
    stage_luts_and_window(d, l1, l2, l2_in_lds, threadIdx.x, blockDim.x, win, win_base, win_len, lane);
    __syncthreads();
    if (wave_first >= d.total_intervals)
        return; // the whole wave; lanes past the last interval of a partly used wave stay (quad exchange)
    HuffShared s;
    s.l1 = l1;
    s.l2 = l2;
    s.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries);
    s.win = win;
    s.win_base = win_base;
    s.win_len = win_len;
    s.du_slots = reinterpret_cast<uint8_t *>(win) + align16(window_words * 4u);
    decode_wave_fused_layout<HS, VS, MC>(d, s, wave_first + lane, lane);
}

// (4:4:4 and 4:4:0: MCUs in pairs -- rows of 64 bytes; the last MCU of an odd restart interval alone -- from two MCUs an interval on: two waves to a SIMD;
// singly otherwise)
__global__ void __launch_bounds__(512)
decode_fused_444_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    fused_layout_kernel_body<1, 1, 2>(descs, l2_in_lds, window_words);
}
__global__ void __launch_bounds__(768)
decode_fused_444_single_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    fused_layout_kernel_body<1, 1, 1>(descs, l2_in_lds, window_words);
}
__global__ void __launch_bounds__(512)
decode_fused_440_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    fused_layout_kernel_body<1, 2, 2>(descs, l2_in_lds, window_words);
}
__global__ void __launch_bounds__(768)
decode_fused_440_single_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    fused_layout_kernel_body<1, 2, 1>(descs, l2_in_lds, window_words);
}
// ... and 2x2 (4:2:0: six data units of samples per lane, two waves to a SIMD)
__global__ void __launch_bounds__(512)
decode_fused_420_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    fused_layout_kernel_body<2, 2, 1>(descs, l2_in_lds, window_words);
}

// The batch kernel for restart intervals too long for whole-interval windows (decode_wave_fused_422_stream: `rows`
// words of every lane's stream at a time, staged MCU by MCU).  A grid row per image, or -- frames of one stream --
// the flat grid with resident waves that draw their units from the queue.
// (WAVE: a struct with the wave's body -- Wave422Stream, or WaveLayoutStream<HS, VS, MC> for the extension layouts)
template <class WAVE>
__device__ __forceinline__ void fused_stream_kernel_body(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t rows,
                                                         uint32_t stage_after, uint32_t stage_below, uint32_t waves_per_image, uint32_t images,
                                                         uint32_t *queue)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x / kWave))), lane = threadIdx.x % kWave;
    // Two grid shapes, as in decode_fused_422_kernel: (workgroups per image, images), or -- waves_per_image != 0:
    // every image has that many waves and the same LUT bytes -- one dimension over the waves of all images, so
    // that small images (a 960x720 frame with DRI = 30 is three waves) still make workgroups of twelve.
    uint32_t image = blockIdx.y, wave_first = (blockIdx.x * (blockDim.x / kWave) + wave) * kWave;
    bool has_work = true;
    if (waves_per_image) {
        const uint32_t flat = blockIdx.x * (blockDim.x / kWave) + wave;
        image = flat / waves_per_image;
        wave_first = (flat % waves_per_image) * kWave;
        has_work = image < images;
        image = has_work ? image : images - 1u; // (it still helps staging the LUTs)
    }
    image = uint32_t(__builtin_amdgcn_readfirstlane(int(image)));
    wave_first = uint32_t(__builtin_amdgcn_readfirstlane(int(wave_first)));
    const ImageDesc &d = descs[image];
    if (!waves_per_image && blockIdx.x * blockDim.x >= d.total_intervals)
        return; // the whole workgroup
    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    const uint32_t wave_area = rows * kWave * 4u + kWave * kDuSlotBytes;
    uint8_t *wave_base = smem + align16((kL1Entries + l2_in_lds) * 2u);
    uint32_t *win = reinterpret_cast<uint32_t *>(wave_base + wave * wave_area);
    stage_luts_and_window(d, l1, l2, l2_in_lds, threadIdx.x, blockDim.x, win, 0u, 0u, lane);
    __syncthreads();
    if (!has_work || wave_first >= d.total_intervals)
        return; // the whole wave; lanes past the last interval of a partly used wave stay (quad exchange)
    HuffShared s;
    s.l1 = l1;
    s.l2 = l2;
    s.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries);
    s.win = win;
    s.win_base = 0u;
    s.win_len = 0u; // (no whole-interval window: the reference reader's words come from global memory)
    s.du_slots = reinterpret_cast<uint8_t *>(win) + rows * kWave * 4u;
    if (!waves_per_image) {
        WAVE::decode(d, s, rows, stage_after, stage_below, wave_first + lane, lane);
        return;
    }
    // The flat grid is no larger than what is resident at once; a wave goes on to further units of 64 intervals on its
    // own (rows and slots are its own, the tables are every image's): no CU waits for the slowest wave of a workgroup.
    // (which units: drawn from the queue, if there is one, at a unit's start and looked at at its end -- see WindowAhead)
    const uint32_t stride = gridDim.x * (blockDim.x / kWave), units = waves_per_image * images;
    for (uint32_t flat = blockIdx.x * (blockDim.x / kWave) + wave;;) {
        uint32_t drawn = 0u;
        if (queue && lane == 0u)
            drawn = atomicAdd(queue, 1u);
        WAVE::decode(descs[image], s, rows, stage_after, stage_below, wave_first + lane, lane);
        flat = queue ? stride + uint32_t(__builtin_amdgcn_readfirstlane(int(drawn))) : flat + stride; // (wave-uniform)
        if (flat >= units)
            break;
        image = uint32_t(__builtin_amdgcn_readfirstlane(int(flat / waves_per_image)));
        wave_first = uint32_t(__builtin_amdgcn_readfirstlane(int((flat % waves_per_image) * kWave)));
    }
}

struct Wave422Stream {
    static __device__ __forceinline__ void decode(const ImageDesc &d, const HuffShared &s, uint32_t rows, uint32_t stage_after,
                                                  uint32_t stage_below, uint32_t interval, uint32_t lane)
    {
        decode_wave_fused_422_stream(d, s, rows, stage_after, stage_below, interval, lane);
    }
};
template <int HS, int VS, int MC>
struct WaveLayoutStream {
    static __device__ __forceinline__ void decode(const ImageDesc &d, const HuffShared &s, uint32_t rows, uint32_t stage_after,
                                                  uint32_t stage_below, uint32_t interval, uint32_t lane)
    {
        decode_wave_fused_layout<HS, VS, MC, true>(d, s, interval, lane, rows, stage_after, stage_below);
    }
};
// The first kernel of the walk + lane-per-MCU route (kernels_body.h): a lane per restart interval, entropy decode only,
// the decoder's state at every MCU's start into ImageDesc::mcu_word / mcu_state.  The launch shapes of the streamed
// batch kernel (fused_stream_kernel_body): a grid row per image, or the flat grid with resident waves and the units' queue.
// LDS: [L1][L2 + direct tables][walk tables, if the images have them][96 bytes nobody reads][per wave: 64 lists | the slow road's window | rows + 3]
constexpr uint32_t kWalkSpareRows = 3; // (the walk reads three rows at its position: one behind the last staged, never used, and room to run over)
__global__ void __launch_bounds__(768)
walk_mcus_422_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t rows, uint32_t stage_below,
                     uint32_t waves_per_image, uint32_t images, uint32_t *queue, uint32_t with_walk_tables, uint32_t chunk)
{
    extern __shared__ __attribute__((aligned(32))) uint8_t smem[];
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x / kWave))), lane = threadIdx.x % kWave;
    uint32_t image = blockIdx.y, wave_first = (blockIdx.x * (blockDim.x / kWave) + wave) * kWave;
    bool has_work = true;
    if (waves_per_image) {
        const uint32_t flat = blockIdx.x * (blockDim.x / kWave) + wave;
        image = flat / waves_per_image;
        wave_first = (flat % waves_per_image) * kWave;
        has_work = image < images;
        image = has_work ? image : images - 1u; // (it still helps staging the tables)
    }
    image = uint32_t(__builtin_amdgcn_readfirstlane(int(image)));
    wave_first = uint32_t(__builtin_amdgcn_readfirstlane(int(wave_first)));
    const ImageDesc &d = descs[image];
    if (!waves_per_image && blockIdx.x * blockDim.x >= d.total_intervals)
        return; // the whole workgroup
    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    // (as offsets from smem, see decode_coop_team_422_kernel: behind an integer round trip a pointer is no longer LDS)
    const uint32_t walk_off = (align16((kL1Entries + l2_in_lds) * 2u) + 31u) & ~31u;
    uint32_t *walk = reinterpret_cast<uint32_t *>(smem + walk_off);
    const bool tables = with_walk_tables != 0u && d.walk != nullptr;
    const uint32_t dump_off = walk_off + (with_walk_tables ? kWalkWords * 4u : 0u);
    int16_t *dump = reinterpret_cast<int16_t *>(smem + dump_off);
    const uint32_t wave_area = uint32_t(kWave) * walk_list_bytes(chunk) + kWalkSlowWords * 4u + (rows + kWalkSpareRows) * kWalkRowBytes;
    uint32_t *lists = reinterpret_cast<uint32_t *>(smem + dump_off + 96u + wave * wave_area);
    uint32_t *slow_window = lists + uint32_t(kWave) * walk_list_bytes(chunk) / 4u;
    uint32_t *win = slow_window + kWalkSlowWords;
    stage_luts(d, l1, l2, l2_in_lds, threadIdx.x, blockDim.x, 2u * kDcFastEntries);
    if (tables)
        copy_words_to_lds(walk, d.walk, kWalkWords, threadIdx.x, blockDim.x);
    __syncthreads();
    if (!has_work || wave_first >= d.total_intervals)
        return; // the whole wave
    HuffShared s;
    s.l1 = l1;
    s.l2 = l2;
    s.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries + 2u * kDcFastEntries);
    s.win = win;
    s.win_base = 0u;
    s.win_len = 0u;
    s.du_slots = nullptr;
    WalkTabs tabs;
    walk_tabs(d, s, tables ? walk : nullptr, tabs);
    if (!waves_per_image) {
        walk_wave_422(d, s, tabs, lists, slow_window, dump, rows, stage_below, chunk, wave_first + lane, lane);
        return;
    }
    // (the flat grid: resident waves that go on to further units of 64 intervals, drawn from the queue if there is one --
    // the images of such a launch carry the same tables)
    const uint32_t stride = gridDim.x * (blockDim.x / kWave), units = waves_per_image * images;
    for (uint32_t flat = blockIdx.x * (blockDim.x / kWave) + wave;;) {
        uint32_t drawn = 0u;
        if (queue && lane == 0u)
            drawn = atomicAdd(queue, 1u);
        walk_wave_422(descs[image], s, tabs, lists, slow_window, dump, rows, stage_below, chunk, wave_first + lane, lane);
        flat = queue ? stride + uint32_t(__builtin_amdgcn_readfirstlane(int(drawn))) : flat + stride; // (wave-uniform)
        if (flat >= units)
            break;
        image = uint32_t(__builtin_amdgcn_readfirstlane(int(flat / waves_per_image)));
        wave_first = uint32_t(__builtin_amdgcn_readfirstlane(int((flat % waves_per_image) * kWave)));
    }
}
__global__ void __launch_bounds__(768)
decode_fused_422_stream_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t rows, uint32_t stage_after,
                               uint32_t stage_below, uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    fused_stream_kernel_body<Wave422Stream>(descs, l2_in_lds, rows, stage_after, stage_below, waves_per_image, images, queue);
}
// ... and of the extension layouts' kernels (4:4:4 / 4:4:0 in pairs -- restart intervals of two MCUs or more --, 4:2:0)
__global__ void __launch_bounds__(512)
decode_fused_444_stream_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t rows, uint32_t stage_after,
                               uint32_t stage_below, uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    fused_stream_kernel_body<WaveLayoutStream<1, 1, 2>>(descs, l2_in_lds, rows, stage_after, stage_below, waves_per_image, images, queue);
}
__global__ void __launch_bounds__(512)
decode_fused_440_stream_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t rows, uint32_t stage_after,
                               uint32_t stage_below, uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    fused_stream_kernel_body<WaveLayoutStream<1, 2, 2>>(descs, l2_in_lds, rows, stage_after, stage_below, waves_per_image, images, queue);
}
__global__ void __launch_bounds__(512)
decode_fused_420_stream_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t rows, uint32_t stage_after,
                               uint32_t stage_below, uint32_t waves_per_image, uint32_t images, uint32_t *queue)
{
    fused_stream_kernel_body<WaveLayoutStream<2, 2, 1>>(descs, l2_in_lds, rows, stage_after, stage_below, waves_per_image, images, queue);
}

// Latency-oriented variant of the fused path for launches that cannot fill
// the chip (a single 4K frame with DRI=4 is 254 waves for 1024 SIMDs): every
// workgroup is a PAIR of waves working on the same 64 restart intervals.
// Wave 0 only entropy-decodes; wave 1 takes each finished data unit out of LDS
// and runs IDCT + composite while wave 0 is already decoding the next one.
// Coefficient slots and DC terms are double-buffered in LDS; one workgroup
// barrier per data unit hands a buffer over.  Same arithmetic, same bytes out.
__global__ void __launch_bounds__(128, 2)
decode_pair_422_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ImageDesc &d = descs[blockIdx.y];
    const uint32_t first_interval = blockIdx.x * kWave;
    if (first_interval >= d.total_intervals)
        return;

    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    uint8_t *area = smem + align16((kL1Entries + l2_in_lds) * 2u);
    uint32_t *win = reinterpret_cast<uint32_t *>(area);
    uint8_t *slots = area + align16(window_words * 4u);          // 2 x 64 slots
    int32_t *dcs = reinterpret_cast<int32_t *>(slots + 2u * kWave * kDuSlotBytes); // 2 x 64 DC terms

    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    uint32_t win_base = 0, win_len = 0;
    wave_window(d, first_interval, window_words, win_base, win_len);
    // the decoder wave issues the window's loads (16 bytes per lane, eight in flight) and both waves
    // copy the LUTs under their latency: one memory round trip behind the start positions
    if (wave == 0)
        stage_luts_and_window(d, l1, l2, l2_in_lds, threadIdx.x, blockDim.x, win, win_base, win_len, lane);
    else
        stage_luts(d, l1, l2, l2_in_lds, threadIdx.x, blockDim.x);
    for (uint32_t i = threadIdx.x; i < 2u * kWave * kDuSlotBytes / 4u; i += blockDim.x)
        reinterpret_cast<slot_word_t *>(slots)[i] = 0u;
    __syncthreads();

    const uint32_t interval = first_interval + lane;
    const bool active = interval < d.total_intervals;
    const uint32_t du_total = d.restart_interval * 4u;

    HuffShared s;
    s.l1 = l1;
    s.l2 = l2;
    s.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries);
    s.win = win;
    s.win_base = win_base;
    s.win_len = win_len;
    s.du_slots = slots;

    if (wave == 0) {
        EntropyState e;
        if (active)
            entropy_init(e, d, s, interval);
#pragma unroll 1
        for (uint32_t du = 0; du < du_total; du++) {
            const uint32_t k = du & 3u, comp = k < 2u ? 0u : k - 1u, set = du & 1u;
            if (active) {
                uint8_t *slot = slots + (set * kWave + lane) * kDuSlotBytes;
                dcs[set * kWave + lane] = entropy_data_unit(e, d, s, comp, reinterpret_cast<int16_t *>(slot));
            }
            __syncthreads(); // hand data unit `du` to the transformer
        }
    } else {
        PixelState t;
        pixel_init(t, d, active ? interval : 0u, active);
#pragma unroll 1
        for (uint32_t du = 0; du < du_total; du++) {
            const uint32_t k = du & 3u, comp = k < 2u ? 0u : k - 1u, set = du & 1u;
            __syncthreads(); // data unit `du` is complete
            uint8_t *set_slots = slots + set * kWave * kDuSlotBytes;
            if (active)
                pixel_transform(t, d, comp, k, set_slots + lane * kDuSlotBytes, dcs[set * kWave + lane]);
            if (k == 3u)
                composite_mcus_422<false>(t, d, set_slots, lane);
        }
    }
}

// One lane per data unit for the IDCT, then the same lanes regroup (through
// LDS) so that every wave-wide store writes 16 MCUs x 64 contiguous bytes.
__global__ void __launch_bounds__(256)
idct_composite_kernel(const ImageDesc *__restrict__ descs)
{
    __shared__ float quant[3 * kRetained];
    __shared__ uint32_t px[4 * kWave * kPxSlotWords];

    const ImageDesc &d = descs[blockIdx.y];
    const uint32_t first_du = blockIdx.x * blockDim.x;
    if (first_du >= d.total_dus)
        return;
    if (threadIdx.x < 3 * kRetained)
        quant[threadIdx.x] = d.quant[threadIdx.x / kRetained][threadIdx.x % kRetained];
    __syncthreads();

    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    uint32_t *wave_px = px + wave * kWave * kPxSlotWords;
    const uint32_t du = first_du + threadIdx.x;
    if (du < d.total_dus) {
        const uint32_t k = du % d.dus_per_mcu;
        const uint32_t comp = (d.comp_of_du >> (2u * k)) & 3u;
        // 64-byte record, four 16-byte loads
        auto *src = CG_GLOBAL(const Vec4u, reinterpret_cast<const Vec4u *>(d.ac + size_t(du) * kRetained));
        uint32_t rec[kRetained / 2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const Vec4u v = src[i];
            rec[4 * i + 0] = v.x;
            rec[4 * i + 1] = v.y;
            rec[4 * i + 2] = v.z;
            rec[4 * i + 3] = v.w;
        }
        uint32_t out[16];
        idct_data_unit(rec, CG_GLOBAL(const int32_t, d.dc)[du], quant + comp * kRetained, out);
        uint32_t *slot = wave_px + lane * kPxSlotWords;
#pragma unroll
        for (int i = 0; i < 16; i++)
            slot[i] = out[i];
    }
    __syncthreads();

    const uint32_t total_mcus = d.total_dus / d.dus_per_mcu;
    const uint32_t first_mcu = (first_du + wave * kWave) / 4u; // 4:2:2: 4 data units per MCU
    composite_422(d, wave_px, first_mcu, total_mcus, lane);
}

#if defined(CG_AC_STAMPS)
hipError_t read_ac_stamps(unsigned long long out[4], bool reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ac_stamps), sizeof(unsigned long long) * 4);
    if (e == hipSuccess && reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_ac_stamps), z, sizeof z);
    }
    return e;
}
#endif

// Extension layouts: composite from the sample records entropy_samples_kernel wrote.  A workgroup
// takes a strip of 32 MCUs of one MCU row: their records (contiguous in memory) go to LDS with
// 16-byte loads, then every lane converts four pixels of a row at a time, so that a wave-wide
// store writes 1 KB of one pixel row.
__global__ void __launch_bounds__(256)
composite_generic_kernel(const ImageDesc *__restrict__ descs)
{
    __shared__ __attribute__((aligned(16))) uint8_t strip[kStripMcus * kMaxDusPerMcu * kRetained * 2];
    const ImageDesc &d = descs[blockIdx.z];
    const uint32_t mcu_row = blockIdx.y, first = blockIdx.x * kStripMcus;
    if (mcu_row * d.mcu_h >= d.out_h || first >= d.width_mcus)
        return; // uniform for the workgroup
    const uint32_t in_strip = umin(kStripMcus, d.width_mcus - first);
    const uint32_t mcu0 = mcu_row * d.width_mcus + first;
    const uint32_t total_mcus = d.total_intervals * d.restart_interval;
    const uint32_t decoded = mcu0 < total_mcus ? umin(in_strip, total_mcus - mcu0) : 0u;
    const uint32_t nvec = decoded * d.dus_per_mcu * 4u; // 16-byte pieces
    auto *src = CG_GLOBAL(const Vec4u, reinterpret_cast<const Vec4u *>(d.ac + size_t(mcu0) * d.dus_per_mcu * kRetained));
    for (uint32_t v = threadIdx.x; v < nvec; v += blockDim.x)
        reinterpret_cast<Vec4u *>(strip)[v] = src[v];
    __syncthreads();
    // 4-pixel groups per row of a full strip: 64 (8-pixel MCUs) or 128; rows per pass 4 or 2
    const uint32_t gsh = d.mcu_w == 16u ? 7u : 6u;
    const uint32_t xg = threadIdx.x & ((1u << gsh) - 1u), r0 = threadIdx.x >> gsh, rstep = blockDim.x >> gsh;
    if (xg * 4u >= in_strip * d.mcu_w)
        return;
    for (uint32_t r = r0; r < d.mcu_h; r += rstep)
        composite_generic_4px(d, first * d.mcu_w + xg * 4u, mcu_row * d.mcu_h + r, strip, mcu0);
}

namespace {
constexpr uint32_t kMaxWavesFused = 12; // 3 per SIMD: what 168 VGPRs allow
constexpr uint32_t kMaxWavesSplit = 16; // entropy_kernel: 4 per SIMD
} // namespace

// Waves per workgroup of the extension-layout kernels (their registers allow two waves per SIMD): 4:2:0 eight -- its
// windows leave room for one workgroup per CU; the paired 8-pixel-MCU kernels four, two workgroups per CU (64 x 4K,
// 4:4:4 / 4:4:0: 1.095 / 0.864 ms per launch against 1.144 / 0.896 with one workgroup of eight; three or five waves,
// i.e. nine or ten per CU, are slower than either, and so are twelve -- three workgroups of four with streamed windows:
// 1.40 / 1.09 ms).
uint32_t fused_layout_wave_cap(uint32_t hs, uint32_t vs, bool pairs) { return hs == 2 && vs == 2 ? 8u : (hs == 1 && pairs ? 4u : 12u); }

HuffLdsPlan plan_huffman(uint32_t max_intervals, uint32_t images, uint32_t max_l2,
                         uint32_t max_wave_words, bool fused, uint32_t wave_cap)
{
    HuffLdsPlan p{};
    // everything behind L1: the L2 LUT and the two direct AC tables (at most 24 KB of LDS)
    p.l2_entries_in_lds = max_l2 < 12288u ? (max_l2 + 1u) & ~1u : 12288u;
    const uint32_t tables = (((kL1Entries + p.l2_entries_in_lds) * 2u) + 15u) & ~15u;
    const uint32_t slots = kWave * kDuSlotBytes;

    // Window: the largest word span of any wave's 64 intervals (the host knows
    // every start offset) plus the per-data-unit slack; waves that need more
    // than the cap fall back to global reads for the excess.
    uint32_t w = max_wave_words + kDuWordSlack + 4u;
    if (w < 256u)
        w = 256u;
    p.window_cut = w > 6144u;
    if (w > 6144u)
        w = 6144u;
    if (const char *e = lab_env("COMPEG_WINDOW_CAP")) // experiment knob (words)
        w = w < uint32_t(atoi(e)) ? w : uint32_t(atoi(e));
    p.window_words = (w + 3u) & ~3u;
    const uint32_t wave_area = ((p.window_words * 4u + 15u) & ~15u) + slots;

    // Workgroup shape.  As many waves per workgroup as one CU's share of the launch (256 CUs), so that a
    // launch that fits the chip runs in one round with one workgroup per CU (one 4K frame with DRI=4 is 254
    // waves: one wave per workgroup, a CU each); at most what the LDS holds and 12 (3 per SIMD) -- a launch
    // that oversubscribes the chip fills every CU with one such workgroup at a time.
    const DeviceLimits lim = device_limits();
    const uint32_t lds_per_cu = lim.lds_bytes, cu_count = lim.cus;
    const uint64_t total_waves = uint64_t((max_intervals + kWave - 1) / kWave) * images;
    const uint32_t fit = (lds_per_cu - tables) / wave_area;
    p.waves_that_fit = fit;
    const uint32_t most = std::max(1u, std::min(std::min(fit, fused ? kMaxWavesFused : kMaxWavesSplit), wave_cap ? wave_cap : 64u));
    const uint64_t per_cu = (total_waves + cu_count - 1u) / cu_count;
    uint32_t wpb = uint32_t(std::min<uint64_t>(std::max<uint64_t>(per_cu, 1u), most));
    // (every image rounds up to whole workgroups: a few more waves per workgroup can save a second round)
    auto groups = [&](uint32_t w) { return uint64_t((max_intervals + w * kWave - 1) / (w * kWave)) * images; };
    auto groups_per_cu = [&](uint32_t w) {
        return std::max(1u, std::min(lds_per_cu / (tables + w * wave_area), most / w));
    };
    uint64_t rounds = 1;
    if (per_cu > most) {
        // two or three rounds: split the waves evenly over them (16 4K frames are 15.9 waves per CU: two
        // rounds of 8 rather than one of 12 and one of 4 that takes as long as a full one)
        rounds = (per_cu + most - 1) / most;
        if (rounds <= 3)
            wpb = uint32_t(std::min<uint64_t>((per_cu + rounds - 1) / rounds, most));
    }
    while (wpb < most && groups(wpb) > uint64_t(cu_count) * rounds * groups_per_cu(wpb))
        wpb++;
    if (const char *e = lab_env("COMPEG_WPB")) // experiment knob
        wpb = atoi(e) > 0 ? uint32_t(atoi(e)) : wpb;
    p.waves_per_block = wpb;
    p.total_bytes = tables + p.waves_per_block * wave_area;
    if (const char *e = lab_env("COMPEG_LDS_PAD")) // experiment knob: lowers occupancy
        p.total_bytes += uint32_t(atoi(e));
    if (getenv("COMPEG_VERBOSE"))
        fprintf(stderr, "[compeg] plan: images=%u intervals=%u waves/block=%u window=%u words l2=%u lds=%u B\n",
                images, max_intervals, p.waves_per_block, p.window_words, p.l2_entries_in_lds, p.total_bytes);
    return p;
}

hipError_t launch_huffman(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                          const HuffLdsPlan &plan, hipStream_t stream)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    const uint32_t threads = plan.waves_per_block * kWave;
    dim3 grid((max_intervals + threads - 1) / threads, images, 1);
    const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void *>(huffman_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    hipLaunchKernelGGL(huffman_kernel, grid, dim3(threads), plan.total_bytes, stream, descs,
                       plan.l2_entries_in_lds, plan.window_words);
    return hipGetLastError();
}

hipError_t launch_entropy(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                          const HuffLdsPlan &plan, hipStream_t stream)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    const uint32_t threads = plan.waves_per_block * kWave;
    dim3 grid((max_intervals + threads - 1) / threads, images, 1);
    const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void *>(entropy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    hipLaunchKernelGGL(entropy_kernel, grid, dim3(threads), plan.total_bytes, stream, descs,
                       plan.l2_entries_in_lds, plan.window_words);
    return hipGetLastError();
}

hipError_t launch_fused_layout(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const HuffLdsPlan &plan,
                               uint32_t hs, uint32_t vs, bool pairs, hipStream_t stream)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    using Kernel = void (*)(const ImageDesc *, uint32_t, uint32_t);
    const Kernel kernel = hs == 1 && vs == 1   ? (pairs ? decode_fused_444_kernel : decode_fused_444_single_kernel)
                          : hs == 1 && vs == 2 ? (pairs ? decode_fused_440_kernel : decode_fused_440_single_kernel)
                          : hs == 2 && vs == 2 ? decode_fused_420_kernel
                                               : nullptr;
    if (!kernel || plan.waves_per_block > fused_layout_wave_cap(hs, vs, pairs))
        return hipErrorInvalidValue;
    const uint32_t threads = plan.waves_per_block * kWave;
    dim3 grid((max_intervals + threads - 1) / threads, images, 1);
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    hipLaunchKernelGGL(kernel, grid, dim3(threads), plan.total_bytes, stream, descs, plan.l2_entries_in_lds, plan.window_words);
    return hipGetLastError();
}

hipError_t launch_fused_422(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                            const HuffLdsPlan &plan, hipStream_t stream, bool uniform, bool one_mcu_intervals, uint32_t *queue, bool from_records)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    static const int wide_knob = [] {
        const char *e = lab_env("COMPEG_WIDE"); // experiment knob: 0 = the quad exchange for every restart interval, 2 = the wave-wide one
        return e ? atoi(e) : 1;
    }();
    const auto kernel = from_records ? decode_fused_422_mcu_rec_kernel
                        : (one_mcu_intervals && wide_knob != 0) || wide_knob == 2 ? decode_fused_422_mcu_kernel : decode_fused_422_kernel;
    const uint32_t wave_limit = kMaxWavesFused;
    const uint32_t threads = plan.waves_per_block * kWave;
    // uniform: every image has max_intervals intervals and the same LUT bytes -> workgroups may span images
    static const bool flat_allowed = [] {
        const char *e = lab_env("COMPEG_FLAT"); // experiment knob: 0 = always one grid row per image
        return e ? atoi(e) != 0 : true;
    }();
    const uint32_t waves_per_image = (uniform && flat_allowed) ? (max_intervals + kWave - 1) / kWave : 0u;
    const uint64_t flat_groups = (uint64_t(waves_per_image) * images + plan.waves_per_block - 1) / plan.waves_per_block;
    dim3 grid((max_intervals + threads - 1) / threads, images, 1);
    if (waves_per_image && flat_groups <= 0x7fffffffu) {
        // at most as many workgroups as are resident at once; their waves loop over the rest (see the kernel)
        static const int resident_cap = [] {
            const char *e = lab_env("COMPEG_RESIDENT"); // experiment knob: 0 = a workgroup per 12 units as before, n > 1 = at most n workgroups (part of the chip)
            return e ? atoi(e) : 1;
        }();
        // (the CUs of the device the launch goes to: a wrong count would cost time, not results -- the waves' stride
        // is the grid's size whatever it is)
        const DeviceLimits lim = device_limits();
        const uint32_t per_cu = std::max(1u, std::min(lim.lds_bytes / plan.total_bytes, wave_limit / plan.waves_per_block));
        const uint32_t cus = lim.cus;
        const uint64_t resident = uint64_t(cus) * per_cu;
        grid = dim3(uint32_t(resident_cap ? std::min(flat_groups, resident_cap > 1 ? uint64_t(resident_cap) : resident) : flat_groups), 1, 1);
    }
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    const bool flat = grid.y == 1 && waves_per_image;
    static const bool queue_allowed = [] {
        const char *e = lab_env("COMPEG_QUEUE"); // experiment knob: 0 = every wave takes every stride-th unit
        return e ? atoi(e) != 0 : true;
    }();
    // The units' queue: where the waves have more than one draw each.  Units of one MCU a lane (DRI = 1) are drawn four
    // at a time: one by one the draw and the later request for the starts cost more than the sharing brings (8 x 8K:
    // 0.426 against 0.397 ms without a queue; 256 x 4K DRI = 4 2.74 against 2.86, 300 x 720p 0.46 against 0.51).
    const uint32_t group = kernel == decode_fused_422_mcu_kernel || kernel == decode_fused_422_mcu_rec_kernel ? kMcuQueueGroup : 1u;
    uint32_t *q = flat && queue_allowed && queue && uint64_t(waves_per_image) * images > uint64_t(grid.x) * plan.waves_per_block * group
                      ? queue
                      : nullptr;
    if (q) {
        const hipError_t z = hipMemsetAsync(q, 0, sizeof(uint32_t), stream);
        if (z != hipSuccess)
            return z;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(threads), plan.total_bytes, stream, descs, plan.l2_entries_in_lds, plan.window_words,
                       flat ? waves_per_image : 0u, images, q);
    return hipGetLastError();
}

// Where whole-interval windows are longer than a window can be, or leave a CU fewer than eight of its twelve waves
// and fewer than the launch would put there (256 frames of 960x720, whole windows / streamed, ms per launch: DRI 6
// 0.335 / 0.412 -- nine waves fit --, 8 0.397 / 0.336, 10 0.469 / 0.366, 16 1.22 / 0.42, 30 1.14 / 0.64, 60 2.2 / 1.18).
bool stream_plan_preferred(const HuffLdsPlan &plan, uint32_t max_intervals, uint32_t images, uint32_t cu_waves, uint32_t group_waves)
{
    const DeviceLimits lim = device_limits();
    const uint64_t waves = uint64_t((max_intervals + kWave - 1) / kWave) * images;
    if (!cu_waves) {
        const uint64_t per_cu = std::min<uint64_t>(kMaxWavesFused, (waves + lim.cus - 1u) / lim.cus);
        return plan.window_cut || (plan.waves_that_fit < 8u && per_cu > plan.waves_that_fit);
    }
    // the extension layouts' kernels (cu_waves of them on a CU, in workgroups of at most group_waves, each with the
    // tables of its own): streamed wherever whole windows cost the CU a wave the launch would put there
    group_waves = group_waves ? std::min(group_waves, cu_waves) : cu_waves;
    const uint32_t groups = cu_waves / group_waves;
    const uint32_t tables = (((kL1Entries + plan.l2_entries_in_lds) * 2u) + 15u) & ~15u;
    const uint32_t wave_area = ((plan.window_words * 4u + 15u) & ~15u) + kWave * kDuSlotBytes;
    const uint32_t room = lim.lds_bytes / groups > tables ? lim.lds_bytes / groups - tables : 0u;
    const uint32_t fit = groups * std::min(group_waves, room / wave_area);
    const uint64_t per_cu = std::min<uint64_t>(cu_waves, (waves + lim.cus - 1u) / lim.cus);
    return plan.window_cut || (fit < cu_waves && per_cu > fit);
}

// Rows for the streamed window: room for `mcu_words` (the launch's average MCU, rounded up) four times over and
// the words a reader holds beyond, as many as leave the CU its twelve waves, never fewer than sixteen.
StreamPlan plan_stream(uint32_t max_intervals, uint32_t images, uint32_t max_l2, uint32_t mcu_words, bool uniform, uint32_t cu_waves,
                       uint32_t group_waves)
{
    // cu_waves: waves a CU holds of this kernel (twelve; eight of the extension layouts'); group_waves: the most a
    // workgroup has (the paired layout kernels: four, two workgroups to a CU -- fused_layout_wave_cap)
    cu_waves = cu_waves ? cu_waves : kMaxWavesFused;
    group_waves = group_waves ? std::min(group_waves, cu_waves) : cu_waves;
    const uint32_t groups_per_cu = cu_waves / group_waves;
    StreamPlan p;
    p.l2_entries_in_lds = max_l2 < 12288u ? (max_l2 + 1u) & ~1u : 12288u;
    const uint32_t tables = (((kL1Entries + p.l2_entries_in_lds) * 2u) + 15u) & ~15u;
    const DeviceLimits lim = device_limits();
    // How often the rows are staged anew -- behind every MCU, behind its luma and its chroma data units, behind every
    // data unit -- and how many: three times what a lane reads on average between two stagings (luma: 0.7 of an
    // MCU's words at most; one data unit: 0.45) and the words a reader holds beyond.  The coarsest step whose rows
    // leave a CU its twelve waves (or all but one of them); where none does (10 bit per pixel), every MCU and at most 64
    // rows.  (256 x 960x720, DRI = 10, ms per launch by step: 1.7 bit per pixel 0.361 / 0.370 / 0.381; 4.6 bit 1.04 /
    // 1.08 / 0.73; 10 bit 1.48 / 1.54 / 1.62.)
    // (several workgroups to a CU: a little less than the arithmetic says -- three of 54 272 bytes did not share a CU)
    const uint32_t lds_room = lim.lds_bytes - (groups_per_cu > 1u ? groups_per_cu * 2048u : 0u);
    const uint32_t most_rows = ((lds_room - groups_per_cu * tables) / cu_waves - kWave * kDuSlotBytes) / (kWave * 4u);
    const uint32_t need[3] = {3u * mcu_words + 4u, 3u * ((7u * mcu_words + 9u) / 10u) + 4u, 3u * ((9u * mcu_words + 19u) / 20u) + 4u};
    uint32_t step = need[0] <= most_rows ? 0u : (need[1] <= most_rows ? 1u : (need[2] <= most_rows + 6u ? 2u : 0u));
    if (const char *e = lab_env("COMPEG_STREAM_STEP")) // experiment knob: 0 / 1 / 2
        step = uint32_t(std::max(0, std::min(2, atoi(e))));
    uint32_t rows = std::max(12u, std::min(64u, need[step]));
    if (groups_per_cu > 1u)
        rows = std::min(rows, std::max(12u, most_rows)); // (so that the workgroups do share the CU)
    if (const char *e = lab_env("COMPEG_STREAM_ROWS")) // experiment knob
        rows = uint32_t(std::max(4, std::min(128, atoi(e))));
    p.stage_after = step == 0u ? 0x8u : (step == 1u ? 0xau : 0xfu);
    // staged anew when a lane has less than one of those averages and the reader's words in front of it (256 x
    // 960x720 DRI = 10, 2048 frames, ms per launch: always 3.01, below 14 words 2.73, below 10 2.67, below 7 2.66; a fetch
    // ahead into the L2 for the next staging, never waited for, changed nothing)
    p.stage_below = std::min(rows, (need[step] - 4u) / 3u + 3u);
    if (const char *e = lab_env("COMPEG_STREAM_BELOW")) // experiment knob
        p.stage_below = uint32_t(std::max(0, atoi(e)));
    const uint32_t wave_area = rows * kWave * 4u + kWave * kDuSlotBytes;
    const uint32_t waves_per_image = (max_intervals + kWave - 1) / kWave;
    const uint64_t total_waves = uint64_t(waves_per_image) * images;
    // Waves per workgroup: one CU's share of the launch (a launch smaller than the chip spreads over all of its CUs:
    // 256 frames of 960x720 with DRI = 30 are 768 waves), at most what the LDS holds and twelve; the per-image grid
    // never more than an image has.  (Workgroups smaller than the share, two to a CU, measured no better.)
    const uint32_t fit = std::max(1u, (lds_room / groups_per_cu - tables) / wave_area);
    uint32_t best = uint32_t(std::min<uint64_t>(std::max<uint64_t>((total_waves + lim.cus - 1) / lim.cus, 1u), std::min(fit, group_waves)));
    if (!uniform)
        best = std::min(best, std::max(1u, waves_per_image));
    // Thirteen to sixteen units a CU: two rounds of eight resident waves (two to a SIMD) rather than one of twelve and a
    // nearly empty one -- a wave's unit takes a fifth longer three to a SIMD, and the second round's few units a whole
    // unit's time again (256 x 1280x720, us per launch, twelve / eight waves a CU: DRI = 9 -- 3328 units -- 601 / 502,
    // DRI = 8 -- 3840 -- 569 / 474; 4608 units, 18 a CU: 668 / 734, left at twelve).
    const uint64_t units_per_cu = (total_waves + lim.cus - 1) / lim.cus;
    if (uniform && cu_waves == kMaxWavesFused && group_waves == cu_waves && units_per_cu >= 13u && units_per_cu <= 16u && fit >= 8u)
        best = 8u;
    if (const char *e = lab_env("COMPEG_WPB")) // experiment knob
        best = atoi(e) > 0 ? uint32_t(atoi(e)) : best;
    // With LDS to spare at that size, more rows: the stagings get rarer by as much as they get larger, and every
    // one of them waits for the stores in front of it (sparse streams, short MCU steps -- 256 x 1080p q50 DRI = 16:
    // 0.90 ms with 12 rows, 0.83 with the 27 that fit).
    const uint32_t spare_rows = best ? ((lds_room / groups_per_cu - tables) / best - kWave * kDuSlotBytes) / (kWave * 4u) : rows;
    if (!lab_env("COMPEG_STREAM_ROWS") && spare_rows > rows)
        rows = std::min(64u, spare_rows);
    const uint32_t area = rows * kWave * 4u + kWave * kDuSlotBytes;
    p.rows = rows;
    p.waves_per_block = best;
    p.total_bytes = tables + best * area;
    p.waves_per_image = uniform ? waves_per_image : 0u;
    p.cu_waves = cu_waves;
    if (getenv("COMPEG_VERBOSE"))
        fprintf(stderr, "[compeg] stream plan: images=%u intervals=%u waves/block=%u rows=%u staged behind data units %#x below %u words lds=%u B%s\n",
                images, max_intervals, best, rows, p.stage_after, p.stage_below, p.total_bytes, uniform ? " flat" : "");
    return p;
}

hipError_t launch_fused_stream(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const StreamPlan &plan,
                                   hipStream_t stream, uint32_t hs, uint32_t vs, uint32_t *queue)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    using Kernel = void (*)(const ImageDesc *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t *);
    const Kernel kernel = hs == 2 && vs == 1   ? decode_fused_422_stream_kernel
                          : hs == 1 && vs == 1 ? decode_fused_444_stream_kernel
                          : hs == 1 && vs == 2 ? decode_fused_440_stream_kernel
                          : hs == 2 && vs == 2 ? decode_fused_420_stream_kernel
                                               : nullptr;
    if (!kernel)
        return hipErrorInvalidValue;
    const uint32_t threads = plan.waves_per_block * kWave;
    dim3 grid((max_intervals + threads - 1) / threads, images, 1);
    const uint64_t flat_groups = (uint64_t(plan.waves_per_image) * images + plan.waves_per_block - 1) / plan.waves_per_block;
    const bool flat = plan.waves_per_image != 0u && flat_groups <= 0x7fffffffu;
    if (flat) {
        // at most as many workgroups as are resident at once; their waves loop over the rest (see the kernel)
        const DeviceLimits lim = device_limits();
        const uint32_t per_cu = std::max(1u, std::min(lim.lds_bytes / plan.total_bytes, plan.cu_waves / plan.waves_per_block));
        grid = dim3(uint32_t(std::min<uint64_t>(flat_groups, uint64_t(lim.cus) * per_cu)), 1, 1);
    }
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    // the units' queue (launch_fused_422): where the resident waves have more than one unit each
    uint32_t *q = flat && queue && flat_groups > grid.x && !lab_env("COMPEG_NO_QUEUE") ? queue : nullptr;
    if (q) {
        const hipError_t z = hipMemsetAsync(q, 0, sizeof(uint32_t), stream);
        if (z != hipSuccess)
            return z;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(threads), plan.total_bytes, stream, descs,
                       plan.l2_entries_in_lds, plan.rows, plan.stage_after, plan.stage_below, flat ? plan.waves_per_image : 0u, images, q);
    return hipGetLastError();
}

// The walk kernel's shape: a CU's share of the launch's waves in one workgroup (at most twelve), and all the LDS the
// tables leave for their rows -- the walk waits for every staging (no IDCT to land it under), so the fewer the better.
WalkPlan plan_walk(uint32_t max_intervals, uint32_t images, uint32_t max_l2, uint32_t mcu_words, uint32_t restart_interval, bool uniform, bool walk_tables)
{
    WalkPlan p;
    const DeviceLimits lim = device_limits();
    p.l2_entries_in_lds = (std::min(max_l2, 12288u) + 2u * kDcFastEntries + 1u) & ~1u;
    p.walk_tables = walk_tables;
    const uint32_t tables = (((((kL1Entries + p.l2_entries_in_lds) * 2u) + 15u) & ~15u) + 31u & ~31u) + (walk_tables ? kWalkWords * 4u : 0u) + 96u;
    const uint32_t waves_per_image = (max_intervals + kWave - 1) / kWave;
    const uint64_t total_waves = uint64_t(waves_per_image) * images;
    uint32_t wpb = uint32_t(std::min<uint64_t>(std::max<uint64_t>((total_waves + lim.cus - 1) / lim.cus, 1u), kMaxWavesFused));
    if (!uniform)
        wpb = std::min(wpb, std::max(1u, waves_per_image));
    if (const char *e = lab_env("COMPEG_WALK_WPB")) // experiment knob
        wpb = uint32_t(std::max(1, std::min(12, atoi(e))));
    // A wave's share of the LDS (walk_mcus_422_kernel): its lanes' lists -- 16 rows' worth per MCU of the chunk the lanes walk
    // between two looks at what they found --, the slow road's window, its rows and the spare ones.  The longer the chunk,
    // the less the lanes wait for each other (they meet at a chunk's end) and the rarer everything that happens per
    // chunk; the rows have to hold a chunk two and a half times over (staged anew when some lane has less than one and
    // a half in front of it).
    const uint32_t share = ((lim.lds_bytes - tables) / wpb - kWalkSlowWords * 4u) / kWalkRowBytes - kWalkSpareRows; // in rows
    const uint32_t per_mcu = 16u + (5u * std::max(mcu_words, 1u) + 1u) / 2u;
    uint32_t chunk = std::max(1u, std::min(std::min(kWalkMaxChunk, restart_interval), (share > 4u ? share - 4u : 0u) / per_mcu));
    if (const char *e = lab_env("COMPEG_WALK_CHUNK")) // experiment knob
        chunk = uint32_t(std::max(1, std::min(int(kWalkMaxChunk), atoi(e))));
    const uint32_t list_rows = (uint32_t(kWave) * walk_list_bytes(chunk) + kWalkRowBytes - 1u) / kWalkRowBytes;
    uint32_t rows = share > list_rows ? share - list_rows : 0u;
    rows = std::max(8u, std::min(rows, 192u));
    if (const char *e = lab_env("COMPEG_WALK_ROWS")) // experiment knob
        rows = uint32_t(std::max(4, std::min(256, atoi(e))));
    p.rows = rows;
    p.chunk = chunk;
    p.stage_below = std::min(rows / 2u, (3u * chunk * mcu_words + 1u) / 2u + 8u);
    if (const char *e = lab_env("COMPEG_WALK_BELOW")) // experiment knob
        p.stage_below = uint32_t(std::max(0, atoi(e)));
    p.waves_per_block = wpb;
    p.total_bytes = tables + wpb * (uint32_t(kWave) * walk_list_bytes(chunk) + kWalkSlowWords * 4u + (rows + kWalkSpareRows) * kWalkRowBytes);
    p.waves_per_image = uniform ? waves_per_image : 0u;
    if (getenv("COMPEG_VERBOSE"))
        fprintf(stderr, "[compeg] walk plan: images=%u intervals=%u waves/block=%u rows=%u below %u words chunk=%u MCUs lds=%u B%s%s\n", images, max_intervals, wpb,
                rows, p.stage_below, chunk, p.total_bytes, uniform ? " flat" : "", walk_tables ? " walk tables" : "");
    return p;
}

hipError_t launch_walk_mcus(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const WalkPlan &plan, hipStream_t stream, uint32_t *queue)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    const uint32_t threads = plan.waves_per_block * kWave;
    dim3 grid((max_intervals + threads - 1) / threads, images, 1);
    const uint64_t flat_groups = (uint64_t(plan.waves_per_image) * images + plan.waves_per_block - 1) / plan.waves_per_block;
    const bool flat = plan.waves_per_image != 0u && flat_groups <= 0x7fffffffu;
    if (flat) {
        const DeviceLimits lim = device_limits();
        const uint32_t per_cu = std::max(1u, std::min(lim.lds_bytes / plan.total_bytes, kMaxWavesFused / plan.waves_per_block));
        grid = dim3(uint32_t(std::min<uint64_t>(flat_groups, uint64_t(lim.cus) * per_cu)), 1, 1);
    }
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(walk_mcus_422_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    uint32_t *q = flat && queue && flat_groups > grid.x ? queue : nullptr;
    if (q) {
        const hipError_t z = hipMemsetAsync(q, 0, sizeof(uint32_t), stream);
        if (z != hipSuccess)
            return z;
    }
    hipLaunchKernelGGL(walk_mcus_422_kernel, grid, dim3(threads), plan.total_bytes, stream, descs, plan.l2_entries_in_lds, plan.rows, plan.stage_below,
                       flat ? plan.waves_per_image : 0u, images, q, plan.walk_tables ? 1u : 0u, plan.chunk);
    return hipGetLastError();
}

hipError_t launch_pair_422(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                           const HuffLdsPlan &plan, hipStream_t stream)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    dim3 grid((max_intervals + kWave - 1) / kWave, images, 1);
    const uint32_t tables = (((kL1Entries + plan.l2_entries_in_lds) * 2u) + 15u) & ~15u;
    const uint32_t lds = tables + ((plan.window_words * 4u + 15u) & ~15u) + 2u * kWave * kDuSlotBytes +
                         2u * kWave * 4u;
    hipLaunchKernelGGL(decode_pair_422_kernel, grid, dim3(2 * kWave), lds, stream, descs,
                       plan.l2_entries_in_lds, plan.window_words);
    return hipGetLastError();
}

// The cooperative kernel's work split over the waves of a workgroup ("team" form): one wave of four walks
// 4 x 64 data units' worth of intervals (16 with DRI = 4, a lane each -- the other three waves wait at a
// barrier), then each of the four decodes 64 of those data units.  The walk is then not
// replicated in four waves with four busy lanes each: a quarter of the walking instructions, walker waves that
// have their SIMD to themselves, and no speculation (hence no tail).
// LDS: [L1][L2 + direct tables][quantisers][walk tables][per team of 4 waves: window | walk bookkeeping | flags | 4 x (64 slots | 64 DC differences)]
constexpr uint32_t kCoopTeamWaves = 4;

// The walk tables of every image of a launch, from its direct tables (global memory to global memory: once per
// set of Huffman tables, not per frame -- the runtime keeps them).
__global__ void __launch_bounds__(1024) walk_tables_kernel(const ImageDesc *__restrict__ descs)
{
    const ImageDesc &d = descs[blockIdx.y];
    if (!d.walk || !(d.coop_ok || d.mcu_ok))
        return;
    HuffShared h{};
    h.l2 = d.l2;
    CoopTables t;
    coop_tables(d, h, t);
    uint32_t *out = const_cast<uint32_t *>(d.walk);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < kWalkWords; i += gridDim.x * blockDim.x)
        out[i] = t.walk_ok ? coop_walk_word(t.ac_fast, t.dc_fast, t.walk_ids, i) : 0u;
}

hipError_t launch_walk_tables(const ImageDesc *descs, uint32_t images, hipStream_t stream)
{
    hipLaunchKernelGGL(walk_tables_kernel, dim3(kWalkWords / 1024u, images, 1), dim3(1024), 0, stream, descs);
    return hipGetLastError();
}

constexpr uint32_t kCoopTeamFlagWords = 4; // (coop_body.h: kTeamWalk ...)
// A team's share of LDS: [window][walk bookkeeping: coop_misc_words(rounds)][flags][4 x (64 slots | 64 DC differences)]
// [the walker's lists, where they are longer than the bytes of its slots: intervals of more than 64 MCUs]
__host__ __device__ __forceinline__ uint32_t coop_team_area(uint32_t window_words, const CoopShape &sh)
{
    return ((window_words * 4u + 15u) & ~15u) + (coop_misc_words(sh.rounds) + kCoopTeamFlagWords) * 4u +
           kCoopTeamWaves * (kWave * kDuSlotBytes + kWave * 4u) +
           (sh.list_cap > kCoopSlotListCap ? uint32_t(kWave) * sh.list_cap * 4u : 0u);
}

// One wave tells the others of its team (through LDS; a workgroup barrier would tie the teams together).
__device__ __forceinline__ void team_signal(uint32_t *flag, uint32_t lane, uint32_t value)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0u)
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void team_wait(uint32_t *flag, uint32_t value)
{
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < value)
        __builtin_amdgcn_s_sleep(4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ void __launch_bounds__(1024)
decode_coop_team_422_kernel(const ImageDesc *__restrict__ descs, uint32_t l2_in_lds, uint32_t window_words, uint32_t spec_shift,
                            uint32_t teams, uint32_t quarters_on, uint32_t group_waves)
{
    // (the workgroup is always 16 waves: those beyond the teams' help with the staging and leave)
    extern __shared__ __attribute__((aligned(32))) uint8_t smem[];
    const ImageDesc &d = descs[blockIdx.y];
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x / kWave))), lane = threadIdx.x % kWave;
    const uint32_t team = wave / kCoopTeamWaves, member = wave % kCoopTeamWaves;
    CoopClock clk;
    coop_clock_start(clk);
    CoopGeom g;
    // (with walk tables nobody speculates, a lane walks its whole interval -- up to 16 data units: beyond that a lane's
    // chain is longer than the speculative walks of its interval's lanes, measured with DRI = 16: 164 against 108 us)
    coop_geom(d, blockIdx.x * teams + team, g, d.walk && d.restart_interval <= kCoopLeanMaxRestart ? 31u : spec_shift, group_waves);
    if (blockIdx.x * teams * g.ipw >= d.total_intervals)
        return; // the whole workgroup
    const CoopShape shape = coop_shape(d.restart_interval, group_waves);

    uint16_t *l1 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *l2 = l1 + kL1Entries;
    float *quant = reinterpret_cast<float *>(smem + align16((kL1Entries + l2_in_lds) * 2u));
    // (32-byte aligned: a walk table's name carries a shift in its low five bits.  As an offset from smem, not by
    // rounding the pointer: behind an integer round trip the compiler no longer knows that this -- and everything
    // laid out behind it -- is LDS, and every access becomes a flat one)
    const uint32_t quant_off = align16((kL1Entries + l2_in_lds) * 2u);
    uint32_t *walk = reinterpret_cast<uint32_t *>(smem + ((quant_off + 3u * kCoopQuantStride * 4u + 31u) & ~31u));
    uint8_t *team_base = reinterpret_cast<uint8_t *>(walk + kWalkWords) + team * coop_team_area(window_words, shape);
    uint32_t *win = reinterpret_cast<uint32_t *>(team_base);
    uint32_t *misc = reinterpret_cast<uint32_t *>(team_base + align16(window_words * 4u));
    uint32_t *flags = misc + coop_misc_words(shape.rounds);
    uint8_t *mine = reinterpret_cast<uint8_t *>(flags + kCoopTeamFlagWords) + member * (kWave * kDuSlotBytes + kWave * 4u);
    const bool helper = team >= teams;
    if (!helper && member == 0u && lane < kCoopTeamFlagWords)
        flags[lane] = 0u;

    // Staging.  Every load of a thread is issued before its first LDS write (copy loops that wait for each load in
    // turn cost a trip to memory per iteration: 10 k cycles for the 52 KB of tables with four waves), in this
    // order: where the team's window begins; the tables (which need nothing but the descriptor); then -- waiting
    // for the first two loads only -- the window and the walker's interval starts.
    uint32_t win_base = 0, win_len = 0, first_word = 0, end_word = 0;
    const bool has_window = !helper && g.intervals != 0u;
    if (has_window)
        coop_window_fetch(d, g, first_word, end_word);
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    const uint32_t n2 = umin(l2_in_lds, d.fast_off + 2u * kFastEntries + 2u * kDcFastEntries), n2w = (n2 + 1u) / 2u;
    auto *g_l1 = CG_GLOBAL(const Dwords4, reinterpret_cast<const Dwords4 *>(d.l1));
    auto *g_l2 = CG_GLOBAL(const Dwords4, reinterpret_cast<const Dwords4 *>(d.l2));
    auto *g_wk = CG_GLOBAL(const Dwords4, reinterpret_cast<const Dwords4 *>(d.walk));
    SlotVec *s_l1 = reinterpret_cast<SlotVec *>(l1), *s_l2 = reinterpret_cast<SlotVec *>(l2), *s_wk = reinterpret_cast<SlotVec *>(walk);
    const uint32_t v_l1 = 4u * 128u / 4u, v_l2 = n2w / 4u, v_wk = d.walk ? kWalkWords / 4u : 0u;
    constexpr uint32_t kRounds = 2; // (1024 threads: the tables in one go)
    Dwords4 a1{}, a2[kRounds], aw[kRounds];
    const bool l1_mine = tid < v_l1;
    if (l1_mine)
        a1 = g_l1[tid];
#pragma unroll
    for (uint32_t k = 0; k < kRounds; k++) {
        const uint32_t i = k * nth + tid;
        if (i < v_l2)
            a2[k] = g_l2[i];
        if (i < v_wk)
            aw[k] = g_wk[i];
    }
    if (has_window)
        coop_window_from(d, first_word, end_word, window_words, win_base, win_len);
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    clk.extra[0] = __builtin_readcyclecounter() - clk.tprev;
#endif
    const uint32_t nvec = (win_len + 3u) / 4u, wlane = member * uint32_t(kWave) + lane; // (the team's 256 lanes)
    SlotVec w[kRounds];
#pragma unroll
    for (uint32_t k = 0; k < kRounds; k++)
        if (k * 256u + wlane < nvec)
            w[k] = scan_words4(d, win_base + 4u * (k * 256u + wlane));
    // (the walker's lanes look up where their intervals begin while all this is on its way)
    CoopLane my_walk{};
    // (the walkers of a workgroup's teams are different waves of theirs: waves go to the SIMDs round robin, so they do
    // not share one; with quarters -- coop_decode_quarter_422 -- wave m of team t takes quarter (m - t) & 3, the walker the last)
    const bool walker = member == ((team + 3u) & (kCoopTeamWaves - 1u));
    if (!helper && walker && g.intervals) {
        HuffShared hw{};
        hw.win_base = win_base;
        hw.win_len = win_len;
        coop_lane(d, hw, g, lane, my_walk);
    }
    if (l1_mine)
        s_l1[tid] = SlotVec{a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (uint32_t k = 0; k < kRounds; k++) {
        const uint32_t i = k * nth + tid;
        if (i < v_l2)
            s_l2[i] = SlotVec{a2[k].x, a2[k].y, a2[k].z, a2[k].w};
        if (i < v_wk)
            s_wk[i] = SlotVec{aw[k].x, aw[k].y, aw[k].z, aw[k].w};
        if (k * 256u + wlane < nvec)
            reinterpret_cast<SlotVec *>(win)[k * 256u + wlane] = w[k];
    }
    // what is left over (with 1024 threads only of windows of more than 2048 words; table 4 of L1 is all-zero; the
    // last words of the L2 copy do not fill a vector)
    for (uint32_t i = kRounds * nth + tid; i < v_l2; i += nth)
        s_l2[i] = SlotVec{g_l2[i].x, g_l2[i].y, g_l2[i].z, g_l2[i].w};
    for (uint32_t i = kRounds * nth + tid; i < v_wk; i += nth)
        s_wk[i] = SlotVec{g_wk[i].x, g_wk[i].y, g_wk[i].z, g_wk[i].w};
    for (uint32_t i = kRounds * 256u + wlane; i < nvec; i += 256u)
        reinterpret_cast<SlotVec *>(win)[i] = scan_words4(d, win_base + 4u * i);
    for (uint32_t i = tid; i < 128u; i += nth)
        reinterpret_cast<slot_word_t *>(l1)[4u * 128u + i] = 0u;
    for (uint32_t k = v_l2 * 4u + tid; k < n2w; k += nth)
        reinterpret_cast<slot_word_t *>(l2)[k] = CG_GLOBAL(const uint32_t, reinterpret_cast<const uint32_t *>(d.l2))[k];
    if (threadIdx.x < 3u * kRetained)
        quant[(threadIdx.x / kRetained) * kCoopQuantStride + threadIdx.x % kRetained] =
            d.quant[threadIdx.x / kRetained][threadIdx.x % kRetained];
    // (what the phases need of the descriptor: asked for in front of the barrier, there behind it)
    CoopShared cs;
    cs.h.l1 = l1;
    cs.h.l2 = l2;
    cs.h.l2_staged = umin(l2_in_lds, d.fast_off + 2u * kFastEntries + 2u * kDcFastEntries);
    cs.h.win = win;
    cs.h.win_base = win_base;
    cs.h.win_len = win_len;
    cs.h.du_slots = mine;
    uint8_t *first_mine = reinterpret_cast<uint8_t *>(flags + kCoopTeamFlagWords);
    constexpr uint32_t kMineBytes = kWave * kDuSlotBytes + kWave * 4u;
    // (the walker's lists: its own slot area, as in the one-wave form; the other waves read them there)
    cs.lists = reinterpret_cast<uint32_t *>(first_mine + ((team + 3u) & (kCoopTeamWaves - 1u)) * kMineBytes);
    if (shape.list_cap > kCoopSlotListCap)
        cs.lists = reinterpret_cast<uint32_t *>(first_mine + kCoopTeamWaves * kMineBytes); // (longer than that: their own area)
    cs.team_diffs = reinterpret_cast<int32_t *>(first_mine + kWave * kDuSlotBytes);
    cs.team_diffs_stride = kMineBytes / 4u;
    cs.team_in_wg = team;
    coop_bind_misc(cs, misc);
    cs.flags = flags;
    cs.diffs = reinterpret_cast<int32_t *>(mine + kWave * kDuSlotBytes);
    cs.quant = quant;
    CoopTables t;
    coop_tables(d, cs.h, t);
    t.walk = d.walk ? walk : nullptr;
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    clk.extra[1] = __builtin_readcyclecounter() - clk.tprev;
#endif
    __syncthreads();
    if (helper)
        return;
#if defined(CG_COOP_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    clk.extra[2] = __builtin_readcyclecounter() - clk.tprev;
#endif

    const uint32_t team_index = blockIdx.x * teams + team;
    // intervals of 16 data units, walked through the walk tables: the decoding starts under the walk, quarter by quarter
    // (worth it where the teams of a workgroup compete for the CU: a team alone on its CU ends with its walker's own
    // round either way -- 34.4 against 34.9 us for one 1080p frame)
    const bool quarters = quarters_on != 0u && teams > 1u && t.walk != nullptr && t.walk_ok && g.dpi == 16u && g.ipw == 16u && g.count == 1u;
    if (walker) {
        // (the walk is the team's critical path, and with quarters its SIMD is busy with other teams' decoding waves:
        // its instructions go first)
        __builtin_amdgcn_s_setprio(3);
        if (g.intervals)
            coop_walk_422<1>(d, cs, t, g, lane, team_index, clk, &my_walk, quarters);
        if (!quarters)
            __builtin_amdgcn_s_setprio(0); // (with quarters its own quarter is the team's last: it keeps going first)
        team_signal(flags, lane, 2u);
    } else if (!quarters) {
        team_wait(flags, 2u);
    }
    CG_COOP_STAMP(7); // (the waves that do not walk: their wait)
    if (quarters) {
        // (the later a quarter, the nearer it is to the team's end: it goes first among the decoding waves of its SIMD)
        const uint32_t quarter = (member - team) & (kCoopTeamWaves - 1u);
        if (quarter == 1u)
            __builtin_amdgcn_s_setprio(1);
        else if (quarter == 2u)
            __builtin_amdgcn_s_setprio(2);
        coop_decode_quarter_422<1>(d, cs, t, g, lane, team_index, quarter, clk);
        if (walker && g.intervals)
            coop_team_serial_422<1>(d, cs, t, g, lane, kCoopTeamWaves);
    } else if (g.intervals) {
        // the walk's rounds of 64 data units, dealt out to the team's waves (more than one each where a single
        // interval is longer than 256 data units)
        for (uint32_t r = member; r < g.rounds; r += kCoopTeamWaves)
            coop_decode_round_422<1>(d, cs, t, g, lane, team_index, r, clk);
        if (walker)
            coop_team_serial_422<1>(d, cs, t, g, lane, (g.dus + uint32_t(kWave) - 1u) / uint32_t(kWave));
    }
    coop_clock_store(clk, d, team_index * kCoopTeamWaves + member, lane);
}

// Every image of the launch must have ImageDesc::coop_ok and the same restart interval.
CoopPlan plan_coop(uint32_t max_intervals, uint32_t images, uint32_t restart_interval, uint32_t max_l2,
                   const CoopSpans &spans)
{
    CoopPlan p{};
    if (restart_interval == 0 || restart_interval > kCoopMaxRestart)
        return p;
    // A team's intervals have to fit its window.  With windows of up to kCoopCosyWindow words four teams share a CU:
    // 4 x 64 data units' worth of intervals fit one up to 4 bit per pixel, denser streams get teams of half and of a
    // quarter as many intervals (fewer rounds: two or one of the team's waves decode).  What does not fit even then
    // -- long intervals: one per MCU row of a 4K frame is 240 MCUs -- gets the largest window there is, a team
    // alone on its CU.  A group longer than that would go through the serial decoder, one lane for the whole
    // team: the other kernels do better.
    constexpr uint32_t kCoopCosyWindow = 2040;
    uint32_t k = 0;
    while (k < 3u && spans.words[k] + kDuWordSlack + 4u + kCoopEndSlack > kCoopCosyWindow)
        k++;
    if (k == 3u) {
        k = 2u;
        if (spans.words[k] + kDuWordSlack + 4u + kCoopEndSlack > kCoopMaxWindow)
            return p;
    }
    p.group_waves = kCoopTeamWaves >> k;
    const uint32_t max_group_words = spans.words[k];
    const CoopShape sh = coop_shape(restart_interval, p.group_waves);
    const DeviceLimits lim = device_limits();
    const uint32_t lds_per_cu = lim.lds_bytes, cu_count = lim.cus;
    p.intervals_per_wave = sh.ipw;
    p.l2_entries_in_lds = (max_l2 + 2u * kDcFastEntries + 1u) & ~1u;
    uint32_t w = max_group_words + kDuWordSlack + 4u + kCoopEndSlack;
    w = std::max(w, 128u);
    w = std::min(w, kCoopMaxWindow);
    p.window_words = (w + 3u) & ~3u;
    uint32_t tables = ((((kL1Entries + p.l2_entries_in_lds) * 2u) + 15u) & ~15u) + 3u * kCoopQuantStride * 4u;
    // four teams of four waves: one workgroup per CU, one copy of the tables
    tables = ((tables + 31u) & ~31u) + kWalkWords * 4u;
    const uint64_t all_teams = uint64_t((max_intervals + p.intervals_per_wave - 1) / p.intervals_per_wave) * images;
    // A launch that needs four teams per CU to be resident at once gets them even if the window asked for is
    // a little too large (it is an estimate where the scan was preprocessed on the device): down to three
    // quarters of it.  A team whose intervals do not fit its window still decodes -- the walks that leave it
    // hand their interval to the serial decoder.
    if (all_teams > 3ull * cu_count && tables + 4u * coop_team_area(p.window_words, sh) > lds_per_cu) {
        const uint32_t room = (lds_per_cu - tables) / 4u, fixed = coop_team_area(0, sh);
        const uint32_t fit = room > fixed ? ((room - fixed) / 4u) & ~3u : 0u;
        if (fit >= p.window_words - p.window_words / 4u)
            p.window_words = fit;
    }
    const uint32_t team_area = coop_team_area(p.window_words, sh);
    uint32_t teams = 4;
    while (teams > 1 && tables + teams * team_area > lds_per_cu)
        teams--;
    if (tables + teams * team_area > lds_per_cu)
        return p;
    // one workgroup per CU while that covers the launch (a small launch spreads over the CUs; the teams of a
    // workgroup share one copy of the tables and the sixteen waves that stage it)
    const uint32_t fit = teams;
    p.fit_teams = fit;
    p.places = cu_count * fit;
    teams = all_teams <= cu_count ? 1u : (all_teams <= 2ull * cu_count ? 2u : 4u);
    teams = teams < fit ? teams : fit;
    p.waves_per_block = teams * kCoopTeamWaves;
    p.total_bytes = tables + teams * team_area;
    p.total_waves = uint32_t(std::min<uint64_t>(all_teams * kCoopTeamWaves, 0xffffffffu));
    p.usable = true;
    if (getenv("COMPEG_VERBOSE"))
        fprintf(stderr, "[compeg] coop team plan: images=%u intervals=%u restart interval=%u per team=%u rounds=%u teams/block=%u window=%u words lds=%u B\n",
                images, max_intervals, restart_interval, p.intervals_per_wave, sh.rounds, teams, p.window_words, p.total_bytes);
    return p;
}

hipError_t launch_coop_422(const ImageDesc *descs, uint32_t images, uint32_t max_intervals, const CoopPlan &plan,
                           hipStream_t stream)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    static const uint32_t spec_shift = [] {
        const char *e = lab_env("COMPEG_COOP_SPEC_SHIFT"); // experiment knob: fewer speculative subsequences
        return e ? uint32_t(atoi(e)) : 0u;
    }();
    const uint32_t teams = plan.waves_per_block / kCoopTeamWaves, per_block = plan.intervals_per_wave * teams;
    dim3 grid((max_intervals + per_block - 1) / per_block, images, 1);
    const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void *>(decode_coop_team_422_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    static const uint32_t quarters_on = [] {
        const char *e = lab_env("COMPEG_COOP_QUARTERS"); // experiment knob: 0 = the decoding waits for the walk's end
        return e ? uint32_t(atoi(e) != 0) : 1u;
    }();
    hipLaunchKernelGGL(decode_coop_team_422_kernel, grid, dim3(1024), plan.total_bytes, stream, descs,
                       plan.l2_entries_in_lds, plan.window_words, spec_shift, teams, quarters_on, plan.group_waves);
    return hipGetLastError();
}

hipError_t launch_idct_composite(const ImageDesc *descs, uint32_t images, uint32_t max_dus,
                                 hipStream_t stream)
{
    if (images == 0 || max_dus == 0)
        return hipSuccess;
    dim3 grid((max_dus + 255) / 256, images, 1);
    hipLaunchKernelGGL(idct_composite_kernel, grid, dim3(256), 0, stream, descs);
    return hipGetLastError();
}

hipError_t launch_entropy_samples(const ImageDesc *descs, uint32_t images, uint32_t max_intervals,
                                  const HuffLdsPlan &plan, hipStream_t stream)
{
    if (images == 0 || max_intervals == 0)
        return hipSuccess;
    const uint32_t threads = plan.waves_per_block * kWave; // planned like the fused kernel: at most 12 waves
    dim3 grid((max_intervals + threads - 1) / threads, images, 1);
    const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void *>(entropy_samples_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
        int(device_limits().lds_bytes));
    if (attr != hipSuccess)
        return attr;
    hipLaunchKernelGGL(entropy_samples_kernel, grid, dim3(threads), plan.total_bytes, stream, descs,
                       plan.l2_entries_in_lds, plan.window_words);
    return hipGetLastError();
}

hipError_t launch_generic_composite(const ImageDesc *descs, uint32_t images, uint32_t max_w, uint32_t max_h,
                                    hipStream_t stream)
{
    if (images == 0 || max_w == 0 || max_h == 0)
        return hipSuccess;
    // y is limited to 65535 rows per launch dimension: more than any baseline JPEG has
    // grid sized for the smallest MCU (8 x 8): workgroups past an image's own strips / MCU rows exit at once
    hipLaunchKernelGGL(composite_generic_kernel, dim3(((max_w + 7) / 8 + kStripMcus - 1) / kStripMcus, (max_h + 7) / 8, images),
                       dim3(256), 0, stream, descs);
    return hipGetLastError();
}

} // namespace compeg
