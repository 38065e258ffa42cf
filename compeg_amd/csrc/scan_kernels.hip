// Device-side scan preprocessing (the reference author's TODO #1).
//
// Produces, from the raw entropy-coded segment in HBM, exactly what the
// reference's ScanBuffer::process produces on the host (src/scan.rs:33-128):
// FF 00 -> FF, every other FF xx pair dropped and counted as a restart marker,
// every restart interval padded with zeros to a 32-bit word, and the word
// offset of every interval in start_positions.
//
// The reference loop is byte-serial.  Here every byte's role follows from the
// length r of the run of FF bytes in front of it (a byte is the second half of
// an FF xx pair iff r is odd), so all bytes are classified independently:
//
//   count_kernel   per 4 KiB tile: what it contributes to the layout    (grid: tiles x images)
//   tile_scan      the same for everything in front of each tile        (one block per image)
//   emit_kernel    every kept byte to its place; padding zeroed; start positions
//
// The layout (which interval a byte belongs to, where that interval starts in
// the word-aligned output) composes associatively from left to right -- see
// `Stretch` -- so three launches do: no per-marker arrays, no pass over them.
//
// Integer / byte work, HBM-bound (a few bytes of traffic per input byte).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "scan_kernels.h"

namespace compeg {
namespace {

constexpr uint32_t kTileBytes = 4096;
constexpr uint32_t kThreads = 256;
constexpr uint32_t kBytesPerThread = kTileBytes / kThreads; // 16
// A thread whose chunk follows an FF walks back byte by byte to find the run's length.  Entropy-coded data
// holds FF 00, FF RSTn and at most a few fill FFs, so real runs are a few bytes long; the bound keeps a
// hostile segment (megabytes of FF) from costing every one of its chunks tens of thousands of dependent
// loads before the image is handed to the host anyway.
constexpr uint32_t kMaxLookBack = 512;

// What a thread knows about its 16 bytes after classification.
struct Chunk {
    uint32_t w[4];     // the 16 bytes, byte i in bits 8*(i%4) of w[i/4]
    uint32_t n;        // how many of them lie inside the segment
    uint32_t kept;     // bit i: byte i contributes one output byte (itself; FF for an FF 00 pair)
    uint32_t marker;   // bit i: byte i is the first byte of an FF xx pair that ends an interval
    uint32_t foreign;  // bit i: byte i follows an FF and is neither 00, RSTn nor FF: a marker that ends the segment
};

__device__ __forceinline__ uint32_t byte_of(const uint32_t (&w)[4], uint32_t i)
{
    return (w[i >> 2] >> ((i & 3u) * 8u)) & 0xffu;
}

// Length of the FF run that ends right in front of byte `pos`, for runs that
// reach back beyond what the neighbouring lanes hold.  Bounded: a run longer
// than kMaxLookBack sets the image's overflow flag and the host preprocessor
// takes over for that image.
__device__ uint32_t ff_run_before_slow(const ScanDesc &d, uint32_t pos)
{
    uint32_t r = 0;
    while (r < pos && r < kMaxLookBack && d.raw[pos - 1 - r] == 0xff)
        r++;
    if (r >= kMaxLookBack && r < pos)
        atomicOr(&d.result[3], 1u);
    return r;
}

// Loads and classifies the 16 bytes at segment offset g (g is a multiple of
// 16).  The segment starts at an arbitrary byte of the JPEG, so the thread
// loads the 5 aligned dwords that cover its bytes (coalesced 16 + 4 bytes per
// lane; the padded input arena makes the over-read safe) and shifts them
// into place.  A byte is the second half of an FF xx pair iff the run of FF
// bytes in front of it has odd length.
__device__ __forceinline__ void load_classify(const ScanDesc &d, uint32_t g, Chunk &c)
{
    c.n = g < d.len ? min(kBytesPerThread, d.len - g) : 0u;
    c.kept = c.marker = c.foreign = 0u;
    c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0u;
    uint32_t next = 0; // the byte behind the chunk
    if (c.n) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(d.raw) + g;
        const uint32_t mis = uint32_t(addr & 3u);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(addr - mis);
        uint32_t a[5];
        for (int i = 0; i < 5; i++)
            a[i] = src[i];
        for (int i = 0; i < 4; i++)
            c.w[i] = mis ? (a[i] >> (mis * 8u)) | (a[i + 1] << (32u - mis * 8u)) : a[i];
        next = mis ? (a[4] >> (mis * 8u)) & 0xffu : a[4] & 0xffu;
        if (g + 16u >= d.len)
            next = 0; // nothing behind the last byte of the segment
    }
    // FF run in front of the chunk: usually 0 (previous byte is not FF)
    uint32_t r = 0;
    if (c.n && g > 0 && d.raw[g - 1] == 0xff)
        r = ff_run_before_slow(d, g);

    // Bit-parallel classification on 16-bit masks (bit i = byte i).
    // F: byte is FF.  Z: byte is 00.
    uint32_t F = 0, Z = 0, R = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t x = c.w[k], nx = ~x, rx = (x ^ 0xd0d0d0d0u) & 0xf8f8f8f8u;
        // exact per-byte "== 0" tests, result in bit 7 of every byte
        const uint32_t zf = ~(((nx & 0x7f7f7f7fu) + 0x7f7f7f7fu) | nx) & 0x80808080u; // bytes of x equal to FF
        const uint32_t zz = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bytes of x equal to 00
        const uint32_t zr = ~(((rx & 0x7f7f7f7fu) + 0x7f7f7f7fu) | rx) & 0x80808080u; // bytes D0 .. D7 (RSTn)
        // gather bits 7, 15, 23, 31 into a nibble
        F |= (((zf >> 7) * 0x00204081u) >> 21 & 0xfu) << (4 * k);
        Z |= (((zz >> 7) * 0x00204081u) >> 21 & 0xfu) << (4 * k);
        R |= (((zr >> 7) * 0x00204081u) >> 21 & 0xfu) << (4 * k);
    }
    const uint32_t valid = c.n >= 16u ? 0xffffu : ((1u << c.n) - 1u);
    F &= valid;
    // (the byte in front of the chunk: FF iff the run in front of it is not empty -- computed above, r)
    c.foreign = ((F << 1) | (r ? 1u : 0u)) & ~F & ~Z & ~R & valid;
    // Lead FFs are the FFs at even distance from the start of their run (a run
    // that continues from the previous chunk starts "odd" when r is odd).
    const uint32_t starts = F & ~(F << 1);
    uint32_t even_starts = starts & 0x5555u, odd_starts = starts & 0xaaaau;
    if ((r & 1u) && (F & 1u)) {
        even_starts &= ~1u;
        odd_starts |= 1u;
    }
    (void)odd_starts;
    const uint32_t in_even_runs = F & ~(F + even_starts); // FFs of runs whose leads sit on even bits
    const uint32_t lead = (in_even_runs & 0x5555u) | (F & ~in_even_runs & 0xaaaau);
    const uint32_t partner = ((lead << 1) | (r & 1u)) & 0xffffu; // second halves of pairs: dropped
    // what follows each byte: zero / exists at all
    const uint32_t next_zero = ((Z >> 1) | ((next == 0u ? 1u : 0u) << 15)) & 0xffffu;
    const uint32_t remaining = d.len - g; // bytes from g to the end of the segment (>= n)
    const uint32_t has_next = remaining > 16u ? 0xffffu : (remaining >= 2u ? ((1u << (remaining - 1u)) - 1u) : 0u);
    c.kept = ((~F & ~partner) | (lead & next_zero & has_next)) & valid;
    c.marker = lead & ~next_zero & has_next & valid;
}

// ---- the scan's algebra -------------------------------------------------------
//
// What a stretch of the segment contributes to the output layout, in a form
// that composes left to right (associative, not commutative), so that tiles
// and threads can be summarised independently and then scanned:
//   markers  restart markers inside the stretch
//   head     kept bytes in front of its first marker (all of them if it has none)
//   words    output words of the intervals that begin and end inside it
//   tail     kept bytes behind its last marker (all of them if it has none)
//   kept     kept bytes in all
struct Stretch {
    uint32_t markers, head, words, tail, kept;
};

static_assert(sizeof(Stretch) == kScanTileStateBytes, "tile_state stride");

__device__ __forceinline__ Stretch stretch_none() { return Stretch{0u, 0u, 0u, 0u, 0u}; }

__device__ __forceinline__ Stretch join(const Stretch &l, const Stretch &r)
{
    Stretch o;
    o.kept = l.kept + r.kept;
    o.markers = l.markers + r.markers;
    const bool lm = l.markers != 0u, rm = r.markers != 0u;
    o.head = lm ? l.head : l.head + r.head;
    o.tail = rm ? r.tail : l.tail + r.tail;
    // the interval that straddles the seam is complete only with markers on both sides
    o.words = l.words + r.words + ((lm && rm) ? (l.tail + r.head + 3u) / 4u : 0u);
    return o;
}

__device__ __forceinline__ Stretch stretch_shfl_up(const Stretch &x, uint32_t o)
{
    return Stretch{__shfl_up(x.markers, o), __shfl_up(x.head, o), __shfl_up(x.words, o), __shfl_up(x.tail, o),
                   __shfl_up(x.kept, o)};
}

// Where the output stands in front of a stretch, given everything before it:
// the open interval's index, its start word and how many bytes it holds so far.
struct Cursor {
    uint32_t interval, start_word, bytes;
};

__device__ __forceinline__ Cursor cursor_behind(const Stretch &before)
{
    if (before.markers == 0u)
        return Cursor{0u, 0u, before.kept};
    return Cursor{before.markers, (before.head + 3u) / 4u + before.words, before.tail};
}

// A thread's 16 classified bytes as a stretch.
__device__ __forceinline__ Stretch stretch_of(const Chunk &c)
{
    Stretch s;
    s.kept = __popc(c.kept);
    s.markers = __popc(c.marker);
    s.words = 0u;
    if (s.markers == 0u) {
        s.head = s.tail = s.kept;
        return s;
    }
    const uint32_t first = uint32_t(__ffs(int(c.marker))) - 1u, last = 31u - uint32_t(__clz(int(c.marker)));
    s.head = __popc(c.kept & ((1u << first) - 1u));
    s.tail = __popc(c.kept & ~((2u << last) - 1u));
    uint32_t prev = first;
    for (uint32_t mk = c.marker & (c.marker - 1u); mk; mk &= mk - 1u) { // more than one marker: rare
        const uint32_t i = uint32_t(__ffs(int(mk))) - 1u;
        s.words += (__popc(c.kept & ((1u << i) - 1u) & ~((2u << prev) - 1u)) + 3u) / 4u;
        prev = i;
    }
    return s;
}

// Exclusive scan of the block's 256 stretches (thread order); `total` = all of them.
__device__ __forceinline__ Stretch block_exclusive_scan(const Stretch &mine, Stretch *lds, Stretch &total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    Stretch x = mine;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const Stretch y = stretch_shfl_up(x, o);
        if (lane >= o)
            x = join(y, x);
    }
    if (lane == 63)
        lds[wave] = x;
    __syncthreads();
    Stretch before = stretch_none();
    total = stretch_none();
    for (uint32_t w = 0; w < kThreads / 64u; w++) {
        const Stretch t = lds[w];
        if (w < wave)
            before = join(before, t);
        total = join(total, t);
    }
    __syncthreads();
    Stretch prev = stretch_shfl_up(x, 1);
    if (lane == 0)
        prev = stretch_none();
    return join(before, prev);
}

// per tile: its stretch
__global__ void __launch_bounds__(kThreads) count_kernel(const ScanDesc *descs)
{
    __shared__ Stretch lds[kThreads / 64u];
    const ScanDesc &d = descs[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    if (tile >= d.ntiles)
        return;
    const uint32_t g = tile * kTileBytes + threadIdx.x * kBytesPerThread;
    Chunk c;
    load_classify(d, g, c);
    // A byte behind an FF that is neither 00 (stuffing), D0..D7 (RSTn) nor another FF (fill) is a marker that ends the
    // entropy-coded segment in the reference's parser (src/file.rs:163-201).  A segment whose end was taken from the
    // file's final EOI without walking it on the host (borrowed, copy-free uploads: runtime.cpp) must not hold one:
    // flag bit 1 sends the image back to the host front-end.
    if (c.foreign)
        atomicOr(&d.result[3], 2u);
    Stretch total;
    block_exclusive_scan(stretch_of(c), lds, total);
    if (threadIdx.x == 0)
        reinterpret_cast<Stretch *>(d.tile_state)[tile] = total;
}

// One block per image: every tile's stretch is replaced by the stretch of all
// tiles in front of it.  result[0] = number of intervals counted (markers + 1),
// result[1] = kept bytes, result[2] = output words.
__global__ void __launch_bounds__(kThreads) tile_scan_kernel(const ScanDesc *descs)
{
    __shared__ Stretch lds[kThreads / 64u];
    const ScanDesc &d = descs[blockIdx.x];
    Stretch *tiles = reinterpret_cast<Stretch *>(d.tile_state);
    Stretch carry = stretch_none();
    for (uint32_t base = 0; base < d.ntiles; base += kThreads) {
        const uint32_t i = base + threadIdx.x;
        const Stretch mine = i < d.ntiles ? tiles[i] : stretch_none();
        Stretch total;
        const Stretch before = block_exclusive_scan(mine, lds, total);
        if (i < d.ntiles)
            tiles[i] = join(carry, before);
        carry = join(carry, total);
    }
    if (threadIdx.x == 0) {
        const uint32_t count = carry.markers + 1u;
        const Cursor end = cursor_behind(carry);
        const uint32_t words = end.start_word + (end.bytes + 3u) / 4u;
        d.result[0] = count;
        d.result[1] = carry.kept;
        d.result[2] = words;
        if (d.patch_nwords)
            *d.patch_nwords = words;
        if (d.patch_nstarts)
            *d.patch_nstarts = min(count, d.slots);
        // entry 0 of the start positions keeps its initial 0 unless a wrapped index
        // (m = k * slots) overwrites it in emit_kernel
        d.starts_out[0] = 0u;
    }
}

// Output staging of one tile: 4096 kept bytes at most, plus up to 3 padding
// bytes behind each of its (at most 2048) markers.
constexpr uint32_t kOutStageBytes = kTileBytes + 3u * (kTileBytes / 2u) + 32u;

// Every kept byte to 4 * start_word + bytes of its interval's cursor; a marker
// closes the interval (the gap up to the next word stays zero) and notes the
// next one's start.  The reference stores start m at index (m & mask) of a
// power-of-two array and keeps min(count, slots) entries, the last writer of a
// slot winning (src/scan.rs:46-56,111): only the last `slots` markers write.
__global__ void __launch_bounds__(kThreads) emit_kernel(const ScanDesc *descs)
{
    __shared__ Stretch lds[kThreads / 64u];
    __shared__ uint32_t span[2];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kOutStageBytes];
    const ScanDesc &d = descs[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    if (tile >= d.ntiles)
        return;
    const uint32_t count = d.result[0];
    const uint32_t g = tile * kTileBytes + threadIdx.x * kBytesPerThread;
    Chunk c;
    load_classify(d, g, c);
    Stretch total;
    const Stretch in_tile = block_exclusive_scan(stretch_of(c), lds, total);
    Cursor at = cursor_behind(join(reinterpret_cast<const Stretch *>(d.tile_state)[tile], in_tile));

    // zero what this tile can touch: its kept bytes, 3 padding bytes per marker, alignment slack
    const uint32_t zero_dwords = min((total.kept + 3u * total.markers + 11u) / 4u, kOutStageBytes / 4u);
    for (uint32_t i = threadIdx.x; i < zero_dwords; i += kThreads)
        reinterpret_cast<uint32_t *>(stage)[i] = 0u; // padding bytes are zeros
    uint32_t end = at.start_word * 4u + at.bytes; // absolute output offset of the thread's next kept byte
    if (threadIdx.x == 0)
        span[0] = end; // where this tile's output begins
    __syncthreads();
    const uint32_t lo = span[0];
    const uint32_t origin = lo & ~3u; // staging words line up with the output's words
    if (c.kept == 0xffffu) {
        // common case, 16 plain bytes: OR them into the (zeroed) staging words at
        // whatever byte alignment the output position has
        const uint32_t o = end - origin, sh = (o & 3u) * 8u;
        uint32_t *dst = reinterpret_cast<uint32_t *>(stage) + (o >> 2);
        if (sh == 0u) {
            for (int k = 0; k < 4; k++)
                dst[k] = c.w[k];
        } else {
            atomicOr(&dst[0], c.w[0] << sh);
            for (int k = 1; k < 4; k++)
                atomicOr(&dst[k], (c.w[k] << sh) | (c.w[k - 1] >> (32u - sh)));
            atomicOr(&dst[4], c.w[3] >> (32u - sh));
        }
        end += 16u;
    } else {
        for (uint32_t i = 0; i < kBytesPerThread; i++) {
            if (c.kept >> i & 1u) {
                stage[end - origin] = uint8_t(byte_of(c.w, i)); // an FF 00 pair emits its FF
                at.bytes++;
                end++;
            } else if (c.marker >> i & 1u) {
                at.interval++;
                at.start_word += (at.bytes + 3u) / 4u;
                at.bytes = 0u;
                end = at.start_word * 4u; // next word boundary: the gap stays zero
                if (at.interval + d.slots >= count)
                    d.starts_out[at.interval & (d.slots - 1u)] = at.start_word;
            }
        }
    }
    // the last thread that holds segment bytes knows where the tile's output ends
    if (c.n && (g + c.n == d.len || threadIdx.x == kThreads - 1)) {
        // a tile that ends on a marker stops at the next interval's start; the
        // final tile pads its last interval to a whole word
        span[1] = (g + c.n == d.len) ? (end + 3u) & ~3u : end;
    }
    __syncthreads();
    const uint32_t hi = span[1];
    if (hi <= lo)
        return;
    // copy out: single bytes up to the first word boundary and after the last,
    // coalesced dwords in between
    const uint32_t body_lo = (lo + 3u) & ~3u, body_hi = hi & ~3u;
    if (body_lo >= body_hi) {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += kThreads)
            d.words_out[i] = stage[i - origin];
        return;
    }
    if (threadIdx.x < body_lo - lo)
        d.words_out[lo + threadIdx.x] = stage[lo - origin + threadIdx.x];
    if (threadIdx.x < hi - body_hi)
        d.words_out[body_hi + threadIdx.x] = stage[body_hi - origin + threadIdx.x];
    uint32_t *out32 = reinterpret_cast<uint32_t *>(d.words_out);
    const uint32_t *stage32 = reinterpret_cast<const uint32_t *>(stage);
    for (uint32_t i = body_lo / 4u + threadIdx.x; i < body_hi / 4u; i += kThreads)
        out32[i] = stage32[i - origin / 4u];
}

// One block per image: the widest stretch of words that 64 consecutive intervals (one decoder wave) cover,
// exactly what max_wave_span() (desc.cpp) computes from the start positions on the host.
__global__ void __launch_bounds__(kThreads) span_kernel(const ScanDesc *descs)
{
    __shared__ uint32_t best;
    const ScanDesc &d = descs[blockIdx.x];
    const uint32_t nstarts = min(d.result[0], d.slots), nwords = d.result[2], intervals = d.expected;
    if (threadIdx.x == 0)
        best = 0u;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t first = threadIdx.x * 64u; first < intervals; first += kThreads * 64u) {
        const uint32_t lo = first < nstarts ? d.starts_out[first] : 0u;
        const uint32_t after = first + 64u;
        const uint32_t hi = (after < intervals && after < nstarts) ? d.starts_out[after] : nwords;
        if (hi > lo)
            mine = max(mine, hi - lo);
    }
    atomicMax(&best, mine);
    __syncthreads();
    if (threadIdx.x == 0)
        d.result[4] = best;
}

// Host-to-device copy done by the compute queue itself: `src` is pinned host
// memory, read over PCIe with 16-byte loads.  Used for the single-image path,
// where a copy-engine transfer costs more in hand-over between the engines
// than in bytes (no gaps between the pieces and the kernels behind them; more
// loads in flight per lane did not help, fewer workgroups did).
typedef uint32_t PullVec __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(kThreads) pull_kernel(PullVec *__restrict__ dst, const PullVec *__restrict__ src,
                                                        uint32_t n16)
{
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n16; i += gridDim.x * kThreads)
        dst[i] = __builtin_nontemporal_load(src + i);
}

// Up to three transfers in one launch (the tail of a single decode's uploads: the rest of the scan, the start
// positions, descriptor + tables -- each alone is a launch and a PCIe round trip of its own).
struct PullSegs {
    PullVec *dst[3];
    const PullVec *src[3];
    uint32_t n16[3];
};
__global__ void __launch_bounds__(kThreads) pull3_kernel(PullSegs segs)
{
#pragma unroll
    for (int k = 0; k < 3; k++)
        for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < segs.n16[k]; i += gridDim.x * kThreads)
            segs.dst[k][i] = __builtin_nontemporal_load(segs.src[k] + i);
}

} // namespace

hipError_t launch_pull3(void *const dst[3], const void *const pinned_src[3], const size_t bytes[3], hipStream_t stream)
{
    PullSegs segs{};
    size_t most = 0;
    for (int k = 0; k < 3; k++) {
        const size_t n16 = (bytes[k] + 15) / 16;
        if (n16 > 0xffffffffu)
            return hipErrorInvalidValue;
        segs.dst[k] = static_cast<PullVec *>(dst[k]);
        segs.src[k] = static_cast<const PullVec *>(pinned_src[k]);
        segs.n16[k] = uint32_t(n16);
        most = std::max(most, n16);
    }
    if (most == 0)
        return hipSuccess;
    const uint32_t blocks = uint32_t(std::min<size_t>((most + kThreads - 1) / kThreads, 48));
    hipLaunchKernelGGL(pull3_kernel, dim3(blocks), dim3(kThreads), 0, stream, segs);
    return hipGetLastError();
}

hipError_t launch_pull(void *dst, const void *pinned_src, size_t bytes, hipStream_t stream)
{
    const size_t n16 = (bytes + 15) / 16;
    if (n16 == 0)
        return hipSuccess;
    if (n16 > 0xffffffffu)
        return hipErrorInvalidValue;
    // few workgroups: 32-64 of them move 1.6 MB in 42 us (launch and wait included), 2048 in 55 us,
    // the copy engine in 38 us (tools/probes/pcie_pull.hip)
    const uint32_t blocks = uint32_t(std::min<size_t>((n16 + kThreads - 1) / kThreads, 48));
    hipLaunchKernelGGL(pull_kernel, dim3(blocks), dim3(kThreads), 0, stream, static_cast<PullVec *>(dst),
                       static_cast<const PullVec *>(pinned_src), uint32_t(n16));
    return hipGetLastError();
}

uint32_t scan_tiles(uint32_t len)
{
    return (len + kTileBytes - 1) / kTileBytes;
}

hipError_t launch_scan(const ScanDesc *descs, uint32_t images, uint32_t max_tiles, hipStream_t stream, bool with_span)
{
    if (images == 0)
        return hipSuccess;
    if (max_tiles) {
        hipLaunchKernelGGL(count_kernel, dim3(max_tiles, images), dim3(kThreads), 0, stream, descs);
    }
    hipLaunchKernelGGL(tile_scan_kernel, dim3(images), dim3(kThreads), 0, stream, descs);
    if (max_tiles) {
        hipLaunchKernelGGL(emit_kernel, dim3(max_tiles, images), dim3(kThreads), 0, stream, descs);
    }
    if (with_span)
        hipLaunchKernelGGL(span_kernel, dim3(images), dim3(kThreads), 0, stream, descs);
    return hipGetLastError();
}

} // namespace compeg
