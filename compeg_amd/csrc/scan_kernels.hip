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
//   count_kernel   per 4 KiB tile: kept bytes and markers              (grid: tiles x images)
//   tile_scan      exclusive prefix over an image's tiles              (one block per image)
//   marker_kernel  P[m] = number of kept bytes in front of marker m
//   interval_scan  start word of interval m = sum of ceil(len_j / 4)   (one block per image)
//   emit_kernel    every kept byte to 4*start[m] + (p - P[m]); padding zeroed
//
// Integer / byte work, HBM-bound (a few bytes of traffic per input byte).
#include <hip/hip_runtime.h>

#include "scan_kernels.h"

namespace compeg {
namespace {

constexpr uint32_t kTileBytes = 4096;
constexpr uint32_t kThreads = 256;
constexpr uint32_t kBytesPerThread = kTileBytes / kThreads; // 16
constexpr uint32_t kMaxLookBack = 1u << 16;

// What a thread knows about its 16 bytes after classification.
struct Chunk {
    uint32_t w[4];     // the 16 bytes, byte i in bits 8*(i%4) of w[i/4]
    uint32_t n;        // how many of them lie inside the segment
    uint32_t kept;     // bit i: byte i contributes one output byte (itself; FF for an FF 00 pair)
    uint32_t marker;   // bit i: byte i is the first byte of an FF xx pair that ends an interval
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
    c.kept = c.marker = 0u;
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
    uint32_t F = 0, Z = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t x = c.w[k], nx = ~x;
        // exact per-byte "== 0" tests, result in bit 7 of every byte
        const uint32_t zf = ~(((nx & 0x7f7f7f7fu) + 0x7f7f7f7fu) | nx) & 0x80808080u; // bytes of x equal to FF
        const uint32_t zz = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bytes of x equal to 00
        // gather bits 7, 15, 23, 31 into a nibble
        F |= (((zf >> 7) * 0x00204081u) >> 21 & 0xfu) << (4 * k);
        Z |= (((zz >> 7) * 0x00204081u) >> 21 & 0xfu) << (4 * k);
    }
    const uint32_t valid = c.n >= 16u ? 0xffffu : ((1u << c.n) - 1u);
    F &= valid;
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

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &total)
{
    // 256 threads: wave-level shuffles, then 4 wave totals through LDS
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = v;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if (lane >= o)
            x += y;
    }
    if (lane == 63)
        lds[wave] = x;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++)
        base += lds[w];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + x - v;
}

__global__ void __launch_bounds__(kThreads) count_kernel(const ScanDesc *descs)
{
    __shared__ uint32_t lds[4];
    const ScanDesc &d = descs[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    if (tile >= d.ntiles)
        return;
    const uint32_t g = tile * kTileBytes + threadIdx.x * kBytesPerThread;
    Chunk c;
    load_classify(d, g, c);
    const uint32_t kept = __popc(c.kept), markers = __popc(c.marker);
    uint32_t tk, tm;
    block_exclusive_scan(kept, lds, tk);
    block_exclusive_scan(markers, lds, tm);
    if (threadIdx.x == 0) {
        d.tile_kept[tile] = tk;
        d.tile_markers[tile] = tm;
    }
}

// One block per image: exclusive prefix over the image's tiles (in place) and
// the totals.  result[0] = number of intervals counted (markers + 1),
// result[1] = kept bytes.
__global__ void __launch_bounds__(kThreads) tile_scan_kernel(const ScanDesc *descs)
{
    __shared__ uint32_t lds[4];
    const ScanDesc &d = descs[blockIdx.x];
    uint32_t carry_k = 0, carry_m = 0;
    for (uint32_t base = 0; base < d.ntiles; base += kThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t k = i < d.ntiles ? d.tile_kept[i] : 0u, m = i < d.ntiles ? d.tile_markers[i] : 0u;
        uint32_t tk, tm;
        const uint32_t ek = block_exclusive_scan(k, lds, tk), em = block_exclusive_scan(m, lds, tm);
        if (i < d.ntiles) {
            d.tile_kept[i] = carry_k + ek;
            d.tile_markers[i] = carry_m + em;
        }
        carry_k += tk;
        carry_m += tm;
    }
    if (threadIdx.x == 0) {
        d.result[0] = carry_m + 1u;
        d.result[1] = carry_k;
        d.marker_pos[0] = 0u;
    }
}

// P[m] for every marker: kept bytes in front of it (m counts from 1).
__global__ void __launch_bounds__(kThreads) marker_kernel(const ScanDesc *descs)
{
    __shared__ uint32_t lds[4];
    const ScanDesc &d = descs[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    if (tile >= d.ntiles)
        return;
    const uint32_t g = tile * kTileBytes + threadIdx.x * kBytesPerThread;
    Chunk c;
    load_classify(d, g, c);
    const uint32_t kept = __popc(c.kept), markers = __popc(c.marker);
    uint32_t tk, tm;
    uint32_t p = d.tile_kept[tile] + block_exclusive_scan(kept, lds, tk);
    uint32_t m = d.tile_markers[tile] + block_exclusive_scan(markers, lds, tm) + 1u;
    for (uint32_t mk = c.marker; mk; mk &= mk - 1u) { // markers are rare: visit only those
        const uint32_t i = uint32_t(__ffs(int(mk))) - 1u;
        if (m < d.marker_capacity)
            d.marker_pos[m] = p + __popc(c.kept & ((1u << i) - 1u));
        m++;
    }
}

// One block per image: start word of every interval.  len_m = P[m+1] - P[m]
// (P[count] = kept bytes); start_m = sum_{j<m} ceil(len_j / 4).  The reference
// stores start m at index (m & mask) of a power-of-two array and keeps
// min(count, slots) entries (src/scan.rs:46-56,111).
//
// A single 4K frame has 16 200 intervals and this block is all that runs, so
// its latency is the point: 1024 threads, chunks of 8192 intervals; P[] comes
// in with coalesced loads through LDS, every thread scans 8 consecutive
// entries (index i lives at i + i / 8: conflict-free for that access), and the
// starts leave through the same LDS array with coalesced stores.
constexpr uint32_t kIvThreads = 1024, kIvPerThread = 8, kIvChunk = kIvThreads * kIvPerThread;

__device__ __forceinline__ uint32_t iv_slot(uint32_t i) { return i + i / kIvPerThread; }

__global__ void __launch_bounds__(kIvThreads) interval_scan_kernel(const ScanDesc *descs)
{
    __shared__ uint32_t pos[kIvChunk + kIvChunk / kIvPerThread + 2];
    __shared__ uint32_t wave_total[kIvThreads / 64];
    const ScanDesc &d = descs[blockIdx.x];
    const uint32_t count = min(d.result[0], d.marker_capacity);
    const uint32_t kept = d.result[1];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t carry = 0; // words in front of the chunk
    for (uint32_t base = 0; base < count; base += kIvChunk) {
        const uint32_t n = min(kIvChunk, count - base);
        // P[base .. base + n], the entry behind the last interval being the kept total
        for (uint32_t i = t; i <= n; i += kIvThreads)
            pos[iv_slot(i)] = base + i < count ? d.marker_pos[base + i] : kept;
        __syncthreads();
        uint32_t p[kIvPerThread + 1];
        const uint32_t first = t * kIvPerThread;
        for (uint32_t k = 0; k <= kIvPerThread; k++)
            p[k] = first + k <= n ? pos[iv_slot(first + k)] : 0u;
        uint32_t mine = 0;
        for (uint32_t k = 0; k < kIvPerThread; k++)
            mine += first + k < n ? (p[k + 1] - p[k] + 3u) / 4u : 0u;
        // block-wide exclusive scan of the per-thread totals
        uint32_t x = mine;
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o)
                x += y;
        }
        if (lane == 63)
            wave_total[wave] = x;
        __syncthreads(); // also: every thread has read its pos[] entries
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < kIvThreads / 64; w++) {
            const uint32_t v = wave_total[w];
            before += w < wave ? v : 0u;
            total += v;
        }
        uint32_t start = carry + before + x - mine;
        for (uint32_t k = 0; k < kIvPerThread; k++) {
            if (first + k < n) {
                pos[iv_slot(first + k)] = start;
                start += (p[k + 1] - p[k] + 3u) / 4u;
            }
        }
        __syncthreads();
        for (uint32_t i = t; i < n; i += kIvThreads) {
            const uint32_t m = base + i, s = pos[iv_slot(i)];
            d.interval_start[m] = s;
            // the last writer of a slot wins in the reference's sequential loop
            if (m + d.slots >= count && m != 0)
                d.starts_out[m & (d.slots - 1u)] = s;
        }
        carry += total;
        __syncthreads(); // pos[] and wave_total[] are reused by the next chunk
    }
    if (t == 0) {
        d.result[2] = carry; // total output words
        if (d.patch_nwords)
            *d.patch_nwords = carry;
        if (d.patch_nstarts)
            *d.patch_nstarts = min(count, d.slots);
        // entry 0 keeps its initial 0 unless a wrapped index (m = k * slots) hit it above
        if (count <= d.slots)
            d.starts_out[0] = 0u;
    }
}

// Output staging of one tile: 4096 kept bytes at most, plus up to 3 padding
// bytes behind each of its (at most 2048) markers.
constexpr uint32_t kOutStageBytes = kTileBytes + 3u * (kTileBytes / 2u) + 32u;

__global__ void __launch_bounds__(kThreads) emit_kernel(const ScanDesc *descs)
{
    __shared__ uint32_t lds[4];
    __shared__ uint32_t span[2];
    __shared__ __attribute__((aligned(16))) uint8_t stage[kOutStageBytes];
    const ScanDesc &d = descs[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    if (tile >= d.ntiles)
        return;
    const uint32_t count = min(d.result[0], d.marker_capacity);
    const uint32_t g = tile * kTileBytes + threadIdx.x * kBytesPerThread;
    Chunk c;
    load_classify(d, g, c);
    const uint32_t kept = __popc(c.kept), markers = __popc(c.marker);
    uint32_t tk, tm;
    uint32_t p = d.tile_kept[tile] + block_exclusive_scan(kept, lds, tk);
    uint32_t m = d.tile_markers[tile] + block_exclusive_scan(markers, lds, tm); // current interval

    // zero what this tile can touch: its kept bytes, 3 padding bytes per marker, alignment slack
    const uint32_t zero_dwords = min((tk + 3u * tm + 11u) / 4u, kOutStageBytes / 4u);
    for (uint32_t i = threadIdx.x; i < zero_dwords; i += kThreads)
        reinterpret_cast<uint32_t *>(stage)[i] = 0u; // padding bytes are zeros
    // absolute output offset of a kept byte: 4 * start[m] + (p - P[m])
    uint32_t pm = m < count ? d.marker_pos[m] : 0u, sm = m < count ? d.interval_start[m] : 0u;
    if (threadIdx.x == 0)
        span[0] = sm * 4u + (p - pm); // where this tile's output begins
    __syncthreads();
    const uint32_t lo = span[0];
    const uint32_t origin = lo & ~3u; // staging words line up with the output's words
    uint32_t end = sm * 4u + (p - pm);
    if (m < count && c.kept == 0xffffu) {
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
    } else if (m < count) {
        for (uint32_t i = 0; i < kBytesPerThread; i++) {
            if (c.kept >> i & 1u) {
                stage[end - origin] = uint8_t(byte_of(c.w, i)); // an FF 00 pair emits its FF
                p++;
                end++;
            } else if (c.marker >> i & 1u) {
                m++;
                if (m >= count)
                    break;
                pm = d.marker_pos[m];
                sm = d.interval_start[m];
                end = sm * 4u + (p - pm); // next word boundary: the gap stays zero
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

} // namespace

uint32_t scan_tiles(uint32_t len)
{
    return (len + kTileBytes - 1) / kTileBytes;
}

hipError_t launch_scan(const ScanDesc *descs, uint32_t images, uint32_t max_tiles, hipStream_t stream)
{
    if (images == 0)
        return hipSuccess;
    if (max_tiles) {
        hipLaunchKernelGGL(count_kernel, dim3(max_tiles, images), dim3(kThreads), 0, stream, descs);
    }
    hipLaunchKernelGGL(tile_scan_kernel, dim3(images), dim3(kThreads), 0, stream, descs);
    if (max_tiles) {
        hipLaunchKernelGGL(marker_kernel, dim3(max_tiles, images), dim3(kThreads), 0, stream, descs);
    }
    hipLaunchKernelGGL(interval_scan_kernel, dim3(images), dim3(kIvThreads), 0, stream, descs);
    if (max_tiles) {
        hipLaunchKernelGGL(emit_kernel, dim3(max_tiles, images), dim3(kThreads), 0, stream, descs);
    }
    return hipGetLastError();
}

} // namespace compeg
