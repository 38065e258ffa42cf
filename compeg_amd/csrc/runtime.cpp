// See runtime.h.
#include "runtime.h"
#include "lab.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace compeg {

Status hip_status(hipError_t e, const char *what)
{
    if (e == hipSuccess)
        return Status{};
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return Status::error(COMPEG_E_HIP, buf);
}

#define CG_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return hip_status(e_, #expr);                                                          \
    } while (0)

#define CG_TRY(expr)                                                                               \
    do {                                                                                           \
        Status s_ = (expr);                                                                        \
        if (!s_.ok())                                                                              \
            return s_;                                                                             \
    } while (0)

DeviceBuffer::~DeviceBuffer()
{
    if (ptr)
        (void)hipFree(ptr);
}

Status DeviceBuffer::reserve(size_t bytes, bool *reallocated)
{
    if (reallocated)
        *reallocated = false;
    if (bytes <= capacity && ptr)
        return Status{};
    if (bytes == 0)
        bytes = 256;
    if (ptr) {
        CG_HIP(hipFree(ptr)); // waits for work that may still use the old block
        ptr = nullptr;
        capacity = 0;
    }
    CG_HIP(hipMalloc(&ptr, bytes));
    capacity = bytes;
    if (reallocated)
        *reallocated = true;
    return Status{};
}

void *DeviceBuffer::release()
{
    void *p = ptr;
    ptr = nullptr;
    capacity = 0;
    return p;
}

PinnedBuffer::~PinnedBuffer()
{
    if (ptr)
        (void)hipHostFree(ptr);
}

Status PinnedBuffer::reserve(size_t bytes)
{
    if (bytes <= capacity && ptr)
        return Status{};
    if (ptr) {
        CG_HIP(hipHostFree(ptr));
        ptr = nullptr;
        capacity = 0;
    }
    bytes = std::max<size_t>(bytes + bytes / 4, 64 * 1024);
    CG_HIP(hipHostMalloc(&ptr, bytes, hipHostMallocDefault));
    capacity = bytes;
    return Status{};
}

// Development switch: COMPEG_PIPELINE=split selects the two-kernel pipeline
// (huffman_kernel -> HBM coefficients -> idct_composite_kernel); the default
// is the fused single kernel.
bool use_fused_pipeline()
{
    static const bool fused = [] {
        const char *e = lab_env("COMPEG_PIPELINE");
        return !(e && strcmp(e, "split") == 0);
    }();
    return fused;
}

// Launches too small to fill the chip (no more decoder waves than there are
// SIMDs) use the paired-wave kernel: it halves the critical path of a wave
// at the price of a second wave per 64 intervals.
bool use_pair_kernel(uint32_t max_intervals, uint32_t images)
{
    static const int forced = [] {
        const char *e = lab_env("COMPEG_PAIR"); // experiment knob: 0 / 1
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0)
        return forced != 0;
    const uint64_t waves = uint64_t((max_intervals + kWave - 1) / kWave) * images;
    return waves <= 1024; // (measured: four 4K frames, 1016 waves, 89 us paired / 101 us fused; five: 133 / 116)
}

// ... and of those, the ones whose images qualify (ImageDesc::coop_ok, one restart interval for the whole
// launch) take the cooperative kernel, which spends the idle lanes inside the intervals (coop_body.h).
// The batch kernel with the window in its streamed form (kernels.h) instead of whole-interval windows.
// Readable bytes behind the words of the last image of a buffer: the streamed window's rows are fetched without a
// look at where a scan ends (up to 64 words beyond its last; never used).
constexpr size_t kStreamSlackBytes = 1024;

// A batch image's output: rows of whole MCUs (16 pixels each way) -- an MCU the image's edge cuts is stored whole, its
// outside into padding; rows begin on 64-byte boundaries whatever the width (device_types.h: out_alloc_h).
static uint32_t output_pitch(uint32_t width) { return (width + 15u) / 16u * 64u; }
static uint32_t output_rows(uint32_t height) { return (height + 15u) / 16u * 16u; }
static size_t output_bytes(uint32_t width, uint32_t height) { return size_t(output_pitch(width)) * output_rows(height); }

bool use_stream_kernel(const HuffLdsPlan &plan, uint32_t max_intervals, uint32_t images, uint32_t cu_waves = 0, uint32_t group_waves = 0)
{
    static const int forced = [] {
        const char *e = lab_env("COMPEG_STREAM"); // experiment knob: 0 / 1
        return e ? atoi(e) : -1;
    }();
    return forced >= 0 ? forced != 0 : stream_plan_preferred(plan, max_intervals, images, cu_waves, group_waves);
}

// The extension layouts' kernels: waves a CU holds (their registers: two to a SIMD) and the most a workgroup has
// (fused_layout_wave_cap); a streamed form exists for 4:2:0 and for 4:4:4 / 4:4:0 with restart intervals of two MCUs or more.
constexpr uint32_t kLayoutCuWaves = 8;
bool layout_has_stream_kernel(uint32_t hs, uint32_t vs, bool pairs) { return (hs == 2 && vs == 2) || (hs == 1 && pairs); }

bool use_coop_kernel(uint32_t max_intervals, uint32_t images, uint32_t restart_interval)
{
    static const int forced = [] {
        const char *e = lab_env("COMPEG_COOP"); // experiment knob: 0 / 1
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0)
        return forced != 0;
    // Measured (DRI = 4, kernel time by HIP events, paired or fused / cooperative in its team form): one frame
    // 1280x720 59 / 32 us, 1920x1080 56 / 34, 3840x2160 59 / 37; four 1080p frames 62 / 38; two 4K frames 74 / 63,
    // three 76 / 88, four 81 / 117.  The cooperative kernel's teams (256 data units each) are resident four to a
    // CU: up to two rounds of them it is ahead.
    const uint64_t data_units = uint64_t(max_intervals) * images * 4u * restart_interval;
    return data_units <= 2ull * 1024u * 256u;
}

// What the cooperative kernel plans its windows with: the largest word span of a team's group of intervals, for
// each of the three group sizes it knows (kernels.h: CoopSpans).
CoopSpans coop_spans_exact(const uint32_t *starts, size_t nstarts, size_t nwords, uint32_t intervals, uint32_t restart_interval)
{
    CoopSpans sp{};
    for (uint32_t k = 0; k < 3u; k++) {
        const uint32_t ipw = coop_shape(restart_interval, 4u >> k).ipw;
        // (a smaller group's span is no larger: where the larger group fits the kernel's ordinary window already, its
        // figure serves as the bound and the pass over the start positions is saved)
        const bool reuse = k && (ipw == coop_shape(restart_interval, 4u >> (k - 1u)).ipw || sp.words[k - 1u] <= 1024u);
        sp.words[k] = reuse ? sp.words[k - 1u] : max_wave_span(starts, nstarts, nwords, intervals, ipw);
    }
    return sp;
}

// Device preprocessing reports the largest word span of 64 consecutive intervals only: a group's share of it and
// half as much again.  (A group that is longer than that still decodes: the walks that leave the window hand their
// interval to the serial decoder.)
// generous: span_of_64 is a measured maximum (batches), not itself an estimate with slack in it (a decoder's
// blocking decode: twice the average span)
CoopSpans coop_spans_estimate(uint32_t span_of_64, uint32_t restart_interval, bool generous)
{
    CoopSpans sp{};
    for (uint32_t k = 0; k < 3u; k++) {
        const uint32_t ipw = coop_shape(restart_interval, 4u >> k).ipw;
        const uint64_t share = uint64_t(span_of_64) * ipw / kWave;
        sp.words[k] = ipw >= uint32_t(kWave) ? span_of_64
                                             : uint32_t(std::min<uint64_t>(generous ? span_of_64 : 0x7fffffffu, share + (generous ? share / 2 : 0) + 64));
    }
    return sp;
}

void coop_spans_max(CoopSpans &into, const CoopSpans &other)
{
    for (uint32_t k = 0; k < 3u; k++)
        into.words[k] = std::max(into.words[k], other.words[k]);
}

// Single-image device preprocessing: the raw segment is fetched from the pinned staging buffer by a
// kernel (COMPEG_PULL=0: by the copy engine).
bool pull_copies()
{
    static const bool pull = [] {
        const char *e = lab_env("COMPEG_PULL");
        return e ? atoi(e) != 0 : true;
    }();
    return pull;
}

namespace {

void *pinned_alloc(size_t n)
{
    void *p = nullptr;
    return hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void pinned_free(void *p) { (void)hipHostFree(p); }

bool is_422(const ImageData &img)
{
    return img.metadata.max_hsample == 2 && img.metadata.max_vsample == 1;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The threads behind run_on_threads: started once and kept, asleep between two calls.  (Thirty-two fresh threads per
// batch upload came to life one after the other over more than a millisecond of a ten-millisecond upload.)  One call
// at a time uses them; a second caller meanwhile gets threads of its own.
class WorkerPool {
public:
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread &t : threads_)
            t.join();
    }

    // work(t) for t = 1 .. nthreads-1 on pool threads; false: the pool is busy
    bool run(unsigned nthreads, void (*call)(void *, unsigned), void *work)
    {
        std::unique_lock<std::mutex> one(busy_, std::try_to_lock);
        if (!one.owns_lock())
            return false;
        {
            std::lock_guard<std::mutex> g(m_);
            while (threads_.size() + 1 < nthreads) {
                const unsigned index = unsigned(threads_.size());
                threads_.emplace_back([this, index] { loop(index); });
            }
            call_ = call;
            work_ = work;
            wanted_ = nthreads - 1;
            pending_ = nthreads - 1;
            generation_++;
        }
        wake_.notify_all();
        call(work, 0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        return true;
    }

private:
    void loop(unsigned index)
    {
        uint64_t seen = 0;
        for (;;) {
            void (*call)(void *, unsigned);
            void *work;
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return stop_ || (generation_ != seen && index < wanted_); });
                if (stop_)
                    return;
                seen = generation_;
                call = call_;
                work = work_;
            }
            call(work, index + 1);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0)
                    done_.notify_all();
            }
        }
    }

    std::mutex busy_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> threads_;
    void (*call_)(void *, unsigned) = nullptr;
    void *work_ = nullptr;
    unsigned wanted_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// (one pool for every kind of work: a local static of the template below would be a pool per instantiation)
WorkerPool &worker_pool()
{
    static WorkerPool pool;
    return pool;
}

// work(t) for t = 0 .. nthreads-1, t = 0 on the caller's thread
template <typename Work>
void run_on_threads(unsigned nthreads, Work &work)
{
    if (nthreads <= 1) {
        work(0);
        return;
    }
    WorkerPool &pool = worker_pool();
    if (pool.run(nthreads, [](void *w, unsigned t) { (*static_cast<Work *>(w))(t); }, &work))
        return;
    std::vector<std::thread> own;
    for (unsigned t = 0; t < nthreads; t++)
        own.emplace_back([&work, t] { work(t); });
    for (std::thread &th : own)
        th.join();
}

// COMPEG_TRACE=1: one stderr line per enqueue with the host time spent in each step (us)
struct EnqueueTrace {
    bool on;
    std::chrono::steady_clock::time_point last;
    std::string line;
    EnqueueTrace()
    {
        static const bool enabled = getenv("COMPEG_TRACE") != nullptr;
        on = enabled;
        if (on)
            last = std::chrono::steady_clock::now();
    }
    static double us_since(std::chrono::steady_clock::time_point t0)
    {
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    void mark(const char *what)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof buf, " %s=%.1f", what, std::chrono::duration<double, std::micro>(now - last).count());
        line += buf;
        last = now;
    }
    ~EnqueueTrace()
    {
        if (on)
            fprintf(stderr, "[compeg] enqueue:%s\n", line.c_str());
    }
};

// Per-image LUT blob: [L1 2048 B][L2, padded to 4 B][two 11-bit direct AC tables]
size_t table_blob_bytes(const ImageData &img)
{
    return COMPEG_HUFFMAN_L1_BYTES + align_up(img.l2.size() * 2, 4) + img.ac_fast.size() * 2 + img.dc_fast.size() * 2;
}

void write_tables(uint8_t *dst, const ImageData &img)
{
    memcpy(dst, img.l1, COMPEG_HUFFMAN_L1_BYTES);
    uint8_t *l2 = dst + COMPEG_HUFFMAN_L1_BYTES;
    const size_t l2_bytes = align_up(img.l2.size() * 2, 4);
    memset(l2, 0, l2_bytes);
    if (!img.l2.empty())
        memcpy(l2, img.l2.data(), img.l2.size() * 2);
    memcpy(l2 + l2_bytes, img.ac_fast.data(), img.ac_fast.size() * 2);
    memcpy(l2 + l2_bytes + img.ac_fast.size() * 2, img.dc_fast.data(), img.dc_fast.size() * 2);
}

// LUT entries a kernel should stage in LDS behind L1: the L2 LUT and the direct AC tables
uint32_t staged_lut_entries(const ImageData &img)
{
    return uint32_t(align_up(img.l2.size() * 2, 4) / 2 + img.ac_fast.size());
}

} // namespace

} // namespace compeg

using namespace compeg;

compeg_decoder::compeg_decoder() : scan(pinned_alloc, pinned_free)
{
    // helpers for the host scan preprocessor (one image is all a Decoder has to be parallel over)
    const unsigned hw = std::thread::hardware_concurrency();
    unsigned threads = hw >= 32 ? 8 : (hw >= 8 ? 4 : 1);
    bool asked = false;
    if (const char *e = getenv("COMPEG_SCAN_THREADS")) {
        threads = unsigned(std::max(1, std::min(16, atoi(e))));
        asked = true;
    }
    scan.set_threads(threads, !asked); // a default nobody asked for checks itself on first use
}

compeg_decoder::~compeg_decoder()
{
    if (last_stream || upload_pending)
        (void)hipStreamSynchronize(last_stream);
    if (upload_done)
        (void)hipEventDestroy(upload_done);
    if (decode_done)
        (void)hipEventDestroy(decode_done);
    if (gpu)
        compeg_gpu_release(gpu);
}

// Device-side variant of the preprocess step of enqueue: raw segment to HBM
// (pinned staging, async copy), scan kernels, then the result words (interval
// count, output size, flags) come back.  Two ways:
//  * before_submit empty: one small synchronous read-back here, and the caller
//    builds the image descriptor from it;
//  * before_submit given (decode_blocking): no synchronisation.  The callback
//    uploads the image descriptor once the output addresses are known, the scan
//    kernels patch the two counts into it (patch_nwords / patch_nstarts), the
//    result words travel to pinned host memory behind them, and the caller
//    checks them after the decode (finish_deferred).
Status compeg_decoder::preprocess_on_device(const ImageData &img, hipStream_t stream, uint32_t &nwords,
                                            uint32_t &nstarts, uint32_t &span, bool &fell_back, size_t blob_bytes,
                                            const BlobWriter &before_submit)
{
    fell_back = false;
    if (img.scan_len > 0xfffffff0u) {
        fell_back = true;
        return Status{};
    }
    const uint32_t len = uint32_t(img.scan_len), expected = img.metadata.total_restart_intervals;
    uint32_t slots = 1;
    while (slots < expected)
        slots <<= 1;
    const uint32_t ntiles = scan_tiles(len);
    size_t total = 64;
    auto take = [&](size_t bytes) {
        const size_t at = total;
        total += align_up(bytes, 256);
        return at;
    };
    // everything the host writes sits in front of the raw bytes and travels with their first piece: the
    // zeroed result words (first 16 bytes), the scan descriptor and, for a deferred decode, the image's blob
    const size_t o_res = 0, o_desc = take(sizeof(ScanDesc)), o_blob = take(before_submit ? blob_bytes : 0),
                 o_raw = take(size_t(len) + 64), o_ts = take(size_t(ntiles) * kScanTileStateBytes + 32),
                 o_st = take(size_t(slots) * 4), o_w = take(size_t(len) + len / 3 + 64);
    CG_TRY(scan_arena.reserve(total + kStreamSlackBytes)); // (words are the arena's last part)
    CG_TRY(raw_stage.reserve(o_raw + len + 64));
    uint8_t *da = static_cast<uint8_t *>(scan_arena.ptr), *hs = static_cast<uint8_t *>(raw_stage.ptr);
    ScanDesc s;
    s.raw = da + o_raw;
    s.len = len;
    s.ntiles = ntiles;
    s.slots = slots;
    s.tile_state = reinterpret_cast<uint32_t *>(da + o_ts);
    s.starts_out = reinterpret_cast<uint32_t *>(da + o_st);
    s.words_out = da + o_w;
    s.result = reinterpret_cast<uint32_t *>(da + o_res);
    dev_words = da + o_w;
    dev_starts = da + o_st;
    EnqueueTrace trace;
    memset(hs, 0, o_blob);
    if (before_submit)
        CG_TRY(before_submit(hs + o_blob, da + o_blob, &s.patch_nwords, &s.patch_nstarts));
    memcpy(hs + o_desc, &s, sizeof s);
    // the segment goes through the pinned staging buffer: copied by the scan buffer's threads if it
    // has helpers (then one transfer follows), else piece by piece, so that the transfer of one piece
    // runs under the host copy of the next (few pieces: every transfer has a fixed cost)
    const size_t piece = scan.threads() > 1 ? size_t(len) + 1 : size_t(640u << 10);
    for (size_t at = 0; at < len || at == 0; at += piece) {
        const size_t n = std::min<size_t>(piece, len - at);
        scan.copy(hs + o_raw + at, img.scan_data() + at, n);
        const size_t from = at ? o_raw + at : 0; // the first piece carries the descriptor
        if (pull_copies())
            CG_HIP(launch_pull(da + from, hs + from, o_raw + at + n - from, stream));
        else
            CG_HIP(hipMemcpyAsync(da + from, hs + from, o_raw + at + n - from, hipMemcpyHostToDevice, stream));
        if (len == 0)
            break;
    }
    trace.mark("  stage_and_copy");
    CG_HIP(launch_scan(reinterpret_cast<const ScanDesc *>(da + o_desc), 1, ntiles, stream));
    CG_TRY(scan_result.reserve(16));
    uint32_t *res = static_cast<uint32_t *>(scan_result.ptr);
    scan_result_dev = da + o_res;
    deferred_expected = expected;
    if (!before_submit)
        CG_HIP(hipMemcpyAsync(res, scan_result_dev, 16, hipMemcpyDeviceToHost, stream));
    trace.mark("submit");
    if (before_submit) {
        // (the result words are fetched behind the decode kernel: fetch_deferred_result)
        // the per-wave span is unknown: size the window generously from the largest possible average
        const uint64_t most_words = (uint64_t(len) + 3u * uint64_t(expected)) / 4u + 1u;
        const uint64_t avg = expected ? (most_words + expected - 1) / expected : most_words;
        span = uint32_t(std::min<uint64_t>(2 * avg * kWave + 64, 0x7fffffffu));
        nwords = nstarts = 0; // patched into the descriptor on the device
        deferred_check = true;
        return Status{};
    }
    CG_HIP(hipStreamSynchronize(stream));
    trace.mark("sync");
    CG_TRY(check_scan_result(fell_back));
    if (fell_back)
        return Status{};
    nwords = res[2];
    nstarts = std::min(res[0], slots);
    // the per-wave span is not read back: size the window generously from the average
    const uint64_t avg = expected ? (uint64_t(nwords) + expected - 1) / expected : nwords;
    span = uint32_t(std::min<uint64_t>(2 * avg * kWave + 64, 0x7fffffffu));
    return Status{};
}

// Reads the scan kernels' result words (in pinned memory, complete once the stream has passed
// the copy): hand the image back to the host preprocessor, or note a count mismatch.
Status compeg_decoder::check_scan_result(bool &fell_back)
{
    const uint32_t *res = static_cast<const uint32_t *>(scan_result.ptr);
    fell_back = (res[3] & 1u) != 0u; // an FF run beyond the kernels' look-back bound
    if (!fell_back && res[0] != deferred_expected) {
        char msg[128];
        snprintf(msg, sizeof msg, "restart interval count mismatch: counted %u, expected %u", res[0],
                 deferred_expected);
        warning = msg; // the reference drops this error (lib.rs:391-394)
    }
    return Status{};
}

// Second half of a decode whose scan was preprocessed without a read-back: the stream has
// finished; if the scan kernels gave the image up, decode it again through the host preprocessor.
Status compeg_decoder::finish_deferred(const ImageData &img, hipStream_t stream)
{
    if (!deferred_check)
        return Status{};
    deferred_check = false;
    bool fell_back = false;
    CG_TRY(check_scan_result(fell_back));
    if (!fell_back)
        return Status{};
    const bool keep = device_preprocess;
    device_preprocess = false;
    Status st = enqueue(img, stream, nullptr, false);
    device_preprocess = keep;
    CG_TRY(st);
    CG_HIP(hipStreamSynchronize(stream));
    return Status{};
}

// Counterpart of Decoder::enqueue (src/lib.rs:385-477).
Status compeg_decoder::enqueue(const ImageData &img, hipStream_t stream, bool *changed, bool may_defer)
{
    EnqueueTrace trace;
    const auto t_enter = std::chrono::steady_clock::now();
    stage_times = compeg_stage_times{0.0, 0.0, 0.0};
    double preprocess_us = 0.0;
    // (every way out of this function leaves the split in stage_times)
    struct Closer {
        compeg_decoder *self;
        std::chrono::steady_clock::time_point t0;
        const double *pre;
        ~Closer()
        {
            self->stage_times.preprocess_us = *pre;
            self->stage_times.enqueue_writes_us = EnqueueTrace::us_since(t0) - *pre;
        }
    } closer{this, t_enter, &preprocess_us};
    CG_HIP(hipSetDevice(gpu->device));
    warning.clear();

    // One decoder, one set of device buffers (descriptor, LUTs, scan words, output): a decode recorded on
    // another stream than the previous one must not overwrite them while that one may still be running.
    // The reference gets this ordering from wgpu's queue; here the new stream waits for the old decode.
    if (!decode_done)
        CG_HIP(hipEventCreateWithFlags(&decode_done, hipEventDisableTiming));
    if (decode_pending && stream != last_stream)
        CG_HIP(hipStreamWaitEvent(stream, decode_done, 0));

    // DynamicTexture::reserve (dynamic.rs:214-248): recreate at exactly the
    // requested size when either dimension is too small.
    bool realloc_out = false;
    if (img.width > out_w || img.height > out_h || !out.ptr) {
        // (rows of whole MCUs, 16 pixels each way: an MCU the extent's edge cuts is stored whole, rows begin on 64-byte
        // boundaries whatever the width -- 256 x 1080x1920 portrait frames 1608 -> 8xx us, device_types.h: out_alloc_h)
        const size_t pitch = align_up(size_t(img.width), 16) * 4;
        const size_t alloc_rows = align_up(size_t(img.height), 16);
        // never shrink the allocation itself, only the logical extent
        bool fresh = false;
        CG_TRY(out.reserve(std::max<size_t>(pitch * alloc_rows, 256), &fresh));
        // wgpu zero-initialises new textures; texels no MCU covers (a truncated
        // last restart interval, lib.rs:785) therefore read as 0 in the reference --
        // also when the new logical texture lands in an allocation that is already there
        (void)fresh;
        CG_HIP(hipMemsetAsync(out.ptr, 0, std::min(out.capacity, std::max<size_t>(pitch * alloc_rows, 256)), stream));
        out_w = img.width;
        out_h = img.height;
        out_pitch = pitch;
        out_alloc_h = uint32_t(alloc_rows);
        realloc_out = true;
    }
    if (changed)
        *changed = realloc_out;

    if (!upload_done)
        CG_HIP(hipEventCreateWithFlags(&upload_done, hipEventDisableTiming));
    if (upload_pending) {
        CG_HIP(hipEventSynchronize(upload_done));
        upload_pending = false;
    }

    const Metadata &md = img.metadata;
    deferred_check = false;
    const uint32_t total_dus = img.total_dus();
    // The walk + lane-per-MCU route may take this image (decided below, with the scan in hand): its records' place, and
    // behind descriptor and tables the image's second descriptor, in which every MCU is an "interval" (kernels_body.h).
    const uint64_t route_mcus = uint64_t(md.total_restart_intervals) * md.restart_interval;
    const bool route_possible = is_422(img) && use_fused_pipeline() && md.restart_interval >= 8u && route_mcus < (1ull << 28);
    const size_t view_off = align_up(sizeof(ImageDesc), 256) + align_up(table_blob_bytes(img), 256);
    const size_t blob_bytes = route_possible ? view_off + align_up(sizeof(ImageDesc), 256) : align_up(sizeof(ImageDesc), 256) + table_blob_bytes(img);
    if (route_possible) {
        CG_TRY(mcu_words.reserve((size_t(route_mcus) + kWave) * 4 + 256));
        CG_TRY(mcu_states.reserve(size_t(route_mcus) * sizeof(McuState) + 256));
    }
    CG_TRY(host_blob.reserve(blob_bytes));
    CG_TRY(dev_blob.reserve(blob_bytes + 16));
    CG_TRY(ac.reserve(size_t(total_dus) * kRetained * 2 + 64));
    CG_TRY(dc.reserve(size_t(total_dus) * 4 + 64));
    uint8_t *hb = static_cast<uint8_t *>(host_blob.ptr);
    uint8_t *db = static_cast<uint8_t *>(dev_blob.ptr);
    const size_t l1_off = align_up(sizeof(ImageDesc), 256), l2_off = l1_off + COMPEG_HUFFMAN_L1_BYTES;

    // the cooperative kernel may take this image: its walk tables' place
    const bool want_walk = is_422(img) && use_fused_pipeline() &&
                           (route_possible || use_coop_kernel(md.total_restart_intervals, 1, md.restart_interval));
    if (want_walk)
        CG_TRY(walk_tables.reserve(kWalkTableBytes));

    // image descriptor + LUTs into the staging blob at hb, as they will sit at db
    auto write_blob = [&](const void *words_ptr, const void *starts_ptr, size_t n_words, size_t n_starts) {
        ImageDesc &d = *reinterpret_cast<ImageDesc *>(hb);
        fill_desc(img, d);
        d.walk = want_walk && (d.coop_ok || (route_possible && d.mcu_ok)) ? static_cast<const uint32_t *>(walk_tables.ptr) : nullptr;
        d.words = static_cast<const uint32_t *>(words_ptr);
        d.starts = static_cast<const uint32_t *>(starts_ptr);
        d.nwords = uint32_t(n_words);
        d.nstarts = uint32_t(n_starts);
        d.l1 = reinterpret_cast<const uint16_t *>(db + l1_off);
        d.l2 = reinterpret_cast<const uint16_t *>(db + l2_off);
        d.ac = static_cast<int16_t *>(ac.ptr);
        d.dc = static_cast<int32_t *>(dc.ptr);
        d.out = static_cast<uint8_t *>(out.ptr);
        d.out_w = out_w;
        d.out_h = out_h;
        d.out_pitch = uint32_t(out_pitch);
        d.out_alloc_h = out_alloc_h;
        write_tables(hb + l1_off, img);
        if (route_possible) {
            d.mcu_word = static_cast<uint32_t *>(mcu_words.ptr);
            d.mcu_state = static_cast<McuState *>(mcu_states.ptr);
            ImageDesc &v = *reinterpret_cast<ImageDesc *>(hb + view_off);
            v = d;
            v.starts = d.mcu_word;
            v.nstarts = d.total_mcus;
            v.total_intervals = d.total_mcus;
            v.restart_interval = 1;
            v.walk = nullptr;
            v.coop_ok = 0;
        }
    };

    uint32_t dev_nwords = 0, dev_nstarts = 0, dev_span = 0;
    bool on_device = device_preprocess && use_fused_pipeline();
    bool blob_uploaded = false;
    const auto t_pre = std::chrono::steady_clock::now();
    if (on_device) {
        bool fell_back = false;
        BlobWriter before_submit;
        // (an image only the walk route decodes well -- intervals longer than the cooperative kernel takes, up to no DRI
        // at all -- gets its scan's counts read back even in a blocking decode: the route's second descriptor needs them,
        // and a lane per interval would take the whole image's time several times over)
        const bool defer_scan = may_defer && !(route_possible && md.restart_interval > 256u);
        if (defer_scan)
            before_submit = [&](uint8_t *host_at, uint8_t *dev_at, uint32_t **patch_nwords,
                                uint32_t **patch_nstarts) -> Status {
                // the blob rides in front of the raw segment; the scan kernels fill in the two counts
                // behind that copy, in stream order
                hb = host_at;
                db = dev_at;
                *patch_nwords = reinterpret_cast<uint32_t *>(db + offsetof(ImageDesc, nwords));
                *patch_nstarts = reinterpret_cast<uint32_t *>(db + offsetof(ImageDesc, nstarts));
                blob_uploaded = true;
                write_blob(dev_words, dev_starts, 0, 0);
                return Status{};
            };
        CG_TRY(preprocess_on_device(img, stream, dev_nwords, dev_nstarts, dev_span, fell_back, blob_bytes,
                                    before_submit));
        on_device = !fell_back;
        if (fell_back) { // (reported before anything was written)
            blob_uploaded = false;
            hb = static_cast<uint8_t *>(host_blob.ptr);
            db = static_cast<uint8_t *>(dev_blob.ptr);
        }
    }
    size_t shipped = 0; // bytes of the preprocessed scan already on their way
    bool blob_prebuilt = false;
    if (!on_device) {
        // the output's worst case (scan.rs:38-44), so that pieces can leave while the scan is running
        CG_TRY(words.reserve(ScanBuffer::output_capacity(img.scan_len) + kStreamSlackBytes));
        CG_TRY(starts.reserve(ScanBuffer::start_slots(md.total_restart_intervals) * 4 + 16));
        hipError_t ship_error = hipSuccess;
        ScanBuffer::Progress ship;
        if (pull_copies())
            ship = [&](size_t final_bytes) {
                if (final_bytes == 0 && !blob_prebuilt) {
                    // (the scan's first round is under way on the helpers: what does not depend on it; the two
                    // counts follow below)
                    write_blob(words.ptr, starts.ptr, 0, 0);
                    blob_prebuilt = true;
                }
                if (final_bytes > shipped && ship_error == hipSuccess)
                    ship_error = launch_pull(static_cast<uint8_t *>(words.ptr) + shipped, scan.data() + shipped,
                                             final_bytes - shipped, stream);
                shipped = std::max(shipped, final_bytes);
            };
        Status pre = scan.process(img.scan_data(), img.scan_len, md.total_restart_intervals, ship, 384u << 10);
        CG_HIP(ship_error);
        if (!pre.ok()) {
            if (pre.code != COMPEG_E_COUNT_MISMATCH)
                return pre;
            warning = pre.message; // the reference drops this error (lib.rs:391-394)
        }
    }
    preprocess_us = EnqueueTrace::us_since(t_pre);
    trace.mark(on_device ? "device_preprocess" : "host_preprocess");
    const size_t n_words = on_device ? dev_nwords : scan.nwords();
    const size_t n_starts = on_device ? dev_nstarts : scan.nstarts();
    const bool pull = !on_device && pull_copies(); // everything the host path uploads sits in pinned memory
    // The rest of the preprocessed scan, the start positions, descriptor + tables: one launch for the three (each alone
    // is a launch and a PCIe round trip of its own).  A large rest leaves on its own first -- the card fetches it
    // while the host makes descriptor and tables.
    size_t rest_at = shipped, rest_bytes = pull && n_words * 4 > shipped ? n_words * 4 - shipped : 0;
    size_t rest_alone = blob_prebuilt ? size_t(-1) : size_t(256u << 10); // (nothing left to make: one launch for all)
    if (const char *e = lab_env("COMPEG_REST_ALONE")) // experiment knob: bytes
        rest_alone = size_t(atol(e));
    if (pull && rest_bytes > rest_alone) {
        CG_HIP(launch_pull(static_cast<uint8_t *>(words.ptr) + rest_at, scan.data() + rest_at, rest_bytes, stream));
        rest_bytes = 0;
    }
    trace.mark("copies");
    if (blob_prebuilt) {
        ImageDesc &d = *reinterpret_cast<ImageDesc *>(hb);
        d.nwords = uint32_t(n_words);
        d.nstarts = uint32_t(n_starts);
        if (route_possible)
            reinterpret_cast<ImageDesc *>(hb + view_off)->nwords = uint32_t(n_words);
    } else if (!blob_uploaded) {
        write_blob(on_device ? dev_words : words.ptr, on_device ? dev_starts : starts.ptr, n_words, n_starts);
    }
    if (pull) {
        void *const dsts[3] = {static_cast<uint8_t *>(words.ptr) + rest_at, starts.ptr, db};
        const void *const srcs[3] = {scan.data() + rest_at, scan.starts(), hb};
        const size_t sizes[3] = {rest_bytes, n_starts * 4, blob_uploaded ? 0 : blob_bytes};
        CG_HIP(launch_pull3(dsts, srcs, sizes, stream));
    } else if (!blob_uploaded) {
        CG_HIP(hipMemcpyAsync(db, hb, blob_bytes, hipMemcpyHostToDevice, stream));
    }
    trace.mark("tables");
    if (pull) {
    } else if (!on_device) {
        if (n_starts)
            CG_HIP(hipMemcpyAsync(starts.ptr, scan.starts(), n_starts * 4, hipMemcpyHostToDevice, stream));
        if (n_words)
            CG_HIP(hipMemcpyAsync(words.ptr, scan.words(), n_words * 4, hipMemcpyHostToDevice, stream));
    }
    // (a blocking decode waits for the stream itself: a marker between the uploads and the decode kernel would only
    // keep the kernel waiting for it -- 3 us of the 6 between the last pull's end and the kernel's start)
    if (!may_defer) {
        CG_HIP(hipEventRecord(upload_done, stream));
        upload_pending = true;
    }
    last_stream = stream;
    last_md = md;
    last_desc_dev = db;
    have_last = true;

    last_kernel = COMPEG_KERNEL_NONE;
    if (total_dus == 0) {
        if (deferred_check)
            CG_HIP(hipMemcpyAsync(scan_result.ptr, scan_result_dev, 16, hipMemcpyDeviceToHost, stream));
        if (!may_defer) {
            CG_HIP(hipEventRecord(decode_done, stream));
            decode_pending = true;
        }
        return Status{};
    }
    const uint32_t span = on_device ? dev_span
                                    : max_wave_span(scan.starts(), scan.nstarts(), scan.nwords(),
                                                    md.total_restart_intervals);
    const bool fused = use_fused_pipeline() && is_422(img);
    // (the extension pipeline's first kernel carries the IDCT: planned like the fused kernel)
    const uint32_t luma_h = md.components[0].hsample, luma_v = md.components[0].vsample;
    // (8-pixel MCUs in pairs: rows of 64 bytes; of an odd interval the last MCU alone.  Intervals of one MCU: the single form)
    const bool mcu_pairs = luma_h == 1 && md.restart_interval >= 2u;
    const HuffLdsPlan plan = plan_huffman(md.total_restart_intervals, 1, staged_lut_entries(img), span,
                                          fused || !is_422(img), is_422(img) ? 0u : fused_layout_wave_cap(luma_h, luma_v, mcu_pairs));
    last_span = span;
    last_plan = plan;
    trace.mark("plan");
    if (!is_422(img)) {
        // extension layouts (4:4:4, 4:4:0, 4:2:0): one fused kernel per layout (the development pipeline keeps the
        // two-kernel route: entropy stage with the IDCT in place, generic composite)
        if (use_fused_pipeline() && layout_has_stream_kernel(luma_h, luma_v, mcu_pairs) &&
            use_stream_kernel(plan, md.total_restart_intervals, 1, kLayoutCuWaves, fused_layout_wave_cap(luma_h, luma_v, mcu_pairs))) {
            const uint64_t mcus = std::max<uint64_t>(1u, uint64_t(md.total_restart_intervals) * std::max(1u, uint32_t(md.restart_interval)));
            const StreamPlan sp = plan_stream(md.total_restart_intervals, 1, staged_lut_entries(img), uint32_t((img.scan_len / 4u + mcus - 1u) / mcus),
                                              true, kLayoutCuWaves, fused_layout_wave_cap(luma_h, luma_v, mcu_pairs));
            CG_HIP(launch_fused_stream(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals, sp, stream, luma_h, luma_v));
            last_kernel = COMPEG_KERNEL_FUSED_STREAM;
        } else if (use_fused_pipeline()) {
            CG_HIP(launch_fused_layout(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals, plan, luma_h, luma_v, mcu_pairs, stream));
            last_kernel = COMPEG_KERNEL_FUSED_LAYOUT;
        } else {
            CG_HIP(launch_entropy_samples(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals,
                                          plan, stream));
            CG_HIP(launch_generic_composite(reinterpret_cast<const ImageDesc *>(db), 1, out_w, out_h, stream));
            last_kernel = COMPEG_KERNEL_GENERIC;
        }
        coefficients_valid = false;
    } else if (fused) {
        CoopPlan coop{};
        if (reinterpret_cast<const ImageDesc *>(hb)->coop_ok && use_coop_kernel(md.total_restart_intervals, 1, md.restart_interval)) {
            // (on the device path `dev_span` is itself an estimate, twice the average span of 64 intervals)
            const CoopSpans spans = on_device ? coop_spans_estimate(dev_span, md.restart_interval, false)
                                              : coop_spans_exact(scan.starts(), scan.nstarts(), scan.nwords(),
                                                                 md.total_restart_intervals, md.restart_interval);
            coop = plan_coop(md.total_restart_intervals, 1, md.restart_interval, staged_lut_entries(img), spans);
        }
        const ImageDesc &hd = *reinterpret_cast<const ImageDesc *>(hb);
        // the walk tables: made from the direct tables and from which of them each component uses
        auto ensure_walk_tables = [&]() -> Status {
            if (!hd.walk)
                return Status{};
            const size_t tb = table_blob_bytes(img), extra = sizeof hd.fast_table + sizeof hd.dc_fast_table + sizeof hd.fast_off;
            uint8_t ids[sizeof hd.fast_table + sizeof hd.dc_fast_table + sizeof hd.fast_off];
            memcpy(ids, hd.fast_table, sizeof hd.fast_table);
            memcpy(ids + sizeof hd.fast_table, hd.dc_fast_table, sizeof hd.dc_fast_table);
            memcpy(ids + sizeof hd.fast_table + sizeof hd.dc_fast_table, &hd.fast_off, sizeof hd.fast_off);
            if (walk_key.size() != tb + extra || memcmp(walk_key.data(), hb + l1_off, tb) != 0 ||
                memcmp(walk_key.data() + tb, ids, extra) != 0) {
                walk_key.assign(hb + l1_off, hb + l1_off + tb);
                walk_key.insert(walk_key.end(), ids, ids + extra);
                CG_HIP(launch_walk_tables(reinterpret_cast<const ImageDesc *>(db), 1, stream));
            }
            return Status{};
        };
        // The walk + lane-per-MCU route (use_mcu_route: the same terms for one image): where the cooperative kernel is not
        // the better of the two (coop_preferred) or cannot take the image, a lane per interval would leave the chip
        // its SIMDs a wave each at most, and the walk's rows hold a few MCUs.  (Not where the descriptors went up in front of
        // the scan kernels -- a blocking decode with device preprocessing: the second descriptor's word count is theirs to
        // fill in, and they know of one descriptor.)
        const uint64_t mcu_words_avg = (img.scan_len / 4u + std::max<uint64_t>(route_mcus, 1u) - 1u) / std::max<uint64_t>(route_mcus, 1u);
        const bool route = route_possible && !blob_uploaded && hd.mcu_ok && hd.walk && !lab_env("COMPEG_NO_DECODER_ROUTE") &&
                           !(coop.usable && coop_preferred(coop, md.total_restart_intervals, 1, md.restart_interval)) &&
                           (md.total_restart_intervals + kWave - 1) / kWave <= 1024u && mcu_words_avg <= 24u &&
                           14.5 * md.restart_interval - 40.0 > double(route_mcus) / 5300.0;
        if (route) {
            CG_TRY(ensure_walk_tables());
            const uint32_t l2n = staged_lut_entries(img);
            CG_HIP(launch_walk_mcus(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals,
                                    plan_walk(md.total_restart_intervals, 1, l2n, uint32_t(mcu_words_avg), md.restart_interval, true, true), stream, nullptr));
            // (the second kernel's window: the words of 64 consecutive MCUs -- three times the average of the largest
            // span of 64 intervals' MCUs and a little, at most all of it: compeg_batch::make_walk_tables)
            const uint64_t avg64 = (uint64_t(span) + md.restart_interval - 1) / md.restart_interval;
            const uint32_t mcu_span = uint32_t(std::min<uint64_t>(3 * avg64 + 64, span));
            const HuffLdsPlan mcu_plan = plan_huffman(hd.total_mcus, 1, l2n, mcu_span, true);
            CG_HIP(launch_fused_422(reinterpret_cast<const ImageDesc *>(db + view_off), 1, hd.total_mcus, mcu_plan, stream, true, true, nullptr, true));
            last_kernel = COMPEG_KERNEL_WALK_MCU;
        } else if (coop.usable) {
            CG_TRY(ensure_walk_tables());
            CG_HIP(launch_coop_422(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals, coop, stream));
            last_kernel = COMPEG_KERNEL_COOP_TEAM;
        } else if (use_stream_kernel(plan, md.total_restart_intervals, 1)) {
            const uint64_t mcus = std::max<uint64_t>(1u, uint64_t(md.total_restart_intervals) * std::max(1u, uint32_t(md.restart_interval)));
            const StreamPlan sp = plan_stream(md.total_restart_intervals, 1, staged_lut_entries(img),
                                              uint32_t((img.scan_len / 4u + mcus - 1u) / mcus), true);
            CG_HIP(launch_fused_stream(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals, sp, stream));
            last_kernel = COMPEG_KERNEL_FUSED_STREAM;
        } else if (use_pair_kernel(md.total_restart_intervals, 1)) {
            CG_HIP(launch_pair_422(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals,
                                   plan, stream));
            last_kernel = COMPEG_KERNEL_PAIR;
        } else {
            CG_HIP(launch_fused_422(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals,
                                    plan, stream, false, md.restart_interval == 1u));
            last_kernel = COMPEG_KERNEL_FUSED;
        }
        coefficients_valid = false;
    } else {
        last_kernel = COMPEG_KERNEL_SPLIT;
        CG_HIP(launch_entropy(reinterpret_cast<const ImageDesc *>(db), 1, md.total_restart_intervals,
                              plan, stream));
        CG_HIP(launch_idct_composite(reinterpret_cast<const ImageDesc *>(db), 1, total_dus, stream));
        coefficients_valid = true;
    }
    if (deferred_check) // behind the decode kernel: no copy engine between the scan kernels and it
        CG_HIP(hipMemcpyAsync(scan_result.ptr, scan_result_dev, 16, hipMemcpyDeviceToHost, stream));
    if (!may_defer) {
        CG_HIP(hipEventRecord(decode_done, stream));
        decode_pending = true;
    }
    trace.mark("launch");
    return Status{};
}

// ScanBuffer::process on the device: copy the raw segment to HBM, run the scan
// kernels, copy words / start positions back into this buffer's host arrays.
Status compeg::ScanBuffer::process_on_gpu(compeg_gpu *gpu, const uint8_t *scan, size_t len,
                                          uint32_t expected)
{
    if (len > 0xfffffff0u)
        return Status::error(COMPEG_E_INVALID_ARG, "scan segment too large");
    CG_HIP(hipSetDevice(gpu->device));
    uint32_t slots = 1;
    while (slots < expected)
        slots <<= 1;
    const uint32_t ntiles = scan_tiles(uint32_t(len));
    DeviceBuffer arena, descbuf;
    size_t total = 64;
    auto take = [&](size_t bytes) {
        const size_t at = total;
        total += align_up(bytes, 256);
        return at;
    };
    const size_t o_raw = take(len + 64), o_ts = take(size_t(ntiles) * kScanTileStateBytes + 32),
                 o_st = take(size_t(slots) * 4), o_w = take(len + len / 3 + 64), o_res = take(kScanResultBytes);
    CG_TRY(arena.reserve(total));
    CG_TRY(descbuf.reserve(sizeof(ScanDesc)));
    uint8_t *da = static_cast<uint8_t *>(arena.ptr);
    CG_HIP(hipMemset(da, 0, total));
    if (len)
        CG_HIP(hipMemcpy(da + o_raw, scan, len, hipMemcpyHostToDevice));
    ScanDesc s;
    s.raw = da + o_raw;
    s.len = uint32_t(len);
    s.ntiles = ntiles;
    s.slots = slots;
    s.tile_state = reinterpret_cast<uint32_t *>(da + o_ts);
    s.starts_out = reinterpret_cast<uint32_t *>(da + o_st);
    s.words_out = da + o_w;
    s.result = reinterpret_cast<uint32_t *>(da + o_res);
    CG_HIP(hipMemcpy(descbuf.ptr, &s, sizeof s, hipMemcpyHostToDevice));
    CG_HIP(launch_scan(static_cast<const ScanDesc *>(descbuf.ptr), 1, ntiles, gpu->stream));
    CG_HIP(hipStreamSynchronize(gpu->stream));
    uint32_t res[4];
    CG_HIP(hipMemcpy(res, da + o_res, 16, hipMemcpyDeviceToHost));
    if (res[3] & 1u)
        return process(scan, len, expected); // FF run beyond the kernels' look-back bound
    const size_t out_cap = ((len + len / 3 + 3) / 4) * 4;
    if (!words_.reserve(out_cap + 8) || !starts_.reserve(size_t(slots) * 4))
        return Status::error(COMPEG_E_HIP, "out of host memory in ScanBuffer");
    nwords_ = res[2];
    nstarts_ = std::min(res[0], slots);
    if (nwords_)
        CG_HIP(hipMemcpy(words_.data, da + o_w, nwords_ * 4, hipMemcpyDeviceToHost));
    if (nstarts_)
        CG_HIP(hipMemcpy(starts_.data, da + o_st, nstarts_ * 4, hipMemcpyDeviceToHost));
    if (res[0] != expected) {
        char msg[128];
        snprintf(msg, sizeof msg, "restart interval count mismatch: counted %u, expected %u", res[0], expected);
        return Status::error(COMPEG_E_COUNT_MISMATCH, msg);
    }
    return Status{};
}

// The streams the batches' host-to-device transfers run on: four per device, shared by every batch of the process, so
// that the transfers of uploads begun one after the other (compeg_batch_upload_jpegs_begin) follow each other on the
// link instead of sharing it -- the first upload's data arrives first.  (256 transfers of 1.6 MB: 43 GB/s on one
// stream, 54-56 on two to four.)  Kept for the life of the process.
hipError_t shared_copy_streams(int device, std::vector<hipStream_t> &out)
{
    static std::mutex m;
    static std::vector<hipStream_t> pool[64];
    std::lock_guard<std::mutex> lock(m);
    if (device < 0 || device >= 64)
        return hipErrorInvalidDevice;
    while (pool[device].size() < 4) {
        hipStream_t c = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
        if (e != hipSuccess)
            return e;
        pool[device].push_back(c);
    }
    out = pool[device];
    return hipSuccess;
}

compeg_batch::~compeg_batch()
{
    (void)hipStreamSynchronize(last_stream);
    if (gpu)
        (void)hipStreamSynchronize(gpu->stream);
    for (hipStream_t c : copy_streams)
        (void)hipStreamSynchronize(c);
    for (hipEvent_t e : events)
        (void)hipEventDestroy(e);
    if (decode_done)
        (void)hipEventDestroy(decode_done);
    if (gpu)
        compeg_gpu_release(gpu);
}

namespace {

// What the layout of a batch needs to know about an image before its threads start: exact for parsed images,
// upper bounds (from the file's headers and its length) for images that the threads are going to parse.
// (an upload that laid its arenas out from the files' headers and lengths and found a file that does not fit:
// the caller parses first and lays out afterwards)
constexpr const char *kLayoutBoundExceeded = "layout bound exceeded";

struct UploadItem {
    uint32_t width = 0, height = 0;
    uint32_t intervals = 0, total_dus = 0;
    size_t tables_cap = 0, scan_cap = 0; // bytes of the LUT blob / of the entropy-coded segment, at most
    bool is422 = true;
    bool covered = true; // every MCU of the image belongs to a restart interval (no truncated last interval, lib.rs:784-785)
};

UploadItem item_of(const ImageData &img)
{
    UploadItem it;
    it.width = img.width;
    it.height = img.height;
    it.intervals = img.metadata.total_restart_intervals;
    it.total_dus = img.total_dus();
    it.tables_cap = table_blob_bytes(img);
    it.scan_cap = img.scan_len;
    it.is422 = is_422(img);
    {
        const Metadata &md = img.metadata;
        const uint32_t mw = md.max_hsample * 8u, mh = md.max_vsample * 8u;
        const uint64_t mcus = (mw && mh) ? uint64_t((img.width + mw - 1) / mw) * ((img.height + mh - 1) / mh) : 0;
        it.covered = uint64_t(md.total_restart_intervals) * md.restart_interval >= mcus;
    }
    return it;
}

// The same from the headers alone (SOF0, DRI; everything up to SOS): no walk over the entropy-coded data.
// false: anything unusual -- the caller then parses first and lays out afterwards.
bool peek_item(const uint8_t *j, size_t len, unsigned flags, UploadItem &it)
{
    if (len < 4 || j[0] != 0xff || j[1] != 0xd8)
        return false;
    size_t p = 2;
    uint32_t ri = 0, w = 0, h = 0, hs = 0, vs = 0, comps = 0;
    bool sof = false;
    while (p + 4 <= len) {
        if (j[p] != 0xff)
            return false;
        const uint8_t m = j[p + 1];
        if (m == 0xff) { // fill byte
            p++;
            continue;
        }
        if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7))
            return false;
        const size_t seg = (size_t(j[p + 2]) << 8) | j[p + 3];
        if (seg < 2 || p + 2 + seg > len)
            return false;
        const uint8_t *d = j + p + 4;
        if (m == 0xc0) {
            if (sof || seg < 8 + 3)
                return false;
            sof = true;
            h = (uint32_t(d[1]) << 8) | d[2];
            w = (uint32_t(d[3]) << 8) | d[4];
            comps = d[5];
            if (comps != 3 || seg < 8 + 3 * comps)
                return false;
            hs = d[7] >> 4;
            vs = d[7] & 15;
        } else if (m == 0xdd) {
            if (seg < 4)
                return false;
            ri = (uint32_t(d[0]) << 8) | d[1];
        } else if (m == 0xda) {
            if (!sof || w == 0 || h == 0)
                return false;
            const bool ok422 = hs == 2 && vs == 1;
            if (!ok422 && !(flags & COMPEG_PARSE_ANY_LUMA_SAMPLING))
                return false;
            if (hs < 1 || hs > 2 || vs < 1 || vs > 2)
                return false;
            const uint32_t wm = ((w + 7) / 8 + hs - 1) / hs, hm = ((h + 7) / 8 + vs - 1) / vs;
            const uint64_t mcus = uint64_t(wm) * hm;
            const uint64_t r = ri ? ri : mcus;
            if (r == 0 || mcus / r > kMaxRestartIntervals)
                return false;
            it.width = w;
            it.height = h;
            it.intervals = uint32_t(mcus / r);
            it.total_dus = uint32_t(it.intervals * r * (hs * vs + 2));
            it.is422 = ok422;
            it.covered = uint64_t(it.intervals) * r >= mcus;
            // five LUT sections at their largest: L1, an L2 LUT of 32767 entries, the direct tables
            it.tables_cap = COMPEG_HUFFMAN_L1_BYTES + 65536 + 2 * kFastEntries * 2 + 2 * kDcFastEntries * 2;
            it.scan_cap = len - (p + 2 + seg);
            return true;
        }
        p += 2 + seg;
    }
    return false;
}

} // namespace

Status compeg_batch::upload(const ImageData *const *images, size_t n, int threads)
{
    (void)finish_upload();
    std::vector<UploadItem> items(n);
    for (size_t i = 0; i < n; i++)
        items[i] = item_of(*images[i]);
    if (preprocess_mode != 0) {
        if (!use_fused_pipeline())
            return Status::error(COMPEG_E_INVALID_ARG, "device preprocessing needs the fused pipeline");
        std::vector<FeedSource> src(n);
        for (size_t i = 0; i < n; i++)
            src[i] = FeedSource{images[i]->scan_data(), images[i]->scan_len, images[i]->metadata.total_restart_intervals};
        return upload_device_scan(n, threads, src.data(), images, nullptr, nullptr);
    }
    return upload_host(n, threads, items.data(), [&](size_t i, Status &) { return images[i]; });
}

// generic_layout, uniform, largest output: what decode() wants to know about the images of the batch
void compeg_batch::note_batch_properties(const ImageData *const *images, size_t n)
{
    generic_layout = false;
    layout_h = n ? images[0]->metadata.components[0].hsample : 0;
    layout_v = n ? images[0]->metadata.components[0].vsample : 0;
    layout_pairs = true;
    one_mcu_intervals = n > 0;
    min_restart_interval = n ? 0xffffffffu : 0u;
    max_restart_interval = 0u;
    total_waves = 0;
    uint64_t scan_bytes = 0, mcus = 0;
    max_out_w = max_out_h = 0;
    // frames of one stream: the same number of restart intervals and byte-identical LUTs in every image
    // (the fused kernel's workgroups may then span image boundaries)
    uniform = n > 0;
    for (size_t i = 0; i < n; i++) {
        const ImageData &img = *images[i], &first = *images[0];
        generic_layout = generic_layout || !is_422(img);
        if (img.metadata.components[0].hsample != layout_h || img.metadata.components[0].vsample != layout_v)
            layout_h = layout_v = 0; // (mixed samplings in one batch)
        layout_pairs = layout_pairs && img.metadata.restart_interval >= 2u;
        one_mcu_intervals = one_mcu_intervals && img.metadata.restart_interval == 1u;
        min_restart_interval = std::min(min_restart_interval, uint32_t(img.metadata.restart_interval));
        max_restart_interval = std::max(max_restart_interval, uint32_t(img.metadata.restart_interval));
        total_waves += (img.metadata.total_restart_intervals + kWave - 1) / kWave;
        scan_bytes += img.scan_len;
        mcus += uint64_t(img.metadata.total_restart_intervals) * std::max(1u, uint32_t(img.metadata.restart_interval));
        max_out_w = std::max(max_out_w, img.width);
        max_out_h = std::max(max_out_h, img.height);
        uniform = uniform && img.metadata.total_restart_intervals == first.metadata.total_restart_intervals &&
                  img.l2 == first.l2 && img.ac_fast == first.ac_fast && img.dc_fast == first.dc_fast &&
                  memcmp(img.l1, first.l1, sizeof img.l1) == 0;
    }
    stream_mcu_words = uint32_t((scan_bytes / 4u + std::max<uint64_t>(mcus, 1u) - 1u) / std::max<uint64_t>(mcus, 1u));
}

// Host path of an upload.  image_of(i, status) hands out image i -- parsing it first when the batch is fed with
// JPEG bytes -- on whichever worker thread takes the image; items: what the layout may assume about it.
Status compeg_batch::upload_host(size_t n, int threads, const void *items_, const ImageSource &image_of)
{
    const UploadItem *items = static_cast<const UploadItem *>(items_);
    CG_HIP(hipSetDevice(gpu->device));
    CG_HIP(hipStreamSynchronize(last_stream));
    count = 0;
    if (n > 65535)
        return Status::error(COMPEG_E_INVALID_ARG, "at most 65535 images per batch");
    static const bool trace_on = getenv("COMPEG_TRACE_BATCH") != nullptr; // one stderr line per upload: host time of its steps
    const auto t_up0 = std::chrono::steady_clock::now();
    auto t_last = t_up0;
    std::string trace_line;
    auto mark = [&](const char *what) {
        if (!trace_on)
            return;
        const auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof buf, " %s=%.2f", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        trace_line += buf;
        t_last = now;
    };

    // One input arena, per image [L1][L2 + direct tables][start positions][words], 256-byte aligned, laid
    // out for the worst case of every scan (scan.rs:38-44) so that nothing depends on another image's size:
    // the threads preprocess straight into the pinned staging copy of the arena and send every image off as
    // soon as it is done -- the transfers run under the preprocessing of the images that follow.
    descs.assign(n, ImageDesc{});
    out_offset.assign(n, 0);
    std::vector<size_t> in_off(n);
    size_t in_total = 0, ac_total = 0, dc_total = 0, out_total = 0;
    bool any_generic = false;
    for (size_t i = 0; i < n; i++) {
        const UploadItem &it = items[i];
        in_off[i] = in_total;
        in_total += align_up(it.tables_cap + ScanBuffer::start_slots(it.intervals) * 4 +
                                 ScanBuffer::output_capacity(it.scan_cap) + 16, 256);
        out_offset[i] = out_total;
        out_total += align_up(output_bytes(it.width, it.height), 256);
        ac_total += size_t(it.total_dus) * kRetained * 2;
        dc_total += size_t(it.total_dus) * 4;
        any_generic = any_generic || !it.is422;
    }

    const bool fused = use_fused_pipeline() && !any_generic;
    CG_TRY(inputs.reserve(in_total + kStreamSlackBytes));
    const bool stamps = lab_env("COMPEG_STAMPS") != nullptr; // diagnostic builds park cycle stamps in dc
    if (!fused) { // the fused kernel keeps coefficients on chip
        CG_TRY(ac.reserve(ac_total + 256));
        CG_TRY(dc.reserve(dc_total + 256));
    } else if (stamps) {
        CG_TRY(dc.reserve(dc_total + 256));
    }
    bool fresh_out = false;
    CG_TRY(out.reserve(out_total + 256, &fresh_out));
    hipStream_t st = gpu->stream;
    // Texels no MCU covers (a truncated last restart interval) read 0 in the reference's fresh texture: the output is
    // cleared when it is new, and whenever some image of the upload has such texels.  Where every MCU of every image
    // is decoded -- frame after frame of a stream -- each upload writes all of its texels and nothing stale can show,
    // whatever the uploads before it looked like.
    bool all_covered = true;
    for (size_t i = 0; i < n; i++)
        all_covered = all_covered && items[i].covered;
    if (fresh_out || !all_covered)
        CG_HIP(hipMemsetAsync(out.ptr, 0, out.capacity, st)); // (the card does this while the host preprocesses)
    CG_HIP(shared_copy_streams(gpu->device, copy_streams));
    CG_TRY(dev_descs.reserve(n * sizeof(ImageDesc) + 256));
    CG_TRY(stage.reserve(in_total + 256));
    uint8_t *hs = static_cast<uint8_t *>(stage.ptr);
    uint8_t *di = static_cast<uint8_t *>(inputs.ptr);
    std::vector<size_t> ac_at(n), dc_at(n);
    for (size_t i = 0, a = 0, c = 0; i < n; i++) {
        ac_at[i] = a;
        dc_at[i] = c;
        a += size_t(items[i].total_dus) * kRetained * 2;
        c += size_t(items[i].total_dus) * 4;
    }
    mark("layout+reserve");

    std::vector<Status> results(n);
    std::vector<const ImageData *> got(n, nullptr);
    std::vector<uint32_t> spans(n, 0);
    std::vector<CoopSpans> group_spans(n);
    std::vector<uint64_t> alg(n, 0);
    std::atomic<int> hip_error{int(hipSuccess)};
    unsigned nthreads = threads > 0 ? unsigned(threads) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    nthreads = unsigned(std::min<size_t>(nthreads, std::max<size_t>(n, 1)));
    const int device = gpu->device;
    std::atomic<uint64_t> thread_us[3] = {{0}, {0}, {0}}; // COMPEG_TRACE_BATCH: thread time in parse / preprocess / copy call
    std::atomic<size_t> next_image{0}; // (images are handed out as threads come free: they do not take equally long)
    auto work = [&](unsigned t) {
        if (hipSetDevice(device) != hipSuccess) {
            hip_error = int(hipErrorInvalidDevice);
            return;
        }
        for (size_t i; (i = next_image.fetch_add(1)) < n;) {
            const auto tw0 = std::chrono::steady_clock::now();
            const ImageData *imgp = image_of(i, results[i]);
            if (trace_on)
                thread_us[0] += uint64_t(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw0).count());
            if (!imgp)
                continue;
            const ImageData &img = *imgp;
            const UploadItem &it = items[i];
            // (an image that does not fit what the layout assumed about it: the caller lays out again, exactly)
            if (img.width != it.width || img.height != it.height || img.metadata.total_restart_intervals != it.intervals ||
                table_blob_bytes(img) > it.tables_cap || img.scan_len > it.scan_cap || is_422(img) != it.is422 ||
                img.total_dus() > it.total_dus) {
                results[i] = Status::error(COMPEG_E_INVALID_ARG, kLayoutBoundExceeded);
                continue;
            }
            got[i] = imgp;
            ImageDesc &d = descs[i];
            fill_desc(img, d);
            size_t o = in_off[i];
            write_tables(hs + o, img);
            d.l1 = reinterpret_cast<const uint16_t *>(di + o);
            d.l2 = reinterpret_cast<const uint16_t *>(di + o + COMPEG_HUFFMAN_L1_BYTES);
            o += table_blob_bytes(img);
            const size_t slots = ScanBuffer::start_slots(img.metadata.total_restart_intervals);
            uint32_t *starts_at = reinterpret_cast<uint32_t *>(hs + o);
            uint8_t *words_at = hs + o + slots * 4;
            size_t nwords = 0, nstarts = 0;
            const auto tw1 = std::chrono::steady_clock::now();
            results[i] = ScanBuffer::process_to(img.scan_data(), img.scan_len, img.metadata.total_restart_intervals,
                                                words_at, starts_at, nwords, nstarts);
            if (trace_on)
                thread_us[1] += uint64_t(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw1).count());
            if (!results[i].ok() && results[i].code != COMPEG_E_COUNT_MISMATCH)
                continue;
            d.starts = reinterpret_cast<const uint32_t *>(di + o);
            d.nstarts = uint32_t(nstarts);
            d.words = reinterpret_cast<const uint32_t *>(di + o + slots * 4);
            d.nwords = uint32_t(nwords);
            d.ac = fused ? nullptr : reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ac.ptr) + ac_at[i]);
            d.dc = (fused && !stamps) ? nullptr : reinterpret_cast<int32_t *>(static_cast<uint8_t *>(dc.ptr) + dc_at[i]);
            d.out = static_cast<uint8_t *>(out.ptr) + out_offset[i];
            d.out_w = img.width;
            d.out_h = img.height;
            d.out_pitch = output_pitch(img.width);
            d.out_alloc_h = output_rows(img.height);
            spans[i] = max_wave_span(starts_at, nstarts, nwords, img.metadata.total_restart_intervals);
            if (d.coop_ok)
                group_spans[i] = coop_spans_exact(starts_at, nstarts, nwords, img.metadata.total_restart_intervals,
                                                  img.metadata.restart_interval);
            alg[i] = 4ull * nwords + 4ull * img.metadata.total_restart_intervals + COMPEG_METADATA_BYTES +
                     COMPEG_HUFFMAN_L1_BYTES + img.l2.size() * 2 + 4ull * img.width * img.height;
            // this image's part of the arena, as far as it is used
            const size_t used = table_blob_bytes(img) + slots * 4 + nwords * 4;
            const auto tw2 = std::chrono::steady_clock::now();
            const hipError_t e = hipMemcpyAsync(di + in_off[i], hs + in_off[i], used, hipMemcpyHostToDevice,
                                                copy_streams[t % copy_streams.size()]);
            if (trace_on)
                thread_us[2] += uint64_t(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw2).count());
            if (e != hipSuccess)
                hip_error = int(e);
        }
    };
    run_on_threads(nthreads, work);
    mark("parse+preprocess+issue");
    // (before any way out: copies in flight still read the pinned arena, which the next upload rewrites)
    hipError_t copies = hipSuccess;
    for (hipStream_t c : copy_streams) {
        const hipError_t e = hipStreamSynchronize(c);
        copies = copies == hipSuccess ? e : copies;
    }
    CG_HIP(hipError_t(hip_error.load()));
    CG_HIP(copies);
    mark("copies_done");
    max_intervals = max_dus = max_l2 = 0;
    algorithmic_bytes = pixels = 0;
    max_span = 0;
    for (size_t i = 0; i < n; i++) {
        if (!results[i].ok() && results[i].code != COMPEG_E_COUNT_MISMATCH) {
            (void)hipStreamSynchronize(st);
            return results[i];
        }
        const ImageData &img = *got[i];
        max_intervals = std::max(max_intervals, img.metadata.total_restart_intervals);
        max_dus = std::max(max_dus, img.total_dus());
        max_l2 = std::max<uint32_t>(max_l2, staged_lut_entries(img));
        pixels += uint64_t(img.width) * img.height;
        max_span = std::max(max_span, spans[i]);
        algorithmic_bytes += alg[i];
    }
    note_batch_properties(got.data(), n);
    coop_r = n ? got[0]->metadata.restart_interval : 0;
    coop_spans = CoopSpans{};
    for (size_t i = 0; i < n; i++) {
        if (!descs[i].coop_ok || got[i]->metadata.restart_interval != coop_r)
            coop_r = 0;
        coop_spans_max(coop_spans, group_spans[i]);
    }
    CG_TRY(make_walk_tables(st, n)); // (the batch holds images only once all of this has succeeded: count stays 0 on a failure)
    CG_HIP(hipStreamSynchronize(st)); // descs (pageable) and the staging arena may be reused from here on
    mark("descs");
    if (trace_on)
        fprintf(stderr, "[compeg] batch upload (%zu images, %u threads):%s total=%.2f ms; thread time: parse %.2f preprocess %.2f copy call %.2f ms\n",
                n, nthreads, trace_line.c_str(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count(),
                thread_us[0].load() / 1e3, thread_us[1].load() / 1e3, thread_us[2].load() / 1e3);
    last_stream = st;
    count = n;
    decodes_timed = 0;
    return Status{};
}

// Host-fed use: JPEG bytes in.  ImageData::new (src/lib.rs:597-824) for every image runs on the worker threads,
// in the same pass as its preprocessing: the layout of the batch is made from the files' headers and lengths
// (upper bounds), so that nothing has to wait for all images to be parsed.  A file whose headers are unusual in
// any way takes the plain road: parse everything (in parallel), then upload().
Status compeg_batch::upload_jpegs(const uint8_t *const *jpegs_in, const size_t *lens_in, size_t n, int threads, unsigned flags,
                                  bool begin_only)
{
    (void)finish_upload(); // (an earlier upload left half done: its transfers read the caller's bytes and the arena)
    // (kept in the batch: an upload whose second step comes later -- finish_upload -- still needs them then)
    feed_jpegs.assign(jpegs_in, jpegs_in + n);
    feed_lens.assign(lens_in, lens_in + n);
    feed_flags = flags;
    feed_fresh.clear();
    feed_fresh.resize(n);
    const uint8_t *const *jpegs = feed_jpegs.data();
    const size_t *lens = feed_lens.data();
    std::vector<std::unique_ptr<ImageData>> &fresh = feed_fresh;
    std::vector<UploadItem> items(n);
    bool peeked = preprocess_mode == 0;
    for (size_t i = 0; i < n && peeked; i++)
        peeked = peek_item(jpegs[i], lens[i], flags, items[i]);
    auto parse_one = [this](size_t i, Status &st) -> const ImageData * {
        ImageData *img = nullptr;
        st = ImageData::parse(feed_jpegs[i], feed_lens[i], false, &img, feed_flags);
        feed_fresh[i].reset(img);
        if (!st.ok())
            st = Status::error(st.code, "image " + std::to_string(i) + ": " + st.message);
        return st.ok() ? img : nullptr;
    };
    Status s;
    bool done = false;
    if (preprocess_mode != 0 && use_fused_pipeline()) {
        // Device preprocessing: the files go up as they are, from where they are if that is page-locked memory; the
        // host reads headers only -- the end of each entropy-coded segment is taken from the file's final EOI and
        // checked by the scan kernels, so that no thread walks the bytes -- and does so while the transfers run (their
        // layout needs no more than a peek at SOF0 / DRI).  An image the kernels hand back is parsed again in full.
        bool laid_out = true;
        std::vector<FeedSource> src(n);
        for (size_t i = 0; i < n && laid_out; i++) {
            laid_out = peek_item(jpegs[i], lens[i], flags, items[i]);
            src[i] = FeedSource{jpegs[i], lens[i], items[i].intervals};
        }
        auto parse_headers = [&](size_t i, Status &st) -> const ImageData * {
            ImageData *img = nullptr;
            st = ImageData::parse(jpegs[i], lens[i], false, &img, flags | kParseDeferScanEnd);
            fresh[i].reset(img);
            if (!st.ok())
                st = Status::error(st.code, "image " + std::to_string(i) + ": " + st.message);
            return st.ok() ? img : nullptr;
        };
        if (laid_out) {
            s = upload_device_scan(n, threads, src.data(), nullptr, parse_headers, parse_one, begin_only);
            done = s.ok() || s.message != kLayoutBoundExceeded;
            if (s.ok() && pending_finish)
                return s; // (finish_upload does the rest, and hands the parsed images over)
        }
        if (!done) {
            // (unusual headers: parse first, lay out afterwards)
            std::vector<Status> results(n);
            unsigned nthreads = threads > 0 ? unsigned(threads) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
            nthreads = unsigned(std::min<size_t>(nthreads, std::max<size_t>(n, 1)));
            auto work = [&](unsigned t) {
                for (size_t i = t; i < n; i += nthreads)
                    parse_headers(i, results[i]);
            };
            run_on_threads(nthreads, work);
            std::vector<const ImageData *> ptrs(n);
            for (size_t i = 0; i < n; i++) {
                if (!results[i].ok())
                    return results[i];
                ptrs[i] = fresh[i].get();
                src[i] = FeedSource{ptrs[i]->scan_data(), ptrs[i]->scan_len, ptrs[i]->metadata.total_restart_intervals};
            }
            s = upload_device_scan(n, threads, src.data(), ptrs.data(), nullptr, parse_one);
        }
        parsed.swap(fresh);
        return s;
    }
    if (peeked) {
        s = upload_host(n, threads, items.data(), parse_one);
        done = s.ok() || s.message != kLayoutBoundExceeded;
    }
    if (!done) {
        std::vector<Status> results(n);
        unsigned nthreads = threads > 0 ? unsigned(threads) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        nthreads = unsigned(std::min<size_t>(nthreads, std::max<size_t>(n, 1)));
        auto work = [&](unsigned t) {
            for (size_t i = t; i < n; i += nthreads)
                if (!fresh[i])
                    parse_one(i, results[i]);
        };
        run_on_threads(nthreads, work);
        std::vector<const ImageData *> ptrs(n);
        for (size_t i = 0; i < n; i++) {
            if (!results[i].ok())
                return results[i];
            ptrs[i] = fresh[i].get();
        }
        s = upload(ptrs.data(), n, threads);
    }
    parsed.swap(fresh); // (the previous upload's images go; nothing on the device refers to them)
    return s;
}

// The second step of an upload begun with begin_only (nothing to do after any other upload).
Status compeg_batch::finish_upload()
{
    if (!pending_finish)
        return Status{};
    const std::function<Status()> finish = std::move(pending_finish);
    pending_finish = nullptr;
    Status s = finish();
    parsed.swap(feed_fresh);
    feed_fresh.clear();
    return s;
}

// Device-side preprocessing: raw entropy-coded segments go to HBM as they are
// and the scan kernels (scan_kernels.hip) produce words / start positions in
// the reference layout.
//
// Nothing of a segment is touched on the host when the caller's bytes lie in pinned (page-locked) memory --
// compeg_host_alloc / compeg_host_register, or any hipHostMalloc'ed / registered range: the DMA engines read it where
// it is, like the reference uploads straight from the bytes it borrowed (src/lib.rs:577-595, 397-407).  Pageable
// bytes are copied into the batch's pinned arena by the worker threads first.  The images' tables (25 KB each) are
// staged side by side and go up in one transfer.
//   src      what is sent for image i: its entropy-coded segment (images parsed by the caller: `given`), or the whole
//            file (compeg_batch_upload_jpegs: the transfers start at once -- their layout needs the files' lengths
//            only -- and `parse` reads the headers on the worker threads meanwhile)
//   reparse  images whose segment end was left to the device (ImageData::scan_end_deferred) and turned out to end
//            earlier (another marker inside: the scan kernels' flag bit 1) are parsed again in full through it
Status compeg_batch::upload_device_scan(size_t n, int threads, const FeedSource *src, const ImageData *const *given,
                                        const ImageSource &parse, const ImageSource &reparse, bool defer_finish)
{
    pending_finish = nullptr;
    CG_HIP(hipSetDevice(gpu->device));
    CG_HIP(hipStreamSynchronize(last_stream));
    count = 0;
    if (n > 65535)
        return Status::error(COMPEG_E_INVALID_ARG, "at most 65535 images per batch");
    const auto t_up0 = std::chrono::steady_clock::now();
    auto ms_since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count(); };
    std::vector<const ImageData *> images(n, nullptr);
    if (given)
        images.assign(given, given + n);

    // which bytes the DMA engines can read where they are
    // (a source counts as page-locked when its first AND its last byte are: one that runs past the end of its
    // registration goes the staged road.  alloc[i]: the page-locked allocation the source lies in, where the runtime can
    // say -- two sources with a gap between them are sent as one transfer only inside one allocation)
    std::vector<uint8_t> pinned(n, 0);
    struct Span {
        const uint8_t *base = nullptr, *end = nullptr;
    };
    std::vector<Span> alloc(n);
    bool any_pageable = false;
    auto locked_at = [](const void *p, hipPointerAttribute_t &attr) {
        const bool yes = hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError(); // (a pageable pointer is "invalid value" to the query: not an error here)
        return yes;
    };
    for (size_t i = 0; i < n; i++) {
        if (src[i].len > 0xfffffff0u)
            return Status::error(COMPEG_E_INVALID_ARG, "scan segment too large");
        hipPointerAttribute_t first{}, last{};
        const bool is_pinned = src[i].len != 0 && locked_at(src[i].bytes, first) && locked_at(src[i].bytes + src[i].len - 1, last);
        pinned[i] = is_pinned ? 1 : 0;
        any_pageable = any_pageable || !is_pinned;
        if (is_pinned && first.devicePointer) {
            hipDeviceptr_t base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange(&base, &size, hipDeviceptr_t(first.devicePointer)) == hipSuccess && size) {
                // (as host addresses: the source's offset inside the allocation is the same on both sides)
                const uint8_t *host_base = src[i].bytes - (static_cast<const uint8_t *>(first.devicePointer) - static_cast<const uint8_t *>(base));
                alloc[i] = Span{host_base, host_base + size};
            }
            (void)hipGetLastError();
        }
    }
    // Page-locked sources that follow each other in memory -- the frames of a capture ring, a receive arena -- go up
    // as ONE transfer per run (one copy of 32 MB moves at the link's rate, sixteen copies of 1.6 MB at three quarters
    // of it); the device copy keeps the host's spacing.  Runs end at the quarters of the batch (the scan kernels of a
    // quarter start when it has arrived).
    constexpr size_t kGroups = 4, kCopyStreams = 4, kMaxRunBytes = size_t(48) << 20, kMaxGap = size_t(64) << 10;
    const auto group_of = [&](size_t i) { return i * kGroups / std::max<size_t>(n, 1); };
    struct Run {
        size_t first, last;      // images
        const uint8_t *host;     // first byte sent
        size_t bytes, dev;       // bytes sent; where they go in the arena
    };
    std::vector<Run> runs;
    std::vector<size_t> run_of(n, 0);
    for (size_t i = 0; i < n; i++) {
        bool joins = false;
        if (!runs.empty() && pinned[i] && pinned[i - 1]) {
            const Run &r = runs.back();
            const uint8_t *end = r.host + r.bytes;
            // (a gap is read by the transfer too: none, or both sources inside one page-locked allocation -- two
            // separately registered buffers may have anything, or nothing, between them)
            const bool one_allocation = alloc[i].base && alloc[i].base == alloc[i - 1].base && alloc[i].end == alloc[i - 1].end &&
                                        src[i].bytes + src[i].len <= alloc[i].end && r.host >= alloc[i].base;
            joins = group_of(i) == group_of(r.first) && src[i].bytes >= end && size_t(src[i].bytes - end) <= kMaxGap &&
                    (src[i].bytes == end || one_allocation) && size_t(src[i].bytes - r.host) + src[i].len <= kMaxRunBytes;
        }
        if (joins) {
            runs.back().last = i;
            runs.back().bytes = size_t(src[i].bytes - runs.back().host) + src[i].len;
        } else {
            runs.push_back(Run{i, i, src[i].bytes, src[i].len, 0});
        }
        run_of[i] = runs.size() - 1;
    }

    // Device arena: [results][all images' LUTs][the runs, 64 readable bytes around each] and per image the scan kernels'
    // scratch and output, sized from the length of what is sent (an upper bound of the segment's).  The pinned
    // staging arena: [all images' LUTs][the pageable sources].
    struct Layout {
        size_t sent, tables, tile_state, starts, words; // `sent`: where the image's source bytes lie in the arena
        size_t staged;
        size_t tile_cap, slot_cap;
    };
    std::vector<Layout> lay(n);
    size_t total = 0;
    auto take = [&](size_t bytes) {
        const size_t at = total;
        total += align_up(bytes, 256);
        return at;
    };
    const size_t o_results = take(n * kScanResultBytes);
    // (tables of a file whose headers are not read yet: room for 32 KB -- Annex-K tables take 19 -- or the upload goes
    // the parse-first way)
    constexpr size_t kUnparsedTablesCap = size_t(32) << 10;
    size_t tables_bytes = 0;
    for (size_t i = 0; i < n; i++) {
        lay[i].tables = tables_bytes;
        tables_bytes += align_up(given ? table_blob_bytes(*given[i]) : kUnparsedTablesCap, 256);
    }
    const size_t o_tables = take(tables_bytes);
    size_t staged_total = align_up(tables_bytes, 256);
    for (Run &r : runs)
        r.dev = take(64 + align_up(r.bytes + 64, 16)) + 64;
    for (size_t i = 0; i < n; i++) {
        Layout &L = lay[i];
        const size_t len = src[i].len;
        L.tables += o_tables;
        L.sent = runs[run_of[i]].dev + size_t(src[i].bytes - runs[run_of[i]].host);
        L.tile_cap = scan_tiles(uint32_t(len));
        L.tile_state = take(L.tile_cap * kScanTileStateBytes + 32);
        L.slot_cap = ScanBuffer::start_slots(src[i].intervals);
        L.starts = take(L.slot_cap * 4);
        L.words = take(len + len / 3 + 64);
        L.staged = staged_total; // (used by pageable sources)
        staged_total += align_up(len + 64, 256);
    }
    CG_TRY(scan_arena.reserve(total + kStreamSlackBytes));
    CG_TRY(scan_descs.reserve(n * sizeof(ScanDesc) + 256));
    hipStream_t st = gpu->stream;
    CG_TRY(dev_descs.reserve(n * sizeof(ImageDesc) + 256));
    CG_HIP(shared_copy_streams(gpu->device, copy_streams)); // (kCopyStreams of them)
    uint8_t *da = static_cast<uint8_t *>(scan_arena.ptr);
    CG_HIP(hipMemsetAsync(da + o_results, 0, n * kScanResultBytes, st));
    CG_TRY(stage.reserve((any_pageable ? staged_total : align_up(tables_bytes, 256)) + 256));
    uint8_t *hs = static_cast<uint8_t *>(stage.ptr);
    const double t_layout = ms_since();

    // The worker threads read the headers (files), stage the tables and the pageable sources; this thread sends the
    // runs off in order, round robin over the copy streams, and starts the scan kernels of a quarter of the images
    // as soon as that quarter has arrived and its headers are read: they run under the transfers of the next one.
    std::vector<std::atomic<uint8_t>> staged_ready(n), parsed_ready(n);
    for (size_t i = 0; i < n; i++) {
        staged_ready[i].store(pinned[i], std::memory_order_relaxed);
        parsed_ready[i].store(given ? 1 : 0, std::memory_order_relaxed);
    }
    unsigned nthreads = threads > 0 ? unsigned(threads) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    nthreads = unsigned(std::min<size_t>(nthreads, std::max<size_t>(n, 1)));
    std::atomic<size_t> next_image{0};
    std::atomic<int> hip_error{int(hipSuccess)};
    std::atomic<bool> failed{false};
    std::vector<Status> results(n);
    const int device = gpu->device;
    auto prepare_images = [&] {
        for (;;) {
            const size_t i = next_image.fetch_add(1, std::memory_order_relaxed);
            if (i >= n)
                return;
            if (!given) {
                images[i] = parse(i, results[i]);
                if (!images[i] || table_blob_bytes(*images[i]) > kUnparsedTablesCap) {
                    if (images[i] && results[i].ok())
                        results[i] = Status::error(COMPEG_E_INVALID_ARG, kLayoutBoundExceeded);
                    images[i] = nullptr;
                    failed.store(true, std::memory_order_relaxed);
                }
                parsed_ready[i].store(1, std::memory_order_release);
            }
            if (!pinned[i]) {
                memcpy(hs + lay[i].staged, src[i].bytes, src[i].len);
                staged_ready[i].store(1, std::memory_order_release);
            }
            if (images[i])
                write_tables(hs + (lay[i].tables - o_tables), *images[i]);
        }
    };
    std::vector<ScanDesc> sd(n);
    std::vector<hipEvent_t> arrived;
    auto send_runs = [&]() -> hipError_t {
        hipError_t err = hipSuccess;
        size_t sent = 0, next_run = 0;
        for (size_t g = 0; g < kGroups && err == hipSuccess; g++) {
            size_t upto = sent;
            while (upto < n && group_of(upto) == g)
                upto++;
            for (; next_run < runs.size() && runs[next_run].first < upto && err == hipSuccess; next_run++) {
                const Run &r = runs[next_run];
                for (size_t i = r.first; i <= r.last; i++)
                    while (!staged_ready[i].load(std::memory_order_acquire))
                        std::this_thread::yield(); // (every worker stages and parses, whatever its device calls say: see `work`)
                if (r.bytes)
                    err = hipMemcpyAsync(da + r.dev, pinned[r.first] ? r.host : hs + lay[r.first].staged, r.bytes,
                                         hipMemcpyHostToDevice, copy_streams[next_run % kCopyStreams]);
            }
            for (size_t k = 0; k < kCopyStreams && err == hipSuccess; k++) {
                hipEvent_t e = nullptr;
                err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
                if (err != hipSuccess)
                    break;
                arrived.push_back(e);
                err = hipEventRecord(e, copy_streams[k]);
                if (err == hipSuccess)
                    err = hipStreamWaitEvent(st, e, 0);
            }
            // the quarter's headers: where each segment lies inside what was sent
            uint32_t group_tiles = 0;
            for (size_t i = sent; i < upto; i++) {
                while (!parsed_ready[i].load(std::memory_order_acquire))
                    std::this_thread::yield();
                if (!images[i])
                    return err; // (a rejected image: the caller sees `failed`)
                const ImageData &img = *images[i];
                const Layout &L = lay[i];
                ScanDesc &d = sd[i];
                d.raw = da + L.sent + (given ? 0 : img.scan_offset);
                d.len = uint32_t(img.scan_len);
                d.ntiles = scan_tiles(d.len);
                d.slots = uint32_t(ScanBuffer::start_slots(img.metadata.total_restart_intervals));
                d.expected = img.metadata.total_restart_intervals;
                d.tile_state = reinterpret_cast<uint32_t *>(da + L.tile_state);
                d.starts_out = reinterpret_cast<uint32_t *>(da + L.starts);
                d.words_out = da + L.words;
                d.result = reinterpret_cast<uint32_t *>(da + o_results + i * kScanResultBytes);
                if (d.slots > L.slot_cap || d.ntiles > L.tile_cap || d.len > src[i].len) {
                    // (headers that say something else than the peek at them did: the parse-first way)
                    results[i] = Status::error(COMPEG_E_INVALID_ARG, kLayoutBoundExceeded);
                    failed.store(true, std::memory_order_relaxed);
                    return err;
                }
                group_tiles = std::max(group_tiles, d.ntiles);
            }
            if (err == hipSuccess && upto > sent) {
                err = hipMemcpyAsync(static_cast<ScanDesc *>(scan_descs.ptr) + sent, sd.data() + sent, (upto - sent) * sizeof(ScanDesc),
                                     hipMemcpyHostToDevice, st);
                if (err == hipSuccess)
                    err = launch_scan(static_cast<const ScanDesc *>(scan_descs.ptr) + sent, uint32_t(upto - sent), group_tiles, st, true);
            }
            sent = upto;
        }
        return err;
    };
    hipError_t sent_status = hipSuccess;
    {
        auto work = [&](unsigned t) {
            // (a thread that cannot make the device current still does its share of the host-only work -- the thread
            // that sends waits for every image to be staged and parsed --, and the sender gives up)
            const bool device_ok = hipSetDevice(device) == hipSuccess;
            if (!device_ok)
                hip_error = int(hipErrorInvalidDevice);
            if (t == 0 && nthreads > 1) {
                if (device_ok)
                    sent_status = send_runs();
                else
                    prepare_images();
            } else {
                prepare_images();
            }
        };
        run_on_threads(nthreads, work);
        if (nthreads <= 1)
            sent_status = send_runs();
        for (hipEvent_t e : arrived)
            (void)hipEventDestroy(e); // (destroyed once it has been reached: HIP defers that)
    }
    // (before any way out: transfers in flight still read the pinned arena / the caller's bytes)
    if (sent_status != hipSuccess || hip_error.load() != int(hipSuccess) || failed.load()) {
        for (hipStream_t c : copy_streams)
            (void)hipStreamSynchronize(c);
        (void)hipStreamSynchronize(st);
        for (size_t i = 0; i < n; i++)
            if (!results[i].ok())
                return results[i];
        CG_HIP(hipError_t(hip_error.load()));
        CG_HIP(sent_status);
        return Status::error(COMPEG_E_INVALID_ARG, "batch upload failed");
    }
    // What is left -- waiting for the transfers and the scan kernels, the decode descriptors from their results -- is
    // the upload's second step: at once, or (compeg_batch_upload_jpegs_begin / compeg_batch_upload_end) when the caller
    // comes back for it, with the next batch's transfers queued behind these in the meantime.
    struct Rest {
        size_t n;
        std::vector<Layout> lay;
        std::vector<ScanDesc> sd;
        std::vector<const ImageData *> images;
        std::vector<uint8_t> pinned;
        size_t runs, o_results, o_tables, tables_bytes;
        unsigned nthreads;
        double t_layout, t_issued;
        std::chrono::steady_clock::time_point t_up0;
        ImageSource reparse;
    };
    auto rest = std::make_shared<Rest>();
    rest->n = n;
    rest->lay = std::move(lay);
    rest->sd = std::move(sd);
    rest->images = std::move(images);
    rest->pinned = std::move(pinned);
    rest->runs = runs.size();
    rest->o_results = o_results;
    rest->o_tables = o_tables;
    rest->tables_bytes = tables_bytes;
    rest->nthreads = nthreads;
    rest->t_layout = t_layout;
    rest->t_issued = ms_since();
    rest->t_up0 = t_up0;
    rest->reparse = reparse;
    auto finish = [this, rest]() -> Status {
        const size_t n = rest->n;
        std::vector<Layout> &lay = rest->lay;
        std::vector<ScanDesc> &sd = rest->sd;
        std::vector<const ImageData *> &images = rest->images;
        const std::vector<uint8_t> &pinned = rest->pinned;
        const size_t o_results = rest->o_results, o_tables = rest->o_tables, tables_bytes = rest->tables_bytes;
        const unsigned nthreads = rest->nthreads;
        const double t_layout = rest->t_layout, t_issued = rest->t_issued;
        const ImageSource &reparse = rest->reparse;
        const auto t_up0 = rest->t_up0;
        auto ms_since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count(); };
        static const bool trace_on = getenv("COMPEG_TRACE_BATCH") != nullptr;
        CG_HIP(hipSetDevice(gpu->device));
        hipStream_t st = gpu->stream;
        uint8_t *da = static_cast<uint8_t *>(scan_arena.ptr), *hs = static_cast<uint8_t *>(stage.ptr);
        struct RunsSize { size_t n; size_t size() const { return n; } } runs{rest->runs};
        CG_HIP(hipMemcpyAsync(da + o_tables, hs, tables_bytes, hipMemcpyHostToDevice, st));
        // all results in one read-back: sizes and window spans for the decode descriptors
        std::vector<uint32_t> res(n * (kScanResultBytes / 4));
        CG_HIP(hipMemcpyAsync(res.data(), da + o_results, n * kScanResultBytes, hipMemcpyDeviceToHost, st));

        // (while the last quarter arrives: everything about the batch that needs the headers only)
        max_tiles = 0;
        max_intervals = max_dus = max_l2 = max_span = 0;
        algorithmic_bytes = pixels = 0;
        size_t out_total = 0;
        bool all_covered = true;
        for (size_t i = 0; i < n; i++) {
            const ImageData &img = *images[i];
            out_total += align_up(output_bytes(img.width, img.height), 256);
            max_tiles = std::max(max_tiles, sd[i].ntiles);
            max_intervals = std::max(max_intervals, img.metadata.total_restart_intervals);
            max_dus = std::max(max_dus, img.total_dus());
            max_l2 = std::max<uint32_t>(max_l2, staged_lut_entries(img));
            pixels += uint64_t(img.width) * img.height;
            all_covered = all_covered && item_of(img).covered;
        }
        note_batch_properties(images.data(), n);
        // (extension layouts: the coefficient records of their kernels, as in upload_host)
        size_t ac_total = 0, dc_total = 0;
        for (size_t i = 0; i < n && generic_layout; i++) {
            ac_total += size_t(images[i]->total_dus()) * kRetained * 2;
            dc_total += size_t(images[i]->total_dus()) * 4;
        }
        if (generic_layout) {
            CG_TRY(ac.reserve(ac_total + 256));
            CG_TRY(dc.reserve(dc_total + 256));
        }
        bool fresh_out = false;
        CG_TRY(out.reserve(out_total + 256, &fresh_out));
        // (texels no MCU covers read 0: see upload_host)
        if (fresh_out || !all_covered)
            CG_HIP(hipMemsetAsync(out.ptr, 0, out.capacity, st));
        CG_HIP(hipStreamSynchronize(st));
        const double t_arrived = ms_since();
        last_stream = st;

        descs.assign(n, ImageDesc{});
        out_offset.assign(n, 0);
        host_fallbacks = 0;
        size_t out_at = 0, ac_at = 0, dc_at = 0;
        for (size_t i = 0; i < n; i++) {
            const Layout &L = lay[i];
            const uint32_t *r = &res[i * (kScanResultBytes / 4)];
            uint32_t nwords = r[2], nstarts = std::min(r[0], sd[i].slots), span = r[4];
            const bool ends_earlier = (r[3] & 2u) != 0u && images[i]->scan_end_deferred;
            if (ends_earlier) {
                // another marker inside what was taken for the segment: the reference's parser ends it there
                // (src/file.rs:163-201) -- this image once more through the whole front-end
                Status ps;
                const ImageData *again = reparse ? reparse(i, ps) : nullptr;
                if (!again)
                    return ps.ok() ? Status::error(COMPEG_E_MALFORMED, "entropy-coded segment ends early") : ps;
                images[i] = again;
            }
            const ImageData &img = *images[i];
            const uint32_t expected = img.metadata.total_restart_intervals;
            if ((r[3] & 1u) || ends_earlier) {
                // pathological FF run (or see above): the host preprocessor (same output format) takes this image
                ScanBuffer sb;
                Status s = sb.process(img.scan_data(), img.scan_len, expected);
                if (!s.ok() && s.code != COMPEG_E_COUNT_MISMATCH)
                    return s;
                nwords = uint32_t(sb.nwords());
                nstarts = uint32_t(sb.nstarts());
                if (nwords)
                    CG_HIP(hipMemcpy(da + L.words, sb.words(), size_t(nwords) * 4, hipMemcpyHostToDevice));
                if (nstarts)
                    CG_HIP(hipMemcpy(da + L.starts, sb.starts(), size_t(nstarts) * 4, hipMemcpyHostToDevice));
                span = max_wave_span(sb.starts(), nstarts, nwords, expected);
                host_fallbacks++;
            }
            ImageDesc &d = descs[i];
            fill_desc(img, d);
            d.l1 = reinterpret_cast<const uint16_t *>(da + L.tables);
            d.l2 = reinterpret_cast<const uint16_t *>(da + L.tables + COMPEG_HUFFMAN_L1_BYTES);
            d.words = reinterpret_cast<const uint32_t *>(da + L.words);
            d.starts = reinterpret_cast<const uint32_t *>(da + L.starts);
            d.nwords = nwords;
            d.nstarts = nstarts;
            d.ac = generic_layout ? reinterpret_cast<int16_t *>(static_cast<uint8_t *>(ac.ptr) + ac_at) : nullptr;
            d.dc = generic_layout ? reinterpret_cast<int32_t *>(static_cast<uint8_t *>(dc.ptr) + dc_at) : nullptr;
            ac_at += size_t(img.total_dus()) * kRetained * 2;
            dc_at += size_t(img.total_dus()) * 4;
            out_offset[i] = out_at;
            d.out = static_cast<uint8_t *>(out.ptr) + out_at;
            out_at += align_up(output_bytes(img.width, img.height), 256);
            d.out_w = img.width;
            d.out_h = img.height;
            d.out_pitch = output_pitch(img.width);
            d.out_alloc_h = output_rows(img.height);
            max_span = std::max(max_span, span);
            if (i == 0) {
                coop_r = img.metadata.restart_interval;
                coop_spans = CoopSpans{};
            }
            if (!d.coop_ok || img.metadata.restart_interval != coop_r) {
                coop_r = 0;
            } else {
                // (the kernels report the span of 64 intervals only: four times the average for a wave's group;
                // a group that is longer than that still decodes, its intervals one lane each)
                coop_spans_max(coop_spans, coop_spans_estimate(span, coop_r, true));
            }
            algorithmic_bytes += 4ull * nwords + 4ull * expected + COMPEG_METADATA_BYTES +
                                 COMPEG_HUFFMAN_L1_BYTES + img.l2.size() * 2 + 4ull * img.width * img.height;
        }
        CG_TRY(make_walk_tables(gpu->stream, n));
        CG_HIP(hipStreamSynchronize(gpu->stream));
        count = n;
        decodes_timed = 0;
        if (trace_on) {
            size_t direct = 0;
            for (size_t i = 0; i < n; i++)
                direct += pinned[i];
            fprintf(stderr, "[compeg] batch upload, device scan (%zu images, %zu read where they are in %zu transfers, %u threads): layout=%.2f issued=%.2f arrived+scanned=%.2f total=%.2f ms\n",
                    n, direct, runs.size(), nthreads, t_layout, t_issued, t_arrived, ms_since());
        }
        return Status{};
    };
    if (defer_finish) {
        pending_finish = finish;
        return Status{};
    }
    return finish();
}

// A launch both the cooperative kernel and the walk + lane-per-MCU route can take: which?  Measured (tools/walk_probe.py
// with COMPEG_WALK=0 COMPEG_COOP=1 / COMPEG_WALK=1, profiles/r04/coop_vs_walk.txt; us per launch, cooperative / route):
//  * up to 7 MCUs an interval the cooperative kernel, a second round of teams included (two 4K frames DRI = 4: 63 / 70);
//  * up to 40 MCUs (its teams walk a lane per interval, like the route's walk): while all teams are resident at once
//    -- as many to a CU as its LDS holds of their windows -- it is ahead (eight 960x720 DRI = 10 frames 60 / 91; one
//    960x720 DRI = 30 90 / 177); with a team more than that the route is (one 4K frame DRI = 10, 1080 teams for 1024
//    places: 97 / 85; two 4K frames DRI = 12 158 / 102, DRI = 20 209 / 136, DRI = 30 276 / 182);
//  * beyond 40 MCUs its walks are speculative and twice as fast as a lane per interval, whatever the interval's length
//    (one 4K frame DRI = 60 197 / 309, DRI = 128 298 / 610, an interval per MCU row -- 240 MCUs -- 528 / 1084; two of
//    them 531 / 1102), and two or three rounds of teams are no worse than the route (eight 1080p frames DRI = 60 331 /
//    319, DRI = 120 475 / 586; sixteen 960x720 frames DRI = 60 263 / 323);
//    (Round 4 found two things that had made its long intervals look worse, both fixed in coop_body.h: a walk's room behind
//    its interval's end -- kCoopEndSlack --, and the end of an image's last interval where the scan goes on behind it: one 4K
//    frame with an interval per MCU row was 1788 us, one 960x720 frame with DRI = 240 1214 -- now 322 / 922.)
namespace compeg {
bool coop_preferred(const CoopPlan &cp, uint32_t max_intervals, uint32_t images, uint32_t restart_interval)
{
    if (!cp.usable)
        return false;
    if (restart_interval < 8u)
        return true;
    const uint64_t teams = uint64_t((max_intervals + cp.intervals_per_wave - 1) / cp.intervals_per_wave) * images;
    const uint64_t places = std::max(1u, cp.places);
    return restart_interval <= kCoopLeanMaxRestart ? teams <= places : teams * 2u <= places * 5u;
}
} // namespace compeg

// The walk + lane-per-MCU route (kernels_body.h) for a batch whose launches are `step` images (the last one `smallest`)?
// Where a lane per restart interval leaves the chip's SIMDs a wave each at most (1024 waves of 64 intervals), its time
// is the length of an interval -- about 20 us per MCU of it, whatever the launch's size.  The route's walk takes a
// quarter of that (5.5 us per MCU of an interval: the walk tables, two symbols a step, no pixels), its second kernel
// decodes 5300 MCUs per us with the whole chip, and the two launches cost about 40 us of prologues between them.
// Measured (walk_probe.py, us per launch, this route / the others): 256 x 960x720 with an interval per MCU row 610 /
// 1249, 64 x 960x720 DRI = 30 410 / 596, 16 x 4K with an interval per MCU row 1292 / 4517, 40 x 1000x990 DRI = 7 185 /
// 178; four 4K frames DRI = 4 99 / 80; and with more than a wave per SIMD 1024 x 960x720 (1440 waves) 1561 / 1506, 256 x
// DRI = 10 (2160 waves) 397 / 351.
bool use_mcu_route(const compeg_batch &b, uint32_t step, uint32_t smallest)
{
    static const int forced = [] {
        const char *e = lab_env("COMPEG_WALK"); // experiment knob: 0 / 1
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0)
        return forced != 0;
    if (b.one_mcu_intervals || b.min_restart_interval < 2u)
        return false; // (a lane per interval is a lane per MCU already)
    // The cooperative kernel's launches: coop_preferred.  (Launches of different sizes -- a chunked decode with a smaller
    // last launch -- keep the cooperative kernel: the records are planned for the whole batch.)
    if (b.coop_r && use_coop_kernel(b.max_intervals, smallest, b.coop_r)) {
        if (step != smallest)
            return false;
        if (coop_preferred(plan_coop(b.max_intervals, step, b.coop_r, b.max_l2, b.coop_spans), b.max_intervals, step, b.coop_r))
            return false;
    }
    const uint64_t waves = uint64_t((b.max_intervals + kWave - 1) / kWave) * step;
    if (waves > 1024u)
        return false;
    // (dense streams: the walk's rows have to hold a few MCUs of every lane -- plan_walk)
    if (b.stream_mcu_words > 24u)
        return false;
    const double mcus = double(b.max_intervals) * b.max_restart_interval * step;
    return 14.5 * b.min_restart_interval - 40.0 > mcus / 5300.0;
}

// Uploads the descriptors; in front of that, gives every image its walk tables if the cooperative kernel may
// decode this batch (uniform batches share one set), and makes them behind the upload.
Status compeg_batch::make_walk_tables(hipStream_t stream, size_t n)
{
    // (whenever some launch of decode() may take the cooperative kernel: launches are `chunk` images, the last one
    // n % chunk -- the smallest of them decides, the kernel takes the small launches)
    const size_t step = chunk ? std::min<size_t>(chunk, n) : n;
    const size_t smallest = step && n % step ? n % step : step;
    // The walk + lane-per-MCU route for this batch's launches?  (decided here, where the records' buffers are made)
    mcu_route = n > 0 && !generic_layout && use_fused_pipeline() && use_mcu_route(*this, uint32_t(step), uint32_t(smallest));
    for (size_t i = 0; i < n && mcu_route; i++)
        mcu_route = descs[i].mcu_ok != 0;
    // (the walk goes through the walk tables too)
    const bool want = n > 0 && !generic_layout && use_fused_pipeline() &&
                      (mcu_route || (coop_r != 0 && use_coop_kernel(max_intervals, uint32_t(smallest), coop_r))) && !lab_env("COMPEG_NO_WALK_TABLES");
    // one set for all: the same tables (uniform) used by the same components
    bool shared = uniform;
    for (size_t i = 1; i < n && shared; i++)
        shared = memcmp(descs[i].fast_table, descs[0].fast_table, sizeof descs[0].fast_table) == 0 &&
                 memcmp(descs[i].dc_fast_table, descs[0].dc_fast_table, sizeof descs[0].dc_fast_table) == 0 &&
                 descs[i].fast_off == descs[0].fast_off;
    if (want)
        CG_TRY(walk_tables.reserve(kWalkTableBytes * (shared ? 1 : n)));
    for (size_t i = 0; i < n; i++)
        descs[i].walk = want ? reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(walk_tables.ptr) +
                                                                  (shared ? 0 : i) * kWalkTableBytes)
                             : nullptr;
    max_mcus = 0;
    mcu_uniform = uniform;
    for (size_t i = 0; i < n && mcu_route; i++) {
        max_mcus = std::max(max_mcus, descs[i].total_mcus);
        mcu_uniform = mcu_uniform && descs[i].total_mcus == descs[0].total_mcus;
    }
    std::vector<ImageDesc> &views = mcu_views; // (a member: the asynchronous copy below reads it after this function has returned)
    views.clear();
    if (mcu_route) {
        // per image: a word index per MCU (and 64 more: the window of an image's last wave asks for the entry behind
        // its MCUs' -- never used) and a state per MCU
        size_t words_total = 0, states_total = 0;
        for (size_t i = 0; i < n; i++) {
            words_total += align_up((size_t(descs[i].total_mcus) + kWave) * 4, 256);
            states_total += align_up(size_t(descs[i].total_mcus) * sizeof(McuState), 256);
        }
        CG_TRY(mcu_words.reserve(words_total + 256));
        CG_TRY(mcu_states.reserve(states_total + 256));
        CG_TRY(mcu_descs.reserve(n * sizeof(ImageDesc) + 256));
        views.resize(n);
        size_t w_at = 0, s_at = 0;
        for (size_t i = 0; i < n; i++) {
            ImageDesc &d = descs[i];
            d.mcu_word = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(mcu_words.ptr) + w_at);
            d.mcu_state = reinterpret_cast<McuState *>(static_cast<uint8_t *>(mcu_states.ptr) + s_at);
            w_at += align_up((size_t(d.total_mcus) + kWave) * 4, 256);
            s_at += align_up(size_t(d.total_mcus) * sizeof(McuState), 256);
            // the image as the second kernel sees it: every "interval" one MCU
            ImageDesc &v = views[i];
            v = d;
            v.starts = d.mcu_word;
            v.nstarts = d.total_mcus;
            v.total_intervals = d.total_mcus;
            v.restart_interval = 1;
            v.walk = nullptr;
            v.coop_ok = 0;
        }
        // Window of the second kernel: the words of 64 consecutive MCUs.  Known is the largest span of 64 consecutive
        // *intervals* (max_span: 64 R MCUs): three times the average of its MCUs and a little, at most all of it.  (A lane
        // whose words lie beyond its wave's window reads them from memory with the reference's reader: slower, the same
        // result.)
        const uint32_t r = std::max(1u, min_restart_interval);
        const uint64_t avg64 = (uint64_t(max_span) + r - 1) / r;
        mcu_span = uint32_t(std::min<uint64_t>(3 * avg64 + 64, max_span));
        CG_HIP(hipMemcpyAsync(mcu_descs.ptr, views.data(), n * sizeof(ImageDesc), hipMemcpyHostToDevice, stream));
    }
    CG_HIP(hipMemcpyAsync(dev_descs.ptr, descs.data(), n * sizeof(ImageDesc), hipMemcpyHostToDevice, stream));
    if (want)
        CG_HIP(launch_walk_tables(static_cast<const ImageDesc *>(dev_descs.ptr), uint32_t(shared ? 1 : n), stream));
    return Status{};
}

Status compeg_batch::decode(hipStream_t stream)
{
    CG_TRY(finish_upload()); // (an upload begun with compeg_batch_upload_jpegs_begin and not ended yet)
    if (count == 0)
        return Status{};
    CG_HIP(hipSetDevice(gpu->device));
    constexpr size_t kMaxTimed = 4096;
    const bool timing = timing_on && decodes_timed < kMaxTimed;
    hipEvent_t *ev = nullptr;
    if (timing) {
        while (events.size() < (decodes_timed + 1) * 3) {
            hipEvent_t e;
            CG_HIP(hipEventCreate(&e));
            events.push_back(e);
        }
        ev = events.data() + decodes_timed * 3;
    }
    CG_TRY(unit_queue.reserve(256));
    // One batch, one set of device buffers and one units' queue: a decode recorded on another stream than the
    // previous one waits for that one (two launches drawing from one queue would each decode part of the units).
    // (the event is recorded only now, behind everything the earlier decodes put on their stream: every event a decode
    // records is a packet the card works through between two kernels -- four of them were 18 us between two
    // single-frame decodes of 32 us)
    if (decode_recorded && stream != last_stream) {
        if (!decode_done)
            CG_HIP(hipEventCreateWithFlags(&decode_done, hipEventDisableTiming));
        CG_HIP(hipEventRecord(decode_done, last_stream));
        CG_HIP(hipStreamWaitEvent(stream, decode_done, 0));
    }
    const ImageDesc *dd = static_cast<const ImageDesc *>(dev_descs.ptr);
    const uint32_t n = uint32_t(count);
    const uint32_t step = chunk ? std::min(chunk, n) : n;
    bool ev1_recorded = false;
    if (timing)
        CG_HIP(hipEventRecord(ev[0], stream));
    if (preprocess_mode == 2 && host_fallbacks == 0)
        CG_HIP(launch_scan(static_cast<const ScanDesc *>(scan_descs.ptr), n, max_tiles, stream));
    for (uint32_t at = 0; at < n; at += step) {
        const uint32_t m = std::min(step, n - at);
        const bool fused = use_fused_pipeline() && !generic_layout;
        const bool one_layout = generic_layout && layout_h != 0 && use_fused_pipeline();
        const bool mcu_pairs = layout_h == 1 && layout_pairs;
        const HuffLdsPlan plan = plan_huffman(max_intervals, m, max_l2, max_span, fused || generic_layout,
                                              one_layout ? fused_layout_wave_cap(layout_h, layout_v, mcu_pairs) : 0u);
        if (one_layout) {
            // every image has the same extension layout: its fused kernel, whole windows or streamed
            const bool streamed = layout_has_stream_kernel(layout_h, layout_v, mcu_pairs) && use_stream_kernel(plan, max_intervals, m, kLayoutCuWaves, fused_layout_wave_cap(layout_h, layout_v, mcu_pairs));
            last_kernel = at ? last_kernel : (streamed ? COMPEG_KERNEL_FUSED_STREAM : COMPEG_KERNEL_FUSED_LAYOUT);
            if (streamed)
                CG_HIP(launch_fused_stream(dd + at, m, max_intervals,
                                               plan_stream(max_intervals, m, max_l2, stream_mcu_words, uniform, kLayoutCuWaves,
                                                           fused_layout_wave_cap(layout_h, layout_v, mcu_pairs)),
                                               stream, layout_h, layout_v, static_cast<uint32_t *>(unit_queue.ptr)));
            else
                CG_HIP(launch_fused_layout(dd + at, m, max_intervals, plan, layout_h, layout_v, mcu_pairs, stream));
            continue;
        }
        if (generic_layout) {
            last_kernel = at ? last_kernel : COMPEG_KERNEL_GENERIC;
            CG_HIP(launch_entropy_samples(dd + at, m, max_intervals, plan, stream));
            if (timing && at == 0) {
                CG_HIP(hipEventRecord(ev[1], stream));
                ev1_recorded = true;
            }
            CG_HIP(launch_generic_composite(dd + at, m, max_out_w, max_out_h, stream));
            continue;
        }
        if (fused && mcu_route) {
            // the walk, a lane per restart interval (streamed windows: any interval length), then a lane per MCU
            const ImageDesc *md = static_cast<const ImageDesc *>(mcu_descs.ptr);
            CG_HIP(launch_walk_mcus(dd + at, m, max_intervals, plan_walk(max_intervals, m, max_l2, stream_mcu_words, min_restart_interval, uniform, descs[0].walk != nullptr),
                                    stream, static_cast<uint32_t *>(unit_queue.ptr)));
            if (timing && at == 0) {
                CG_HIP(hipEventRecord(ev[1], stream));
                ev1_recorded = true;
            }
            const HuffLdsPlan mcu_plan = plan_huffman(max_mcus, m, max_l2, mcu_span, true);
            CG_HIP(launch_fused_422(md + at, m, max_mcus, mcu_plan, stream, mcu_uniform, true, static_cast<uint32_t *>(unit_queue.ptr), true));
            if (at == 0)
                last_kernel = COMPEG_KERNEL_WALK_MCU;
            continue;
        }
        if (fused) {
            CoopPlan coop{};
            if (coop_r && use_coop_kernel(max_intervals, m, coop_r))
                coop = plan_coop(max_intervals, m, coop_r, max_l2, coop_spans);
            // Images of different sizes (a grid row per image, workgroups that wait for their slowest wave): the
            // streamed form in workgroups of four waves, three to a CU, which the hardware shares out as they finish --
            // 256 frames of four sizes 550 -> 630 Gpixel/s, 1024 small ones 344 -> 444; not below DRI = 3 (606 -> 559).
            const bool small_groups = !coop.usable && !uniform && min_restart_interval >= 3u && m == n &&
                                      total_waves > uint64_t(12u) * 256u && !lab_env("COMPEG_NO_SMALL_GROUPS");
            const bool streamed = !coop.usable && (small_groups || use_stream_kernel(plan, max_intervals, m));
            if (coop.usable)
                CG_HIP(launch_coop_422(dd + at, m, max_intervals, coop, stream));
            else if (streamed)
                CG_HIP(launch_fused_stream(dd + at, m, max_intervals,
                                           plan_stream(max_intervals, m, max_l2, stream_mcu_words, uniform, 0, small_groups ? 4u : 0u), stream, 2, 1,
                                           static_cast<uint32_t *>(unit_queue.ptr)));
            else if (use_pair_kernel(max_intervals, m))
                CG_HIP(launch_pair_422(dd + at, m, max_intervals, plan, stream));
            else
                CG_HIP(launch_fused_422(dd + at, m, max_intervals, plan, stream, uniform, one_mcu_intervals, static_cast<uint32_t *>(unit_queue.ptr)));
            if (at == 0)
                last_kernel = coop.usable ? COMPEG_KERNEL_COOP_TEAM
                              : streamed  ? COMPEG_KERNEL_FUSED_STREAM
                                          : (use_pair_kernel(max_intervals, m) ? COMPEG_KERNEL_PAIR : COMPEG_KERNEL_FUSED);
            continue;
        }
        last_kernel = at ? last_kernel : COMPEG_KERNEL_SPLIT;
        CG_HIP(launch_entropy(dd + at, m, max_intervals, plan, stream));
        if (timing && at == 0) {
            CG_HIP(hipEventRecord(ev[1], stream)); // stage split is exact for unchunked decodes
            ev1_recorded = true;
        }
        CG_HIP(launch_idct_composite(dd + at, m, max_dus, stream));
    }
    if (timing) {
        CG_HIP(hipEventRecord(ev[2], stream));
        if (has_stage_event.size() <= decodes_timed)
            has_stage_event.resize(decodes_timed + 1);
        has_stage_event[decodes_timed] = ev1_recorded; // (one-kernel decodes: no event in the middle)
        decodes_timed++;
    }
    decode_recorded = true;
    last_stream = stream;
    return Status{};
}
