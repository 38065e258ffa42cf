// Host scan preprocessor (see scan.h): the reference's byte-serial loop (src/scan.rs:33-128) as a
// vector copy between 0xFF bytes, optionally shared by several threads; byte-identical output.
#include "scan.h"
#include "lab.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace compeg {

HostArena::HostArena() : alloc_fn(malloc), free_fn(free) {}

HostArena::~HostArena()
{
    if (data)
        free_fn(data);
}

bool HostArena::reserve(size_t bytes)
{
    if (bytes <= capacity)
        return true;
    size_t want = capacity ? capacity : 4096;
    while (want < bytes)
        want += want / 2 + 4096;
    void *p = alloc_fn(want);
    if (!p)
        return false;
    if (data)
        free_fn(data);
    data = static_cast<uint8_t *>(p);
    capacity = want;
    return true;
}

namespace {

// The loop, written once and compiled for two vector widths.  Plain bytes are
// copied one vector at a time until a vector holds an FF; the stores run ahead
// of wp by up to a vector (whatever follows overwrites them; the arena has
// slack).  Fewer than a vector left: byte by byte.  COPY_MASK(src, dst) copies
// one vector and returns the bit mask of its FF bytes.
#define COMPEG_SCAN_LOOP(VECTOR_BYTES, COPY_MASK)                                                      \
    size_t wp = 0, ri = 1, rp = 0;                                                                     \
    size_t report_at = progress ? progress_step : ~size_t(0);                                          \
    size_t head = ~size_t(0);                                                                          \
    for (;;) {                                                                                         \
        while (rp + (VECTOR_BYTES) <= len) {                                                           \
            const uint32_t ffs = COPY_MASK(scan + rp, out + wp);                                       \
            if (ffs) {                                                                                 \
                const uint32_t n = uint32_t(__builtin_ctz(ffs));                                       \
                rp += n;                                                                               \
                wp += n;                                                                               \
                break;                                                                                 \
            }                                                                                          \
            rp += (VECTOR_BYTES);                                                                      \
            wp += (VECTOR_BYTES);                                                                      \
        }                                                                                              \
        if (rp + (VECTOR_BYTES) > len) {                                                               \
            while (rp < len && scan[rp] != 0xff)                                                       \
                out[wp++] = scan[rp++];                                                                \
        }                                                                                              \
        if (rp + 1 >= len)                                                                             \
            break; /* no FF left, or a lone FF ends the data (dropped) */                              \
        /* FF 00 emits FF; anything else counts as RSTn (scan.rs:103-112): the output is padded    */ \
        /* with zeros to the next word and the new interval's word offset is noted.  No branch on  */ \
        /* the kind of pair: the two kinds alternate unpredictably.                                */ \
        const bool stuffing = scan[rp + 1] == 0x00;                                                    \
        rp += 2;                                                                                       \
        store_u32(out + wp, stuffing ? 0xffu : 0u); /* FF, or up to three padding zeros */             \
        const size_t next = stuffing ? wp + 1 : (wp + 3) & ~size_t(3);                                 \
        uint32_t &slot = starts[ri & mask];                                                            \
        slot = stuffing ? slot : uint32_t(next / 4);                                                   \
        head = (!stuffing && ri == 1) ? wp : head;                                                     \
        ri += stuffing ? 0 : 1;                                                                        \
        wp = next;                                                                                     \
        if (wp >= report_at) { /* everything below wp is final */                                     \
            progress(wp & ~size_t(15));                                                                \
            report_at = wp + progress_step;                                                            \
        }                                                                                              \
    }                                                                                                  \
    end.wp = wp;                                                                                       \
    end.ri = ri;                                                                                       \
    end.head = head == ~size_t(0) ? wp : head;

struct ScanEnd {
    size_t wp, ri;
    size_t head; // kept bytes in front of the first marker (all of them if there is none)
};

inline void store_u32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline uint32_t copy_mask_avx2(const uint8_t *src, uint8_t *dst)
{
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src));
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(dst), v);
    return uint32_t(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8(char(0xff)))));
}

inline uint32_t copy_mask_sse2(const uint8_t *src, uint8_t *dst)
{
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src));
    _mm_storeu_si128(reinterpret_cast<__m128i *>(dst), v);
    return uint32_t(_mm_movemask_epi8(_mm_cmpeq_epi8(v, _mm_set1_epi8(char(0xff)))));
}

__attribute__((target("avx2"))) void scan_avx2(const uint8_t *scan, size_t len, uint8_t *out, uint32_t *starts,
                                               size_t mask, ScanEnd &end, const ScanBuffer::Progress &progress,
                                               size_t progress_step)
{
    COMPEG_SCAN_LOOP(32, copy_mask_avx2)
}

void scan_sse2(const uint8_t *scan, size_t len, uint8_t *out, uint32_t *starts, size_t mask, ScanEnd &end,
               const ScanBuffer::Progress &progress, size_t progress_step)
{
    COMPEG_SCAN_LOOP(16, copy_mask_sse2)
}
#else
inline uint32_t copy_mask_none(const uint8_t *, uint8_t *) { return 0; }

void scan_bytes(const uint8_t *scan, size_t len, uint8_t *out, uint32_t *starts, size_t mask, ScanEnd &end,
                const ScanBuffer::Progress &progress, size_t progress_step)
{
    COMPEG_SCAN_LOOP(len + 1, copy_mask_none)
}
#endif

void scan_range(const uint8_t *scan, size_t len, uint8_t *out, uint32_t *starts, size_t mask, ScanEnd &end,
                const ScanBuffer::Progress &progress, size_t progress_step)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2)
        scan_avx2(scan, len, out, starts, mask, end, progress, progress_step);
    else
        scan_sse2(scan, len, out, starts, mask, end, progress, progress_step);
#else
    scan_bytes(scan, len, out, starts, mask, end, progress, progress_step);
#endif
}

// Spin-wait step.  No PAUSE: inside a virtual machine a loop of PAUSEs is what the hypervisor watches
// for to take the CPU away (pause-loop exiting), and a helper that lost its CPU is late by a
// scheduling quantum, not by a cache miss.
inline void cpu_relax()
{
    std::atomic_signal_fence(std::memory_order_seq_cst);
}

} // namespace

// ---- several threads on one segment ------------------------------------------
//
// The segment is cut into one piece per thread.  A piece cannot know where its
// output goes (every interval is padded to a word, so positions depend on
// everything in front), but it can be preprocessed on its own into a private
// buffer: the bytes in front of its first marker (its head), then whole
// word-aligned intervals, then the unpadded bytes behind its last marker.  A
// serial step of a few instructions per piece turns the pieces' sizes into
// output positions, and the threads copy their pieces into place.  Whether the
// first byte of a piece is the second half of an FF xx pair follows from the
// length of the FF run in front of it (odd: yes), as in the device kernels.
struct ScanTeam {
    struct Piece {
        size_t begin = 0, end = 0;  // byte range of the segment
        std::vector<uint8_t> out;   // private output
        std::vector<uint32_t> starts; // entry j: word offset (in `out`) of the interval the piece's marker j opens, j >= 1
        ScanEnd done{0, 1, 0};
        // where the piece's output begins: open interval, its start word, bytes it holds so far
        size_t interval = 0, start_word = 0, bytes = 0;
    };
    std::vector<Piece> pieces;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<uint64_t> generation{0};
    std::atomic<unsigned> pending{0};
    std::atomic<bool> stop{false};
    const std::function<void(unsigned)> *job = nullptr;
    int claimed = 0; // helpers counted against the process-wide budget of default (unasked-for) helpers

    explicit ScanTeam(unsigned threads) : pieces(threads)
    {
        for (unsigned k = 1; k < threads; k++)
            workers.emplace_back([this, k] { work(k); });
    }
    ~ScanTeam()
    {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
        }
        cv.notify_all();
        for (std::thread &t : workers)
            t.join();
        g_default_helpers.fetch_sub(claimed);
    }
    // Helper threads that decoders start by default, all decoders of the process together: never more than
    // the machine has spare hardware threads ("any number of decoders per gpu" must not mean any number of
    // spinning helpers).  Thread counts asked for explicitly are not limited.
    static std::atomic<int> g_default_helpers;
    void work(unsigned k)
    {
        uint64_t seen = 0;
        for (;;) {
            // a decoder fed frame after frame finds its helpers still spinning; otherwise they sleep
            for (unsigned spins = 0; generation.load(std::memory_order_acquire) == seen && !stop; spins++) {
                if (spins < 400000) {
                    cpu_relax();
                } else {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return generation.load() != seen || stop; });
                }
            }
            if (stop)
                return;
            seen = generation.load(std::memory_order_acquire);
            (*job)(k);
            pending.fetch_sub(1, std::memory_order_release);
        }
    }
    // job(k) on every thread, k = 0 on the caller's
    void run(const std::function<void(unsigned)> &j)
    {
        job = &j;
        pending.store(unsigned(workers.size()), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> l(m);
            generation.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        j(0);
        while (pending.load(std::memory_order_acquire))
            cpu_relax();
    }
};

std::atomic<int> ScanTeam::g_default_helpers{0};

ScanBuffer::ScanBuffer() = default;
ScanBuffer::ScanBuffer(HostArena::AllocFn a, HostArena::FreeFn f) : words_(a, f), starts_(a, f) {}
ScanBuffer::~ScanBuffer() = default;

void ScanBuffer::set_threads(unsigned threads, bool self_check)
{
    threads = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    if (team_ && team_->pieces.size() == threads) {
        team_checked_ = team_checked_ || !self_check;
        return;
    }
    team_.reset();
    team_checked_ = !self_check;
    int claim = 0;
    if (self_check && threads > 1) {
        const int spare = int(std::thread::hardware_concurrency()) - 1;
        const int want = int(threads) - 1;
        const int before = ScanTeam::g_default_helpers.fetch_add(want);
        claim = std::max(0, std::min(want, spare - before));
        if (claim < want)
            ScanTeam::g_default_helpers.fetch_sub(want - claim);
        threads = unsigned(claim) + 1u;
    }
    if (threads > 1) {
        team_.reset(new ScanTeam(threads));
        team_->claimed = claim;
    }
}

unsigned ScanBuffer::threads() const { return team_ ? unsigned(team_->pieces.size()) : 1u; }

void ScanBuffer::copy(void *dst, const void *src, size_t bytes)
{
    const size_t n = team_ ? team_->pieces.size() : 1;
    if (n < 2 || bytes < n * (64u << 10)) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes + n - 1) / n + 63) & ~size_t(63); // (rounded up before it is aligned: n pieces cover every byte)
    team_->run([&](unsigned k) {
        const size_t at = per * k;
        if (at < bytes)
            memcpy(static_cast<uint8_t *>(dst) + at, static_cast<const uint8_t *>(src) + at,
                   std::min(per, bytes - at));
    });
}

// false: not worth it or not possible (the caller takes the one-thread loop)
//
// With a progress callback the segment is taken in rounds, each one shared by all threads, and the callback hears
// about the output of a round as soon as it is in place: the decoder ships it while the next round is being scanned.
// The callback runs on the calling thread as the first thing of its share of the next round, so that the helpers
// never wait for the launches it makes; in the first round it is called with 0 ("nothing yet: prepare what does not
// depend on the scan").
bool ScanBuffer::process_with_team(const uint8_t *scan, size_t len, uint32_t expected, uint8_t *out, uint32_t *starts,
                                   size_t slots, size_t &wp_out, size_t &ri_out, const Progress &progress)
{
    ScanTeam &team = *team_;
    const size_t n = team.pieces.size();
    if (len < n * (64u << 10))
        return false;
    size_t rounds = 1;
    if (progress) {
        size_t want = 2; // (a round costs two rendezvous of the team: three or four rounds measured 4-15 us slower)
        if (const char *e = lab_env("COMPEG_SCAN_ROUNDS")) // experiment knob
            want = size_t(std::max(1, atoi(e)));
        rounds = std::max<size_t>(1, std::min(want, len / (n * (48u << 10))));
    }
    // the calling thread's share of a round, in quarters of a helper's
    size_t own = 4; // (a smaller share for the thread that makes the launches measured no better)
    if (const char *e = lab_env("COMPEG_SCAN_OWN")) // experiment knob
        own = size_t(std::max(0, std::min(4, atoi(e))));
    // a cut that falls inside an FF xx pair moves behind the partner byte
    auto cut = [&](size_t at) {
        if (at == 0 || at >= len)
            return at >= len ? len : at;
        size_t run = 0;
        while (run < at && scan[at - 1 - run] == 0xff)
            run++;
        return at + (run & 1u);
    };
    (void)expected;
    const size_t mask = slots - 1;
    size_t interval = 0, start_word = 0, bytes = 0; // the output's cursor: open interval, its start word, bytes in it
    // (two rounds: the first a little longer than the second -- 9/16 of the segment.  The decoder ships a round's output
    // while the next round is scanned; the second round's launch has to be made, and to reach the card, before the first
    // round's transfer ends: with equal halves of a 4K frame's scan the card idled 4 us between the two)
    auto round_start = [&](size_t r) { return rounds == 2 ? (r == 0 ? size_t(0) : (r == 1 ? len / 16 * 9 : len)) : len / rounds * r; };
    for (size_t r = 0; r < rounds; r++) {
        const size_t lo = round_start(r), hi = r + 1 == rounds ? len : round_start(r + 1);
        const size_t shares = own + 4 * (n - 1); // (in quarters of a helper's piece)
        for (size_t k = 0; k <= n; k++) {
            const size_t at = cut(k == n ? hi : lo + (hi - lo) / shares * (k ? own + 4 * (k - 1) : 0));
            if (k < n)
                team.pieces[k].begin = at;
            if (k > 0)
                team.pieces[k - 1].end = at;
        }
        const size_t reported = (start_word * 4 + bytes) & ~size_t(15);
        team.run([&](unsigned k) {
            if (k == 0 && progress)
                progress(r ? reported : 0);
            ScanTeam::Piece &p = team.pieces[k];
            const size_t range = p.end - p.begin;
            if (p.out.size() < range + range / 3 + 80)
                p.out.resize(range + range / 3 + 80);
            // a marker takes two bytes, so a piece holds at most range / 2 of them: its private table is sized
            // for that and the marker index can never wrap.  (A table sized from the expected count could, on
            // segments whose markers crowd into one piece, and giving up then -- after an earlier round's output
            // had been reported and was perhaps being shipped -- would have had the one-thread loop rewrite
            // reported bytes with its look-ahead stores.)
            size_t cap = 1024;
            while (cap < range / 2 + 2)
                cap <<= 1;
            if (p.starts.size() < cap)
                p.starts.resize(cap);
            p.done = ScanEnd{0, 1, 0};
            scan_range(scan + p.begin, range, p.out.data(), p.starts.data(), cap - 1, p.done, {}, 0);
        });
        // positions
        for (ScanTeam::Piece &p : team.pieces) {
            p.interval = interval;
            p.start_word = start_word;
            p.bytes = bytes;
            const size_t markers = p.done.ri - 1;
            if (markers == 0) {
                bytes += p.done.head;
            } else {
                const size_t first = p.starts[1], last = p.starts[markers];
                interval += markers;
                start_word += (bytes + p.done.head + 3) / 4 + (last - first);
                bytes = p.done.wp - last * 4;
            }
        }
        // The reference keeps the last writer of every slot (scan.rs:46-56,111).  Rounds follow each other, so
        // later ones overwrite earlier ones as in the reference; inside a round only its last `slots` markers write.
        const size_t count_so_far = interval + 1;
        team.run([&](unsigned k) {
            const ScanTeam::Piece &p = team.pieces[k];
            const size_t markers = p.done.ri - 1;
            uint8_t *dst = out + p.start_word * 4 + p.bytes;
            memcpy(dst, p.out.data(), p.done.head);
            if (markers == 0)
                return;
            const size_t seam_end = p.start_word * 4 + p.bytes + p.done.head, body = (seam_end + 3) & ~size_t(3);
            memset(out + seam_end, 0, body - seam_end);
            const size_t first = p.starts[1];
            memcpy(out + body, p.out.data() + first * 4, p.done.wp - first * 4);
            for (size_t j = 1; j <= markers; j++) {
                const size_t global = p.interval + j;
                if (global + slots >= count_so_far)
                    starts[global & mask] = uint32_t(body / 4 + (p.starts[j] - first));
            }
        });
    }
    wp_out = start_word * 4 + bytes;
    ri_out = interval + 1;
    return true;
}

size_t ScanBuffer::output_capacity(size_t len)
{
    // Worst case: a 1-byte interval behind a 2-byte marker occupies a whole
    // word, i.e. 4 bytes out for 3 in (scan.rs:38-44); plus the vector loop's slack.
    return ((len + len / 3 + 3) / 4) * 4 + 72;
}

size_t ScanBuffer::start_slots(uint32_t expected)
{
    size_t slots = 1;
    while (slots < expected)
        slots <<= 1;
    return slots;
}

Status ScanBuffer::process_to(const uint8_t *scan, size_t len, uint32_t expected, uint8_t *out, uint32_t *starts,
                              size_t &nwords, size_t &nstarts)
{
    const size_t slots = start_slots(expected);
    memset(starts, 0, slots * 4);
    ScanEnd end{0, 1, 0};
    scan_range(scan, len, out, starts, slots - 1, end, {}, 0);
    nwords = (end.wp + 3) / 4;
    store_u32(out + end.wp, 0u);
    nstarts = end.ri < slots ? end.ri : slots;
    if (end.ri != expected) {
        char msg[128];
        snprintf(msg, sizeof msg, "restart interval count mismatch: counted %zu, expected %u", end.ri, expected);
        return Status::error(COMPEG_E_COUNT_MISMATCH, msg);
    }
    return Status{};
}

Status ScanBuffer::process(const uint8_t *scan, size_t len, uint32_t expected, const Progress &progress,
                           size_t progress_step)
{
    // Worst case: a 1-byte interval behind a 2-byte marker occupies a whole
    // word, i.e. 4 bytes out for 3 in (scan.rs:38-44).
    const size_t out_cap = ((len + len / 3 + 3) / 4) * 4;
    size_t slots = 1;
    while (slots < expected)
        slots <<= 1;
    if (!words_.reserve(out_cap + 72) || !starts_.reserve(slots * 4))
        return Status::error(COMPEG_E_HIP, "out of host memory in ScanBuffer");
    uint8_t *out = words_.data;
    uint32_t *starts = reinterpret_cast<uint32_t *>(starts_.data);
    memset(starts, 0, slots * 4);
    const size_t mask = slots - 1;

    size_t wp = 0, ri = 1;
    // The first segment large enough for the helpers is also timed on one thread: where threads are scarce or
    // slow to wake (a small CPU quota, some virtual machines) sharing the work costs more than it saves, and
    // the buffer then keeps to the calling thread.
    if (team_ && !team_checked_ && len >= team_->pieces.size() * (64u << 10)) {
        team_checked_ = true;
        using clock = std::chrono::steady_clock;
        const auto t0 = clock::now();
        ScanEnd alone{0, 1, 0};
        scan_range(scan, len, out, starts, mask, alone, {}, 0);
        const auto t1 = clock::now();
        memset(starts, 0, slots * 4);
        size_t twp = 0, tri = 1;
        // (once to wake the helpers and touch their buffers, once for the clock)
        bool shared = process_with_team(scan, len, expected, out, starts, slots, twp, tri, {});
        const auto t2 = clock::now();
        memset(starts, 0, slots * 4);
        shared = shared && process_with_team(scan, len, expected, out, starts, slots, twp, tri, {});
        const auto t3 = clock::now();
        if (!shared || (t3 - t2) > (t1 - t0))
            team_.reset();
        memset(starts, 0, slots * 4);
    }
    if (!team_ || !process_with_team(scan, len, expected, out, starts, slots, wp, ri, progress)) {
        memset(starts, 0, slots * 4); // (a team that gave up may have written some)
        ScanEnd end{0, 1, 0};
        scan_range(scan, len, out, starts, mask, end, progress, progress_step);
        wp = end.wp;
        ri = end.ri;
    }
    const size_t nwords = (wp + 3) / 4;
    store_u32(out + wp, 0u);
    wp = nwords * 4;
    nwords_ = nwords;
    nstarts_ = ri < slots ? ri : slots;

    if (ri != expected) {
        char msg[128];
        snprintf(msg, sizeof msg, "restart interval count mismatch: counted %zu, expected %u", ri,
                 expected);
        return Status::error(COMPEG_E_COUNT_MISMATCH, msg);
    }
    return Status{};
}

} // namespace compeg
