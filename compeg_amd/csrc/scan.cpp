#include "scan.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

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
        ri += stuffing ? 0 : 1;                                                                        \
        wp = next;                                                                                     \
        if (wp >= report_at) { /* everything below wp is final */                                     \
            progress(wp & ~size_t(15));                                                                \
            report_at = wp + progress_step;                                                            \
        }                                                                                              \
    }                                                                                                  \
    end.wp = wp;                                                                                       \
    end.ri = ri;

struct ScanEnd {
    size_t wp, ri;
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

} // namespace

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

    ScanEnd end{0, 1};
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2)
        scan_avx2(scan, len, out, starts, mask, end, progress, progress_step);
    else
        scan_sse2(scan, len, out, starts, mask, end, progress, progress_step);
#else
    scan_bytes(scan, len, out, starts, mask, end, progress, progress_step);
#endif
    size_t wp = end.wp;
    const size_t ri = end.ri;
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
