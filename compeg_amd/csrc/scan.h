// Host scan preprocessor: strips RSTn markers and FF 00 stuffing from the
// entropy-coded segment, aligns every restart interval to a 32-bit word and
// records each interval's word offset.  Byte-identical to the reference's
// ScanBuffer::process on a fresh buffer (src/scan.rs:33-128).
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>

#include "front.h"

struct compeg_gpu;

namespace compeg {

// Grow-only byte arena; the decoder plugs in pinned host memory so that the
// preprocessed scan can be DMA'd without a bounce copy.
struct HostArena {
    using AllocFn = void *(*)(size_t);
    using FreeFn = void (*)(void *);
    uint8_t *data = nullptr;
    size_t capacity = 0;
    AllocFn alloc_fn;
    FreeFn free_fn;

    HostArena(AllocFn a, FreeFn f) : alloc_fn(a), free_fn(f) {}
    HostArena();
    ~HostArena();
    HostArena(const HostArena &) = delete;
    HostArena &operator=(const HostArena &) = delete;
    bool reserve(size_t bytes); // contents are not preserved
};

struct ScanTeam;

class ScanBuffer {
  public:
    ScanBuffer();
    ScanBuffer(HostArena::AllocFn a, HostArena::FreeFn f);
    ~ScanBuffer();

    // Threads that share the work of one process() call on segments of 64 KiB per thread and more
    // (1 = the calling thread alone, the default).  The helpers live as long as the buffer.
    // self_check: the first such call is timed both ways and the helpers are dropped if sharing the
    // work is slower than the calling thread alone (for thread counts nobody asked for explicitly).
    void set_threads(unsigned threads, bool self_check = false);
    // memcpy shared by the same threads (the decoder stages raw segments with it)
    void copy(void *dst, const void *src, size_t bytes);
    unsigned threads() const;

    // COMPEG_E_COUNT_MISMATCH leaves the truncated result in place, like the
    // reference (scan.rs:55-63).
    // progress (optional) is called from inside the loop whenever about `progress_step` more
    // output bytes are final, with the number of final bytes (a multiple of 16): the decoder
    // ships them to the GPU while the rest of the segment is still being scanned.
    using Progress = std::function<void(size_t final_bytes)>;
    Status process(const uint8_t *scan, size_t len, uint32_t expected_intervals, const Progress &progress = {},
                   size_t progress_step = 0);
    // The same result in memory of the caller's (one thread, no helpers): `out` holds
    // output_capacity(len) bytes, `starts` start_slots(expected) words.
    static size_t output_capacity(size_t len);
    static size_t start_slots(uint32_t expected_intervals);
    static Status process_to(const uint8_t *scan, size_t len, uint32_t expected_intervals, uint8_t *out,
                             uint32_t *starts, size_t &nwords, size_t &nstarts);
    // Same buffers, filled by the device-side scan kernels (runtime.cpp).
    Status process_on_gpu(struct ::compeg_gpu *gpu, const uint8_t *scan, size_t len,
                          uint32_t expected_intervals);

    const uint8_t *data() const { return words_.data; }
    size_t data_bytes() const { return nwords_ * 4; }
    const uint32_t *words() const { return reinterpret_cast<const uint32_t *>(words_.data); }
    size_t nwords() const { return nwords_; }
    const uint32_t *starts() const { return reinterpret_cast<const uint32_t *>(starts_.data); }
    size_t nstarts() const { return nstarts_; }

  private:
    bool process_with_team(const uint8_t *scan, size_t len, uint32_t expected, uint8_t *out, uint32_t *starts,
                           size_t slots, size_t &wp, size_t &ri, const Progress &progress);
    HostArena words_, starts_;
    size_t nwords_ = 0, nstarts_ = 0;
    std::unique_ptr<ScanTeam> team_;
    bool team_checked_ = false; // the helpers have been timed against the calling thread alone
};

} // namespace compeg
