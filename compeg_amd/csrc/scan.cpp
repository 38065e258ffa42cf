#include "scan.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

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

Status ScanBuffer::process(const uint8_t *scan, size_t len, uint32_t expected)
{
    // Worst case: a 1-byte interval behind a 2-byte marker occupies a whole
    // word, i.e. 4 bytes out for 3 in (scan.rs:38-44).
    const size_t out_cap = ((len + len / 3 + 3) / 4) * 4;
    size_t slots = 1;
    while (slots < expected)
        slots <<= 1;
    if (!words_.reserve(out_cap + 8) || !starts_.reserve(slots * 4))
        return Status::error(COMPEG_E_HIP, "out of host memory in ScanBuffer");
    uint8_t *out = words_.data;
    uint32_t *starts = reinterpret_cast<uint32_t *>(starts_.data);
    memset(starts, 0, slots * 4);
    const size_t mask = slots - 1;

    size_t wp = 0, ri = 1, rp = 0;
    while (rp < len) {
        // copy the run up to the next FF in one go
        const uint8_t *ff = static_cast<const uint8_t *>(memchr(scan + rp, 0xff, len - rp));
        const size_t run = ff ? size_t(ff - (scan + rp)) : len - rp;
        memcpy(out + wp, scan + rp, run);
        wp += run;
        rp += run;
        if (!ff || rp + 1 >= len)
            break; // no FF left, or a lone FF ends the data (dropped)
        const uint8_t m = scan[rp + 1];
        rp += 2;
        if (m == 0x00) {
            out[wp++] = 0xff;
        } else {
            // anything else counts as RSTn (scan.rs:103-112): pad with zeros
            // to the next word and note where the new interval starts
            while (wp & 3)
                out[wp++] = 0;
            starts[ri & mask] = uint32_t(wp / 4);
            ri++;
        }
    }
    const size_t nwords = (wp + 3) / 4;
    while (wp & 3)
        out[wp++] = 0;
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
