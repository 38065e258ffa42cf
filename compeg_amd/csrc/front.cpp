// Host front-end: see front.h.  Produces, for every accepted file, the same
// Metadata / L1 / L2 bytes and scan-data range as the reference's
// ImageData::new, and rejects what it rejects with the same message.
#include "front.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace compeg {
namespace {

const char *kEof = "reached end of data while decoding JPEG stream"; // file.rs:281-283

// ITU T.81 Annex K.3 (the reference falls back to these when a DHT is
// absent: lib.rs:608-613, huffman.rs:121-177).
const uint8_t kDcLumaCounts[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaCounts[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcSymbols[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaCounts[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
const uint8_t kAcChromaCounts[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
const uint8_t kAcLumaSymbols[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61,
    0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
    0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25,
    0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
    0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
    0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8,
    0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaSymbols[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61,
    0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
    0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18,
    0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
    0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63,
    0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7,
    0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
std::string fmt(const char *f, ...)
{
    char buf[320];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

Status unsupported(std::string m) { return Status::error(COMPEG_E_UNSUPPORTED, std::move(m)); }
Status malformed(std::string m) { return Status::error(COMPEG_E_MALFORMED, std::move(m)); }

// Bounded view over the file with the reference Reader's failure behaviour
// (file.rs:269-355): reading at or past `end` is the EOF error.
struct Cursor {
    const uint8_t *base;
    size_t end;
    size_t at;

    size_t left() const { return end - at; }
    bool byte(uint8_t &v)
    {
        if (at >= end)
            return false;
        v = base[at++];
        return true;
    }
    bool be16(unsigned &v)
    {
        uint8_t a, b;
        if (!byte(a) || !byte(b))
            return false;
        v = unsigned(a) << 8 | b;
        return true;
    }
};

const char *sof_name(uint8_t marker)
{
    static const char *const names[16] = {"SOF0", "SOF1",  "SOF2",  "SOF3",  "?",     "SOF5",
                                          "SOF6", "SOF7",  "?",     "SOF9",  "SOF10", "SOF11",
                                          "?",    "SOF13", "SOF14", "SOF15"};
    return names[marker & 15];
}

bool is_sof(uint8_t m)
{
    return m >= 0xc0 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc;
}

// Finds the end of the entropy-coded segment that starts at `from`
// (file.rs:163-201): FF 00 and FF D0..D7 belong to the segment, FF fill bytes
// in front of them too; any other marker ends it (the segment then stops in
// front of the last FF).  Returns false when the file ends first.
bool find_scan_end_bytes(const uint8_t *buf, size_t len, size_t from, size_t &end_out)
{
    size_t p = from;
    for (;;) {
        const void *hit = p < len ? memchr(buf + p, 0xff, len - p) : nullptr;
        if (!hit)
            return false;
        p = size_t(static_cast<const uint8_t *>(hit) - buf);
        size_t q = p + 1;
        while (q < len && buf[q] == 0xff)
            q++;
        if (q >= len)
            return false;
        uint8_t m = buf[q];
        if (m == 0x00 || (m >= 0xd0 && m <= 0xd7)) {
            p = q + 1;
        } else {
            end_out = q - 1;
            return true;
        }
    }
}

#if defined(__x86_64__)
// The same walk 32 bytes at a time: a restart-interval stream has an FF every hundred bytes or so, and a
// memchr call per FF costs more than the bytes between them (a 4K frame: 0.23 ms, as much as preprocessing
// it).  Anything but the plain cases -- FF 00, FF RSTn -- is left to the byte loop above, from that FF on.
__attribute__((target("avx2"))) bool find_scan_end_avx2(const uint8_t *buf, size_t len, size_t from, size_t &end_out)
{
    size_t p = from;
    const __m256i ff = _mm256_set1_epi8(char(0xff)), zero = _mm256_setzero_si256();
    const __m256i hi5 = _mm256_set1_epi8(char(0xf8)), rst = _mm256_set1_epi8(char(0xd0));
    uint64_t carry = 0; // the byte in front of this vector is an FF
    while (p + 32 <= len) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(buf + p));
        const uint64_t is_ff = uint32_t(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, ff)));
        // bytes that may follow an FF inside the segment: 00 and D0..D7
        const uint32_t fine = uint32_t(_mm256_movemask_epi8(_mm256_or_si256(
            _mm256_cmpeq_epi8(v, zero), _mm256_cmpeq_epi8(_mm256_and_si256(v, hi5), rst))));
        const uint32_t other = uint32_t((is_ff << 1) | carry) & ~fine; // something else follows an FF (one branch per vector)
        if (other)
            return find_scan_end_bytes(buf, len, p + uint32_t(__builtin_ctz(other)) - 1, end_out);
        carry = is_ff >> 31;
        p += 32;
    }
    return find_scan_end_bytes(buf, len, p - size_t(carry), end_out);
}
#endif

bool find_scan_end(const uint8_t *buf, size_t len, size_t from, size_t &end_out)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2)
        return find_scan_end_avx2(buf, len, from, end_out);
#endif
    return find_scan_end_bytes(buf, len, from, end_out);
}

} // namespace

bool HuffmanLut::build(const uint8_t counts[16], const uint8_t *symbols, size_t nsymbols)
{
    // Canonical (Annex C) code assignment.  First pass: validate and find the
    // 8-bit prefixes that need an L2 block; blocks are numbered by ascending
    // prefix, which is the order the reference concatenates them in
    // (huffman.rs:107-116).
    struct Code {
        uint32_t code;
        uint8_t len, sym;
    };
    std::vector<Code> codes;
    size_t total = 0;
    for (int i = 0; i < 16; i++)
        total += counts[i];
    if (total > nsymbols)
        return false;
    codes.reserve(total);
    uint32_t next = 0;
    size_t k = 0;
    bool needs_block[256] = {false};
    for (int len = 1; len <= 16; len++) {
        next <<= 1;
        for (unsigned c = 0; c < counts[len - 1]; c++) {
            if (next >= (1u << len))
                return false; // over-subscribed: not a prefix code
            codes.push_back({next, uint8_t(len), symbols[k++]});
            if (len > 8)
                needs_block[next >> (len - 8)] = true;
            next++;
        }
    }
    int block_of[256];
    size_t nblocks = 0;
    for (int p = 0; p < 256; p++)
        block_of[p] = needs_block[p] ? int(nblocks++) : -1;
    if (nblocks > 128)
        return false; // delegate index is 15 bits (huffman.rs:294-297)

    memset(l1, 0, sizeof l1);
    l2.assign(nblocks * 256, 0);
    for (int p = 0; p < 256; p++)
        if (block_of[p] >= 0)
            l1[p] = uint16_t(0x8000u | unsigned(block_of[p]) * 256u);
    for (const Code &c : codes) {
        const uint16_t entry = uint16_t(unsigned(c.len) << 8 | c.sym);
        if (c.len <= 8) {
            const unsigned first = c.code << (8 - c.len), span = 1u << (8 - c.len);
            for (unsigned i = 0; i < span; i++)
                l1[first + i] = entry;
        } else {
            const unsigned prefix = c.code >> (c.len - 8);
            const unsigned first = (c.code << (16 - c.len)) & 0xffu, span = 1u << (16 - c.len);
            uint16_t *blk = l2.data() + size_t(block_of[prefix]) * 256;
            for (unsigned i = 0; i < span; i++)
                blk[first + i] = entry;
        }
    }
    return true;
}

HuffmanLut HuffmanLut::annex_k(int which)
{
    HuffmanLut t;
    switch (which) {
    case 0:
        t.build(kDcLumaCounts, kDcSymbols, 12);
        break;
    case 1:
        t.build(kAcLumaCounts, kAcLumaSymbols, 162);
        break;
    case 2:
        t.build(kDcChromaCounts, kDcSymbols, 12);
        break;
    default:
        t.build(kAcChromaCounts, kAcChromaSymbols, 162);
        break;
    }
    return t;
}

Status ImageData::parse(const uint8_t *jpeg, size_t len, bool copy, ImageData **out, unsigned flags)
{
    *out = nullptr;
    if (!jpeg && len)
        return Status::error(COMPEG_E_INVALID_ARG, "jpeg pointer is NULL");

    ImageData *img = new ImageData();
    struct Guard {
        ImageData *p;
        ~Guard() { delete p; }
    } guard{img};
    if (copy) {
        img->owned.assign(jpeg, jpeg + len);
        jpeg = img->owned.data();
    }
    img->jpeg = jpeg;
    img->jpeg_len = len;

    Cursor file{jpeg, len, 0};
    uint8_t b0, b1;
    if (!file.byte(b0))
        return malformed(kEof);
    if (b0 != 0xff)
        return malformed("JPEG image does not start with SOI marker");
    if (!file.byte(b1))
        return malformed(kEof);
    if (b1 != 0xd8)
        return malformed("JPEG image does not start with SOI marker");

    HuffmanLut tables[4] = {HuffmanLut::annex_k(0), HuffmanLut::annex_k(1), HuffmanLut::annex_k(2),
                            HuffmanLut::annex_k(3)};
    Metadata &md = img->metadata;
    memset(&md, 0, sizeof md);

    bool have_frame = false, have_scan = false, have_dri = false;
    uint32_t dri = 0;
    uint8_t frame_ids[3] = {0, 0, 0}, hv[3] = {0, 0, 0}, tq[3] = {0, 0, 0};
    uint8_t td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    unsigned frame_w = 0, frame_h = 0;

    for (;;) {
        // next marker: skip to an FF, the byte after it names the segment
        uint8_t v;
        do {
            if (!file.byte(v))
                return malformed(kEof);
        } while (v != 0xff);
        uint8_t marker;
        if (!file.byte(marker))
            return malformed(kEof);
        if (marker == 0x00)
            return malformed("invalid ff 00 marker");
        if (marker == 0xd9)
            break;

        unsigned seglen;
        if (!file.be16(seglen))
            return malformed(kEof);
        if (seglen < 2)
            return malformed(fmt("invalid segment length %u", seglen));
        seglen -= 2;
        if (file.left() < seglen)
            return malformed(kEof);
        Cursor seg{jpeg, file.at + seglen, file.at};
        size_t resume = seg.end;

        if (marker == 0xdb) { // DQT: as many 65-byte tables as fit (file.rs:108-121)
            for (size_t n = seg.left() / 65; n; n--) {
                const uint8_t *t = jpeg + seg.at;
                seg.at += 65;
                const unsigned pq = t[0] >> 4, dst = t[0] & 15;
                if (pq != 0)
                    return unsupported(fmt(
                        "invalid quantization table precision Pq=%u (only 0 is allowed)", pq));
                if (dst > 3)
                    return unsupported(fmt(
                        "invalid quantization table destination Tq=%u (0-3 are allowed)", dst));
                for (int i = 0; i < 64; i++)
                    md.qtables[dst][i] = t[1 + i];
            }
        } else if (marker == 0xc4) { // DHT (file.rs:123-138, lib.rs:701-720)
            while (seg.left() >= 18) {
                const uint8_t *h = jpeg + seg.at;
                seg.at += 17;
                size_t nsym = 0;
                for (int i = 0; i < 16; i++)
                    nsym += h[1 + i];
                if (seg.left() < nsym)
                    return malformed(kEof);
                // The reference collects every table of the segment before
                // judging any of them, so a truncated later table wins over a
                // bad earlier one; validate in a second sweep below.
                seg.at += nsym;
            }
            Cursor again{jpeg, seg.end, file.at};
            while (again.left() >= 18) {
                const uint8_t *h = jpeg + again.at;
                size_t nsym = 0;
                for (int i = 0; i < 16; i++)
                    nsym += h[1 + i];
                again.at += 17 + nsym;
                const unsigned th = h[0] & 15, tc = h[0] >> 4;
                if (th > 1)
                    return unsupported(
                        fmt("DHT Th=%u, only 0 and 1 are allowed for baseline JPEGs", th));
                if (tc > 1)
                    return unsupported(fmt("invalid table class Tc=%u (only 0 and 1 are valid)", tc));
                HuffmanLut lut;
                if (!lut.build(h + 1, h + 17, nsym))
                    return malformed("malformed huffman table (not a prefix code)");
                tables[th << 1 | tc] = std::move(lut);
            }
        } else if (is_sof(marker)) { // file.rs:140-153, lib.rs:626-676
            uint8_t precision, ncomp;
            unsigned y, x;
            if (!seg.byte(precision) || !seg.be16(y) || !seg.be16(x) || !seg.byte(ncomp))
                return malformed(kEof);
            if (seg.left() < size_t(ncomp) * 3)
                return malformed("frame header component list exceeds its segment");
            if (marker != 0xc0)
                return unsupported(fmt("not a baseline JPEG (SOF=%s)", sof_name(marker)));
            if (precision != 8)
                return unsupported(fmt("sample precision of %u bits is not supported", precision));
            if (have_frame)
                return unsupported("encountered multiple SOF markers");
            if (ncomp != 3)
                return unsupported(fmt(
                    "frame with %u components not supported (only 3 components are supported)",
                    ncomp));
            const uint8_t *c = jpeg + seg.at;
            for (int i = 0; i < 3; i++) {
                frame_ids[i] = c[i * 3];
                hv[i] = c[i * 3 + 1];
                tq[i] = c[i * 3 + 2];
            }
            if (tq[0] > 3 || tq[1] > 3 || tq[2] > 3)
                return unsupported(fmt(
                    "invalid quantization table selection [%u,%u,%u] (only tables 0-3 are valid)",
                    tq[0], tq[1], tq[2]));
            const bool any_luma = (flags & COMPEG_PARSE_ANY_LUMA_SAMPLING) &&
                                  (hv[0] == 0x11 || hv[0] == 0x21 || hv[0] == 0x12 || hv[0] == 0x22);
            if (hv[0] != 0x21 && !any_luma)
                return unsupported(
                    fmt("invalid sampling factors %ux%u for Y component (expected 2x1)",
                        hv[0] >> 4, hv[0] & 15));
            if (hv[1] != 0x11 || hv[2] != 0x11)
                return unsupported(
                    fmt("invalid U/V sampling factors %ux%u and %ux%u (expected 1x1)", hv[1] >> 4,
                        hv[1] & 15, hv[2] >> 4, hv[2] & 15));
            have_frame = true;
            frame_w = x;
            frame_h = y;
        } else if (marker == 0xda) { // file.rs:155-209, lib.rs:726-756
            uint8_t ncomp;
            if (!seg.byte(ncomp))
                return malformed(kEof);
            if (seg.left() < size_t(ncomp) * 2)
                return malformed("scan header component list exceeds its segment");
            const uint8_t *c = jpeg + seg.at;
            seg.at += size_t(ncomp) * 2;
            uint8_t ss, se, ahal;
            if (!seg.byte(ss) || !seg.byte(se) || !seg.byte(ahal))
                return malformed(kEof);
            const size_t data_start = seg.at;
            size_t data_end;
            if ((flags & kParseDeferScanEnd) && len >= data_start + 2 && jpeg[len - 2] == 0xff && jpeg[len - 1] == 0xd9) {
                data_end = len - 2; // (where the search ends if no other marker lies in between: front.h)
                img->scan_end_deferred = true;
            } else if (!find_scan_end(jpeg, len, data_start, data_end))
                return malformed(kEof);
            if (ss != 0 || se != 63 || ahal != 0)
                return unsupported("non-baseline scan header");
            if (!have_frame)
                return unsupported("SOS not preceded by SOF header");
            if (ncomp != 3)
                return unsupported(fmt(
                    "scan with %u components not supported (only 3 components are supported)",
                    ncomp));
            if (c[0] != frame_ids[0] || c[2] != frame_ids[1] || c[4] != frame_ids[2])
                return unsupported(
                    fmt("scan component index mismatch (expected component order [%u, %u, %u], "
                        "got [%u, %u, %u])",
                        frame_ids[0], frame_ids[1], frame_ids[2], c[0], c[2], c[4]));
            for (int i = 0; i < 3; i++) {
                td[i] = c[i * 2 + 1] >> 4;
                ta[i] = c[i * 2 + 1] & 15;
            }
            img->scan_offset = data_start;
            img->scan_len = data_end - data_start;
            have_scan = true; // a later SOS replaces this one (lib.rs:753)
            if (data_end > resume)
                resume = data_end;
        } else if (marker == 0xdd) {
            if (!seg.be16(dri))
                return malformed(kEof);
            have_dri = true;
        } else if (marker == 0xe0) {
            // APP0: the reference decodes a JFIF header here and fails on a
            // bad density unit or a short thumbnail (file.rs:226-268).
            if (seg.left() >= 5 && memcmp(jpeg + seg.at, "JFIF\0", 5) == 0) {
                seg.at += 5;
                uint8_t f[9];
                for (int i = 0; i < 3; i++)
                    if (!seg.byte(f[i]))
                        return malformed(kEof);
                if (f[2] > 2)
                    return malformed(fmt("JFIF header specifies invalid density unit %u", f[2]));
                for (int i = 3; i < 9; i++)
                    if (!seg.byte(f[i]))
                        return malformed(kEof);
                if (seg.left() < size_t(f[7]) * f[8] * 3)
                    return malformed(kEof);
            }
        }
        // every other segment (APPn, COM, unknown) is skipped by its length
        file.at = resume;
    }

    if (!have_frame || !have_scan)
        return unsupported("missing SOS/SOI marker");

    if (frame_w + 7 > 0xffff || frame_h + 7 > 0xffff)
        return malformed("image dimensions overflow 16-bit arithmetic");
    // 2, 1, 4 for the only layout the reference accepts (Y 2x1, Cb/Cr 1x1)
    md.max_hsample = hv[0] >> 4;
    md.max_vsample = hv[0] & 15;
    md.dus_per_mcu = md.max_hsample * md.max_vsample + 2;
    const uint32_t width_dus = (frame_w + 7) / 8, height_dus = (frame_h + 7) / 8;
    md.width_mcus = (width_dus + md.max_hsample - 1) / md.max_hsample;
    const uint32_t height_mcus = (height_dus + md.max_vsample - 1) / md.max_vsample;
    const uint32_t mcus = md.width_mcus * height_mcus;
    md.restart_interval = have_dri ? dri : mcus;
    if (md.restart_interval == 0)
        return malformed("restart interval of 0 MCUs (empty image or DRI with Ri=0)");
    md.total_restart_intervals = mcus / md.restart_interval;
    if (md.total_restart_intervals > kMaxRestartIntervals)
        return unsupported(fmt("number of restart intervals exceeds limit (%u > %u)",
                               md.total_restart_intervals, kMaxRestartIntervals));
    for (int i = 0; i < 3; i++) {
        md.components[i].hsample = hv[i] >> 4;
        md.components[i].vsample = hv[i] & 15;
        md.components[i].qtable = tq[i];
        md.components[i].dchuff = uint32_t(td[i]) << 1;
        md.components[i].achuff = uint32_t(ta[i]) << 1 | 1;
    }
    md.retained_coefficients = kRetainedCoefficients;
    img->width = frame_w;
    img->height = frame_h;

    // Concatenate the four LUTs: [Th0 DC, Th0 AC, Th1 DC, Th1 AC]; delegates
    // of later tables are rebased by the L2 entries in front (huffman.rs:247-271).
    size_t rebase = 0;
    for (int t = 0; t < 4; t++) {
        if (t > 0 && rebase > 0xffff)
            return malformed("huffman tables exceed the 15-bit L2 index space");
        for (int i = 0; i < 256; i++) {
            uint16_t e = tables[t].l1[i];
            if (e & 0x8000) {
                const size_t idx = size_t(e & 0x7fff) + rebase;
                if (idx > 0x7fff)
                    return malformed("huffman tables exceed the 15-bit L2 index space");
                e = uint16_t(0x8000u | idx);
            }
            img->l1[t * 256 + i] = e;
        }
        img->l2.insert(img->l2.end(), tables[t].l2.begin(), tables[t].l2.end());
        rebase += tables[t].l2.size();
    }

    // 11-bit direct table for the AC codes (tables 1 and 3): an entry is the
    // two-level lookup's result when that does not depend on the bits behind
    // the prefix, i.e. for codes of at most 11 bits (and for prefixes no code
    // starts with); longer codes escape to the two-level tables.
    img->flags = flags & ~kParseDeferScanEnd;
    const uint32_t zrl_advance = (flags & COMPEG_PARSE_STANDARD_ENTROPY) ? 16u : 17u; // quirk Q2
    img->ac_fast.assign(2 * kFastEntries, uint16_t(kFastEscape));
    for (int t = 0; t < 2; t++) {
        const uint16_t *l1 = img->l1 + (2 * t + 1) * 256;
        for (uint32_t x = 0; x < kFastEntries; x++) {
            const uint16_t e1 = l1[x >> (kFastBits - 8)];
            uint16_t e = uint16_t(kFastEscape);
            if (!(e1 & 0x8000)) {
                e = uint16_t(fast_entry(e1, zrl_advance));
            } else {
                // all 16-bit continuations of this prefix must agree
                const uint32_t lowbits = 16 - kFastBits;
                const size_t first = size_t(e1 & 0x7fff) + ((x << lowbits) & 0xff);
                uint16_t v = first < img->l2.size() ? img->l2[first] : 0;
                bool same = true;
                for (uint32_t i = 1; i < (1u << lowbits) && same; i++) {
                    const size_t idx = first + i;
                    same = (idx < img->l2.size() ? img->l2[idx] : 0) == v;
                }
                if (same && (v >> 8) <= kFastBits)
                    e = uint16_t(fast_entry(v, zrl_advance));
            }
            img->ac_fast[size_t(t) * kFastEntries + x] = e;
        }
    }

    // Direct DC tables for the cooperative kernel, same idea: 9-bit prefixes of the two DC tables.
    img->dc_fast.assign(2 * kDcFastEntries, uint16_t(kFastEscape));
    for (int t = 0; t < 2; t++) {
        const uint16_t *l1 = img->l1 + (2 * t) * 256;
        for (uint32_t x = 0; x < kDcFastEntries; x++) {
            const uint16_t e1 = l1[x >> (kDcFastBits - 8)];
            uint16_t v = e1;
            bool usable = !(e1 & 0x8000);
            if (!usable) {
                // all 16-bit continuations of this prefix must agree
                const uint32_t lowbits = 16 - kDcFastBits;
                const size_t first = size_t(e1 & 0x7fff) + ((x << lowbits) & 0xff);
                v = first < img->l2.size() ? img->l2[first] : 0;
                usable = true;
                for (uint32_t i = 1; i < (1u << lowbits) && usable; i++) {
                    const size_t idx = first + i;
                    usable = (idx < img->l2.size() ? img->l2[idx] : 0) == v;
                }
                usable = usable && (v >> 8) <= kDcFastBits;
            }
            const uint32_t len = v >> 8, cat = v & 0xffu;
            if (usable && cat <= 15u && len + cat <= 31u)
                img->dc_fast[size_t(t) * kDcFastEntries + x] = uint16_t((1u << 9) | ((len + cat) << 4) | cat);
        }
    }

    guard.p = nullptr;
    *out = img;
    return Status{};
}

} // namespace compeg
