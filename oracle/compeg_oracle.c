/*
 * compeg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See compeg_oracle.h for the rules on who may use this file.
 *
 * Build with strict IEEE f32: -O2 -ffp-contract=off, never -ffast-math.
 * "ref:" comments cite the upstream file:line a function restates.
 */
#include "compeg_oracle.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* scan preprocessing -- ref: src/scan.rs:15-128                             */
/* ======================================================================== */

struct orc_scanbuf {
    /* Both vectors keep their old contents on resize (only growth is
     * zero-filled), like Vec::resize in the reference (quirk Q7). */
    uint32_t *words;
    size_t words_len, words_cap;
    uint32_t *starts;
    size_t starts_len, starts_cap;
};

static void vec_resize_u32(uint32_t **p, size_t *len, size_t *cap, size_t n)
{
    if (n > *cap) {
        size_t ncap = *cap ? *cap : 16;
        while (ncap < n)
            ncap *= 2;
        *p = (uint32_t *)realloc(*p, ncap * sizeof(uint32_t));
        *cap = ncap;
    }
    if (n > *len)
        memset(*p + *len, 0, (n - *len) * sizeof(uint32_t));
    *len = n;
}

orc_scanbuf *orc_scanbuf_new(void)
{
    return (orc_scanbuf *)calloc(1, sizeof(orc_scanbuf));
}

void orc_scanbuf_free(orc_scanbuf *sb)
{
    if (!sb)
        return;
    free(sb->words);
    free(sb->starts);
    free(sb);
}

static size_t next_pow2(size_t v)
{
    size_t p = 1;
    while (p < v)
        p <<= 1;
    return p; /* next_power_of_two(0) == 1 in Rust as well */
}

/* ref: src/scan.rs:33-66 (process) and :85-128 (preprocess_scalar) */
int orc_scanbuf_process(orc_scanbuf *sb, const uint8_t *scan, size_t len,
                        uint32_t expected_intervals, char *err)
{
    size_t out_bytes = len + len / 3;
    vec_resize_u32(&sb->words, &sb->words_len, &sb->words_cap, (out_bytes + 3) / 4);

    size_t splen = next_pow2((size_t)expected_intervals);
    size_t mask = splen - 1;
    vec_resize_u32(&sb->starts, &sb->starts_len, &sb->starts_cap, splen);

    uint8_t *out = (uint8_t *)sb->words;
    size_t ri = 1; /* entry 0 is the implicit first interval */
    size_t wp = 0;
    size_t i = 0;
    while (i < len) {
        uint8_t b = scan[i++];
        if (b != 0xff) {
            out[wp++] = b;
            continue;
        }
        if (i >= len)
            break; /* lone trailing FF is dropped */
        uint8_t m = scan[i++];
        if (m == 0x00) {
            out[wp++] = 0xff;
        } else {
            /* any other FF xx counts as RSTn: pad to a word, note its start */
            wp = (wp + 3) & ~(size_t)3;
            sb->starts[ri & mask] = (uint32_t)(wp / 4);
            ri++;
        }
    }

    /* truncate() never grows */
    size_t nwords = (wp + 3) / 4;
    if (nwords < sb->words_len)
        sb->words_len = nwords;
    if (ri < sb->starts_len)
        sb->starts_len = ri;

    if (ri != (size_t)expected_intervals) {
        if (err)
            snprintf(err, ORC_ERRLEN,
                     "restart interval count mismatch: counted %zu, expected %u", ri,
                     expected_intervals);
        return 1;
    }
    return 0;
}

const uint8_t *orc_scanbuf_data(const orc_scanbuf *sb, size_t *nbytes)
{
    *nbytes = sb->words_len * 4;
    return (const uint8_t *)sb->words;
}

const uint8_t *orc_scanbuf_starts(const orc_scanbuf *sb, size_t *nbytes)
{
    *nbytes = sb->starts_len * 4;
    return (const uint8_t *)sb->starts;
}

/* ======================================================================== */
/* Huffman LUTs -- ref: src/huffman.rs:23-119 (build), :179-190 (lookup)      */
/* entry encoding (little-endian u16): bits<<8 | value; L1 delegate =         */
/* 0x8000 | index of the prefix's 256-entry block in L2 (ref :282-319)        */
/* ======================================================================== */

struct orc_table {
    uint16_t l1[256];
    uint16_t *l2;
    size_t l2_len; /* entries */
};

orc_table *orc_table_build(const uint8_t li[16], const uint8_t *vij, size_t nvij)
{
    /* slot i: either an immediate L1 entry or a private 256-entry block */
    uint16_t imm[256];
    uint16_t *blk[256];
    memset(imm, 0, sizeof imm);
    memset(blk, 0, sizeof blk);

    int bad = 0;
    uint32_t code = 0; /* u16 in the reference: shifts drop bits, += 1 wraps (release build) */
    size_t k = 0;
    for (int len = 1; len <= 16 && !bad; len++) {
        code = (code << 1) & 0xffff;
        for (unsigned c = 0; c < li[len - 1] && !bad; c++) {
            if (k >= nvij) {
                bad = 1;
                break;
            }
            uint16_t res = (uint16_t)((len << 8) | vij[k++]);
            if (len <= 8) {
                uint32_t padded = (code << (8 - len)) & 0xffff;
                uint32_t copies = 1u << (8 - len);
                for (uint32_t l = 0; l < copies; l++) {
                    uint32_t idx = padded | l;
                    if (idx > 255) {
                        bad = 1;
                        break;
                    }
                    if (blk[idx]) { /* reference overwrites the Slot; L2 block dropped */
                        free(blk[idx]);
                        blk[idx] = NULL;
                    }
                    imm[idx] = res;
                }
            } else {
                uint32_t msb = code >> (len - 8);
                if (msb > 255) {
                    bad = 1;
                    break;
                }
                if (!blk[msb]) {
                    if (imm[msb] != 0) { /* assert_eq!(entry, NULL) */
                        bad = 1;
                        break;
                    }
                    blk[msb] = (uint16_t *)calloc(256, sizeof(uint16_t));
                }
                uint32_t padded = (code << (16 - len)) & 0xffff;
                uint32_t lsb = padded & 0xff;
                uint32_t copies = 1u << (16 - len);
                for (uint32_t l = 0; l < copies; l++) {
                    uint32_t idx = lsb | l;
                    if (idx > 255 || blk[msb][idx] != 0) {
                        bad = 1;
                        break;
                    }
                    blk[msb][idx] = res;
                }
            }
            code = (code + 1) & 0xffff; /* release-build wrap; debug builds panic here */
        }
    }

    orc_table *t = NULL;
    if (!bad) {
        t = (orc_table *)calloc(1, sizeof *t);
        size_t nblk = 0;
        for (int i = 0; i < 256; i++)
            nblk += blk[i] != NULL;
        if (nblk * 256 > 0x7fff + 1) /* delegate index is 15 bits */
            bad = 1;
        t->l2 = (uint16_t *)malloc((nblk ? nblk : 1) * 256 * sizeof(uint16_t));
        for (int i = 0; i < 256 && !bad; i++) {
            if (blk[i]) {
                if (t->l2_len > 0x7fff) {
                    bad = 1;
                    break;
                }
                t->l1[i] = (uint16_t)(0x8000 | t->l2_len);
                memcpy(t->l2 + t->l2_len, blk[i], 256 * sizeof(uint16_t));
                t->l2_len += 256;
            } else {
                t->l1[i] = imm[i];
            }
        }
        if (bad) {
            orc_table_free(t);
            t = NULL;
        }
    }
    for (int i = 0; i < 256; i++)
        free(blk[i]);
    return t;
}

void orc_table_free(orc_table *t)
{
    if (!t)
        return;
    free(t->l2);
    free(t);
}

uint16_t orc_table_lookup(const orc_table *t, uint16_t code)
{
    uint16_t e = t->l1[code >> 8];
    if (e & 0x8000)
        e = t->l2[(size_t)(e & 0x7fff) + (code & 0xff)];
    return e;
}

size_t orc_table_l2_len(const orc_table *t)
{
    return t->l2_len;
}

static size_t sappend(char *out, size_t cap, size_t pos, const char *fmt, ...)
{
    char tmp[128];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (n < 0)
        n = 0;
    for (int i = 0; i < n; i++) {
        if (pos + 1 < cap)
            out[pos] = tmp[i];
        pos++;
    }
    if (cap)
        out[pos < cap ? pos : cap - 1] = 0;
    return pos;
}

/* ref: src/huffman.rs:192-232 (iter + Debug): walk the 16-bit code space,
 * print each distinct code as "<bits> -> <value hex>". */
size_t orc_table_debug(const orc_table *t, char *out, size_t cap)
{
    size_t pos = 0;
    uint32_t cur = 0;
    int first = 1;
    if (cap)
        out[0] = 0;
    while (cur <= 0xffff) {
        uint16_t e = orc_table_lookup(t, (uint16_t)cur);
        unsigned bits = e >> 8;
        if (bits == 0) {
            cur++;
            continue;
        }
        if (!first)
            pos = sappend(out, cap, pos, "\n");
        first = 0;
        uint32_t code = cur >> (16 - bits);
        for (int b = (int)bits - 1; b >= 0; b--)
            pos = sappend(out, cap, pos, "%c", ((code >> b) & 1) ? '1' : '0');
        pos = sappend(out, cap, pos, " -> %02x", e & 0xff);
        cur += 1u << (16 - bits);
    }
    return pos;
}

/* Annex K tables -- ref: src/huffman.rs:121-177 (values are from ITU T.81 K.3) */
static const uint8_t K_DC_L_LI[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t K_DC_C_LI[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t K_DC_V[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t K_AC_L_LI[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const uint8_t K_AC_C_LI[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const uint8_t K_AC_L_V[162] = {
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
static const uint8_t K_AC_C_V[162] = {
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

orc_table *orc_table_default(int which)
{
    switch (which) {
    case 0:
        return orc_table_build(K_DC_L_LI, K_DC_V, 12);
    case 1:
        return orc_table_build(K_AC_L_LI, K_AC_L_V, 162);
    case 2:
        return orc_table_build(K_DC_C_LI, K_DC_V, 12);
    case 3:
        return orc_table_build(K_AC_C_LI, K_AC_C_V, 162);
    }
    return NULL;
}

/* ======================================================================== */
/* bit reader -- ref: src/huffman.wgsl:35-79 (authoritative),                */
/* src/bits.rs:18-67 (CPU twin).  WGSL semantics: all u32 arithmetic wraps,  */
/* shift counts are taken modulo 32, out-of-bounds buffer reads yield 0.     */
/* ======================================================================== */

#define SHL(x, n) ((uint32_t)(x) << ((n) & 31u))
#define SHR(x, n) ((uint32_t)(x) >> ((n) & 31u))

void orc_bits_init(orc_bits *b, const uint32_t *words, size_t nwords, uint32_t start)
{
    b->words = words;
    b->nwords = nwords;
    b->next_word = start;
    b->cur = b->next = b->left = 0;
    orc_bits_refill(b);
}

void orc_bits_refill(orc_bits *b)
{
    if (b->left < 32u) {
        uint32_t w = b->next_word < b->nwords ? b->words[b->next_word] : 0u;
        w = (w & 0xffu) << 24 | (w & 0xff00u) << 8 | (w & 0xff0000u) >> 8 | (w & 0xff000000u) >> 24;
        b->next_word += 1u;
        b->cur |= SHR(w, b->left);
        b->next = SHL(SHL(w, 1u), 31u - b->left);
        b->left += 32u;
    }
}

void orc_bits_consume(orc_bits *b, uint32_t n)
{
    b->cur = SHL(b->cur, n);
    b->cur |= SHR(SHR(b->next, 1u), 31u - n);
    b->next = SHL(b->next, n);
    b->left -= n;
}

uint32_t orc_bits_peek(const orc_bits *b, uint32_t n)
{
    return SHR(SHR(b->cur, 1u), 31u - n);
}

uint32_t orc_bits_huffdecode_table(orc_bits *b, const orc_table *t)
{
    uint16_t e = orc_table_lookup(t, (uint16_t)(b->cur >> 16));
    orc_bits_consume(b, e >> 8);
    return e & 0xffu;
}

/* ref: src/huffman.wgsl:213-216 */
int32_t orc_huff_extend(int32_t v, uint32_t t)
{
    int32_t vt = (int32_t)SHL(1u, t - 1u);
    if (v < vt)
        return (int32_t)((uint32_t)v + SHL(0xffffffffu, t) + 1u);
    return v;
}

/* ======================================================================== */
/* segment parser -- ref: src/file.rs:13-355                                 */
/* ======================================================================== */

typedef struct {
    const uint8_t *buf;
    size_t len; /* readable limit */
    size_t pos;
} rd_t;

#define E_EOF "reached end of data while decoding JPEG stream"

typedef struct {
    char msg[ORC_ERRLEN];
    int failed;
} perr_t;

static int fail(perr_t *e, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e->msg, sizeof e->msg, fmt, ap);
    va_end(ap);
    e->failed = 1;
    return -1;
}

static int rd_peek(const rd_t *r, size_t off, perr_t *e)
{
    if (r->pos + off >= r->len)
        return fail(e, E_EOF);
    return r->buf[r->pos + off];
}

static int rd_u8(rd_t *r, perr_t *e)
{
    int v = rd_peek(r, 0, e);
    if (v >= 0)
        r->pos++;
    return v;
}

static int rd_u16(rd_t *r, perr_t *e)
{
    int a = rd_u8(r, e);
    if (a < 0)
        return -1;
    int b = rd_u8(r, e);
    if (b < 0)
        return -1;
    return a << 8 | b;
}

typedef struct {
    uint8_t ci, hi, vi, tqi;
} frame_comp_t;
typedef struct {
    uint8_t csj, tdj, taj;
} scan_comp_t;

enum { K_NONE, K_DQT, K_DHT, K_DRI, K_SOF, K_SOS, K_APP, K_COM };

typedef struct {
    size_t offset;
    uint8_t marker;
    int kind;
    const uint8_t *raw;
    size_t raw_len;
    /* DQT */
    size_t dqt_count;
    const uint8_t *dqt; /* dqt_count records of 65 bytes */
    /* DHT: tables back to back inside [dht, dht_end) */
    size_t dht_count;
    const uint8_t *dht;
    /* DRI */
    uint16_t ri;
    /* SOF */
    uint8_t p;
    uint16_t y, x;
    size_t ncomp;
    const uint8_t *comps;
    /* SOS */
    uint8_t ss, se, ahal;
    size_t data_off, data_len;
    /* APP */
    int jfif;
    uint8_t jf[7 + 2]; /* major, minor, unit, xd(2), yd(2), xt, yt */
    const uint8_t *thumb;
    size_t thumb_len;
    /* COM */
    const uint8_t *com;
    size_t com_len;
} seg_t;

typedef struct {
    rd_t r;
} parser_t;

static int parser_new(parser_t *p, const uint8_t *buf, size_t len, perr_t *e)
{
    p->r.buf = buf;
    p->r.len = len;
    p->r.pos = 0;
    int a = rd_u8(&p->r, e);
    if (a < 0)
        return -1;
    if (a != 0xff)
        return fail(e, "JPEG image does not start with SOI marker");
    int b = rd_u8(&p->r, e);
    if (b < 0)
        return -1;
    if (b != 0xd8)
        return fail(e, "JPEG image does not start with SOI marker");
    return 0;
}

/* returns 1 = segment produced, 0 = EOI, -1 = error (or "panic: ...") */
static int parser_next(parser_t *p, seg_t *s, perr_t *e)
{
    memset(s, 0, sizeof *s);
    int v;
    do {
        v = rd_u8(&p->r, e);
        if (v < 0)
            return -1;
    } while (v != 0xff);
    s->offset = p->r.pos - 1;
    int marker = rd_u8(&p->r, e);
    if (marker < 0)
        return -1;
    if (marker == 0x00)
        return fail(e, "invalid ff 00 marker");
    if (marker == 0xd9)
        return 0;
    s->marker = (uint8_t)marker;

    int l = rd_u16(&p->r, e);
    if (l < 0)
        return -1;
    if (l < 2)
        return fail(e, "invalid segment length %d", l);
    size_t length = (size_t)l - 2;
    if (p->r.len - p->r.pos < length)
        return fail(e, E_EOF);
    size_t end = p->r.pos + length;
    rd_t r = {p->r.buf, end, p->r.pos};
    s->raw = p->r.buf + s->offset + 4;
    s->raw_len = length;

    switch (marker) {
    case 0xdb: { /* ref: file.rs:108-121 */
        s->kind = K_DQT;
        s->dqt_count = (r.len - r.pos) / 65;
        s->dqt = r.buf + r.pos;
        r.pos += s->dqt_count * 65;
        break;
    }
    case 0xc4: { /* ref: file.rs:123-138 */
        s->kind = K_DHT;
        s->dht = r.buf + r.pos;
        while (r.len - r.pos >= 18) {
            const uint8_t *h = r.buf + r.pos;
            r.pos += 17;
            size_t nv = 0;
            for (int i = 0; i < 16; i++)
                nv += h[1 + i];
            if (r.len - r.pos < nv)
                return fail(e, E_EOF);
            r.pos += nv;
            s->dht_count++;
        }
        break;
    }
    case 0xc0: case 0xc1: case 0xc2: case 0xc3: case 0xc5: case 0xc6: case 0xc7:
    case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf: { /* ref: file.rs:140-153 */
        s->kind = K_SOF;
        int pp = rd_u8(&r, e), yy = rd_u16(&r, e), xx = rd_u16(&r, e), n = rd_u8(&r, e);
        if (pp < 0 || yy < 0 || xx < 0 || n < 0)
            return -1;
        if (r.len - r.pos < (size_t)n * 3)
            return fail(e, "panic: frame component list exceeds segment");
        s->p = (uint8_t)pp;
        s->y = (uint16_t)yy;
        s->x = (uint16_t)xx;
        s->ncomp = (size_t)n;
        s->comps = r.buf + r.pos;
        r.pos += (size_t)n * 3;
        break;
    }
    case 0xda: { /* ref: file.rs:155-209 */
        s->kind = K_SOS;
        int n = rd_u8(&r, e);
        if (n < 0)
            return -1;
        if (r.len - r.pos < (size_t)n * 2)
            return fail(e, "panic: scan component list exceeds segment");
        s->ncomp = (size_t)n;
        s->comps = r.buf + r.pos;
        r.pos += (size_t)n * 2;
        int ss = rd_u8(&r, e), se = rd_u8(&r, e), ah = rd_u8(&r, e);
        if (ss < 0 || se < 0 || ah < 0)
            return -1;
        s->ss = (uint8_t)ss;
        s->se = (uint8_t)se;
        s->ahal = (uint8_t)ah;
        /* entropy-coded data follows the header, not the declared length */
        p->r.pos = r.pos;
        size_t start = p->r.pos;
        for (;;) {
            int b;
            while ((b = rd_peek(&p->r, 0, e)) != 0xff) {
                if (b < 0)
                    return -1;
                p->r.pos++;
            }
            size_t off = 1;
            b = rd_peek(&p->r, off, e);
            if (b < 0)
                return -1;
            while (b == 0xff) {
                off++;
                b = rd_peek(&p->r, off, e);
                if (b < 0)
                    return -1;
            }
            if (b == 0x00 || (b >= 0xd0 && b <= 0xd7)) {
                p->r.pos += off + 1;
            } else {
                p->r.pos += off - 1;
                break;
            }
        }
        s->data_off = start;
        s->data_len = p->r.pos - start;
        break;
    }
    case 0xdd: {
        s->kind = K_DRI;
        int ri = rd_u16(&r, e);
        if (ri < 0)
            return -1;
        s->ri = (uint16_t)ri;
        break;
    }
    case 0xfe:
        s->kind = K_COM;
        s->com = r.buf + r.pos;
        s->com_len = r.len - r.pos;
        r.pos = r.len;
        break;
    default:
        if (marker >= 0xe0 && marker <= 0xef) { /* ref: file.rs:225-268 */
            s->kind = K_APP;
            if (marker == 0xe0 && r.len - r.pos >= 5 && memcmp(r.buf + r.pos, "JFIF\0", 5) == 0) {
                r.pos += 5;
                int f[9];
                for (int i = 0; i < 3; i++)
                    if ((f[i] = rd_u8(&r, e)) < 0)
                        return -1;
                if (f[2] > 2)
                    return fail(e, "JFIF header specifies invalid density unit %d", f[2]);
                for (int i = 3; i < 9; i++)
                    if ((f[i] = rd_u8(&r, e)) < 0)
                        return -1;
                size_t tl = (size_t)f[7] * (size_t)f[8] * 3;
                if (r.len - r.pos < tl)
                    return fail(e, E_EOF);
                for (int i = 0; i < 9; i++)
                    s->jf[i] = (uint8_t)f[i];
                s->jfif = 1;
                s->thumb = r.buf + r.pos;
                s->thumb_len = tl;
            }
            r.pos = r.len;
        } else {
            p->r.pos = end;
        }
        break;
    }
    if (r.pos < end)
        r.pos = end;
    if (r.pos > p->r.pos)
        p->r.pos = r.pos;
    return 1;
}

static size_t dump_bytes(char *out, size_t cap, size_t pos, const uint8_t *b, size_t n, int hex)
{
    pos = sappend(out, cap, pos, "[");
    for (size_t i = 0; i < n; i++)
        pos = sappend(out, cap, pos, hex ? "%s%x" : "%s%u", i ? ", " : "", b[i]);
    return sappend(out, cap, pos, "]");
}

static const char *sof_name(uint8_t m)
{
    static const char *names[16] = {"SOF0",  "SOF1",  "SOF2",  "SOF3", NULL,    "SOF5",
                                    "SOF6",  "SOF7",  NULL,    "SOF9", "SOF10", "SOF11",
                                    NULL,    "SOF13", "SOF14", "SOF15"};
    return names[m & 15];
}

/* ref: src/file/tests.rs:9-58 (dump) plus the Debug impls in src/file.rs */
size_t orc_parser_dump(const uint8_t *jpeg, size_t len, char *out, size_t cap)
{
    size_t pos = 0;
    perr_t e = {{0}, 0};
    parser_t p;
    if (cap)
        out[0] = 0;
    if (parser_new(&p, jpeg, len, &e) < 0)
        goto err;
    for (;;) {
        seg_t s;
        int rc = parser_next(&p, &s, &e);
        if (rc < 0)
            goto err;
        if (rc == 0)
            break;
        pos = sappend(out, cap, pos, "%04zX [FF %02X] ", s.offset, s.marker);
        switch (s.kind) {
        case K_NONE:
            pos = dump_bytes(out, cap, pos, s.raw, s.raw_len, 1);
            break;
        case K_DQT:
            pos = sappend(out, cap, pos, "DQT([");
            for (size_t i = 0; i < s.dqt_count; i++) {
                const uint8_t *q = s.dqt + i * 65;
                pos = sappend(out, cap, pos, "%sQuantizationTable { Pq: %u, Tq: %u, Qk: ",
                              i ? ", " : "", q[0] >> 4, q[0] & 15);
                pos = dump_bytes(out, cap, pos, q + 1, 64, 0);
                pos = sappend(out, cap, pos, " }");
            }
            pos = sappend(out, cap, pos, "])");
            break;
        case K_DHT: {
            pos = sappend(out, cap, pos, "DHT { tables: [");
            const uint8_t *h = s.dht;
            for (size_t i = 0; i < s.dht_count; i++) {
                size_t nv = 0;
                for (int k = 0; k < 16; k++)
                    nv += h[1 + k];
                pos = sappend(out, cap, pos, "%sHuffmanTable { Tc: %u, Th: %u, Li: ",
                              i ? ", " : "", h[0] >> 4, h[0] & 15);
                pos = dump_bytes(out, cap, pos, h + 1, 16, 0);
                pos = sappend(out, cap, pos, ", Vij: ");
                pos = dump_bytes(out, cap, pos, h + 17, nv, 0);
                pos = sappend(out, cap, pos, " }");
                h += 17 + nv;
            }
            pos = sappend(out, cap, pos, "] }");
            break;
        }
        case K_DRI:
            pos = sappend(out, cap, pos, "DRI { Ri: %u }", s.ri);
            break;
        case K_SOF:
            pos = sappend(out, cap, pos, "SOF { sof: %s, P: %u, Y: %u, X: %u, components: [",
                          sof_name(s.marker), s.p, s.y, s.x);
            for (size_t i = 0; i < s.ncomp; i++) {
                const uint8_t *c = s.comps + i * 3;
                pos = sappend(out, cap, pos,
                              "%sFrameComponent { Ci: %u, Hi: %u, Vi: %u, Tqi: %u }",
                              i ? ", " : "", c[0], c[1] >> 4, c[1] & 15, c[2]);
            }
            pos = sappend(out, cap, pos, "] }");
            break;
        case K_SOS:
            pos = sappend(out, cap, pos, "SOS { components: [");
            for (size_t i = 0; i < s.ncomp; i++) {
                const uint8_t *c = s.comps + i * 2;
                pos = sappend(out, cap, pos, "%sScanComponent { Csj: %u, Tdj: %u, Taj: %u }",
                              i ? ", " : "", c[0], c[1] >> 4, c[1] & 15);
            }
            pos = sappend(out, cap, pos, "], Ss: %u, Se: %u, Ah: %u, Al: %u, data: ", s.ss, s.se,
                          s.ahal >> 4, s.ahal & 15);
            pos = dump_bytes(out, cap, pos, jpeg + s.data_off, s.data_len, 0);
            pos = sappend(out, cap, pos, " }");
            break;
        case K_APP:
            if (s.jfif) {
                static const char *units[3] = {"None", "DotsPerInch", "DotsPerCm"};
                pos = sappend(out, cap, pos,
                              "APP { n: %u, kind: Some(JFIF(JFIF { major_version: %u, "
                              "minor_version: %u, unit: %s, ",
                              s.marker - 0xe0, s.jf[0], s.jf[1], units[s.jf[2]]);
                pos = sappend(out, cap, pos,
                              "xdensity: %u, ydensity: %u, xthumbnail: %u, ythumbnail: %u, "
                              "thumbnail: ",
                              s.jf[3] << 8 | s.jf[4], s.jf[5] << 8 | s.jf[6], s.jf[7], s.jf[8]);
                pos = dump_bytes(out, cap, pos, s.thumb, s.thumb_len, 0);
                pos = sappend(out, cap, pos, " })) }");
            } else {
                pos = sappend(out, cap, pos, "APP { n: %u, kind: None } ", s.marker - 0xe0);
                pos = dump_bytes(out, cap, pos, s.raw, s.raw_len, 1);
            }
            break;
        case K_COM:
            pos = sappend(out, cap, pos, "Com(\"");
            for (size_t i = 0; i < s.com_len; i++) {
                uint8_t c = s.com[i];
                if (c == '\t')
                    pos = sappend(out, cap, pos, "\\t");
                else if (c == '\r')
                    pos = sappend(out, cap, pos, "\\r");
                else if (c == '\n')
                    pos = sappend(out, cap, pos, "\\n");
                else if (c == '\'' || c == '"' || c == '\\')
                    pos = sappend(out, cap, pos, "\\%c", c);
                else if (c >= 0x20 && c < 0x7f)
                    pos = sappend(out, cap, pos, "%c", c);
                else
                    pos = sappend(out, cap, pos, "\\x%02x", c);
            }
            pos = sappend(out, cap, pos, "\")");
            break;
        }
        pos = sappend(out, cap, pos, "\n");
    }
    if (p.r.pos < p.r.len) {
        pos = sappend(out, cap, pos, "%zu trailing bytes: ", p.r.len - p.r.pos);
        pos = dump_bytes(out, cap, pos, p.r.buf + p.r.pos, p.r.len - p.r.pos, 1);
        pos = sappend(out, cap, pos, "\n");
    }
    return pos;
err:
    pos = sappend(out, cap, pos, "error: %s\n", e.msg);
    return pos;
}

/* ======================================================================== */
/* image front-end -- ref: src/lib.rs:597-824, src/metadata.rs:3-43          */
/* ======================================================================== */

#define MD_SIZE 1112
#define MD_RI 1024
#define MD_COMP 1028 /* 3 x {vsample,hsample,qtable,dchuff,achuff} */
#define MD_TOTAL 1088
#define MD_WMCU 1092
#define MD_MAXH 1096
#define MD_MAXV 1100
#define MD_DUS 1104
#define MD_RET 1108

static uint32_t md_get(const uint8_t *md, size_t off)
{
    uint32_t v;
    memcpy(&v, md + off, 4);
    return v;
}

static void md_put(uint8_t *md, size_t off, uint32_t v)
{
    memcpy(md + off, &v, 4);
}

struct orc_image {
    uint8_t md[MD_SIZE];
    uint16_t width, height;
    uint16_t l1[1024];
    uint16_t *l2;
    size_t l2_len;
    size_t scan_off, scan_len;
    unsigned flags; /* orc_image_parse_ext */
};

/* flags & 1 (extension, not the reference): luma sampling 1x1, 2x1, 1x2 or 2x2 is accepted, i.e.
 * 4:4:4, 4:2:2, 4:4:0 and 4:2:0; see orc_finalize_pass for what that means for the output.
 * flags & 2 (extension): entropy decoding as ITU-T T.81 has it where the reference deviates --
 * the reader is refilled in front of a DC code as well (quirk Q1) and ZRL skips 16 positions, not
 * 17 (quirk Q2); see orc_huffman_pass_ext. */
orc_image *orc_image_parse(const uint8_t *jpeg, size_t len, char *err)
{
    return orc_image_parse_ext(jpeg, len, 0, err);
}

orc_image *orc_image_parse_ext(const uint8_t *jpeg, size_t len, unsigned flags, char *err)
{
    perr_t e = {{0}, 0};
    orc_table *tables[4] = {orc_table_default(0), orc_table_default(1), orc_table_default(2),
                            orc_table_default(3)};
    uint32_t q[4][64];
    memset(q, 0, sizeof q);
    int have_size = 0, have_ri = 0, have_scan = 0, have_comps = 0;
    uint32_t ri = 0;
    uint16_t width = 0, height = 0;
    frame_comp_t fc[3];
    uint8_t dch[3] = {0, 0, 0}, ach[3] = {0, 0, 0};
    size_t scan_off = 0, scan_len = 0;
    orc_image *img = NULL;

    parser_t p;
    if (parser_new(&p, jpeg, len, &e) < 0)
        goto out;
    for (;;) {
        seg_t s;
        int rc = parser_next(&p, &s, &e);
        if (rc < 0)
            goto out;
        if (rc == 0)
            break;
        switch (s.kind) {
        case K_SOF: {
            if (s.marker != 0xc0) {
                fail(&e, "not a baseline JPEG (SOF=%s)", sof_name(s.marker));
                goto out;
            }
            if (s.p != 8) {
                fail(&e, "sample precision of %u bits is not supported", s.p);
                goto out;
            }
            if (have_comps) {
                fail(&e, "encountered multiple SOF markers");
                goto out;
            }
            if (s.ncomp != 3) {
                fail(&e,
                     "frame with %zu components not supported (only 3 components are supported)",
                     s.ncomp);
                goto out;
            }
            for (int i = 0; i < 3; i++) {
                const uint8_t *c = s.comps + i * 3;
                fc[i].ci = c[0];
                fc[i].hi = c[1] >> 4;
                fc[i].vi = c[1] & 15;
                fc[i].tqi = c[2];
            }
            if (fc[0].tqi > 3 || fc[1].tqi > 3 || fc[2].tqi > 3) {
                fail(&e,
                     "invalid quantization table selection [%u,%u,%u] (only tables 0-3 are valid)",
                     fc[0].tqi, fc[1].tqi, fc[2].tqi);
                goto out;
            }
            if ((flags & 1u) && fc[0].hi >= 1 && fc[0].hi <= 2 && fc[0].vi >= 1 && fc[0].vi <= 2) {
                /* extension: any of the four luma samplings */
            } else if (fc[0].hi != 2 || fc[0].vi != 1) {
                fail(&e, "invalid sampling factors %ux%u for Y component (expected 2x1)",
                     fc[0].hi, fc[0].vi);
                goto out;
            }
            if (fc[1].hi != fc[2].hi || fc[1].vi != fc[2].vi || fc[1].hi != 1 || fc[1].vi != 1) {
                fail(&e, "invalid U/V sampling factors %ux%u and %ux%u (expected 1x1)", fc[1].hi,
                     fc[1].vi, fc[2].hi, fc[2].vi);
                goto out;
            }
            have_comps = 1;
            width = s.x;
            height = s.y;
            have_size = 1;
            break;
        }
        case K_DQT:
            for (size_t i = 0; i < s.dqt_count; i++) {
                const uint8_t *t = s.dqt + i * 65;
                if ((t[0] >> 4) != 0) {
                    fail(&e, "invalid quantization table precision Pq=%u (only 0 is allowed)",
                         t[0] >> 4);
                    goto out;
                }
                if ((t[0] & 15) > 3) {
                    fail(&e, "invalid quantization table destination Tq=%u (0-3 are allowed)",
                         t[0] & 15);
                    goto out;
                }
                for (int k = 0; k < 64; k++)
                    q[t[0] & 15][k] = t[1 + k];
            }
            break;
        case K_DHT: {
            const uint8_t *h = s.dht;
            for (size_t i = 0; i < s.dht_count; i++) {
                size_t nv = 0;
                for (int k = 0; k < 16; k++)
                    nv += h[1 + k];
                unsigned th = h[0] & 15, tc = h[0] >> 4;
                if (th > 1) {
                    fail(&e, "DHT Th=%u, only 0 and 1 are allowed for baseline JPEGs", th);
                    goto out;
                }
                if (tc > 1) {
                    fail(&e, "invalid table class Tc=%u (only 0 and 1 are valid)", tc);
                    goto out;
                }
                orc_table *t = orc_table_build(h + 1, h + 17, nv);
                if (!t) {
                    fail(&e, "panic: malformed huffman table");
                    goto out;
                }
                orc_table_free(tables[th << 1 | tc]);
                tables[th << 1 | tc] = t;
                h += 17 + nv;
            }
            break;
        }
        case K_DRI:
            ri = s.ri;
            have_ri = 1;
            break;
        case K_SOS: {
            if (s.ss != 0 || s.se != 63 || s.ahal != 0) {
                fail(&e, "non-baseline scan header");
                goto out;
            }
            if (!have_comps) {
                fail(&e, "SOS not preceded by SOF header");
                goto out;
            }
            if (s.ncomp != 3) {
                fail(&e,
                     "scan with %zu components not supported (only 3 components are supported)",
                     s.ncomp);
                goto out;
            }
            const uint8_t *c = s.comps;
            if (c[0] != fc[0].ci || c[2] != fc[1].ci || c[4] != fc[2].ci) {
                fail(&e,
                     "scan component index mismatch (expected component order [%u, %u, %u], got "
                     "[%u, %u, %u])",
                     fc[0].ci, fc[1].ci, fc[2].ci, c[0], c[2], c[4]);
                goto out;
            }
            for (int i = 0; i < 3; i++) {
                dch[i] = c[i * 2 + 1] >> 4;
                ach[i] = c[i * 2 + 1] & 15;
            }
            scan_off = s.data_off;
            scan_len = s.data_len;
            have_scan = 1;
            break;
        }
        default:
            break;
        }
    }
    if (!have_size || !have_comps || !have_scan) {
        fail(&e, "missing SOS/SOI marker");
        goto out;
    }

    {
        uint32_t dus_per_mcu = 0, max_h = 0, max_v = 0;
        for (int i = 0; i < 3; i++) {
            dus_per_mcu += (uint32_t)fc[i].hi * fc[i].vi;
            if (fc[i].hi > max_h)
                max_h = fc[i].hi;
            if (fc[i].vi > max_v)
                max_v = fc[i].vi;
        }
        if ((uint32_t)width + 7 > 0xffff || (uint32_t)height + 7 > 0xffff) {
            fail(&e, "panic: u16 overflow in image size"); /* (width + 7) / 8 in u16 */
            goto out;
        }
        uint32_t width_dus = ((uint32_t)width + 7) / 8, height_dus = ((uint32_t)height + 7) / 8;
        uint32_t width_mcus = (width_dus + max_h - 1) / max_h;
        uint32_t height_mcus = (height_dus + max_v - 1) / max_v;
        if (!have_ri)
            ri = height_mcus * width_mcus;
        if (ri == 0) {
            fail(&e, "panic: attempt to divide by zero"); /* lib.rs:785, quirk Q8 */
            goto out;
        }
        uint32_t total = height_mcus * width_mcus / ri;
        if (total > 64u * 65535u) {
            fail(&e, "number of restart intervals exceeds limit (%u > %u)", total, 64u * 65535u);
            goto out;
        }

        img = (orc_image *)calloc(1, sizeof *img);
        for (int t = 0; t < 4; t++)
            for (int k = 0; k < 64; k++)
                md_put(img->md, (size_t)(t * 64 + k) * 4, q[t][k]);
        md_put(img->md, MD_RI, ri);
        for (int i = 0; i < 3; i++) {
            size_t o = MD_COMP + (size_t)i * 20;
            md_put(img->md, o + 0, fc[i].vi);
            md_put(img->md, o + 4, fc[i].hi);
            md_put(img->md, o + 8, fc[i].tqi);
            md_put(img->md, o + 12, (uint32_t)(uint8_t)(dch[i] << 1));
            md_put(img->md, o + 16, (uint32_t)(uint8_t)((ach[i] << 1) | 1));
        }
        md_put(img->md, MD_TOTAL, total);
        md_put(img->md, MD_WMCU, width_mcus);
        md_put(img->md, MD_MAXH, max_h);
        md_put(img->md, MD_MAXV, max_v);
        md_put(img->md, MD_DUS, dus_per_mcu);
        md_put(img->md, MD_RET, 32);
        img->width = width;
        img->height = height;
        img->scan_off = scan_off;
        img->scan_len = scan_len;
        img->flags = flags;

        /* ref: src/huffman.rs:247-271 -- concatenate, rebasing delegate indices */
        size_t total_l2 = 0;
        for (int t = 0; t < 4; t++)
            total_l2 += tables[t]->l2_len;
        img->l2 = (uint16_t *)malloc((total_l2 ? total_l2 : 1) * sizeof(uint16_t));
        size_t off = 0;
        for (int t = 0; t < 4; t++) {
            for (int i = 0; i < 256; i++) {
                uint16_t en = tables[t]->l1[i];
                if ((en & 0x8000) && t != 0) {
                    size_t idx = (size_t)(en & 0x7fff) + off;
                    if (idx > 0x7fff) {
                        fail(&e, "panic: L2 offset overflow");
                        orc_image_free(img);
                        img = NULL;
                        goto out;
                    }
                    en = (uint16_t)(0x8000 | idx);
                }
                img->l1[t * 256 + i] = en;
            }
            memcpy(img->l2 + off, tables[t]->l2, tables[t]->l2_len * sizeof(uint16_t));
            off += tables[t]->l2_len;
            if (off > 0xffff && t < 3) { /* offset.try_into::<u16>().unwrap() */
                fail(&e, "panic: L2 offset overflow");
                orc_image_free(img);
                img = NULL;
                goto out;
            }
        }
        img->l2_len = total_l2;
    }

out:
    for (int t = 0; t < 4; t++)
        orc_table_free(tables[t]);
    if (!img && err)
        snprintf(err, ORC_ERRLEN, "%s", e.msg);
    return img;
}

void orc_image_free(orc_image *img)
{
    if (!img)
        return;
    free(img->l2);
    free(img);
}

uint32_t orc_image_width(const orc_image *img) { return img->width; }
uint32_t orc_image_height(const orc_image *img) { return img->height; }
uint32_t orc_image_parallelism(const orc_image *img) { return md_get(img->md, MD_TOTAL); }
const uint8_t *orc_image_metadata(const orc_image *img) { return img->md; }
const uint8_t *orc_image_l1(const orc_image *img) { return (const uint8_t *)img->l1; }
const uint8_t *orc_image_l2(const orc_image *img, size_t *nbytes)
{
    *nbytes = img->l2_len * 2;
    return (const uint8_t *)img->l2;
}
void orc_image_scan(const orc_image *img, size_t *offset, size_t *len)
{
    *offset = img->scan_off;
    *len = img->scan_len;
}

/* ======================================================================== */
/* huffman pass -- ref: src/huffman.wgsl:81-216                              */
/* One loop iteration = one shader invocation (one restart interval).        */
/* ======================================================================== */

typedef struct {
    const uint8_t *md;
    const uint8_t *l1;
    size_t l1_bytes;
    const uint8_t *l2;
    size_t l2_bytes;
} hctx_t;

/* array<u32> view with robust (zero) out-of-bounds reads */
static uint32_t word_at(const uint8_t *base, size_t nbytes, size_t idx)
{
    if ((idx + 1) * 4 > nbytes)
        return 0;
    uint32_t v;
    memcpy(&v, base + idx * 4, 4);
    return v;
}

static uint32_t huffdecode(orc_bits *b, const hctx_t *c, uint32_t table)
{
    uint32_t code = b->cur >> 16;
    uint32_t l1idx = code >> 8;
    uint32_t entry = word_at(c->l1, c->l1_bytes, (size_t)table * 128 + (l1idx >> 1));
    entry = (entry >> ((l1idx & 1u) * 16u)) & 0xffffu;
    if (entry & 0x8000u) {
        uint32_t l2idx = (entry & 0x7fffu) + (code & 0xffu);
        entry = word_at(c->l2, c->l2_bytes, l2idx >> 1);
        entry = (entry >> ((l2idx & 1u) * 16u)) & 0xffffu;
    }
    orc_bits_consume(b, entry >> 8);
    return entry & 0xffu;
}

void orc_huffman_pass(const uint8_t *md, const uint8_t *l1, const uint8_t *l2, size_t l2_bytes,
                      const uint32_t *words, size_t nwords, const uint32_t *starts,
                      size_t nstarts, int32_t *coef, size_t ncoef)
{
    orc_huffman_pass_ext(md, l1, l2, l2_bytes, words, nwords, starts, nstarts, coef, ncoef, 0);
}

/* flags & 2: the two deviations from T.81 switched off (extension; the reference is flags == 0) */
void orc_huffman_pass_ext(const uint8_t *md, const uint8_t *l1, const uint8_t *l2, size_t l2_bytes,
                          const uint32_t *words, size_t nwords, const uint32_t *starts,
                          size_t nstarts, int32_t *coef, size_t ncoef, unsigned flags)
{
    const int standard = (flags & 2u) != 0;
    hctx_t c = {md, l1, 2048, l2, l2_bytes};
    uint32_t count = md_get(md, MD_TOTAL);
    uint32_t ri = md_get(md, MD_RI);
    uint32_t dus_per_mcu = md_get(md, MD_DUS);
    uint32_t retained = md_get(md, MD_RET);

    memset(coef, 0, ncoef * sizeof(int32_t));

    /* The uniform block's words the loops below use, read once (the shader reads them from the
     * uniform buffer wherever it needs them: the same values every time): the components'
     * fields and the quantisation tables.  A table index beyond the four tables reads words
     * behind them, as the shader's indexing would (md_get inside the loops did the same). */
    uint32_t c_vs[3], c_hs[3], c_qt[3], c_dct[3], c_act[3];
    for (uint32_t comp = 0; comp < 3; comp++) {
        const size_t o = MD_COMP + (size_t)comp * 20;
        c_vs[comp] = md_get(md, o + 0);
        c_hs[comp] = md_get(md, o + 4);
        c_qt[comp] = md_get(md, o + 8);
        c_dct[comp] = md_get(md, o + 12);
        c_act[comp] = md_get(md, o + 16);
    }
    uint32_t quant[3][64];
    for (uint32_t comp = 0; comp < 3; comp++)
        for (uint32_t pos = 0; pos < 64; pos++)
            quant[comp][pos] = md_get(md, ((size_t)c_qt[comp] * 64 + pos) * 4);

    for (uint32_t id = 0; id < count; id++) {
        orc_bits b;
        orc_bits_init(&b, words, nwords, id < nstarts ? starts[id] : 0u);
        int32_t dcpred[3] = {0, 0, 0};
        for (uint32_t i = 0; i < ri; i++) {
            uint32_t mcu_index = id * ri + i;
            uint32_t du_index = mcu_index * dus_per_mcu;
            for (uint32_t comp = 0; comp < 3; comp++) {
                uint32_t vs = c_vs[comp], hs = c_hs[comp];
                uint32_t dct = c_dct[comp], act = c_act[comp];
                for (uint32_t v = 0; v < vs; v++) {
                    for (uint32_t h = 0; h < hs; h++) {
                        uint32_t start = du_index * retained;
                        /* DC: note there is no refill here (quirk Q1) */
                        if (standard)
                            orc_bits_refill(&b);
                        uint32_t dccat = huffdecode(&b, &c, dct);
                        int32_t diff = (int32_t)orc_bits_peek(&b, dccat);
                        orc_bits_consume(&b, dccat);
                        diff = dccat == 0 ? 0 : orc_huff_extend(diff, dccat);
                        dcpred[comp] = (int32_t)((uint32_t)dcpred[comp] + (uint32_t)diff);
                        if ((size_t)start < ncoef)
                            coef[start] = (int32_t)((uint32_t)dcpred[comp] * quant[comp][0]);
                        for (uint32_t pos = 1; pos < 64; pos++) {
                            orc_bits_refill(&b);
                            uint32_t rs = huffdecode(&b, &c, act);
                            if (rs == 0)
                                break;
                            if (rs == 0xf0) {
                                pos += standard ? 15 : 16; /* +1 from the loop: 17 in total (quirk Q2) */
                                continue;
                            }
                            uint32_t ssss = rs & 15u;
                            pos += rs >> 4;
                            int32_t val = (int32_t)orc_bits_peek(&b, ssss);
                            orc_bits_consume(&b, ssss);
                            int32_t cf = orc_huff_extend(val, ssss);
                            if (pos < retained) { /* quirk Q3 */
                                size_t idx = (size_t)start + pos;
                                if (idx < ncoef)
                                    coef[idx] = (int32_t)((uint32_t)cf * quant[comp][pos]);
                            }
                        }
                        du_index++;
                    }
                }
            }
        }
    }
}

/* ======================================================================== */
/* dct pass -- ref: src/dct.wgsl:7-201.  Every arithmetic step is a separate */
/* f32 operation (no FMA); constants are the f32 nearest to the WGSL         */
/* AbstractFloat literal.                                                     */
/* ======================================================================== */

static const float SCALE[8] = {(float)1.0,         (float)1.387039845, (float)1.306562965,
                               (float)1.175875602, (float)1.0,         (float)0.785694958,
                               (float)0.541196100, (float)0.275899379};

static const uint8_t ZIGZAG[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
    41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
    46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

#define C_1_414 ((float)1.414213562)
#define C_1_847 ((float)1.847759065)
#define C_1_082 ((float)1.082392200)
#define C_2_613 ((float)2.613125930)

static float clampf(float v)
{
    /* WGSL clamp(e, low, high) = min(max(e, low), high) */
    float m = v > 0.0f ? v : 0.0f;
    return m < 255.0f ? m : 255.0f;
}

void orc_dct_pass(const uint8_t *md, int32_t *coef, size_t ncoef)
{
    uint32_t total_dus = md_get(md, MD_TOTAL) * md_get(md, MD_RI) * md_get(md, MD_DUS);
    uint32_t retained = md_get(md, MD_RET);

    for (uint32_t du = 0; du < total_dus; du++) {
        size_t go = (size_t)du * retained;
        float ws[64];

        for (uint32_t col = 0; col < 8; col++) { /* lane = column */
            float in[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t row = 0; row < 8; row++) {
                float mul = SCALE[row] * SCALE[col];
                uint32_t i = ZIGZAG[row * 8 + col];
                if (i < retained) {
                    int32_t cv = go + i < ncoef ? coef[go + i] : 0;
                    in[row] = (float)cv * mul;
                }
            }
            float tmp0 = in[0] * 0.125f, tmp1 = in[2] * 0.125f;
            float tmp2 = in[4] * 0.125f, tmp3 = in[6] * 0.125f;
            float tmp10 = tmp0 + tmp2, tmp11 = tmp0 - tmp2;
            float tmp13 = tmp1 + tmp3;
            float tmp12 = (tmp1 - tmp3) * C_1_414 - tmp13;
            float t0 = tmp10 + tmp13, t3 = tmp10 - tmp13;
            float t1 = tmp11 + tmp12, t2 = tmp11 - tmp12;

            float tmp4 = in[1] * 0.125f, tmp5 = in[3] * 0.125f;
            float tmp6 = in[5] * 0.125f, tmp7 = in[7] * 0.125f;
            float z13 = tmp6 + tmp5, z10 = tmp6 - tmp5;
            float z11 = tmp4 + tmp7, z12 = tmp4 - tmp7;
            float t7 = z11 + z13;
            float t11 = (z11 - z13) * C_1_414;
            float z5 = (z10 + z12) * C_1_847;
            float t10 = z5 - z12 * C_1_082;
            float t12 = z5 - z10 * C_2_613;
            float t6 = t12 - t7;
            float t5 = t11 - t6;
            float t4 = t10 - t5;

            ws[0 * 8 + col] = t0 + t7;
            ws[7 * 8 + col] = t0 - t7;
            ws[1 * 8 + col] = t1 + t6;
            ws[6 * 8 + col] = t1 - t6;
            ws[2 * 8 + col] = t2 + t5;
            ws[5 * 8 + col] = t2 - t5;
            ws[3 * 8 + col] = t3 + t4;
            ws[4 * 8 + col] = t3 - t4;
        }

        for (uint32_t row = 0; row < 8; row++) { /* lane = row */
            float *w = ws + row * 8;
            float z5 = w[0] + 128.5f;
            float tmp10 = z5 + w[4], tmp11 = z5 - w[4];
            float tmp13 = w[2] + w[6];
            float tmp12 = (w[2] - w[6]) * C_1_414 - tmp13;
            float tmp0 = tmp10 + tmp13, tmp3 = tmp10 - tmp13;
            float tmp1 = tmp11 + tmp12, tmp2 = tmp11 - tmp12;

            float z13 = w[5] + w[3], z10 = w[5] - w[3];
            float z11 = w[1] + w[7], z12 = w[1] - w[7];
            float tmp7 = z11 + z13;
            float t11 = (z11 - z13) * C_1_414;
            float z5i = (z10 + z12) * C_1_847;
            float t10 = z5i - z12 * C_1_082;
            float t12 = z5i - z10 * C_2_613;
            float tmp6 = t12 - tmp7;
            float tmp5 = t11 - tmp6;
            float tmp4 = t10 - tmp5;

            w[0] = clampf(tmp0 + tmp7);
            w[7] = clampf(tmp0 - tmp7);
            w[1] = clampf(tmp1 + tmp6);
            w[6] = clampf(tmp1 - tmp6);
            w[2] = clampf(tmp2 + tmp5);
            w[5] = clampf(tmp2 - tmp5);
            w[3] = clampf(tmp3 + tmp4);
            w[4] = clampf(tmp3 - tmp4);
        }

        for (uint32_t y = 0; y < 8; y++) {
            const float *w = ws + y * 8;
            uint32_t lo = (uint32_t)w[0] | (uint32_t)w[1] << 8 | (uint32_t)w[2] << 16 |
                          (uint32_t)w[3] << 24;
            uint32_t hi = (uint32_t)w[4] | (uint32_t)w[5] << 8 | (uint32_t)w[6] << 16 |
                          (uint32_t)w[7] << 24;
            if (go + y * 2 + 0 < ncoef)
                coef[go + y * 2 + 0] = (int32_t)lo;
            if (go + y * 2 + 1 < ncoef)
                coef[go + y * 2 + 1] = (int32_t)hi;
        }
    }
}

/* ======================================================================== */
/* finalize pass -- ref: src/dct.wgsl:218-334                                */
/* ======================================================================== */

static int32_t clampi(int32_t v)
{
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

void orc_finalize_pass(const uint8_t *md, const int32_t *coef, size_t ncoef, uint8_t *rgba,
                       uint32_t tex_w, uint32_t tex_h)
{
    uint32_t total_mcus = md_get(md, MD_TOTAL) * md_get(md, MD_RI);
    uint32_t dus_per_mcu = md_get(md, MD_DUS), retained = md_get(md, MD_RET);
    uint32_t width_mcus = md_get(md, MD_WMCU);
    uint32_t max_h = md_get(md, MD_MAXH), max_v = md_get(md, MD_MAXV);
    uint32_t msx = max_h * 8, msy = max_v * 8;

    /* The reference's shader assigns one thread to each of the 8 rows of an MCU and keeps four
     * data units per MCU in workgroup memory: it covers MCUs that are 8 rows tall with at most
     * four data units (4:2:2 -- the only layout its front-end accepts -- and 4:4:4).  With
     * max_v == 1 the loops below are that shader, statement by statement.  For the 16-row MCUs
     * of 4:4:0 / 4:2:0 (extension) the same formulas are continued to rows 8..15: a component's
     * data unit is picked by block row as well as block column, its sample by row / yscale. */
    uint32_t c_vs[3], c_hs[3]; /* (the components' sampling factors: read once, see orc_huffman_pass_ext) */
    for (uint32_t comp = 0; comp < 3; comp++) {
        c_vs[comp] = md_get(md, MD_COMP + (size_t)comp * 20 + 0);
        c_hs[comp] = md_get(md, MD_COMP + (size_t)comp * 20 + 4);
    }
    /* Which data unit, row, word and byte of it a pixel (row, col) of an MCU takes its component
     * from: the shader's index arithmetic, statement by statement, done once per (row, col) of
     * the MCU instead of once per pixel of the image -- the same indices every MCU. */
    if (total_mcus == 0)
        return;
    struct pick {
        uint8_t du, y, word, shift;
    } picks[3][16][16];
    for (uint32_t row = 0; row < msy && row < 16; row++)
        for (uint32_t col = 0; col < msx && col < 16; col++) {
            uint32_t du_offset = 0;
            for (uint32_t comp = 0; comp < 3; comp++) {
                uint32_t vs = c_vs[comp], hs = c_hs[comp];
                uint32_t du = du_offset + (row * vs / msy) * hs + col * hs / msx;
                uint32_t xscale = max_h / hs, yscale = max_v / vs;
                uint32_t x = col / xscale, y = row / yscale;
                picks[comp][row][col].du = (uint8_t)(du < 6 ? du : 0);
                picks[comp][row][col].y = (uint8_t)(y & 7);
                picks[comp][row][col].word = (uint8_t)((x & 7u) > 3u);
                picks[comp][row][col].shift = (uint8_t)((x & 7u) * 8u);
                du_offset += hs * vs;
            }
        }
    for (uint32_t mcu = 0; mcu < total_mcus; mcu++) {
        uint32_t rows[6][8][2]; /* databuf[local_mcu].du[i].rows[row] */
        memset(rows, 0, sizeof rows);
        for (uint32_t row = 0; row < 8; row++) {
            for (uint32_t i = 0; i < dus_per_mcu && i < 6; i++) {
                size_t off = ((size_t)mcu * dus_per_mcu + i) * retained + row * 2;
                rows[i][row][0] = off < ncoef ? (uint32_t)coef[off] : 0;
                rows[i][row][1] = off + 1 < ncoef ? (uint32_t)coef[off + 1] : 0;
            }
        }
        uint32_t mx = mcu % width_mcus, my = mcu / width_mcus;
        for (uint32_t row = 0; row < msy; row++) { /* reference: one thread per MCU row, 8 rows */
            for (uint32_t col = 0; col < msx; col++) {
                uint32_t cx = mx * msx + col, cy = my * msy + row;
                uint32_t comp_val[3];
                for (uint32_t comp = 0; comp < 3; comp++) {
                    const struct pick k = picks[comp][row][col];
                    comp_val[comp] = SHR(rows[k.du][k.y][k.word], k.shift);
                }
                int32_t yy = (int32_t)(comp_val[0] & 0xffu);
                int32_t cb = (int32_t)(comp_val[1] & 0xffu) - 128;
                int32_t cr = (int32_t)(comp_val[2] & 0xffu) - 128;
                /* arithmetic right shifts (quirk Q5) */
                int32_t r = yy + ((45 * cr) >> 5);
                int32_t g = yy - ((11 * cb + 23 * cr) >> 5);
                int32_t b = yy + ((113 * cb) >> 6);
                if (cx < tex_w && cy < tex_h) {
                    uint8_t *px = rgba + ((size_t)cy * tex_w + cx) * 4;
                    px[0] = (uint8_t)clampi(r);
                    px[1] = (uint8_t)clampi(g);
                    px[2] = (uint8_t)clampi(b);
                    px[3] = 255;
                }
            }
        }
    }
}

/* ======================================================================== */
/* whole path -- ref: src/lib.rs:385-450                                     */
/* ======================================================================== */

int orc_image_decode(const orc_image *img, const uint8_t *jpeg, uint8_t *rgba, uint32_t tex_w,
                     uint32_t tex_h, int32_t *coef_out, char *err)
{
    uint32_t total = md_get(img->md, MD_TOTAL);
    uint32_t total_dus = total * md_get(img->md, MD_RI) * md_get(img->md, MD_DUS);
    size_t ncoef = (size_t)total_dus * md_get(img->md, MD_RET);

    /* (the coefficient buffer is kept from frame to frame, one per thread -- the reference's
     * Decoder keeps its DynamicBuffer the same way, src/lib.rs:302-370; orc_huffman_pass_ext
     * clears it like the reference's clear_buffer does.  The scan buffer is a fresh one every
     * time: its output is defined on a fresh buffer) */
    static __thread int32_t *coef = NULL;
    static __thread size_t coef_cap = 0;
    orc_scanbuf *sb = orc_scanbuf_new();
    int rc = orc_scanbuf_process(sb, jpeg + img->scan_off, img->scan_len, total, err);

    if (coef_cap < (ncoef ? ncoef : 1)) {
        free(coef);
        coef_cap = ncoef ? ncoef : 1;
        coef = (int32_t *)malloc(coef_cap * sizeof(int32_t));
    }
    orc_huffman_pass_ext(img->md, (const uint8_t *)img->l1, (const uint8_t *)img->l2,
                         img->l2_len * 2, sb->words, sb->words_len, sb->starts, sb->starts_len, coef,
                         ncoef, img->flags);
    if (coef_out)
        memcpy(coef_out, coef, ncoef * sizeof(int32_t));
    orc_dct_pass(img->md, coef, ncoef);
    orc_finalize_pass(img->md, coef, ncoef, rgba, tex_w, tex_h);
    orc_scanbuf_free(sb);
    return rc;
}
