/*
 * synth_jpeg.c -- deterministic synthetic-input generator for tests and bench.
 *
 * A small baseline-JPEG *encoder* (the decoder under test never sees this
 * code): YCbCr, luma sampling HxV in {1x1, 2x1, 2x2}, chroma 1x1, Annex-K
 * quantisation tables scaled libjpeg-style by a quality factor, Annex-K
 * Huffman tables (optionally omitted from the file, as hardware MJPEG
 * encoders do), restart interval of N MCUs with cycling RST0..7 markers.
 * It plays the role examples/enc.rs plays for the reference's fixtures.
 *
 * Also provides the seeded image synthesiser described in SURVEY.md 8(d).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const uint8_t ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                               12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                               58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static const uint8_t Q_LUMA[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                   14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                   18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t Q_CHROMA[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                     24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* ITU T.81 Annex K.3 typical Huffman tables */
static const uint8_t DC_L_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_C_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_L_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const uint8_t AC_C_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const uint8_t AC_L_VALS[162] = {
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
static const uint8_t AC_C_VALS[162] = {
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

typedef struct {
    uint16_t code[256];
    uint8_t len[256];
} enc_table;

static void make_enc_table(enc_table *t, const uint8_t bits[16], const uint8_t *vals)
{
    memset(t, 0, sizeof *t);
    unsigned code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        for (unsigned i = 0; i < bits[l - 1]; i++) {
            t->code[vals[k]] = (uint16_t)code++;
            t->len[vals[k]] = (uint8_t)l;
            k++;
        }
        code <<= 1;
    }
}

typedef struct {
    uint8_t *out;
    size_t cap, pos;
    uint64_t acc;
    int nbits;
    int overflow;
} bitw;

static void put_byte(bitw *w, uint8_t b)
{
    if (w->pos < w->cap)
        w->out[w->pos] = b;
    else
        w->overflow = 1;
    w->pos++;
}

static void put_bits(bitw *w, unsigned code, int len)
{
    w->acc = (w->acc << len) | (code & ((1u << len) - 1));
    w->nbits += len;
    while (w->nbits >= 8) {
        uint8_t b = (uint8_t)(w->acc >> (w->nbits - 8));
        put_byte(w, b);
        if (b == 0xff)
            put_byte(w, 0x00);
        w->nbits -= 8;
    }
}

static void flush_bits(bitw *w)
{
    if (w->nbits > 0)
        put_bits(w, (1u << (8 - w->nbits)) - 1, 8 - w->nbits);
    w->acc = 0;
    w->nbits = 0;
}

static void put_marker(bitw *w, uint8_t m)
{
    put_byte(w, 0xff);
    put_byte(w, m);
}

static void put_u16(bitw *w, unsigned v)
{
    put_byte(w, (uint8_t)(v >> 8));
    put_byte(w, (uint8_t)v);
}

static int bit_size(int v)
{
    int a = v < 0 ? -v : v, n = 0;
    while (a) {
        n++;
        a >>= 1;
    }
    return n;
}

static void encode_block(bitw *w, const int16_t zz[64], int *pred, const enc_table *dc,
                         const enc_table *ac)
{
    int diff = zz[0] - *pred;
    *pred = zz[0];
    int s = bit_size(diff);
    put_bits(w, dc->code[s], dc->len[s]);
    if (s)
        put_bits(w, (unsigned)(diff < 0 ? diff - 1 : diff), s);
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = zz[k];
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            put_bits(w, ac->code[0xf0], ac->len[0xf0]);
            run -= 16;
        }
        s = bit_size(v);
        int sym = run << 4 | s;
        put_bits(w, ac->code[sym], ac->len[sym]);
        put_bits(w, (unsigned)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run)
        put_bits(w, ac->code[0], ac->len[0]);
}

static float COSF[8][8];
static int cos_ready = 0;

static void init_cos(void)
{
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++)
            COSF[u][x] = (float)(cos((2 * x + 1) * u * M_PI / 16.0) * (u == 0 ? sqrt(0.125) : 0.5));
    cos_ready = 1;
}

static void fdct_quant(const float in[64], const float rq[64], int16_t zz[64])
{
    float tmp[64], outp[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int x = 0; x < 8; x++)
                s += in[y * 8 + x] * COSF[u][x];
            tmp[y * 8 + u] = s;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int y = 0; y < 8; y++)
                s += tmp[y * 8 + u] * COSF[v][y];
            outp[v * 8 + u] = s;
        }
    for (int k = 0; k < 64; k++)
        zz[k] = (int16_t)lrintf(outp[ZZ[k]] * rq[k]); /* rq = 1/q, zig-zag order */
}

static void scale_qtable(const uint8_t base[64], int quality, uint8_t out_zz[64])
{
    if (quality < 1)
        quality = 1;
    if (quality > 100)
        quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int k = 0; k < 64; k++) {
        int v = (base[ZZ[k]] * scale + 50) / 100;
        out_zz[k] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v));
    }
}

/* Flags */
#define SJ_NO_DHT 1u      /* omit DHT segments (decoder must fall back to Annex K) */
#define SJ_NO_EOI 2u      /* truncate before EOI */
#define SJ_JFIF 4u        /* emit a JFIF APP0 header */

/*
 * rgb: h*w*3 bytes.  hs,vs: luma sampling factors.  ri: MCUs per restart
 * interval (0 = no DRI).  Returns the number of bytes needed; the stream is
 * complete only when the return value <= cap.
 */
size_t synth_encode(const uint8_t *rgb, int w, int h, int quality, int hs, int vs, int ri,
                    unsigned flags, uint8_t *out, size_t cap)
{
    uint8_t ql[64], qc[64];
    float rql[64], rqc[64];
    scale_qtable(Q_LUMA, quality, ql);
    scale_qtable(Q_CHROMA, quality, qc);
    for (int k = 0; k < 64; k++) {
        rql[k] = 1.0f / (float)ql[k];
        rqc[k] = 1.0f / (float)qc[k];
    }
    if (!cos_ready)
        init_cos();
    enc_table dcl, dcc, acl, acc;
    make_enc_table(&dcl, DC_L_BITS, DC_VALS);
    make_enc_table(&dcc, DC_C_BITS, DC_VALS);
    make_enc_table(&acl, AC_L_BITS, AC_L_VALS);
    make_enc_table(&acc, AC_C_BITS, AC_C_VALS);

    bitw W = {out, cap, 0, 0, 0, 0};
    bitw *bw = &W;
    put_marker(bw, 0xd8);
    if (flags & SJ_JFIF) {
        static const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
        put_marker(bw, 0xe0);
        put_u16(bw, 16);
        for (int i = 0; i < 14; i++)
            put_byte(bw, jfif[i]);
    }
    for (int t = 0; t < 2; t++) {
        put_marker(bw, 0xdb);
        put_u16(bw, 67);
        put_byte(bw, (uint8_t)t);
        for (int k = 0; k < 64; k++)
            put_byte(bw, t ? qc[k] : ql[k]);
    }
    if (ri > 0) {
        put_marker(bw, 0xdd);
        put_u16(bw, 4);
        put_u16(bw, (unsigned)ri);
    }
    put_marker(bw, 0xc0);
    put_u16(bw, 17);
    put_byte(bw, 8);
    put_u16(bw, (unsigned)h);
    put_u16(bw, (unsigned)w);
    put_byte(bw, 3);
    put_byte(bw, 1);
    put_byte(bw, (uint8_t)(hs << 4 | vs));
    put_byte(bw, 0);
    put_byte(bw, 2);
    put_byte(bw, 0x11);
    put_byte(bw, 1);
    put_byte(bw, 3);
    put_byte(bw, 0x11);
    put_byte(bw, 1);
    if (!(flags & SJ_NO_DHT)) {
        const uint8_t *bits[4] = {DC_L_BITS, AC_L_BITS, DC_C_BITS, AC_C_BITS};
        const uint8_t *vals[4] = {DC_VALS, AC_L_VALS, DC_VALS, AC_C_VALS};
        const int nval[4] = {12, 162, 12, 162};
        const uint8_t tcth[4] = {0x00, 0x10, 0x01, 0x11};
        for (int t = 0; t < 4; t++) {
            put_marker(bw, 0xc4);
            put_u16(bw, (unsigned)(2 + 17 + nval[t]));
            put_byte(bw, tcth[t]);
            for (int i = 0; i < 16; i++)
                put_byte(bw, bits[t][i]);
            for (int i = 0; i < nval[t]; i++)
                put_byte(bw, vals[t][i]);
        }
    }
    put_marker(bw, 0xda);
    put_u16(bw, 12);
    put_byte(bw, 3);
    put_byte(bw, 1);
    put_byte(bw, 0x00);
    put_byte(bw, 2);
    put_byte(bw, 0x11);
    put_byte(bw, 3);
    put_byte(bw, 0x11);
    put_byte(bw, 0);
    put_byte(bw, 63);
    put_byte(bw, 0);

    const int mw = 8 * hs, mh = 8 * vs;
    const int mcus_x = (w + mw - 1) / mw, mcus_y = (h + mh - 1) / mh;
    int pred[3] = {0, 0, 0};
    int count = 0, rst = 0;
    float *Y = (float *)malloc(sizeof(float) * (size_t)mw * mh);
    float *Cb = (float *)malloc(sizeof(float) * (size_t)mw * mh);
    float *Cr = (float *)malloc(sizeof(float) * (size_t)mw * mh);
    for (int my = 0; my < mcus_y; my++) {
        for (int mx = 0; mx < mcus_x; mx++) {
            if (ri > 0 && count == ri) {
                flush_bits(bw);
                put_marker(bw, (uint8_t)(0xd0 + rst));
                rst = (rst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                count = 0;
            }
            count++;
            for (int y = 0; y < mh; y++) {
                int sy = my * mh + y;
                if (sy >= h)
                    sy = h - 1;
                for (int x = 0; x < mw; x++) {
                    int sx = mx * mw + x;
                    if (sx >= w)
                        sx = w - 1;
                    const uint8_t *p = rgb + ((size_t)sy * w + sx) * 3;
                    float r = p[0], g = p[1], b = p[2];
                    Y[y * mw + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
                    Cb[y * mw + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                    Cr[y * mw + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                }
            }
            float blk[64];
            int16_t zz[64];
            for (int v = 0; v < vs; v++)
                for (int u = 0; u < hs; u++) {
                    for (int y = 0; y < 8; y++)
                        for (int x = 0; x < 8; x++)
                            blk[y * 8 + x] = Y[(v * 8 + y) * mw + u * 8 + x];
                    fdct_quant(blk, rql, zz);
                    encode_block(bw, zz, &pred[0], &dcl, &acl);
                }
            for (int c = 0; c < 2; c++) {
                const float *src = c ? Cr : Cb;
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++) {
                        float s = 0;
                        for (int dy = 0; dy < vs; dy++)
                            for (int dx = 0; dx < hs; dx++)
                                s += src[(y * vs + dy) * mw + x * hs + dx];
                        blk[y * 8 + x] = s / (float)(hs * vs);
                    }
                fdct_quant(blk, rqc, zz);
                encode_block(bw, zz, &pred[1 + c], &dcc, &acc);
            }
        }
    }
    free(Y);
    free(Cb);
    free(Cr);
    flush_bits(bw);
    if (!(flags & SJ_NO_EOI))
        put_marker(bw, 0xd9);
    return bw->pos;
}

/* ---- seeded content ------------------------------------------------------ */

static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

/*
 * kind 0: three low-frequency sinusoids per channel + uniform noise of
 *         amplitude `noise` (SURVEY 8d "natural-like", ~1.5-3 bit/px at q85)
 * kind 1: uniform random RGB (stress: long codes, ZRL runs, L2 LUT)
 * kind 2: flat grey with sparse impulses (mostly-EOB blocks, tiny intervals)
 */
void synth_fill(uint8_t *rgb, int w, int h, uint64_t seed, int kind, int noise)
{
    uint64_t s = seed * 0x2545f4914f6cdd1dull + 0xc0ffee;
    double fx[3][3], fy[3][3], ph[3][3], amp[3][3];
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) {
            fx[c][k] = (double)(splitmix(&s) % 1000) / 1000.0 * 6.0 / (w > 1 ? w : 1) * 6.283185307;
            fy[c][k] = (double)(splitmix(&s) % 1000) / 1000.0 * 6.0 / (h > 1 ? h : 1) * 6.283185307;
            ph[c][k] = (double)(splitmix(&s) % 1000) / 1000.0 * 6.283185307;
            amp[c][k] = 20.0 + (double)(splitmix(&s) % 25);
        }
    /* sin(a + b) = sin a cos b + cos a sin b with a = fx*x + ph, b = fy*y */
    float *sx = (float *)malloc(sizeof(float) * 18 * (size_t)(w > 0 ? w : 1));
    float *cx = sx + 9 * (size_t)(w > 0 ? w : 1);
    if (kind == 0)
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++)
                for (int x = 0; x < w; x++) {
                    sx[(size_t)(c * 3 + k) * w + x] = (float)(amp[c][k] * sin(fx[c][k] * x + ph[c][k]));
                    cx[(size_t)(c * 3 + k) * w + x] = (float)(amp[c][k] * cos(fx[c][k] * x + ph[c][k]));
                }
    for (int y = 0; y < h; y++) {
        float sy[9], cy[9];
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++) {
                sy[c * 3 + k] = (float)sin(fy[c][k] * y);
                cy[c * 3 + k] = (float)cos(fy[c][k] * y);
            }
        for (int x = 0; x < w; x++) {
            uint8_t *p = rgb + ((size_t)y * w + x) * 3;
            uint64_t r = splitmix(&s);
            for (int c = 0; c < 3; c++) {
                float v;
                if (kind == 1) {
                    v = (float)((r >> (c * 8)) & 255);
                } else if (kind == 2) {
                    v = ((r >> 40) % 997 == 0) ? (float)((r >> (c * 8)) & 255) : 128.0f;
                } else {
                    v = 128.0f;
                    for (int k = 0; k < 3; k++)
                        v += sx[(size_t)(c * 3 + k) * w + x] * cy[c * 3 + k] +
                             cx[(size_t)(c * 3 + k) * w + x] * sy[c * 3 + k];
                    if (noise > 0)
                        v += (float)((int)((r >> (c * 16)) % (unsigned)(2 * noise + 1)) - noise);
                }
                p[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : (int)v));
            }
        }
    }
    free(sx);
}
