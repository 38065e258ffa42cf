/*
 * compeg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference decoder's algorithm for the baseline
 * 4:2:2 restart-interval JPEG path.  Every function cites the reference
 * file:line it follows (paths relative to the upstream SludgePhD/Compeg tree).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The shipped decoder
 * (compeg_amd/csrc, libcompeg_hip.so) never links or calls it.
 *
 * Parity pinning (see oracle/README.md):
 *   - scan preprocess   : exact, reference KATs src/scan.rs:151-180, benches/scan.dat
 *   - Huffman LUT build : exact, reference snapshots src/huffman.rs:359-546
 *   - bit reader        : exact, reference KATs src/bits.rs:74-130
 *   - segment parser    : exact, 16 reference dumps src/file/test-images/ (*.log)
 *   - decoded pixels    : the reference pins these only to +-3 on two 64x8
 *                         images (src/tests.rs:18,131-135); bit-exactness is
 *                         defined against this restatement of the WGSL
 *                         abstract semantics (wrapping u32/i32, shift mod 32,
 *                         IEEE f32 per operation, no FMA contraction).
 */
#ifndef COMPEG_ORACLE_H
#define COMPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_ERRLEN 256

/* ---- scan preprocessing (src/scan.rs) ---------------------------------- */
typedef struct orc_scanbuf orc_scanbuf;
orc_scanbuf *orc_scanbuf_new(void);
void orc_scanbuf_free(orc_scanbuf *sb);
/* 0 = ok; 1 = restart-interval count mismatch (message in err, buffers still
 * hold the truncated result, exactly as the reference leaves them). */
int orc_scanbuf_process(orc_scanbuf *sb, const uint8_t *scan, size_t len,
                        uint32_t expected_intervals, char *err);
const uint8_t *orc_scanbuf_data(const orc_scanbuf *sb, size_t *nbytes);
const uint8_t *orc_scanbuf_starts(const orc_scanbuf *sb, size_t *nbytes);

/* ---- Huffman LUTs (src/huffman.rs) ------------------------------------- */
typedef struct orc_table orc_table;
/* returns NULL when the reference would panic (malformed table). */
orc_table *orc_table_build(const uint8_t li[16], const uint8_t *vij, size_t nvij);
void orc_table_free(orc_table *t);
uint16_t orc_table_lookup(const orc_table *t, uint16_t code); /* bits<<8|value */
size_t orc_table_l2_len(const orc_table *t);
/* "code -> value" listing in the format of the reference's Debug impl. */
size_t orc_table_debug(const orc_table *t, char *out, size_t cap);
/* which: 0 luma DC, 1 luma AC, 2 chroma DC, 3 chroma AC (Annex K). */
orc_table *orc_table_default(int which);

/* ---- bit reader (src/huffman.wgsl:35-79, src/bits.rs:18-67) ------------- */
typedef struct {
    const uint32_t *words;
    size_t nwords;
    uint32_t next_word, cur, next, left;
} orc_bits;
void orc_bits_init(orc_bits *b, const uint32_t *words, size_t nwords, uint32_t start);
void orc_bits_refill(orc_bits *b);
void orc_bits_consume(orc_bits *b, uint32_t n);
uint32_t orc_bits_peek(const orc_bits *b, uint32_t n);
uint32_t orc_bits_huffdecode_table(orc_bits *b, const orc_table *t);
int32_t orc_huff_extend(int32_t v, uint32_t t);

/* ---- segment parser dump (src/file.rs + src/file/tests.rs:9-58) --------- */
size_t orc_parser_dump(const uint8_t *jpeg, size_t len, char *out, size_t cap);

/* ---- image front-end (src/lib.rs:597-824) ------------------------------- */
typedef struct orc_image orc_image;
/* returns NULL and fills err on rejection.  err text equals the reference's
 * message where it has one; "panic: ..." marks inputs the reference aborts on. */
orc_image *orc_image_parse(const uint8_t *jpeg, size_t len, char *err);
/* flags & 1: also accept luma sampling 1x1 / 1x2 / 2x2; flags & 2: entropy decoding per T.81 (refill
 * in front of DC codes, ZRL = 16 positions) -- both extensions beyond the reference */
orc_image *orc_image_parse_ext(const uint8_t *jpeg, size_t len, unsigned flags, char *err);
void orc_image_free(orc_image *img);
uint32_t orc_image_width(const orc_image *img);
uint32_t orc_image_height(const orc_image *img);
uint32_t orc_image_parallelism(const orc_image *img);
const uint8_t *orc_image_metadata(const orc_image *img); /* 1112-byte block */
const uint8_t *orc_image_l1(const orc_image *img);        /* 2048 bytes */
const uint8_t *orc_image_l2(const orc_image *img, size_t *nbytes);
void orc_image_scan(const orc_image *img, size_t *offset, size_t *len);

/* ---- the three GPU passes as scalar loops ------------------------------- */
/* coefficients: int32[total_dus*retained], zero-filled by the callee first
 * (the reference clears the buffer, src/lib.rs:428). */
void orc_huffman_pass(const uint8_t *metadata, const uint8_t *l1, const uint8_t *l2,
                      size_t l2_bytes, const uint32_t *words, size_t nwords,
                      const uint32_t *starts, size_t nstarts, int32_t *coefficients,
                      size_t ncoef);
void orc_huffman_pass_ext(const uint8_t *metadata, const uint8_t *l1, const uint8_t *l2,
                          size_t l2_bytes, const uint32_t *words, size_t nwords, const uint32_t *starts,
                          size_t nstarts, int32_t *coefficients, size_t ncoef, unsigned flags);
void orc_dct_pass(const uint8_t *metadata, int32_t *coefficients, size_t ncoef);
/* rgba: tex_w*tex_h*4 bytes, stores outside the texture are dropped. */
void orc_finalize_pass(const uint8_t *metadata, const int32_t *coefficients, size_t ncoef,
                       uint8_t *rgba, uint32_t tex_w, uint32_t tex_h);

/* Full reference path for one image: preprocess + huffman + dct + finalize
 * (src/lib.rs:385-450).  coef_out (optional) receives the coefficient buffer
 * as it stands after the huffman pass.  Returns 0, or 1 with err filled when
 * the preprocess step reports a count mismatch (the decode still ran, as in
 * the reference where that error is dropped, src/lib.rs:391-394,532-536). */
int orc_image_decode(const orc_image *img, const uint8_t *jpeg, uint8_t *rgba,
                     uint32_t tex_w, uint32_t tex_h, int32_t *coef_out, char *err);

#ifdef __cplusplus
}
#endif
#endif
