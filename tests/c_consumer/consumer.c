/* A plain C99 consumer of include/compeg_hip.h (test infrastructure): what a non-Python FFI user of the
 * boundary sees.  Compiled with gcc -std=c99 -Wall -Wextra -Werror and linked against libcompeg_hip.so by
 * tests/test_host_parity.py on every run of the CPU suite (no GPU call there) and by tests/test_gpu_parity.py.
 *   consumer file.jpg            prints what ImageData::new reports for the file, then exercises error strings and
 *                                the ScanBuffer known-answer vector of the reference (src/scan.rs:151-159).
 *   consumer --decode file.jpg   the reference's own test flow (src/tests.rs:41-72: open the GPU, parse, decode
 *                                blocking, read the texture back tightly packed) through the C ABI, on the GPU:
 *                                prints the FNV-1a hash of the RGBA bytes. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "compeg_hip.h"

static int fail(const char *what)
{
    fprintf(stderr, "consumer: %s: %s\n", what, compeg_last_error());
    return 1;
}

/* Gpu::open -> ImageData::new -> Decoder::decode_blocking -> texture read-back (src/tests.rs:41-72) */
static int decode_on_gpu(const unsigned char *bytes, size_t n)
{
    compeg_gpu *gpu = NULL;
    compeg_image *img = NULL;
    compeg_decoder *dec = NULL;
    compeg_op *op = NULL;
    if (compeg_gpu_open(-1, &gpu) != COMPEG_OK)
        return fail("compeg_gpu_open");
    if (compeg_image_parse(bytes, n, 0, &img) != COMPEG_OK)
        return fail("compeg_image_parse");
    if (compeg_decoder_new(gpu, &dec) != COMPEG_OK)
        return fail("compeg_decoder_new");
    if (compeg_decoder_decode_blocking(dec, img, &op) != COMPEG_OK)
        return fail("compeg_decoder_decode_blocking");
    const uint32_t w = compeg_image_width(img), h = compeg_image_height(img);
    printf("changed %d kernel %d\n", compeg_op_texture_changed(op), compeg_decoder_last_kernel(dec));
    unsigned char *rgba = (unsigned char *)malloc((size_t)w * h * 4);
    if (!rgba || compeg_decoder_read_output(dec, rgba, w, h) != COMPEG_OK)
        return fail("compeg_decoder_read_output");
    unsigned long long hash = 14695981039346656037ull; /* FNV-1a, 64 bit */
    for (size_t i = 0; i < (size_t)w * h * 4; i++) {
        hash ^= rgba[i];
        hash *= 1099511628211ull;
    }
    printf("decoded %u %u fnv1a %016llx\n", w, h, hash);
    /* a second decode into the same texture: not reallocated (lib.rs:564-573) */
    compeg_op_free(op);
    op = NULL;
    if (compeg_decoder_decode_blocking(dec, img, &op) != COMPEG_OK)
        return fail("compeg_decoder_decode_blocking (second)");
    printf("changed_again %d\n", compeg_op_texture_changed(op));
    free(rgba);
    compeg_op_free(op);
    compeg_decoder_free(dec);
    compeg_image_free(img);
    compeg_gpu_release(gpu);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    const int decode = strcmp(argv[1], "--decode") == 0;
    if (decode && argc < 3)
        return 2;
    FILE *f = fopen(argv[decode ? 2 : 1], "rb");
    if (!f)
        return 2;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *bytes = (unsigned char *)malloc((size_t)n);
    if (!bytes || fread(bytes, 1, (size_t)n, f) != (size_t)n)
        return 2;
    fclose(f);
    if (decode) {
        const int rc_decode = decode_on_gpu(bytes, (size_t)n);
        free(bytes);
        return rc_decode;
    }

    printf("version %s\n", compeg_version());

    /* ImageData::new, borrowing the bytes (Cow::Borrowed) */
    compeg_image *img = NULL;
    if (compeg_image_parse(bytes, (size_t)n, 0, &img) != COMPEG_OK)
        return fail("compeg_image_parse");
    printf("image %u %u %u\n", compeg_image_width(img), compeg_image_height(img), compeg_image_parallelism(img));
    size_t off = 0, len = 0, l2 = 0;
    compeg_image_scan_range(img, &off, &len);
    (void)compeg_image_huffman_l2(img, &l2);
    printf("scan %zu %zu l2 %zu\n", off, len, l2);
    const unsigned char *md = (const unsigned char *)compeg_image_metadata(img);
    unsigned long sum = 0;
    for (size_t i = 0; i < COMPEG_METADATA_BYTES; i++)
        sum = sum * 131u + md[i];
    printf("metadata %lu\n", sum & 0xfffffffful);
    compeg_image_free(img);

    /* errors: status code + thread-local message with the reference's text */
    img = NULL;
    const unsigned char junk[4] = {1, 2, 3, 4};
    int rc = compeg_image_parse(junk, sizeof junk, 1, &img);
    printf("junk %d %s\n", rc, compeg_last_error());
    rc = compeg_image_parse(junk, sizeof junk, 1, NULL);
    printf("null %d\n", rc);

    /* ScanBuffer::process, the reference's known-answer vector */
    compeg_scanbuffer *sb = compeg_scanbuffer_new();
    const unsigned char kat[7] = {0xFF, 0x00, 0x44, 0x55, 0xFF, 0xD0, 0x34};
    if (!sb || compeg_scanbuffer_process(sb, kat, sizeof kat, 2) != COMPEG_OK)
        return fail("compeg_scanbuffer_process");
    size_t nd = 0, ns = 0;
    const unsigned char *d = (const unsigned char *)compeg_scanbuffer_data(sb, &nd);
    const unsigned char *s = (const unsigned char *)compeg_scanbuffer_start_positions(sb, &ns);
    printf("scanbuffer");
    for (size_t i = 0; i < nd; i++)
        printf(" %02x", d[i]);
    printf(" |");
    for (size_t i = 0; i < ns; i++)
        printf(" %02x", s[i]);
    printf("\n");
    rc = compeg_scanbuffer_process(sb, kat, sizeof kat, 1);
    printf("mismatch %d %s\n", rc, compeg_last_error());
    compeg_scanbuffer_free(sb);

    /* the stage-time record is plain data of three doubles */
    compeg_stage_times t = {0.0, 0.0, 0.0};
    printf("stage_times %zu %d\n", sizeof t, compeg_decoder_last_stage_times(NULL, &t));
    free(bytes);
    return 0;
}
