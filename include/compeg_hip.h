/*
 * compeg_hip.h -- C ABI of libcompeg_hip.so: an MI355X-native (gfx950) decoder
 * for baseline 8-bit YCbCr 4:2:2 restart-interval JPEGs.
 *
 * The entry points are what a Rust `compeg`-shaped crate binds in place of the
 * reference's wgpu-backed implementation.  Each group cites the reference
 * interface it replaces (paths relative to SludgePhD/Compeg, v0.5.0).
 * INTEGRATION.md shows the Rust side.
 *
 * Conventions
 *   - every fallible call returns COMPEG_OK (0) or a negative COMPEG_E_* code;
 *     the message is available from compeg_last_error() (thread-local).  Texts
 *     equal the reference's error strings where it has one (src/lib.rs:622-793,
 *     src/file.rs:21-25,43-45,281-283,345-347, src/scan.rs:58-63).
 *   - nothing in this library aborts the process; inputs on which the
 *     reference panics (src/lib.rs:784-785 division by zero, src/huffman.rs
 *     asserts, src/file.rs:316,331-335 slicing) return COMPEG_E_MALFORMED.
 *   - objects are not internally synchronised: one thread at a time per
 *     decoder / scan buffer (the reference's `&mut self`), any number of
 *     decoders per compeg_gpu (the reference's `Arc<Gpu>`).
 *   - `hip_stream` arguments are `hipStream_t` passed as `void*` (NULL = the
 *     default stream) so that the header needs no HIP include.
 */
#ifndef COMPEG_HIP_H
#define COMPEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library itself is built -fvisibility=hidden */
#endif

#define COMPEG_OK 0
#define COMPEG_E_INVALID_ARG (-1) /* NULL handle, bad size ...                       */
#define COMPEG_E_UNSUPPORTED (-2) /* well-formed JPEG outside the supported subset   */
#define COMPEG_E_MALFORMED (-3)   /* broken stream (incl. inputs the reference panics on) */
#define COMPEG_E_COUNT_MISMATCH (-4) /* restart-interval count mismatch (scan.rs:58-63) */
#define COMPEG_E_HIP (-5)         /* HIP runtime failure, no usable gfx950 device    */

/* Size of the uniform block shared by all kernels (src/metadata.rs:21-41). */
#define COMPEG_METADATA_BYTES 1112
#define COMPEG_HUFFMAN_L1_BYTES 2048

typedef struct compeg_gpu compeg_gpu;
typedef struct compeg_decoder compeg_decoder;
typedef struct compeg_image compeg_image;
typedef struct compeg_scanbuffer compeg_scanbuffer;
typedef struct compeg_op compeg_op;
typedef struct compeg_batch compeg_batch;

/* Last error message of the calling thread ("" if none). */
const char *compeg_last_error(void);
/* Library version string, e.g. "compeg-hip 0.1.0 (gfx950)". */
const char *compeg_version(void);

/* ---- Gpu: src/lib.rs:64-270 (`Gpu::open`, `Gpu::from_wgpu`) -------------- */

/* Opens HIP device `device` (-1 = current device) and loads the gfx950 code
 * object.  The handle is reference-counted and immutable, i.e. shareable
 * between threads like `Arc<Gpu>`. */
int compeg_gpu_open(int device, compeg_gpu **out);
/* `from_wgpu` analogue: adopt a caller-owned stream as the queue that
 * start_decode / decode_blocking submit to. */
int compeg_gpu_from_stream(int device, void *hip_stream, compeg_gpu **out);
void compeg_gpu_retain(compeg_gpu *gpu);
void compeg_gpu_release(compeg_gpu *gpu);
int compeg_gpu_device(const compeg_gpu *gpu);
/* Human-readable device name ("AMD Instinct MI355X ..."), valid while gpu lives. */
const char *compeg_gpu_name(const compeg_gpu *gpu);

/* ---- ImageData: src/lib.rs:576-851 ---------------------------------------- */

/* Parses and validates a JPEG (`ImageData::new`).  copy != 0 keeps a private
 * copy of the bytes (Cow::Owned); copy == 0 borrows them (Cow::Borrowed): the
 * caller keeps `jpeg` alive and unchanged while the image is in use. */
int compeg_image_parse(const uint8_t *jpeg, size_t len, int copy, compeg_image **out);
/* Extension: flags widen the accepted subset.  With
 * COMPEG_PARSE_ANY_LUMA_SAMPLING the luma component may be sampled 1x1, 2x1,
 * 1x2 or 2x2 against 1x1 chroma (4:4:4, 4:2:2, 4:4:0, 4:2:0); the reference
 * rejects all but 2x1 (lib.rs:650-660).  4:4:4 decodes as the reference's
 * shaders would decode it if its front-end let it through; for the 16-row
 * MCUs of 4:4:0 / 4:2:0 the shaders' nearest-neighbour chroma rule is
 * continued vertically (oracle/compeg_oracle.c, orc_finalize_pass).  flags == 0
 * is compeg_image_parse. */
#define COMPEG_PARSE_ANY_LUMA_SAMPLING 1u
/* COMPEG_PARSE_STANDARD_ENTROPY: the image is entropy-decoded as ITU-T T.81 has
 * it in the two places where the reference deviates -- the bit reader is topped
 * up in front of DC codes as well (the reference does not, huffman.wgsl:157-160,
 * and loses the rest of a restart interval when a DC code needs more bits than
 * it has buffered), and ZRL skips 16 positions (the reference skips 17,
 * huffman.wgsl:176-179).  Everything else (32 retained coefficients, IDCT,
 * colour conversion) stays the reference's.  On valid streams the output then
 * no longer depends on the restart interval. */
#define COMPEG_PARSE_STANDARD_ENTROPY 2u
int compeg_image_parse_ext(const uint8_t *jpeg, size_t len, int copy, unsigned flags, compeg_image **out);
void compeg_image_free(compeg_image *img);
uint32_t compeg_image_width(const compeg_image *img);       /* lib.rs:828-831 */
uint32_t compeg_image_height(const compeg_image *img);      /* lib.rs:834-837 */
uint32_t compeg_image_parallelism(const compeg_image *img); /* lib.rs:838-846 */
/* What the reference uploads per image (src/lib.rs:397-407), exposed so that a
 * binding's tests can compare them byte for byte: the 1112-byte Metadata
 * block, the 2048-byte L1 LUT, the L2 LUT, and the location of the
 * entropy-coded segment inside the JPEG. */
const uint8_t *compeg_image_metadata(const compeg_image *img);
const uint8_t *compeg_image_huffman_l1(const compeg_image *img);
const uint8_t *compeg_image_huffman_l2(const compeg_image *img, size_t *nbytes);
void compeg_image_scan_range(const compeg_image *img, size_t *offset, size_t *len);

/* ---- ScanBuffer: src/scan.rs:15-77 (doc-hidden re-export, lib.rs:44-46) ---- */

compeg_scanbuffer *compeg_scanbuffer_new(void);
void compeg_scanbuffer_free(compeg_scanbuffer *sb);
/* `ScanBuffer::process`.  On COMPEG_E_COUNT_MISMATCH the buffers still hold
 * the (truncated) result, as in the reference. */
int compeg_scanbuffer_process(compeg_scanbuffer *sb, const uint8_t *scan, size_t len,
                              uint32_t expected_restart_intervals);
/* Extension: `threads` threads (1..16, default 1) share the work of every following process() call on
 * segments of 64 KiB per thread and more; same output.  The helper threads live as long as the buffer. */
int compeg_scanbuffer_set_threads(compeg_scanbuffer *sb, unsigned threads);
/* Same result, computed by the device-side scan kernels: the
 * segment is copied to HBM, preprocessed there and the two buffers are copied
 * back.  Inputs the kernels hand back (FF runs longer than 512 bytes) are
 * processed on the host. */
int compeg_scanbuffer_process_on_gpu(compeg_scanbuffer *sb, compeg_gpu *gpu, const uint8_t *scan,
                                     size_t len, uint32_t expected_restart_intervals);
/* `processed_scan_data()` / `start_positions()`: valid until the next process(). */
const uint8_t *compeg_scanbuffer_data(const compeg_scanbuffer *sb, size_t *nbytes);
const uint8_t *compeg_scanbuffer_start_positions(const compeg_scanbuffer *sb, size_t *nbytes);

/* ---- Decoder / DecodeOp: src/lib.rs:273-574 ------------------------------- */

int compeg_decoder_new(compeg_gpu *gpu, compeg_decoder **out); /* Decoder::new */
void compeg_decoder_free(compeg_decoder *dec);

/* `Decoder::enqueue(&ImageData, &mut CommandEncoder) -> bool` (lib.rs:385):
 * preprocess on the host, upload, and record the decode on `hip_stream` (the
 * command-encoder analogue) without waiting.  *texture_changed (optional) is
 * set to 1 when the output buffer was reallocated (always on the first call).
 * Like the reference, a restart-interval count mismatch does not stop the
 * decode (lib.rs:391-394 drops that error); it is reported through
 * compeg_decoder_last_warning(). */
int compeg_decoder_enqueue(compeg_decoder *dec, const compeg_image *img, void *hip_stream,
                           int *texture_changed);
/* `start_decode` (lib.rs:483-499): enqueue on the gpu's own stream + submit. */
int compeg_decoder_start_decode(compeg_decoder *dec, const compeg_image *img, compeg_op **op);
/* `decode_blocking` (lib.rs:508-529): start_decode + wait.  (With device-side scan preprocessing the
 * blocking call does without the read-back between the scan kernels and the decode kernel: it looks at the
 * scan kernels' verdict after the decode and, for the rare segment they hand back, decodes again through
 * the host preprocessor before returning.) */
int compeg_decoder_decode_blocking(compeg_decoder *dec, const compeg_image *img, compeg_op **op);
const char *compeg_decoder_last_warning(const compeg_decoder *dec);
/* Host time of the stages of the decoder's last decode, in microseconds -- the three timers the reference
 * traces (`t_preprocess` lib.rs:391-396, `t_enqueue_writes` lib.rs:452-475, `t_poll` lib.rs:516-522):
 *   preprocess_us      scan preprocessing on the host (ScanBuffer::process), or staging the raw segment and
 *                      submitting the scan kernels when preprocessing runs on the device;
 *   enqueue_writes_us  everything else `enqueue` does: tables and descriptor, uploads, kernel launch;
 *   poll_us            the wait for the device in decode_blocking (0 after enqueue / start_decode).
 * Plain data, C layout; all zeros before the first decode. */
typedef struct compeg_stage_times {
    double preprocess_us;
    double enqueue_writes_us;
    double poll_us;
} compeg_stage_times;
int compeg_decoder_last_stage_times(const compeg_decoder *dec, compeg_stage_times *out);
/* Diagnostics: which decode kernel the decoder's last enqueue (compeg_batch_last_kernel: the first launch of the
 * batch's last decode) went to -- the dispatch is by launch size and restart interval, and tests pin a case to the
 * path it is meant to cover.  The reference has one pipeline (lib.rs:436-448) and no counterpart. */
#define COMPEG_KERNEL_NONE 0       /* nothing launched yet (or an image without a complete restart interval) */
#define COMPEG_KERNEL_FUSED 1      /* decode_fused_422_kernel: a lane per restart interval, launches that fill the chip */
#define COMPEG_KERNEL_PAIR 2       /* decode_pair_422_kernel: decoder wave + transformer wave per 64 intervals */
#define COMPEG_KERNEL_COOP_TEAM 3  /* decode_coop_team_422_kernel: the lanes of a team work inside the intervals */
#define COMPEG_KERNEL_GENERIC 4    /* entropy_samples_kernel + composite_generic_kernel (batches of mixed layouts) */
#define COMPEG_KERNEL_SPLIT 5      /* entropy_kernel + idct_composite_kernel (development pipeline) */
#define COMPEG_KERNEL_FUSED_LAYOUT 6 /* decode_fused_444 / _440 / _420_kernel (one layout other than 4:2:2) */
#define COMPEG_KERNEL_FUSED_STREAM 7 /* decode_fused_422 / _444 / _440 / _420_stream_kernel: the batch kernels with streamed
                                       windows (long restart intervals, dense streams) */
#define COMPEG_KERNEL_WALK_MCU 8     /* walk_mcus_422_kernel + decode_fused_422_mcu_rec_kernel: a lane per restart interval
                                       finds where the MCUs begin, then a lane per MCU decodes (batches of few or long
                                       restart intervals; single images whose intervals the cooperative kernel does not
                                       take or takes less well: beyond 256 MCUs, no DRI at all, a 4K frame's 1080 teams) */
int compeg_decoder_last_kernel(const compeg_decoder *dec);
/* Extension: on != 0 moves the scan preprocessing of every following decode
 * from the host (the reference's data flow, default) to the device-side scan
 * kernels; the raw entropy-coded segment is uploaded instead of the
 * preprocessed one.  Results are identical. */
int compeg_decoder_set_device_preprocess(compeg_decoder *dec, int on);
/* Extension: threads the host scan preprocessor of this decoder uses for one image (see
 * compeg_scanbuffer_set_threads).  Default: 8 on hosts with 32 hardware threads or more, 4 with 8 or more, else 1;
 * the environment variable COMPEG_SCAN_THREADS overrides the default.  The default (only) checks itself:
 * the first large segment is timed on one thread and on all, and the helpers are dropped if they lose. */
int compeg_decoder_set_scan_threads(compeg_decoder *dec, unsigned threads);

/* `DecodeOp` (lib.rs:541-574).  compeg_op_wait replaces polling the
 * SubmissionIndex.  Ops are freed by the caller. */
int compeg_op_wait(compeg_op *op);
int compeg_op_texture_changed(const compeg_op *op);
void compeg_op_free(compeg_op *op);

/* `Decoder::texture()` (lib.rs:372-374): the device-resident RGBA8 output,
 * row-major, bytes R,G,B,255, `pitch` bytes per row -- the width rounded up to
 * 16 pixels (rows begin on 64-byte boundaries; the allocation has whole MCU rows
 * too: what an MCU at the texture's edge holds beyond it lands in that padding,
 * which nobody should read).  Like the reference's
 * texture it never shrinks, so width/height may exceed the last image; only
 * the image's own WxH corner is defined.  The pointer stays valid until the
 * next decode that reallocates (texture_changed) or the decoder is freed. */
int compeg_decoder_output(const compeg_decoder *dec, void **device_ptr, uint32_t *width,
                          uint32_t *height, size_t *pitch_bytes);
/* `Decoder::into_texture()` (lib.rs:380-383): transfers ownership of the
 * output allocation to the caller (release it with compeg_device_free) and
 * frees the decoder. */
int compeg_decoder_take_output(compeg_decoder *dec, void **device_ptr, uint32_t *width,
                               uint32_t *height, size_t *pitch_bytes);
void compeg_device_free(void *device_ptr);
/* Test-harness helper (src/tests.rs:52-84 does copy_texture_to_buffer):
 * waits for the decoder's pending work and copies the WxH corner to host
 * memory, tightly packed (4*width bytes per row). */
int compeg_decoder_read_output(compeg_decoder *dec, uint8_t *host_rgba, uint32_t width,
                               uint32_t height);
/* Debug read-back (role of DownloadBuffer, src/dynamic.rs:81-163): the
 * coefficient buffer as the reference's huffman pass leaves it,
 * int32[total_dus*32] at du*32 + zigzag_pos, dequantised. */
int compeg_decoder_read_coefficients(compeg_decoder *dec, int32_t *host_coefficients,
                                     size_t count);

/* ---- Batch (extension; not in the reference) -------------------------------
 * Decodes many independent images with one launch sequence: the images'
 * preprocessed scans and tables are made resident in HBM once
 * (compeg_batch_upload), after which every compeg_batch_decode is pure device
 * work.  Outputs are RGBA8 images in one device allocation, each with rows
 * `pitch` bytes apart (compeg_batch_output: 4 x the width rounded up to 16
 * pixels -- rows on 64-byte boundaries; compeg_batch_read_output hands out a
 * tightly packed copy). */
int compeg_batch_new(compeg_gpu *gpu, compeg_batch **out);
void compeg_batch_free(compeg_batch *batch);
/* Host front-end for all images (preprocess or stage on `host_threads` threads,
 * 0 = one per core, at most 16) + upload, every image sent off as soon as it is
 * ready; replaces any previous content. */
int compeg_batch_upload(compeg_batch *batch, const compeg_image *const *images, size_t count,
                        int host_threads);
/* Records the decode of every uploaded image on hip_stream (NULL = the gpu's
 * stream) and returns without waiting.  Decodes of one batch recorded on different
 * streams are ordered one behind the other (the batch's device buffers are one set). */
/* The same from JPEG bytes (host-fed use): `ImageData::new` for every image -- on the worker threads, it walks the
 * whole entropy-coded segment -- then what compeg_batch_upload does.  flags: COMPEG_PARSE_*.  The bytes are
 * borrowed until the call returns.  An image the front-end rejects fails the whole call with its error text,
 * prefixed "image <index>: ".  Two batches on two compeg_gpu handles (two streams) pipeline a stream of frames:
 * the upload of one runs under the decode of the other (bench.py, "end_to_end"). */
int compeg_batch_upload_jpegs(compeg_batch *batch, const uint8_t *const *jpegs, const size_t *lengths, size_t count,
                              int host_threads, unsigned flags);
/* The copy-free road of compeg_batch_upload_jpegs.  `ImageData::new` borrows the caller's bytes (`Cow::Borrowed`,
 * lib.rs:577-595) and the reference uploads straight from them (lib.rs:397-407).  With device preprocessing
 * (compeg_batch_set_device_preprocess 1 or 2) and JPEG bytes in page-locked memory -- from compeg_host_alloc, or
 * any range made known with compeg_host_register (a capture driver's buffers, a receive ring) -- the host reads
 * the headers only: the entropy-coded segments are fetched by the card's DMA engines from where they lie and
 * preprocessed by the scan kernels, which also check what the skipped walk over the segment would have found (an
 * image with a marker other than RSTn inside its segment is parsed again in full on the host).  Pageable bytes
 * work too; the worker threads then copy them into the batch's own pinned arena first.  The bytes must stay
 * valid and unchanged until the call returns. */
/* Measurement aid (bench.py --host-feed-ranks; no device is touched): the host's share of feeding one batch,
 * `reps` times over on `host_threads` threads -- road 0: ImageData::new + ScanBuffer::process of every image into its
 * place in an arena (what compeg_batch_upload_jpegs does on the host with host preprocessing: every byte read once
 * and written once); road 1: the headers only (the copy-free road above); road 2: the headers, and every file copied
 * into its place in the arena (pageable bytes with device preprocessing: the default of compeg_batch_upload_jpegs
 * without compeg_host_register).  *seconds: wall time of all reps. */
int compeg_host_feed_work(const uint8_t *const *jpegs, const size_t *lengths, size_t count, int host_threads, unsigned flags,
                          int road, int reps, double *seconds);
/* compeg_batch_upload_jpegs in two steps, for a feeder that keeps the link busy.  _begin returns as soon as every
 * transfer is queued (copy-free road: after a peek at the headers; on every other road it does the whole upload);
 * _end waits for the transfers and the scan kernels and finishes the batch (compeg_batch_decode does it too if
 * nobody has).  In between, the feeder begins the next batch's upload: its transfers queue up behind these, and the
 * link does not idle while this batch's results are read and its descriptors made.  The bytes must stay valid
 * and unchanged until _end returns; the batch must not be touched otherwise between the two. */
int compeg_batch_upload_jpegs_begin(compeg_batch *batch, const uint8_t *const *jpegs, const size_t *lengths, size_t count,
                                    int host_threads, unsigned flags);
int compeg_batch_upload_end(compeg_batch *batch);
int compeg_host_alloc(size_t bytes, void **out);
void compeg_host_free(void *ptr);
int compeg_host_register(void *ptr, size_t bytes);
int compeg_host_unregister(void *ptr);
int compeg_batch_decode(compeg_batch *batch, void *hip_stream);
/* Where the scans are preprocessed.  0 (default): on the host during
 * compeg_batch_upload, like the reference.  1: raw entropy-coded segments are
 * uploaded and preprocessed once by the scan kernels.  2: like 1, and every
 * compeg_batch_decode re-runs the scan kernels first, i.e. a decode covers the
 * whole path from raw scan bytes in HBM to RGBA.  Set before upload.  Every layout
 * the front-end accepts (the extension layouts of COMPEG_PARSE_ANY_LUMA_SAMPLING too). */
int compeg_batch_set_device_preprocess(compeg_batch *batch, int mode);
/* Images of the last upload that the scan kernels handed back to the host. */
size_t compeg_batch_host_fallbacks(const compeg_batch *batch);
/* Images per kernel-launch pair (0 = the whole batch in one pair, the default).
 * Smaller chunks keep the coefficient intermediates cache-resident. */
int compeg_batch_set_chunk(compeg_batch *batch, uint32_t images_per_launch);
int compeg_batch_wait(compeg_batch *batch);
size_t compeg_batch_count(const compeg_batch *batch);
/* Device pointer / geometry of image `index`'s output. */
int compeg_batch_output(const compeg_batch *batch, size_t index, void **device_ptr,
                        uint32_t *width, uint32_t *height, size_t *pitch_bytes);
int compeg_batch_read_output(compeg_batch *batch, size_t index, uint8_t *host_rgba);
/* Algorithmic bytes of the uploaded batch as SURVEY.md 8(d) defines them:
 * 4*scan_words + 4*intervals + 1112 + 2048 + L2 bytes + 4*W*H per image. */
uint64_t compeg_batch_algorithmic_bytes(const compeg_batch *batch);
uint64_t compeg_batch_pixels(const compeg_batch *batch);
/* Kernel-level timing of every compeg_batch_decode since the last upload or
 * reset (at most 4096 are kept), measured with HIP events recorded on the
 * stream the kernels ran on: number of decodes, summed total milliseconds and
 * (for unchunked decodes) summed per-stage milliseconds
 * [huffman, idct+composite].  Waits for those decodes to finish; reset != 0
 * then starts a new measurement. */
int compeg_batch_timing(compeg_batch *batch, int reset, uint32_t *decodes, double *total_ms,
                        double stage_ms[2]);
/* on == 0: decodes record no timing events (compeg_batch_timing then reports none).  Every event is a packet the card
 * works through between two kernels: back-to-back decodes of a single frame run closer together without them.
 * Default: on. */
int compeg_batch_set_timing(compeg_batch *batch, int on);
int compeg_batch_last_kernel(const compeg_batch *batch); /* COMPEG_KERNEL_* */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* COMPEG_HIP_H */
