// extern "C" surface of libcompeg_hip (include/compeg_hip.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <cstring>
#include <new>
#include <string>

#include "compeg_hip.h"
#include "runtime.h"

using namespace compeg;

struct compeg_image {
    ImageData *data;
};

struct compeg_scanbuffer {
    ScanBuffer buf;
};

namespace {

thread_local std::string g_error;

int fail(const Status &s)
{
    g_error = s.message;
    return s.code;
}

int fail(int code, const char *msg)
{
    g_error = msg;
    return code;
}

int ok()
{
    return COMPEG_OK;
}

// Runs body(), turning C++ exceptions (allocation failure) into an error code
// so that nothing propagates across the C boundary.
template <class F>
int guarded(F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(COMPEG_E_HIP, "out of host memory");
    } catch (const std::exception &e) {
        return fail(COMPEG_E_INVALID_ARG, e.what());
    }
}

int open_gpu(int device, hipStream_t stream, bool adopt, compeg_gpu **out)
{
    if (!out)
        return fail(COMPEG_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(COMPEG_E_HIP, "no HIP device available (libcompeg_hip needs a gfx950 GPU; "
                                  "there is no CPU fallback)");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess)
            device = 0;
    }
    if (device >= ndev)
        return fail(COMPEG_E_INVALID_ARG, "HIP device index out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess)
        return fail(hip_status(e, "hipGetDeviceProperties"));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string m = std::string("device is ") + prop.gcnArchName +
                        ", but libcompeg_hip carries gfx950 (MI355X) code objects only";
        return fail(COMPEG_E_HIP, m.c_str());
    }
    e = hipSetDevice(device);
    if (e != hipSuccess)
        return fail(hip_status(e, "hipSetDevice"));
    compeg_gpu *g = new compeg_gpu();
    g->device = device;
    g->name = prop.name[0] ? std::string(prop.name) + " (" + prop.gcnArchName + ")" : std::string(prop.gcnArchName);
    if (adopt) {
        g->stream = stream;
        g->owns_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete g;
            return fail(hip_status(e, "hipStreamCreate"));
        }
        g->owns_stream = true;
    }
    *out = g;
    return ok();
}

} // namespace

extern "C" {

const char *compeg_last_error(void)
{
    return g_error.c_str();
}

const char *compeg_version(void)
{
    return "compeg-hip 0.1.0 (gfx950)";
}

int compeg_gpu_open(int device, compeg_gpu **out)
{
    return guarded([&] { return open_gpu(device, nullptr, false, out); });
}

int compeg_gpu_from_stream(int device, void *hip_stream, compeg_gpu **out)
{
    return guarded([&] { return open_gpu(device, static_cast<hipStream_t>(hip_stream), true, out); });
}

void compeg_gpu_retain(compeg_gpu *gpu)
{
    if (gpu)
        gpu->refs.fetch_add(1, std::memory_order_relaxed);
}

void compeg_gpu_release(compeg_gpu *gpu)
{
    if (!gpu)
        return;
    if (gpu->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        if (gpu->owns_stream && gpu->stream) {
            (void)hipSetDevice(gpu->device);
            (void)hipStreamSynchronize(gpu->stream);
            (void)hipStreamDestroy(gpu->stream);
        }
        delete gpu;
    }
}

int compeg_gpu_device(const compeg_gpu *gpu)
{
    return gpu ? gpu->device : -1;
}

const char *compeg_gpu_name(const compeg_gpu *gpu)
{
    return gpu ? gpu->name.c_str() : "";
}

/* ---- ImageData ------------------------------------------------------------ */

int compeg_image_parse(const uint8_t *jpeg, size_t len, int copy, compeg_image **out)
{
    return compeg_image_parse_ext(jpeg, len, copy, 0u, out);
}

int compeg_image_parse_ext(const uint8_t *jpeg, size_t len, int copy, unsigned flags, compeg_image **out)
{
    return guarded([&] {
        if (!out)
            return fail(COMPEG_E_INVALID_ARG, "out is NULL");
        *out = nullptr;
        if (flags & ~(COMPEG_PARSE_ANY_LUMA_SAMPLING | COMPEG_PARSE_STANDARD_ENTROPY))
            return fail(COMPEG_E_INVALID_ARG, "unknown parse flags");
        ImageData *d = nullptr;
        Status s = ImageData::parse(jpeg, len, copy != 0, &d, flags);
        if (!s.ok())
            return fail(s);
        *out = new compeg_image{d};
        return ok();
    });
}

void compeg_image_free(compeg_image *img)
{
    if (img) {
        delete img->data;
        delete img;
    }
}

uint32_t compeg_image_width(const compeg_image *img)
{
    return img ? img->data->width : 0;
}

uint32_t compeg_image_height(const compeg_image *img)
{
    return img ? img->data->height : 0;
}

uint32_t compeg_image_parallelism(const compeg_image *img)
{
    return img ? img->data->metadata.total_restart_intervals : 0;
}

const uint8_t *compeg_image_metadata(const compeg_image *img)
{
    return img ? reinterpret_cast<const uint8_t *>(&img->data->metadata) : nullptr;
}

const uint8_t *compeg_image_huffman_l1(const compeg_image *img)
{
    return img ? reinterpret_cast<const uint8_t *>(img->data->l1) : nullptr;
}

const uint8_t *compeg_image_huffman_l2(const compeg_image *img, size_t *nbytes)
{
    if (nbytes)
        *nbytes = img ? img->data->l2.size() * 2 : 0;
    return img ? reinterpret_cast<const uint8_t *>(img->data->l2.data()) : nullptr;
}

void compeg_image_scan_range(const compeg_image *img, size_t *offset, size_t *len)
{
    if (offset)
        *offset = img ? img->data->scan_offset : 0;
    if (len)
        *len = img ? img->data->scan_len : 0;
}

/* ---- ScanBuffer ----------------------------------------------------------- */

compeg_scanbuffer *compeg_scanbuffer_new(void)
{
    try {
        return new compeg_scanbuffer();
    } catch (...) {
        return nullptr;
    }
}

void compeg_scanbuffer_free(compeg_scanbuffer *sb)
{
    delete sb;
}

int compeg_scanbuffer_process(compeg_scanbuffer *sb, const uint8_t *scan, size_t len,
                              uint32_t expected_restart_intervals)
{
    return guarded([&] {
        if (!sb || (!scan && len))
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        Status s = sb->buf.process(scan, len, expected_restart_intervals);
        return s.ok() ? ok() : fail(s);
    });
}

int compeg_scanbuffer_set_threads(compeg_scanbuffer *sb, unsigned threads)
{
    return guarded([&] {
        if (!sb || threads < 1 || threads > 16)
            return fail(COMPEG_E_INVALID_ARG, "threads must be 1..16");
        sb->buf.set_threads(threads);
        return ok();
    });
}

int compeg_scanbuffer_process_on_gpu(compeg_scanbuffer *sb, compeg_gpu *gpu, const uint8_t *scan,
                                     size_t len, uint32_t expected_restart_intervals)
{
    return guarded([&] {
        if (!sb || !gpu || (!scan && len))
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        Status s = sb->buf.process_on_gpu(gpu, scan, len, expected_restart_intervals);
        return s.ok() ? ok() : fail(s);
    });
}

const uint8_t *compeg_scanbuffer_data(const compeg_scanbuffer *sb, size_t *nbytes)
{
    if (nbytes)
        *nbytes = sb ? sb->buf.data_bytes() : 0;
    return sb ? sb->buf.data() : nullptr;
}

const uint8_t *compeg_scanbuffer_start_positions(const compeg_scanbuffer *sb, size_t *nbytes)
{
    if (nbytes)
        *nbytes = sb ? sb->buf.nstarts() * 4 : 0;
    return sb ? reinterpret_cast<const uint8_t *>(sb->buf.starts()) : nullptr;
}

/* ---- Decoder -------------------------------------------------------------- */

int compeg_decoder_new(compeg_gpu *gpu, compeg_decoder **out)
{
    return guarded([&] {
        if (!gpu || !out)
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        hipError_t e = hipSetDevice(gpu->device);
        if (e != hipSuccess)
            return fail(hip_status(e, "hipSetDevice"));
        compeg_decoder *d = new compeg_decoder();
        compeg_gpu_retain(gpu);
        d->gpu = gpu;
        *out = d;
        return ok();
    });
}

void compeg_decoder_free(compeg_decoder *dec)
{
    if (dec) {
        (void)hipSetDevice(dec->gpu->device);
        delete dec;
    }
}

int compeg_decoder_enqueue(compeg_decoder *dec, const compeg_image *img, void *hip_stream,
                           int *texture_changed)
{
    return guarded([&] {
        if (!dec || !img)
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        bool changed = false;
        Status s = dec->enqueue(*img->data, static_cast<hipStream_t>(hip_stream), &changed);
        if (texture_changed)
            *texture_changed = changed ? 1 : 0;
        return s.ok() ? ok() : fail(s);
    });
}

int compeg_decoder_start_decode(compeg_decoder *dec, const compeg_image *img, compeg_op **op)
{
    return guarded([&] {
        if (!dec || !img || !op)
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        *op = nullptr;
        bool changed = false;
        Status s = dec->enqueue(*img->data, dec->gpu->stream, &changed);
        if (!s.ok())
            return fail(s);
        compeg_op *o = new compeg_op();
        o->device = dec->gpu->device;
        o->texture_changed = changed;
        hipError_t e = hipEventCreateWithFlags(&o->done, hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipEventRecord(o->done, dec->gpu->stream);
        if (e != hipSuccess) {
            compeg_op_free(o);
            return fail(hip_status(e, "hipEventRecord"));
        }
        *op = o;
        return ok();
    });
}

int compeg_decoder_decode_blocking(compeg_decoder *dec, const compeg_image *img, compeg_op **op)
{
    // like start_decode + wait; a blocking decode may run the device-side scan preprocessing
    // without the read-back in the middle and look at its outcome afterwards
    return guarded([&] {
        if (!dec || !img || !op)
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        *op = nullptr;
        bool changed = false;
        Status s = dec->enqueue(*img->data, dec->gpu->stream, &changed, true);
        if (!s.ok())
            return fail(s);
        const auto t_poll = std::chrono::steady_clock::now();
        hipError_t e = hipStreamSynchronize(dec->gpu->stream);
        if (e != hipSuccess)
            return fail(hip_status(e, "hipStreamSynchronize"));
        // (the stream is drained: nothing of this decode is pending -- enqueue recorded no events for a blocking decode)
        dec->upload_pending = dec->decode_pending = false;
        const compeg_stage_times first = dec->stage_times; // (a re-decode through the host path would reset them)
        s = dec->finish_deferred(*img->data, dec->gpu->stream);
        if (!s.ok())
            return fail(s);
        dec->stage_times = first;
        dec->stage_times.poll_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_poll).count();
        // (an operation that is complete already: no event to wait for)
        compeg_op *o = new compeg_op();
        o->device = dec->gpu->device;
        o->texture_changed = changed;
        *op = o;
        return ok();
    });
}

int compeg_decoder_set_device_preprocess(compeg_decoder *dec, int on)
{
    if (!dec)
        return fail(COMPEG_E_INVALID_ARG, "dec is NULL");
    dec->device_preprocess = on != 0;
    return ok();
}

int compeg_decoder_set_scan_threads(compeg_decoder *dec, unsigned threads)
{
    return guarded([&] {
        if (!dec || threads < 1 || threads > 16)
            return fail(COMPEG_E_INVALID_ARG, "threads must be 1..16");
        dec->scan.set_threads(threads);
        return ok();
    });
}

const char *compeg_decoder_last_warning(const compeg_decoder *dec)
{
    return dec ? dec->warning.c_str() : "";
}

int compeg_decoder_last_kernel(const compeg_decoder *dec) { return dec ? dec->last_kernel : COMPEG_KERNEL_NONE; }

int compeg_decoder_last_stage_times(const compeg_decoder *dec, compeg_stage_times *out)
{
    if (!dec || !out)
        return fail(COMPEG_E_INVALID_ARG, "NULL argument");
    *out = dec->stage_times;
    return ok();
}

int compeg_op_wait(compeg_op *op)
{
    if (!op)
        return fail(COMPEG_E_INVALID_ARG, "op is NULL");
    if (!op->done)
        return ok(); // (a blocking decode's: complete when it was made)
    hipError_t e = hipEventSynchronize(op->done);
    return e == hipSuccess ? ok() : fail(hip_status(e, "hipEventSynchronize"));
}

int compeg_op_texture_changed(const compeg_op *op)
{
    return op && op->texture_changed ? 1 : 0;
}

void compeg_op_free(compeg_op *op)
{
    if (op) {
        if (op->done)
            (void)hipEventDestroy(op->done);
        delete op;
    }
}

int compeg_decoder_output(const compeg_decoder *dec, void **device_ptr, uint32_t *width,
                          uint32_t *height, size_t *pitch_bytes)
{
    if (!dec)
        return fail(COMPEG_E_INVALID_ARG, "dec is NULL");
    if (device_ptr)
        *device_ptr = dec->out.ptr;
    if (width)
        *width = dec->out_w;
    if (height)
        *height = dec->out_h;
    if (pitch_bytes)
        *pitch_bytes = dec->out_pitch;
    return ok();
}

int compeg_decoder_take_output(compeg_decoder *dec, void **device_ptr, uint32_t *width,
                               uint32_t *height, size_t *pitch_bytes)
{
    if (!dec || !device_ptr)
        return fail(COMPEG_E_INVALID_ARG, "NULL argument");
    (void)hipSetDevice(dec->gpu->device);
    hipError_t e = hipStreamSynchronize(dec->last_stream);
    if (e != hipSuccess)
        return fail(hip_status(e, "hipStreamSynchronize"));
    compeg_decoder_output(dec, nullptr, width, height, pitch_bytes);
    *device_ptr = dec->out.release();
    delete dec;
    return ok();
}

void compeg_device_free(void *device_ptr)
{
    if (device_ptr)
        (void)hipFree(device_ptr);
}

int compeg_decoder_read_output(compeg_decoder *dec, uint8_t *host_rgba, uint32_t width,
                               uint32_t height)
{
    return guarded([&] {
        if (!dec || !host_rgba)
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        if (width > dec->out_w || height > dec->out_h)
            return fail(COMPEG_E_INVALID_ARG, "requested area exceeds the output");
        (void)hipSetDevice(dec->gpu->device);
        hipError_t e = hipStreamSynchronize(dec->last_stream);
        if (e == hipSuccess && width && height)
            e = hipMemcpy2D(host_rgba, size_t(width) * 4, dec->out.ptr, dec->out_pitch,
                            size_t(width) * 4, height, hipMemcpyDeviceToHost);
        return e == hipSuccess ? ok() : fail(hip_status(e, "read_output"));
    });
}

int compeg_decoder_read_coefficients(compeg_decoder *dec, int32_t *host, size_t count)
{
    return guarded([&] {
        if (!dec || !host || !dec->have_last)
            return fail(COMPEG_E_INVALID_ARG, "no decode to read back");
        const Metadata &md = dec->last_md;
        const size_t dus =
            size_t(md.total_restart_intervals) * md.restart_interval * md.dus_per_mcu;
        if (count < dus * kRetained)
            return fail(COMPEG_E_INVALID_ARG, "coefficient buffer too small");
        (void)hipSetDevice(dec->gpu->device);
        hipError_t e = hipStreamSynchronize(dec->last_stream);
        if (e != hipSuccess)
            return fail(hip_status(e, "hipStreamSynchronize"));
        std::vector<int16_t> ac(dus * kRetained);
        std::vector<int32_t> dc(dus);
        if (dus && !dec->coefficients_valid) {
            // the fused kernel keeps coefficients on chip: rerun the entropy
            // stage alone into the scratch buffers for this debug read-back
            const HuffLdsPlan plan = plan_huffman(md.total_restart_intervals, 1, dec->last_plan.l2_entries_in_lds,
                                                  dec->last_span, false);
            e = launch_huffman(static_cast<const ImageDesc *>(dec->last_desc_dev), 1,
                               md.total_restart_intervals, plan, dec->last_stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(dec->last_stream);
            if (e != hipSuccess)
                return fail(hip_status(e, "huffman_kernel (coefficient read-back)"));
            dec->coefficients_valid = true;
        }
        if (dus) {
            e = hipMemcpy(ac.data(), dec->ac.ptr, ac.size() * 2, hipMemcpyDeviceToHost);
            if (e == hipSuccess)
                e = hipMemcpy(dc.data(), dec->dc.ptr, dc.size() * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess)
                return fail(hip_status(e, "hipMemcpy"));
        }
        // component of every data unit inside an MCU, in scan order
        uint32_t comp_of[kMaxDusPerMcu] = {0};
        uint32_t k = 0;
        for (uint32_t c = 0; c < 3; c++)
            for (uint32_t i = 0; i < md.components[c].hsample * md.components[c].vsample &&
                                 k < uint32_t(kMaxDusPerMcu);
                 i++)
                comp_of[k++] = c;
        for (size_t du = 0; du < dus; du++) {
            const uint32_t *q = md.qtables[md.components[comp_of[du % md.dus_per_mcu]].qtable & 3];
            host[du * kRetained] = dc[du];
            for (int z = 1; z < kRetained; z++)
                host[du * kRetained + z] = int32_t(uint32_t(int32_t(ac[du * kRetained + z])) * q[z]);
        }
        return ok();
    });
}

#if defined(CG_AC_STAMPS)
extern "C" __attribute__((visibility("default"))) int compeg_debug_ac_stamps(unsigned long long *out, int reset)
{
    return compeg::read_ac_stamps(out, reset != 0) == hipSuccess ? 0 : -1;
}
#endif

/* Diagnostic builds (-DCG_STAMPS) park per-wave cycle stamps in the dc scratch buffers; these
 * two undeclared helpers read them back.  Not part of the API. */
__attribute__((visibility("default"))) int compeg_debug_read_dc(compeg_decoder *dec, void *host, size_t bytes)
{
    if (!dec || !dec->dc.ptr || bytes > dec->dc.capacity)
        return COMPEG_E_INVALID_ARG;
    (void)hipStreamSynchronize(dec->last_stream);
    return hipMemcpy(host, dec->dc.ptr, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : COMPEG_E_HIP;
}

__attribute__((visibility("default"))) int compeg_debug_read_batch_dc(compeg_batch *b, void *host, size_t bytes)
{
    if (!b || !b->dc.ptr || bytes > b->dc.capacity)
        return COMPEG_E_INVALID_ARG;
    (void)hipStreamSynchronize(b->last_stream);
    return hipMemcpy(host, b->dc.ptr, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : COMPEG_E_HIP;
}

/* ---- Batch ---------------------------------------------------------------- */

int compeg_batch_new(compeg_gpu *gpu, compeg_batch **out)
{
    return guarded([&] {
        if (!gpu || !out)
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        compeg_batch *b = new compeg_batch();
        compeg_gpu_retain(gpu);
        b->gpu = gpu;
        *out = b;
        return ok();
    });
}

void compeg_batch_free(compeg_batch *batch)
{
    if (batch) {
        (void)hipSetDevice(batch->gpu->device);
        delete batch;
    }
}

int compeg_batch_upload(compeg_batch *batch, const compeg_image *const *images, size_t count,
                        int host_threads)
{
    return guarded([&] {
        if (!batch || (!images && count))
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        std::vector<const ImageData *> ptrs(count);
        for (size_t i = 0; i < count; i++) {
            if (!images[i])
                return fail(COMPEG_E_INVALID_ARG, "NULL image in batch");
            ptrs[i] = images[i]->data;
        }
        Status s = batch->upload(ptrs.data(), count, host_threads);
        return s.ok() ? ok() : fail(s);
    });
}

int compeg_batch_upload_jpegs(compeg_batch *batch, const uint8_t *const *jpegs, const size_t *lengths, size_t count,
                              int host_threads, unsigned flags)
{
    return guarded([&] {
        if (!batch || ((!jpegs || !lengths) && count))
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        for (size_t i = 0; i < count; i++)
            if (!jpegs[i])
                return fail(COMPEG_E_INVALID_ARG, "NULL image in batch");
        Status s = batch->upload_jpegs(jpegs, lengths, count, host_threads, flags);
        return s.ok() ? ok() : fail(s);
    });
}

int compeg_batch_upload_jpegs_begin(compeg_batch *batch, const uint8_t *const *jpegs, const size_t *lengths, size_t count,
                                    int host_threads, unsigned flags)
{
    return guarded([&] {
        if (!batch || ((!jpegs || !lengths) && count))
            return fail(COMPEG_E_INVALID_ARG, "NULL argument");
        for (size_t i = 0; i < count; i++)
            if (!jpegs[i])
                return fail(COMPEG_E_INVALID_ARG, "NULL image in batch");
        Status s = batch->upload_jpegs(jpegs, lengths, count, host_threads, flags, true);
        return s.ok() ? ok() : fail(s);
    });
}

int compeg_batch_upload_end(compeg_batch *batch)
{
    return guarded([&] {
        if (!batch)
            return fail(COMPEG_E_INVALID_ARG, "batch is NULL");
        Status s = batch->finish_upload();
        return s.ok() ? ok() : fail(s);
    });
}

int compeg_batch_decode(compeg_batch *batch, void *hip_stream)
{
    return guarded([&] {
        if (!batch)
            return fail(COMPEG_E_INVALID_ARG, "batch is NULL");
        hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : batch->gpu->stream;
        Status s = batch->decode(st);
        return s.ok() ? ok() : fail(s);
    });
}

int compeg_batch_wait(compeg_batch *batch)
{
    if (!batch)
        return fail(COMPEG_E_INVALID_ARG, "batch is NULL");
    (void)hipSetDevice(batch->gpu->device);
    hipError_t e = hipStreamSynchronize(batch->last_stream ? batch->last_stream : batch->gpu->stream);
    return e == hipSuccess ? ok() : fail(hip_status(e, "hipStreamSynchronize"));
}

size_t compeg_batch_count(const compeg_batch *batch)
{
    return batch ? batch->count : 0;
}

int compeg_batch_set_device_preprocess(compeg_batch *batch, int mode)
{
    if (!batch || mode < 0 || mode > 2)
        return fail(COMPEG_E_INVALID_ARG, "bad argument");
    batch->preprocess_mode = mode;
    return ok();
}

size_t compeg_batch_host_fallbacks(const compeg_batch *batch)
{
    return batch ? batch->host_fallbacks : 0;
}

int compeg_batch_set_chunk(compeg_batch *batch, uint32_t images_per_launch)
{
    return guarded([&] {
        if (!batch)
            return fail(COMPEG_E_INVALID_ARG, "batch is NULL");
        if (batch->chunk == images_per_launch)
            return ok();
        batch->chunk = images_per_launch;
        // (the launches change size: some of them may now be the cooperative kernel's, which wants its walk tables, or
        // take the walk + lane-per-MCU route, which wants its records' buffers)
        if (batch->count) {
            if (batch->last_stream && hipStreamSynchronize(batch->last_stream) != hipSuccess)
                return fail(COMPEG_E_HIP, "hipStreamSynchronize failed");
            if (hipSetDevice(batch->gpu->device) != hipSuccess)
                return fail(COMPEG_E_HIP, "hipSetDevice failed");
            compeg::Status st = batch->make_walk_tables(batch->gpu->stream, batch->count);
            if (!st.ok())
                return fail(st);
            if (hipStreamSynchronize(batch->gpu->stream) != hipSuccess)
                return fail(COMPEG_E_HIP, "hipStreamSynchronize failed");
        }
        return ok();
    });
}

int compeg_host_feed_work(const uint8_t *const *jpegs, const size_t *lengths, size_t count, int host_threads, unsigned flags,
                          int road, int reps, double *seconds)
{
    return guarded([&] {
        if (!jpegs || !lengths || !seconds || count == 0 || reps <= 0 || road < 0 || road > 2)
            return fail(COMPEG_E_INVALID_ARG, "bad arguments");
        const unsigned nthreads = unsigned(std::max(1, host_threads));
        // Roads 0 and 2 write what they make where an upload would: every image has a place of its own in one arena
        // (kept between calls: an upload's pinned arena is), so that a batch larger than the caches costs the memory
        // traffic it costs there -- every byte read once and written once.
        static std::mutex arena_lock;
        static std::vector<uint8_t> arena;
        std::vector<size_t> at(count + 1, 0);
        for (size_t i = 0; i < count; i++)
            at[i + 1] = at[i] + ((road == 0 ? compeg::ScanBuffer::output_capacity(lengths[i]) : lengths[i]) + 255) / 256 * 256;
        std::unique_lock<std::mutex> hold(arena_lock);
        if (road != 1 && arena.size() < at[count])
            arena.resize(at[count]);
        std::vector<std::vector<uint32_t>> starts(nthreads);
        std::vector<compeg::Status> status(nthreads);
        std::atomic<size_t> next{0};
        const size_t total = count * size_t(reps);
        auto work = [&](unsigned t) {
            for (;;) {
                const size_t k = next.fetch_add(1, std::memory_order_relaxed);
                if (k >= total || !status[t].ok())
                    return;
                const size_t i = k % count;
                compeg::ImageData *img = nullptr;
                compeg::Status st = compeg::ImageData::parse(jpegs[i], lengths[i], false, &img,
                                                             road != 0 ? flags | compeg::kParseDeferScanEnd : flags);
                std::unique_ptr<compeg::ImageData> owned(img);
                if (st.ok() && road == 0) {
                    const uint32_t expected = img->metadata.total_restart_intervals;
                    starts[t].resize(compeg::ScanBuffer::start_slots(expected));
                    size_t nwords = 0, nstarts = 0;
                    st = compeg::ScanBuffer::process_to(img->scan_data(), img->scan_len, expected, arena.data() + at[i], starts[t].data(),
                                                        nwords, nstarts);
                    if (st.code == COMPEG_E_COUNT_MISMATCH)
                        st = compeg::Status{};
                } else if (st.ok() && road == 2) {
                    memcpy(arena.data() + at[i], jpegs[i], lengths[i]); // (pageable bytes: into the pinned arena, whole)
                }
                if (!st.ok())
                    status[t] = st;
            }
        };
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> threads;
        struct JoinAll {
            std::vector<std::thread> &v;
            ~JoinAll()
            {
                for (std::thread &th : v)
                    if (th.joinable())
                        th.join();
            }
        } join_all{threads}; // (also when starting a thread throws midway)
        threads.reserve(nthreads);
        for (unsigned t = 1; t < nthreads; t++)
            threads.emplace_back(work, t);
        work(0);
        for (std::thread &th : threads)
            th.join();
        *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (const compeg::Status &st : status)
            if (!st.ok())
                return fail(st);
        return ok();
    });
}

int compeg_host_alloc(size_t bytes, void **out)
{
    if (!out)
        return fail(COMPEG_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess || !p)
        return fail(COMPEG_E_HIP, "hipHostMalloc failed");
    *out = p;
    return ok();
}

void compeg_host_free(void *ptr)
{
    if (ptr)
        (void)hipHostFree(ptr);
}

int compeg_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes)
        return fail(COMPEG_E_INVALID_ARG, "nothing to register");
    if (hipHostRegister(ptr, bytes, hipHostRegisterPortable) != hipSuccess)
        return fail(COMPEG_E_HIP, "hipHostRegister failed");
    return ok();
}

int compeg_host_unregister(void *ptr)
{
    if (!ptr)
        return fail(COMPEG_E_INVALID_ARG, "ptr is NULL");
    if (hipHostUnregister(ptr) != hipSuccess)
        return fail(COMPEG_E_HIP, "hipHostUnregister failed");
    return ok();
}

int compeg_batch_last_kernel(const compeg_batch *batch) { return batch ? batch->last_kernel : COMPEG_KERNEL_NONE; }

int compeg_batch_output(const compeg_batch *batch, size_t index, void **device_ptr, uint32_t *width,
                        uint32_t *height, size_t *pitch_bytes)
{
    if (!batch || index >= batch->count)
        return fail(COMPEG_E_INVALID_ARG, "batch index out of range");
    const ImageDesc &d = batch->descs[index];
    if (device_ptr)
        *device_ptr = d.out;
    if (width)
        *width = d.out_w;
    if (height)
        *height = d.out_h;
    if (pitch_bytes)
        *pitch_bytes = d.out_pitch;
    return ok();
}

int compeg_batch_read_output(compeg_batch *batch, size_t index, uint8_t *host_rgba)
{
    return guarded([&] {
        if (!batch || index >= batch->count || !host_rgba)
            return fail(COMPEG_E_INVALID_ARG, "bad argument");
        int rc = compeg_batch_wait(batch);
        if (rc != COMPEG_OK)
            return rc;
        const ImageDesc &d = batch->descs[index];
        // (tightly packed on the host; the device rows are out_pitch apart)
        hipError_t e = d.out_w && d.out_h ? hipMemcpy2D(host_rgba, size_t(d.out_w) * 4, d.out, d.out_pitch, size_t(d.out_w) * 4, d.out_h,
                                                        hipMemcpyDeviceToHost)
                                          : hipSuccess;
        return e == hipSuccess ? ok() : fail(hip_status(e, "hipMemcpy2D"));
    });
}

uint64_t compeg_batch_algorithmic_bytes(const compeg_batch *batch)
{
    return batch ? batch->algorithmic_bytes : 0;
}

uint64_t compeg_batch_pixels(const compeg_batch *batch)
{
    return batch ? batch->pixels : 0;
}

int compeg_batch_set_timing(compeg_batch *batch, int on)
{
    if (!batch)
        return fail(COMPEG_E_INVALID_ARG, "batch is NULL");
    batch->timing_on = on != 0;
    return ok();
}

int compeg_batch_timing(compeg_batch *batch, int reset, uint32_t *decodes, double *total_ms,
                        double stage_ms[2])
{
    return guarded([&] {
        if (!batch)
            return fail(COMPEG_E_INVALID_ARG, "batch is NULL");
        (void)hipSetDevice(batch->gpu->device);
        double t = 0, a = 0, b = 0;
        const bool split = !batch->chunk || batch->chunk >= batch->count;
        for (size_t i = 0; i < batch->decodes_timed; i++) {
            hipEvent_t *ev = batch->events.data() + i * 3;
            hipError_t e = hipEventSynchronize(ev[2]);
            float x = 0, y = 0, z = 0;
            if (e == hipSuccess)
                e = hipEventElapsedTime(&x, ev[0], ev[2]);
            const bool staged = i < batch->has_stage_event.size() && batch->has_stage_event[i];
            if (e == hipSuccess && split && staged) {
                e = hipEventElapsedTime(&y, ev[0], ev[1]);
                if (e == hipSuccess)
                    e = hipEventElapsedTime(&z, ev[1], ev[2]);
            } else if (e == hipSuccess && split) {
                y = x; // (one kernel does the whole path)
            }
            if (e != hipSuccess)
                return fail(hip_status(e, "hipEventElapsedTime"));
            t += x;
            a += y;
            b += z;
        }
        if (decodes)
            *decodes = uint32_t(batch->decodes_timed);
        if (total_ms)
            *total_ms = t;
        if (stage_ms) {
            stage_ms[0] = a;
            stage_ms[1] = b;
        }
        if (reset)
            batch->decodes_timed = 0;
        return ok();
    });
}

} // extern "C"
