//! Raw declarations of `include/compeg_hip.h`, one group per reference type.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_double, c_int, c_uint, c_void};

macro_rules! opaque {
    ($($name:ident),*) => { $(#[repr(C)] pub struct $name { _private: [u8; 0] })* };
}
opaque!(compeg_gpu, compeg_decoder, compeg_image, compeg_scanbuffer, compeg_op, compeg_batch);

/// Host microseconds of a decode's stages (the reference's trace timers, src/lib.rs:391-396,452-475,516-522).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct compeg_stage_times {
    pub preprocess_us: c_double,
    pub enqueue_writes_us: c_double,
    pub poll_us: c_double,
}

pub const COMPEG_OK: c_int = 0;
pub const COMPEG_E_INVALID_ARG: c_int = -1;
pub const COMPEG_E_UNSUPPORTED: c_int = -2;
pub const COMPEG_E_MALFORMED: c_int = -3;
pub const COMPEG_E_COUNT_MISMATCH: c_int = -4;
pub const COMPEG_E_HIP: c_int = -5;
pub const COMPEG_PARSE_ANY_LUMA_SAMPLING: c_uint = 1;
pub const COMPEG_PARSE_STANDARD_ENTROPY: c_uint = 2;
pub const COMPEG_KERNEL_NONE: c_int = 0;
pub const COMPEG_KERNEL_FUSED: c_int = 1;
pub const COMPEG_KERNEL_PAIR: c_int = 2;
pub const COMPEG_KERNEL_COOP_TEAM: c_int = 3;
pub const COMPEG_KERNEL_GENERIC: c_int = 4;
pub const COMPEG_KERNEL_SPLIT: c_int = 5;
pub const COMPEG_KERNEL_FUSED_LAYOUT: c_int = 6;
pub const COMPEG_KERNEL_FUSED_STREAM: c_int = 7;

extern "C" {
    pub fn compeg_last_error() -> *const c_char;
    pub fn compeg_version() -> *const c_char;

    // Gpu
    pub fn compeg_gpu_open(device: c_int, out: *mut *mut compeg_gpu) -> c_int;
    pub fn compeg_gpu_from_stream(device: c_int, hip_stream: *mut c_void, out: *mut *mut compeg_gpu) -> c_int;
    pub fn compeg_gpu_retain(gpu: *mut compeg_gpu);
    pub fn compeg_gpu_release(gpu: *mut compeg_gpu);
    pub fn compeg_gpu_device(gpu: *const compeg_gpu) -> c_int;
    pub fn compeg_gpu_name(gpu: *const compeg_gpu) -> *const c_char;

    // ImageData
    pub fn compeg_image_parse(jpeg: *const u8, len: usize, copy: c_int, out: *mut *mut compeg_image) -> c_int;
    pub fn compeg_image_parse_ext(jpeg: *const u8, len: usize, copy: c_int, flags: c_uint,
                                  out: *mut *mut compeg_image) -> c_int;
    pub fn compeg_image_free(img: *mut compeg_image);
    pub fn compeg_image_width(img: *const compeg_image) -> u32;
    pub fn compeg_image_height(img: *const compeg_image) -> u32;
    pub fn compeg_image_parallelism(img: *const compeg_image) -> u32;
    pub fn compeg_image_metadata(img: *const compeg_image) -> *const u8;
    pub fn compeg_image_huffman_l1(img: *const compeg_image) -> *const u8;
    pub fn compeg_image_huffman_l2(img: *const compeg_image, nbytes: *mut usize) -> *const u8;
    pub fn compeg_image_scan_range(img: *const compeg_image, offset: *mut usize, len: *mut usize);

    // ScanBuffer
    pub fn compeg_scanbuffer_new() -> *mut compeg_scanbuffer;
    pub fn compeg_scanbuffer_free(sb: *mut compeg_scanbuffer);
    pub fn compeg_scanbuffer_process(sb: *mut compeg_scanbuffer, scan: *const u8, len: usize, expected: u32) -> c_int;
    pub fn compeg_scanbuffer_set_threads(sb: *mut compeg_scanbuffer, threads: c_uint) -> c_int;
    pub fn compeg_scanbuffer_process_on_gpu(sb: *mut compeg_scanbuffer, gpu: *mut compeg_gpu, scan: *const u8,
                                            len: usize, expected: u32) -> c_int;
    pub fn compeg_scanbuffer_data(sb: *const compeg_scanbuffer, nbytes: *mut usize) -> *const u8;
    pub fn compeg_scanbuffer_start_positions(sb: *const compeg_scanbuffer, nbytes: *mut usize) -> *const u8;

    // Decoder / DecodeOp
    pub fn compeg_decoder_new(gpu: *mut compeg_gpu, out: *mut *mut compeg_decoder) -> c_int;
    pub fn compeg_decoder_free(dec: *mut compeg_decoder);
    pub fn compeg_decoder_enqueue(dec: *mut compeg_decoder, img: *const compeg_image, hip_stream: *mut c_void,
                                  texture_changed: *mut c_int) -> c_int;
    pub fn compeg_decoder_start_decode(dec: *mut compeg_decoder, img: *const compeg_image,
                                       op: *mut *mut compeg_op) -> c_int;
    pub fn compeg_decoder_decode_blocking(dec: *mut compeg_decoder, img: *const compeg_image,
                                          op: *mut *mut compeg_op) -> c_int;
    pub fn compeg_decoder_last_warning(dec: *const compeg_decoder) -> *const c_char;
    pub fn compeg_decoder_last_stage_times(dec: *const compeg_decoder, out: *mut compeg_stage_times) -> c_int;
    pub fn compeg_decoder_last_kernel(dec: *const compeg_decoder) -> c_int;
    pub fn compeg_decoder_set_device_preprocess(dec: *mut compeg_decoder, on: c_int) -> c_int;
    pub fn compeg_decoder_set_scan_threads(dec: *mut compeg_decoder, threads: c_uint) -> c_int;
    pub fn compeg_op_wait(op: *mut compeg_op) -> c_int;
    pub fn compeg_op_texture_changed(op: *const compeg_op) -> c_int;
    pub fn compeg_op_free(op: *mut compeg_op);
    pub fn compeg_decoder_output(dec: *const compeg_decoder, device_ptr: *mut *mut c_void, width: *mut u32,
                                 height: *mut u32, pitch_bytes: *mut usize) -> c_int;
    pub fn compeg_decoder_take_output(dec: *mut compeg_decoder, device_ptr: *mut *mut c_void, width: *mut u32,
                                      height: *mut u32, pitch_bytes: *mut usize) -> c_int;
    pub fn compeg_device_free(device_ptr: *mut c_void);
    pub fn compeg_decoder_read_output(dec: *mut compeg_decoder, host_rgba: *mut u8, width: u32, height: u32) -> c_int;
    pub fn compeg_decoder_read_coefficients(dec: *mut compeg_decoder, host: *mut i32, count: usize) -> c_int;

    // Batch (extension)
    pub fn compeg_batch_new(gpu: *mut compeg_gpu, out: *mut *mut compeg_batch) -> c_int;
    pub fn compeg_batch_free(batch: *mut compeg_batch);
    pub fn compeg_batch_upload(batch: *mut compeg_batch, images: *const *const compeg_image, count: usize,
                               host_threads: c_int) -> c_int;
    pub fn compeg_batch_upload_jpegs(batch: *mut compeg_batch, jpegs: *const *const u8, lengths: *const usize,
                                     count: usize, host_threads: c_int, flags: c_uint) -> c_int;
    pub fn compeg_host_feed_work(jpegs: *const *const u8, lengths: *const usize, count: usize, host_threads: c_int,
                                 flags: c_uint, road: c_int, reps: c_int, seconds: *mut c_double) -> c_int;
    pub fn compeg_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn compeg_host_free(ptr: *mut c_void);
    pub fn compeg_host_register(ptr: *mut c_void, bytes: usize) -> c_int;
    pub fn compeg_host_unregister(ptr: *mut c_void) -> c_int;
    pub fn compeg_batch_upload_jpegs_begin(batch: *mut compeg_batch, jpegs: *const *const u8, lengths: *const usize,
                                           count: usize, host_threads: c_int, flags: c_uint) -> c_int;
    pub fn compeg_batch_upload_end(batch: *mut compeg_batch) -> c_int;
    pub fn compeg_batch_decode(batch: *mut compeg_batch, hip_stream: *mut c_void) -> c_int;
    pub fn compeg_batch_set_device_preprocess(batch: *mut compeg_batch, mode: c_int) -> c_int;
    pub fn compeg_batch_host_fallbacks(batch: *const compeg_batch) -> usize;
    pub fn compeg_batch_set_chunk(batch: *mut compeg_batch, images_per_launch: u32) -> c_int;
    pub fn compeg_batch_wait(batch: *mut compeg_batch) -> c_int;
    pub fn compeg_batch_count(batch: *const compeg_batch) -> usize;
    pub fn compeg_batch_output(batch: *const compeg_batch, index: usize, device_ptr: *mut *mut c_void,
                               width: *mut u32, height: *mut u32, pitch_bytes: *mut usize) -> c_int;
    pub fn compeg_batch_read_output(batch: *mut compeg_batch, index: usize, host_rgba: *mut u8) -> c_int;
    pub fn compeg_batch_algorithmic_bytes(batch: *const compeg_batch) -> u64;
    pub fn compeg_batch_pixels(batch: *const compeg_batch) -> u64;
    pub fn compeg_batch_set_timing(batch: *mut compeg_batch, on: c_int) -> c_int;
    pub fn compeg_batch_timing(batch: *mut compeg_batch, reset: c_int, decodes: *mut u32, total_ms: *mut c_double,
                               stage_ms: *mut c_double) -> c_int;
    pub fn compeg_batch_last_kernel(batch: *const compeg_batch) -> c_int;
}
