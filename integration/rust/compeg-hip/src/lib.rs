//! `compeg`-shaped API over `libcompeg_hip.so` (MI355X / gfx950).
//!
//! Same public items and call sequences as SludgePhD/Compeg's `src/lib.rs`
//! (`Gpu`, `Decoder`, `DecodeOp`, `ImageData`, the doc-hidden `ScanBuffer`,
//! `Error`/`Result`), with two substitutions a caller can see:
//!
//! * the output is a [`Texture`] (device pointer to RGBA8 rows + extent +
//!   pitch) instead of a `wgpu::Texture`;
//! * `Decoder::enqueue` records on a HIP stream ([`Stream`]) instead of a
//!   `wgpu::CommandEncoder`.
//!
//! [`Batch`] (many images per launch) is an extension the reference does not
//! have.  This crate has not been compiled in the repository's build image
//! (no Rust toolchain there); the C ABI underneath is what the test suite
//! exercises, through the same entry points in the same order.

mod ffi;

use std::borrow::Cow;
use std::ffi::CStr;
use std::fmt;
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};
use std::ptr::{self, NonNull};
use std::sync::Arc;

/// String-only error, `Display` = `Debug` = the message (like `compeg::Error`).
pub struct Error {
    message: String,
    code: i32,
}

impl Error {
    /// The C status behind the message (`COMPEG_E_*`); the reference has no equivalent.
    pub fn code(&self) -> i32 {
        self.code
    }

    fn last(code: c_int) -> Self {
        // thread-local in the library: valid until this thread's next failing call
        let message = unsafe { CStr::from_ptr(ffi::compeg_last_error()) }.to_string_lossy().into_owned();
        Error { message, code }
    }
}

impl fmt::Display for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.write_str(&self.message)
    }
}

impl fmt::Debug for Error {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        f.write_str(&self.message)
    }
}

impl std::error::Error for Error {}

pub type Result<T, E = Error> = std::result::Result<T, E>;

fn check(code: c_int) -> Result<()> {
    if code == ffi::COMPEG_OK {
        Ok(())
    } else {
        Err(Error::last(code))
    }
}

/// A HIP stream owned by the caller (`hipStream_t`); null = the `Gpu`'s own stream.
#[derive(Clone, Copy)]
pub struct Stream(pub *mut c_void);

impl Stream {
    pub const DEFAULT: Stream = Stream(ptr::null_mut());
}

/// Device + stream + loaded gfx950 code object.  Immutable after creation and
/// reference-counted inside the library; share it as `Arc<Gpu>` like the reference.
pub struct Gpu {
    raw: NonNull<ffi::compeg_gpu>,
}

unsafe impl Send for Gpu {}
unsafe impl Sync for Gpu {}

impl Gpu {
    /// Opens the default device.  `async` only to keep the reference's signature.
    pub async fn open() -> Result<Self> {
        Self::open_device(-1)
    }

    /// Opens HIP device `index` (-1 = current default).
    pub fn open_device(index: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::compeg_gpu_open(index, &mut raw) })?;
        Ok(Gpu { raw: NonNull::new(raw).expect("compeg_gpu_open returned null") })
    }

    /// The `from_wgpu` analogue: work is recorded on a stream the caller owns.
    ///
    /// # Safety
    /// `stream` must be a valid `hipStream_t` of device `device` and outlive the `Gpu`.
    pub unsafe fn from_stream(device: i32, stream: Stream) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(ffi::compeg_gpu_from_stream(device, stream.0, &mut raw))?;
        Ok(Gpu { raw: NonNull::new(raw).expect("compeg_gpu_from_stream returned null") })
    }

    pub fn device(&self) -> i32 {
        unsafe { ffi::compeg_gpu_device(self.raw.as_ptr()) }
    }

    pub fn name(&self) -> String {
        unsafe { CStr::from_ptr(ffi::compeg_gpu_name(self.raw.as_ptr())) }.to_string_lossy().into_owned()
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { ffi::compeg_gpu_release(self.raw.as_ptr()) }
    }
}

/// Extensions of `ImageData::with_flags`; all off = the reference's behaviour.
#[derive(Clone, Copy, Default)]
pub struct ParseFlags {
    /// 4:4:4, 4:4:0 and 4:2:0 are accepted as well as 4:2:2.
    pub any_luma_sampling: bool,
    /// Bit reader topped up in front of DC codes, ZRL = 16 positions (ITU-T T.81).
    pub standard_entropy: bool,
}

/// A parsed and validated JPEG (baseline, 8 bit, YCbCr 4:2:2, restart intervals).
/// Borrows or owns the bytes exactly like the reference's `Cow`.
pub struct ImageData<'a> {
    raw: NonNull<ffi::compeg_image>,
    _jpeg: Cow<'a, [u8]>,
}

unsafe impl Send for ImageData<'_> {}
unsafe impl Sync for ImageData<'_> {}

impl<'a> ImageData<'a> {
    pub fn new(jpeg: impl Into<Cow<'a, [u8]>>) -> Result<Self> {
        let jpeg = jpeg.into();
        let mut raw = ptr::null_mut();
        // copy = 0: `_jpeg` keeps the bytes alive for as long as the handle exists
        check(unsafe { ffi::compeg_image_parse(jpeg.as_ptr(), jpeg.len(), 0, &mut raw) })?;
        Ok(ImageData { raw: NonNull::new(raw).expect("compeg_image_parse returned null"), _jpeg: jpeg })
    }

    /// Extension: like `new`, but 4:4:4, 4:4:0 and 4:2:0 are accepted as well as 4:2:2.
    pub fn new_any_sampling(jpeg: impl Into<Cow<'a, [u8]>>) -> Result<Self> {
        Self::with_flags(jpeg, ParseFlags { any_luma_sampling: true, standard_entropy: false })
    }

    /// Extension: `ParseFlags` widen the accepted subset / switch the entropy decoder to ITU-T T.81
    /// behaviour where the reference deviates (see `include/compeg_hip.h`).
    pub fn with_flags(jpeg: impl Into<Cow<'a, [u8]>>, flags: ParseFlags) -> Result<Self> {
        let jpeg = jpeg.into();
        let mut raw = ptr::null_mut();
        let bits = if flags.any_luma_sampling { ffi::COMPEG_PARSE_ANY_LUMA_SAMPLING } else { 0 }
            | if flags.standard_entropy { ffi::COMPEG_PARSE_STANDARD_ENTROPY } else { 0 };
        check(unsafe { ffi::compeg_image_parse_ext(jpeg.as_ptr(), jpeg.len(), 0, bits, &mut raw) })?;
        Ok(ImageData { raw: NonNull::new(raw).expect("compeg_image_parse_ext returned null"), _jpeg: jpeg })
    }

    pub fn width(&self) -> u32 {
        unsafe { ffi::compeg_image_width(self.raw.as_ptr()) }
    }

    pub fn height(&self) -> u32 {
        unsafe { ffi::compeg_image_height(self.raw.as_ptr()) }
    }

    /// Number of restart intervals = lanes that decode in parallel.
    pub fn parallelism(&self) -> u32 {
        unsafe { ffi::compeg_image_parallelism(self.raw.as_ptr()) }
    }
}

impl Drop for ImageData<'_> {
    fn drop(&mut self) {
        unsafe { ffi::compeg_image_free(self.raw.as_ptr()) }
    }
}

/// The decoder's output: RGBA8 (`R, G, B, 255`), row-major, resident in HBM.
/// Like the reference's texture it never shrinks, so `width`/`height` may
/// exceed the last image; only that image's own corner is defined.
#[derive(Clone, Copy)]
pub struct Texture<'a> {
    pub device_ptr: *mut c_void,
    pub width: u32,
    pub height: u32,
    pub pitch_bytes: usize,
    _owner: PhantomData<&'a ()>,
}

/// `Decoder::into_texture()`: the allocation now belongs to the caller.
pub struct OwnedTexture {
    pub device_ptr: *mut c_void,
    pub width: u32,
    pub height: u32,
    pub pitch_bytes: usize,
}

unsafe impl Send for OwnedTexture {}

impl Drop for OwnedTexture {
    fn drop(&mut self) {
        unsafe { ffi::compeg_device_free(self.device_ptr) }
    }
}

/// Owns all device buffers of one decode pipeline and a reusable scan buffer.
/// Every method takes `&mut self`: one thread at a time per decoder, many decoders per `Gpu`.
pub struct Decoder {
    raw: NonNull<ffi::compeg_decoder>,
    _gpu: Arc<Gpu>,
}

unsafe impl Send for Decoder {}

impl Decoder {
    pub fn new(gpu: Arc<Gpu>) -> Self {
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::compeg_decoder_new(gpu.raw.as_ptr(), &mut raw) }).expect("compeg_decoder_new");
        Decoder { raw: NonNull::new(raw).expect("compeg_decoder_new returned null"), _gpu: gpu }
    }

    /// Extension: run the scan preprocessing on the GPU (the raw entropy-coded
    /// segment is uploaded instead of the preprocessed one).  Same results.
    pub fn set_device_preprocess(&mut self, on: bool) {
        check(unsafe { ffi::compeg_decoder_set_device_preprocess(self.raw.as_ptr(), on as c_int) })
            .expect("compeg_decoder_set_device_preprocess");
    }

    /// Extension: threads of this decoder's host scan preprocessor (1..=16).
    pub fn set_scan_threads(&mut self, threads: u32) {
        check(unsafe { ffi::compeg_decoder_set_scan_threads(self.raw.as_ptr(), threads as std::os::raw::c_uint) })
            .expect("compeg_decoder_set_scan_threads");
    }

    /// Preprocesses, uploads and records the decode on `stream` without
    /// waiting; returns whether the output was reallocated (always on the
    /// first call).  The reference records into a `CommandEncoder` here.
    pub fn enqueue(&mut self, data: &ImageData<'_>, stream: Stream) -> bool {
        let mut changed = 0;
        check(unsafe { ffi::compeg_decoder_enqueue(self.raw.as_ptr(), data.raw.as_ptr(), stream.0, &mut changed) })
            .expect("compeg_decoder_enqueue");
        changed != 0
    }

    /// Enqueues on the `Gpu`'s own stream and submits; returns at once.
    pub fn start_decode(&mut self, data: &ImageData<'_>) -> DecodeOp<'_> {
        let mut op = ptr::null_mut();
        check(unsafe { ffi::compeg_decoder_start_decode(self.raw.as_ptr(), data.raw.as_ptr(), &mut op) })
            .expect("compeg_decoder_start_decode");
        DecodeOp { raw: NonNull::new(op).expect("null op"), decoder: self }
    }

    /// `start_decode` + wait.
    pub fn decode_blocking(&mut self, data: &ImageData<'_>) -> DecodeOp<'_> {
        let mut op = ptr::null_mut();
        check(unsafe { ffi::compeg_decoder_decode_blocking(self.raw.as_ptr(), data.raw.as_ptr(), &mut op) })
            .expect("compeg_decoder_decode_blocking");
        DecodeOp { raw: NonNull::new(op).expect("null op"), decoder: self }
    }

    /// The reference drops a restart-interval count mismatch silently; here it can be read back.
    pub fn last_warning(&self) -> Option<String> {
        let p = unsafe { ffi::compeg_decoder_last_warning(self.raw.as_ptr()) };
        if p.is_null() {
            return None;
        }
        let s = unsafe { CStr::from_ptr(p) }.to_string_lossy();
        if s.is_empty() { None } else { Some(s.into_owned()) }
    }

    /// Host time of the stages of the last decode: what the reference traces as `t_preprocess`,
    /// `t_enqueue_writes` and `t_poll` (src/lib.rs:391-396,452-475,516-522).
    pub fn last_stage_times(&self) -> ffi::compeg_stage_times {
        let mut t = ffi::compeg_stage_times::default();
        check(unsafe { ffi::compeg_decoder_last_stage_times(self.raw.as_ptr(), &mut t) })
            .expect("compeg_decoder_last_stage_times");
        t
    }

    pub fn texture(&self) -> Texture<'_> {
        let (mut p, mut w, mut h, mut pitch) = (ptr::null_mut(), 0, 0, 0);
        check(unsafe { ffi::compeg_decoder_output(self.raw.as_ptr(), &mut p, &mut w, &mut h, &mut pitch) })
            .expect("compeg_decoder_output");
        Texture { device_ptr: p, width: w, height: h, pitch_bytes: pitch, _owner: PhantomData }
    }

    pub fn into_texture(self) -> OwnedTexture {
        let (mut p, mut w, mut h, mut pitch) = (ptr::null_mut(), 0, 0, 0);
        // compeg_decoder_take_output frees the decoder: skip our Drop
        let raw = self.raw;
        let gpu = unsafe { ptr::read(&self._gpu) };
        std::mem::forget(self);
        let rc = unsafe { ffi::compeg_decoder_take_output(raw.as_ptr(), &mut p, &mut w, &mut h, &mut pitch) };
        drop(gpu);
        check(rc).expect("compeg_decoder_take_output");
        OwnedTexture { device_ptr: p, width: w, height: h, pitch_bytes: pitch }
    }

    /// Test helper (the reference's tests copy the texture to a buffer): waits and
    /// returns the image's `width x height` corner, tightly packed.
    pub fn read_output(&mut self, width: u32, height: u32) -> Result<Vec<u8>> {
        let mut v = vec![0u8; width as usize * height as usize * 4];
        check(unsafe { ffi::compeg_decoder_read_output(self.raw.as_ptr(), v.as_mut_ptr(), width, height) })?;
        Ok(v)
    }
}

impl Drop for Decoder {
    fn drop(&mut self) {
        unsafe { ffi::compeg_decoder_free(self.raw.as_ptr()) }
    }
}

/// An in-flight decode; `wait` replaces polling the wgpu submission index.
pub struct DecodeOp<'a> {
    raw: NonNull<ffi::compeg_op>,
    decoder: &'a Decoder,
}

impl DecodeOp<'_> {
    pub fn wait(&self) {
        check(unsafe { ffi::compeg_op_wait(self.raw.as_ptr()) }).expect("compeg_op_wait");
    }

    pub fn texture(&self) -> Texture<'_> {
        self.decoder.texture()
    }

    pub fn texture_changed(&self) -> bool {
        unsafe { ffi::compeg_op_texture_changed(self.raw.as_ptr()) != 0 }
    }
}

impl Drop for DecodeOp<'_> {
    fn drop(&mut self) {
        unsafe { ffi::compeg_op_free(self.raw.as_ptr()) }
    }
}

/// Scan preprocessing on its own (the reference exposes it for its benchmark).
#[doc(hidden)]
pub struct ScanBuffer {
    raw: NonNull<ffi::compeg_scanbuffer>,
}

unsafe impl Send for ScanBuffer {}

impl ScanBuffer {
    pub fn new() -> Self {
        ScanBuffer { raw: NonNull::new(unsafe { ffi::compeg_scanbuffer_new() }).expect("compeg_scanbuffer_new") }
    }

    /// `Err` on a restart-interval count mismatch; the buffers then hold the truncated result.
    pub fn process(&mut self, scan_data: &[u8], expected_restart_intervals: u32) -> Result<()> {
        check(unsafe {
            ffi::compeg_scanbuffer_process(self.raw.as_ptr(), scan_data.as_ptr(), scan_data.len(), expected_restart_intervals)
        })
    }

    /// Extension: `threads` threads (1..=16) share every following `process` call; same bytes.
    pub fn set_threads(&mut self, threads: u32) -> Result<()> {
        check(unsafe { ffi::compeg_scanbuffer_set_threads(self.raw.as_ptr(), threads as std::os::raw::c_uint) })
    }

    /// Extension: the same bytes, computed by the device-side scan kernels.
    pub fn process_on_gpu(&mut self, gpu: &Gpu, scan_data: &[u8], expected_restart_intervals: u32) -> Result<()> {
        check(unsafe {
            ffi::compeg_scanbuffer_process_on_gpu(self.raw.as_ptr(), gpu.raw.as_ptr(), scan_data.as_ptr(), scan_data.len(),
                                                  expected_restart_intervals)
        })
    }

    pub fn processed_scan_data(&self) -> &[u8] {
        let mut n = 0;
        let p = unsafe { ffi::compeg_scanbuffer_data(self.raw.as_ptr(), &mut n) };
        if n == 0 { &[] } else { unsafe { std::slice::from_raw_parts(p, n) } }
    }

    pub fn start_positions(&self) -> &[u8] {
        let mut n = 0;
        let p = unsafe { ffi::compeg_scanbuffer_start_positions(self.raw.as_ptr(), &mut n) };
        if n == 0 { &[] } else { unsafe { std::slice::from_raw_parts(p, n) } }
    }
}

impl Default for ScanBuffer {
    fn default() -> Self {
        Self::new()
    }
}

impl Drop for ScanBuffer {
    fn drop(&mut self) {
        unsafe { ffi::compeg_scanbuffer_free(self.raw.as_ptr()) }
    }
}

/// Where a [`Batch`] preprocesses its scans.
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum Preprocess {
    /// On the host during `upload`, like the reference.
    Host = 0,
    /// Raw segments are uploaded and preprocessed once by the scan kernels.
    DeviceOnce = 1,
    /// Every `decode` re-runs the scan kernels first (raw scan bytes in HBM -> RGBA).
    DeviceEveryDecode = 2,
}

/// Extension: many independent images decoded by one launch sequence.  The
/// images' scans and tables are made resident in HBM once; every `decode` is
/// pure device work.  Outputs are RGBA8 images, rows `pitch_bytes` apart (the width rounded up to 16 pixels).
pub struct Batch {
    raw: NonNull<ffi::compeg_batch>,
    _gpu: Arc<Gpu>,
}

unsafe impl Send for Batch {}

impl Batch {
    pub fn new(gpu: Arc<Gpu>) -> Result<Self> {
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::compeg_batch_new(gpu.raw.as_ptr(), &mut raw) })?;
        Ok(Batch { raw: NonNull::new(raw).expect("compeg_batch_new returned null"), _gpu: gpu })
    }

    /// Set before `upload`.
    pub fn set_preprocess(&mut self, mode: Preprocess) -> Result<()> {
        check(unsafe { ffi::compeg_batch_set_device_preprocess(self.raw.as_ptr(), mode as c_int) })
    }

    /// Host front-end for all images (`host_threads` = 0: one per core) + upload; replaces previous content.
    pub fn upload(&mut self, images: &[&ImageData<'_>], host_threads: usize) -> Result<()> {
        let raws: Vec<*const ffi::compeg_image> = images.iter().map(|i| i.raw.as_ptr() as *const _).collect();
        check(unsafe { ffi::compeg_batch_upload(self.raw.as_ptr(), raws.as_ptr(), raws.len(), host_threads as c_int) })
    }

    /// Images of the last upload that the scan kernels handed back to the host.
    pub fn host_fallbacks(&self) -> usize {
        unsafe { ffi::compeg_batch_host_fallbacks(self.raw.as_ptr()) }
    }

    /// Records the decode of every uploaded image on `stream` and returns without waiting.
    pub fn decode(&mut self, stream: Stream) -> Result<()> {
        check(unsafe { ffi::compeg_batch_decode(self.raw.as_ptr(), stream.0) })
    }

    pub fn wait(&mut self) -> Result<()> {
        check(unsafe { ffi::compeg_batch_wait(self.raw.as_ptr()) })
    }

    pub fn len(&self) -> usize {
        unsafe { ffi::compeg_batch_count(self.raw.as_ptr()) }
    }

    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    pub fn texture(&self, index: usize) -> Result<Texture<'_>> {
        let (mut p, mut w, mut h, mut pitch) = (ptr::null_mut(), 0, 0, 0);
        check(unsafe { ffi::compeg_batch_output(self.raw.as_ptr(), index, &mut p, &mut w, &mut h, &mut pitch) })?;
        Ok(Texture { device_ptr: p, width: w, height: h, pitch_bytes: pitch, _owner: PhantomData })
    }

    pub fn pixels(&self) -> u64 {
        unsafe { ffi::compeg_batch_pixels(self.raw.as_ptr()) }
    }
}

impl Drop for Batch {
    fn drop(&mut self) {
        unsafe { ffi::compeg_batch_free(self.raw.as_ptr()) }
    }
}
