//! `cargo run --example decode -- frame.jpg`: the call sequence of the reference's
//! tests (`Gpu::open` -> `ImageData::new` -> `Decoder::decode_blocking` -> read the texture).
use std::sync::Arc;

use compeg::{Decoder, Gpu, ImageData};

fn block_on<F: std::future::Future>(f: F) -> F::Output {
    // `Gpu::open` never suspends; a minimal executor keeps this example dependency-free.
    use std::task::{Context, Poll, RawWaker, RawWakerVTable, Waker};
    fn raw() -> RawWaker {
        fn no(_: *const ()) {}
        fn clone(_: *const ()) -> RawWaker {
            raw()
        }
        static VT: RawWakerVTable = RawWakerVTable::new(clone, no, no, no);
        RawWaker::new(std::ptr::null(), &VT)
    }
    let waker = unsafe { Waker::from_raw(raw()) };
    let mut f = std::pin::pin!(f);
    loop {
        if let Poll::Ready(v) = f.as_mut().poll(&mut Context::from_waker(&waker)) {
            return v;
        }
    }
}

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let path = std::env::args().nth(1).expect("usage: decode <file.jpg>");
    let jpeg = std::fs::read(path)?;
    let gpu = Arc::new(block_on(Gpu::open())?);
    let image = ImageData::new(jpeg.as_slice())?;
    let mut decoder = Decoder::new(gpu.clone());
    decoder.set_device_preprocess(true);
    let op = decoder.decode_blocking(&image);
    let tex = op.texture();
    println!("{}: {}x{} -> RGBA8 at {:?} (pitch {}), texture_changed = {}", gpu.name(), image.width(),
             image.height(), tex.device_ptr, tex.pitch_bytes, op.texture_changed());
    drop(op);
    let rgba = decoder.read_output(image.width(), image.height())?;
    println!("first pixel: {:?}", &rgba[..4]);
    Ok(())
}
