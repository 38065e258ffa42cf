"""Where does the time of one blocking 4K decode go?  Run with COMPEG_TRACE=1 to see the host-side
steps of Decoder::enqueue; this script adds wall-clock figures for the whole call and for wait()."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compeg_amd
from tools import synth


def main():
    w, h, ri = 3840, 2160, 4
    jpeg = synth.make_jpeg(w, h, seed=0xC0FFEE, quality=85, ri=ri)
    img = compeg_amd.ImageData(jpeg)
    gpu = compeg_amd.Gpu.open()
    modes = {"host": (False,), "device": (True,)}.get(sys.argv[1] if len(sys.argv) > 1 else "", (False, True))
    for mode in modes:   # (the profile timeline reads the tail of the last pass)
        dec = compeg_amd.Decoder(gpu)
        dec.set_device_preprocess(mode)
        for _ in range(3):
            dec.decode_blocking(img)
        t_start, t_wait = [], []
        for _ in range(20):
            t0 = time.perf_counter()
            op = dec.start_decode(img)
            t1 = time.perf_counter()
            op.wait()
            t2 = time.perf_counter()
            t_start.append(t1 - t0)
            t_wait.append(t2 - t1)
        t_start.sort()
        t_wait.sort()
        print(f"device_preprocess={mode}: start_decode median {t_start[10]*1e6:.0f} us, wait median {t_wait[10]*1e6:.0f} us, "
              f"total {1e6*(t_start[10]+t_wait[10]):.0f} us; scan bytes {img.scan_range()[1]}", flush=True)
        sb = compeg_amd.ScanBuffer()
        o, n = img.scan_range()
        import numpy as np
        scan = np.frombuffer(jpeg, dtype=np.uint8)[o:o + n]
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            sb.process(scan, img.parallelism())
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"  ScanBuffer.process (host) median {ts[5]*1e6:.0f} us", flush=True)

        tb = []
        for _ in range(20):
            t0 = time.perf_counter()
            dec.decode_blocking(img)
            tb.append(time.perf_counter() - t0)
        tb.sort()
        print(f"  decode_blocking median {tb[10]*1e6:.0f} us, best {tb[0]*1e6:.0f} us", flush=True)

if __name__ == "__main__":
    main()
