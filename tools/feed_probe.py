"""The copy-free feed road alone: compeg_batch_upload_jpegs with device preprocessing from page-locked bytes, back to
back, with and without a decode of the other batch running underneath.  What the PCIe link and the DMA engines give
this transfer pattern (256 segments of 1.6 MB), and what the decode kernel takes from it.
    COMPEG_TRACE_BATCH=1 python tools/feed_probe.py [frames] [reps]"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import compeg_amd as ca
from tools import synth
from concurrent.futures import ThreadPoolExecutor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
with ThreadPoolExecutor(16) as ex:
    jpegs = list(ex.map(lambda i: synth.make_jpeg(3840, 2160, seed=0xC0FFEE + i, quality=85, ri=4), range(64)))
jpegs = [jpegs[i % 64] for i in range(n)]
nbytes = sum(len(j) for j in jpegs)
pinned = ca.HostBuffer(sum(len(j) + 64 for j in jpegs))
views = ca.JpegList(pinned.place(jpegs))
gpus = [ca.Gpu.open(0) for _ in range(2)]
bs = [ca.Batch(g) for g in gpus]
for b in bs:
    b.set_device_preprocess(1)
    b.upload_jpegs(views, host_threads=8)
    b.decode(); b.wait()
for label, with_decode in (("uploads alone", False), ("uploads under the other batch's decode", True)):
    ts = []
    for r in range(2 * reps):
        b = bs[r & 1]
        b.wait()
        t0 = time.perf_counter()
        b.upload_jpegs(views, host_threads=8)
        ts.append(time.perf_counter() - t0)
        if with_decode:
            b.decode()
    for b in bs:
        b.wait()
    ts = np.array(ts[2:]) * 1e3
    print(f"{label}: upload median {np.median(ts):.2f} ms (min {ts.min():.2f}, max {ts.max():.2f}) = {nbytes / np.median(ts) / 1e6:.1f} GB/s over PCIe, "
          f"{n * 3840 * 2160 / np.median(ts) / 1e6:.0f} Gpixel/s", flush=True)

# uploads in two steps, the next one queued ahead: where the feeder's time goes
for b in bs:
    b.wait()
bs[0].upload_jpegs_begin(views, host_threads=8)
rows = []
t_prev = time.perf_counter()
for r in range(1, 14):
    b, o = bs[r & 1], bs[(r + 1) & 1]
    t0 = time.perf_counter(); b.wait()
    t1 = time.perf_counter(); b.upload_jpegs_begin(views, host_threads=8)
    t2 = time.perf_counter(); o.upload_end()
    t3 = time.perf_counter(); o.decode()
    t4 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t_prev))
    t_prev = t4
last = bs[14 & 1]
last.upload_end(); last.decode()
for b in bs:
    b.wait()
print("queued ahead, per iteration [ms]: wait(decode)  begin  end  decode()  period")
for w in rows:
    print("   " + "  ".join(f"{v * 1e3:7.2f}" for v in w))
