#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s decoded on 4K 4:2:2 restart-interval JPEGs (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  For N > 1 the driver launches this file through
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment);
torch.distributed (RCCL) is used for the barrier and the max-over-ranks reduction only: the
decode path has no exchange step -- images are independent and are sharded across the ranks
(SURVEY.md 8e), so there is no data-path collective.

A "step" is one pass of the whole device-side hot path (huffman decode -> IDCT -> 4:2:2
upsample + YCbCr->RGBA) over this rank's batch of synthetic JPEGs.  The preprocessed scans and
tables are resident in HBM before the timed region starts and the RGBA output stays in HBM,
exactly like the reference leaves it in a texture.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The HIP runtime maps a process's streams onto four hardware queues unless told otherwise; the host-fed pipeline of
# `end_to_end` keeps two decode streams and the library's transfer streams busy at once, and streams that share a queue
# wait for each other (measured: 177 against 271 Gpixel/s on the road with uploads queued ahead).  Set before HIP wakes up.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PRIME_SECONDS = 0.08           # the headline step: untimed decodes in front of the --warmup steps (see main)
SUBRECORD_WARM_SECONDS = 0.08  # other_configs: decodes before the timed ones, at least (see bench_config)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--prime-seconds", type=float, default=PRIME_SECONDS,
                   help="untimed decodes for this long in front of the warmup steps (a card out of idle runs its launches "
                        "5-20 %% longer for 30-40 ms); 0 = none")
    p.add_argument("--batch", type=int, default=256,
                   help="images per GPU per step (256 = one GPU's share of BASELINE config 4: "
                        "2048 4K frames over 8 GPUs)")
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--ri", type=int, default=4, help="MCUs per restart interval (DRI)")
    p.add_argument("--quality", type=int, default=85)
    p.add_argument("--kind", type=int, default=0, help="0 natural-like, 1 random RGB, 2 sparse")
    p.add_argument("--sampling", default="2x1",
                   help="luma sampling HxV: 2x1 = 4:2:2 (the reference's only layout, default); 1x1, 1x2, 2x2 "
                        "= 4:4:4, 4:4:0, 4:2:0 through the extension pipeline (not the headline metric)")
    p.add_argument("--distinct", type=int, default=0,
                   help="number of distinct synthetic images (0 = one per batch slot)")
    p.add_argument("--chunk", type=int, default=0, help="images per kernel-launch pair (0 = all)")
    p.add_argument("--preprocess", choices=["host", "device", "device-per-step"], default="host",
                   help="where scans are preprocessed: host at upload (the reference's data flow, default), "
                        "scan kernels once at upload, or scan kernels inside every timed step")
    p.add_argument("--cpu-seconds", type=float, default=12.0,
                   help="budget for the CPU-oracle baseline sample (0 disables it)")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--no-sweep", action="store_true", help="leave other_configs.launch_size_sweep out (profiling runs: its launches share grids with the headline's)")
    p.add_argument("--sweep-only", action="store_true", help="print the launch-size sweep (other_configs.launch_size_sweep) and leave")
    p.add_argument("--host-feed-only", action="store_true", help="print end_to_end.host_feed_scaling (no device touched) and leave")
    p.add_argument("--e2e-reps", type=int, default=5,
                   help="batches per arm (x2) of the host-fed pipeline measurement `end_to_end` (0 disables it)")
    p.add_argument("--no-extra-configs", action="store_true",
                   help="skip the sub-records for BASELINE configs[0], [2] and [4] (they run on rank 0 at N = 1 only)")
    p.add_argument("--host-feed-ranks", default="1,2,4,8",
                   help="rank counts of the `host_feed_scaling` record: that many processes, bound like the ranks of a "
                        "multi-GPU run, do the host's share of feeding their GPU at the same time, no device involved "
                        "(empty: skip)")
    p.add_argument("--host-feed-child", default="", help=argparse.SUPPRESS)
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="development only: run the multi-rank code path with every rank on cuda:0 and gloo "
                        "for the barrier / max (RCCL needs one device per rank); the number is meaningless")
    return p.parse_args()


def make_inputs(args, rank, world, threads):
    from compeg_amd.sharding import shard_bounds
    from tools import synth

    distinct = args.distinct if args.distinct > 0 else args.batch
    distinct = min(distinct, args.batch)
    # the job is one batch of args.batch * world frames (config 4: 2048 = 256 x 8), cut into
    # contiguous per-rank blocks; frame g is synthesised from seed 0xC0FFEE + g
    lo, hi = shard_bounds(args.batch * world, rank, world)
    assert hi - lo == args.batch

    def one(i):
        return synth.make_jpeg(args.width, args.height, seed=0xC0FFEE + lo + i,
                               kind=args.kind, quality=args.quality, ri=args.ri, sampling=args.sampling_hv)

    with ThreadPoolExecutor(threads) as ex:
        jpegs = list(ex.map(one, range(distinct)))
    return [jpegs[i % distinct] for i in range(args.batch)], distinct


def cpu_baseline(jpegs, budget_s, pixels_per_image):
    """The CPU oracle (a line-faithful port of the reference path: scan preprocess + huffman +
    IDCT + composite, single-threaded like the reference's CPU side) on a bounded sample."""
    from oracle import oracle as orc

    done, t0 = 0, time.perf_counter()
    while True:
        img = orc.ImageData(jpegs[done % len(jpegs)], allow_sampling=True)
        img.decode()
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 64:
            break
    return {"value": round(done * pixels_per_image / el / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
            "kind": "port", "ms_per_frame": round(el / done * 1e3, 2),
            "sample": f"{done} frames of the same workload through oracle/libcompeg_oracle.so "
                      f"(parse + scan preprocess + huffman + IDCT + composite), 1 thread, {el:.1f} s"}


def cpu_baseline_all_cores(jpegs, budget_s, pixels_per_image):
    """The same oracle on every host core at once (one frame per call, the ctypes call releases the GIL):
    SURVEY.md 8(d) baseline (ii).  The reference itself is single-threaded; this is what its algorithm does
    with the whole host."""
    from oracle import oracle as orc

    # (every core this process may use -- or as many as the box's cgroup grants CPU time for, where that is fewer: 256
    # threads on a quota of 16 cores spend their time being throttled)
    from compeg_amd.sharding import cpu_quota_cores
    quota = cpu_quota_cores()
    cores = max(1, min(len(os.sched_getaffinity(0)), int(quota) if quota else 1 << 30))
    deadline = time.perf_counter() + budget_s

    def work(t):
        done = 0
        while time.perf_counter() < deadline:
            orc.ImageData(jpegs[(t + done * cores) % len(jpegs)], allow_sampling=True).decode()
            done += 1
        return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores)))
    el = time.perf_counter() - t0
    return {"value": round(done * pixels_per_image / el / 1e6, 3), "unit": "Mpixels/s", "cores": cores,
            "host_cores_visible": os.cpu_count(), "cgroup_cpu_quota_cores": quota,
            "sample": f"{done} frames on {cores} threads, {el:.1f} s"}


def rate_of(fn, nbytes, min_s=0.4):
    """GB/s of input bytes over at least min_s of back-to-back calls (after one untimed call)."""
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el >= min_s:
            return round(n * nbytes / el / 1e9, 3)


def bench_scan_dat(compeg_amd, gpu, frame_jpeg, frame_pixels):
    """BASELINE configs[0], the `cargo bench` analogue (benches/bench.rs:10-21): ScanBuffer::process on
    benches/scan.dat (496 464 bytes, 42 876 restart intervals), reused buffer, GB/s of input -- through the
    oracle's restatement of the reference's byte loop and through the product's three preprocessors.  Plus the
    same on one 4K benchmark frame's entropy-coded segment (A1 alone, what SURVEY.md 8(d) asks beside the
    decode baseline)."""
    from oracle import oracle as orc

    path = os.path.join(ROOT, "tests", "golden", "scan", "scan.dat")
    data = open(path, "rb").read()
    out = {"workload": "ScanBuffer::process(benches/scan.dat, 42876), 496464 B per call", "unit": "GB/s of input"}
    o = orc.ScanBuffer()
    out["oracle_scalar_1_thread"] = rate_of(lambda: o.process(data, 42876), len(data))
    sb = compeg_amd.ScanBuffer()
    out["product_host_1_thread"] = rate_of(lambda: sb.process(data, 42876), len(data))
    sb4 = compeg_amd.ScanBuffer()
    sb4.set_threads(4)      # (the helpers take segments of at least 64 KiB per thread: four fit this file)
    out["product_host_4_threads"] = rate_of(lambda: sb4.process(data, 42876), len(data))
    sbg = compeg_amd.ScanBuffer()
    out["product_gpu_scan_kernels_incl_pcie_both_ways"] = rate_of(lambda: sbg.process_on_gpu(gpu, data, 42876), len(data))
    assert sb.processed_scan_data() == o.processed_scan_data() == sb4.processed_scan_data() == sbg.processed_scan_data()
    assert sb.start_positions() == o.start_positions() == sb4.start_positions() == sbg.start_positions()
    out["outputs_identical"] = True
    # A1 alone on the benchmark frame
    ref = orc.ImageData(frame_jpeg)
    seg = ref.scan_data()
    n_int = ref.parallelism()
    a1 = {"segment_bytes": len(seg), "restart_intervals": n_int}
    a1["oracle_scalar_1_thread_gbs"] = rate_of(lambda: o.process(seg, n_int), len(seg))
    a1["oracle_scalar_1_thread_mpix_s"] = round(a1["oracle_scalar_1_thread_gbs"] * 1e9 / len(seg) * frame_pixels / 1e6, 1)
    a1["product_host_1_thread_gbs"] = rate_of(lambda: sb.process(seg, n_int), len(seg))
    threads = max(1, min(8, (os.cpu_count() or 2) // 2))
    sbn = compeg_amd.ScanBuffer()
    sbn.set_threads(threads)
    a1[f"product_host_{threads}_threads_gbs"] = rate_of(lambda: sbn.process(seg, n_int), len(seg))
    out["a1_on_one_benchmark_frame"] = a1
    return out


def single_frame_kernel_us(compeg_amd, gpu, jpeg, reps=30):
    """Kernel time of one frame alone on the card (one-image batch, HIP events on the decode stream: median), which
    kernel took it, and whether the output equals the oracle's."""
    import statistics

    import numpy as np
    from oracle import oracle as orc

    b = compeg_amd.Batch(gpu)
    b.upload([compeg_amd.ImageData(jpeg, copy=False)], host_threads=1)
    for _ in range(5):
        b.decode()
    b.wait()
    b.timing(reset=True)
    ts = []
    for _ in range(reps):
        b.decode()
        b.wait()
        ts.append(b.timing(reset=True)[1] * 1e3)
    ok = bool(np.array_equal(b.read_output(0), orc.ImageData(jpeg).decode()))
    if not ok:
        raise SystemExit("bench: single frame differs from the oracle")
    alg = b.algorithmic_bytes()
    us = statistics.median(ts)
    return {"kernel": b.last_kernel(), "kernel_us": round(us, 2), "min_us": round(min(ts), 2),
            "roofline": {"bound": "hbm", "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": alg},
            "verified_bit_exact_vs_oracle": ok}


def bench_mjpeg_stream(compeg_amd, gpu, quality, steps, warmup, threads):
    """The reference's real-world input (src/file/test-images/mjpeg.jpg, the webcam MJPEG frames its viewer consumes:
    960x720, DRI = 10, Annex-K tables without DHT) as a bench row: the fixture itself and a synthetic frame of the
    same format as single frames, a DRI = 4 frame of the same size beside them, and a 256-frame batch."""
    from tools import synth

    out = {"workload": "960x720 YUV 4:2:2 baseline JPEG, DRI=10, no DHT (webcam MJPEG stream)"}
    fixture = os.path.join(ROOT, "tests", "golden", "parser", "mjpeg.jpg")
    if os.path.exists(fixture):
        out["reference_fixture_mjpeg_jpg"] = single_frame_kernel_us(compeg_amd, gpu, open(fixture, "rb").read())
    out["single_frame_dri10"] = single_frame_kernel_us(
        compeg_amd, gpu, synth.make_jpeg(960, 720, seed=0xC0FFEE, quality=quality, ri=10, flags=synth.NO_DHT))
    out["single_frame_dri4_same_size"] = single_frame_kernel_us(
        compeg_amd, gpu, synth.make_jpeg(960, 720, seed=0xC0FFEE, quality=quality, ri=4, flags=synth.NO_DHT))
    out["dri10_over_dri4"] = round(out["single_frame_dri10"]["kernel_us"] / out["single_frame_dri4_same_size"]["kernel_us"], 3)
    out["batch_256"] = bench_config(compeg_amd, gpu, 960, 720, 10, quality, 256, steps, warmup, threads, 64,
                                    "256 x 960x720 YUV 4:2:2 baseline JPEG, DRI=10, no DHT, 64 distinct frames",
                                    flags=synth.NO_DHT)
    # the same stream with a restart interval per MCU row (60 MCUs), as many encoders write it: 90 intervals a frame
    # -- 360 waves for the chip's 3072 in a batch of 256 frames, 1440 in one of 1024
    for frames in (256, 1024):
        out[f"batch_{frames}_row_intervals"] = bench_config(
            compeg_amd, gpu, 960, 720, 60, quality, frames, max(3, steps // 2), warmup, threads, 32,
            f"{frames} x 960x720 YUV 4:2:2 baseline JPEG, DRI=60 (one restart interval per MCU row), 32 distinct frames")
    return out


def bench_config(compeg_amd, gpu, width, height, ri, quality, batch, steps, warmup, threads, distinct, label, flags=0,
                 sampling=(2, 1)):
    """One more single-GPU configuration of BASELINE.json, measured like the headline one: resident inputs, `steps`
    timed decodes of the whole batch, kernel time from the batch's HIP events, a spread of slots verified."""
    import numpy as np
    from oracle import oracle as orc
    from tools import synth

    def one(i):
        return synth.make_jpeg(width, height, seed=0xC0FFEE + i, kind=0, quality=quality, ri=ri, flags=flags, sampling=sampling)

    ext = tuple(sampling) != (2, 1)
    with ThreadPoolExecutor(threads) as ex:
        jpegs = list(ex.map(one, range(distinct)))
    images = [compeg_amd.ImageData(j, copy=False, allow_sampling=ext) for j in jpegs]
    b = compeg_amd.Batch(gpu)
    b.upload([images[i % distinct] for i in range(batch)], host_threads=threads)
    # (sub-records only -- the headline step keeps to exactly --warmup decodes: the card comes out of the idle time
    # the host spent making the frames, and its launches get shorter for 30-40 ms (256 x 1080p under rocprofv3:
    # 803 us falling to 669 over 27 launches, profiles/r03/pmc_summary.md); launches of under a millisecond would
    # otherwise be timed on that slope)
    warm_decodes, t_warm = 0, time.perf_counter()
    while warm_decodes < warmup or time.perf_counter() - t_warm < SUBRECORD_WARM_SECONDS:
        b.decode()
        b.wait()
        warm_decodes += 1
    b.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        b.decode()
    b.wait()
    el = time.perf_counter() - t0
    n, ev_ms, _, _ = b.timing(reset=True)
    kernel_ms = ev_ms / max(n, 1)
    alg = b.algorithmic_bytes()
    ok = True
    slots = sorted({0, batch - 1, batch // 2})
    for i in slots:
        ok = ok and bool(np.array_equal(b.read_output(i), orc.ImageData(jpegs[i % distinct], allow_sampling=ext).decode()))
    if not ok:
        raise SystemExit(f"bench: {label}: GPU output differs from the oracle")
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    return {"workload": label, "value": round(b.pixels() * steps / el / 1e6, 1), "unit": "Mpixels/s",
            "ms_per_step": round(el / steps * 1e3, 4), "ms_per_frame": round(el / steps / batch * 1e3, 5),
            "steps": steps, "warmup_decodes": warm_decodes, "bits_per_pixel": round(8 * sum(len(j) for j in jpegs) / distinct / (width * height), 3),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": alg,
                         "kernel_ms": round(kernel_ms, 4)},
            "kernel": b.last_kernel(), "verified_bit_exact_vs_oracle": ok, "verified_slots": slots}


def bench_launch_size_sweep(compeg_amd, gpu, quality, threads):
    """Where the dispatch changes kernels: launches of 1 .. 256 4K frames (DRI = 4) and of 1 .. 1024 frames of the MJPEG
    stream (960x720, DRI = 10) -- us per frame by the batch's HIP events, the kernel chosen, the fraction of the HBM
    roofline -- and the single-frame latency of the other BASELINE sizes (8K DRI = 1: configs[4] as written; 1080p)."""
    import numpy as np
    from tools import synth

    def frames_of(w, h, ri, distinct):
        with ThreadPoolExecutor(threads) as ex:
            jpegs = list(ex.map(lambda i: synth.make_jpeg(w, h, seed=0xBEEF + i, kind=0, quality=quality, ri=ri), range(distinct)))
        return jpegs, [compeg_amd.ImageData(j, copy=False) for j in jpegs]

    def point(images, n, reps=12):
        b = compeg_amd.Batch(gpu)
        b.upload([images[i % len(images)] for i in range(n)], host_threads=threads)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:   # (launches out of idle run long: bench_config)
            b.decode()
            b.wait()
        b.timing(reset=True)
        ts = []
        for _ in range(reps):
            b.decode()
            b.wait()
            _, total, _, _ = b.timing(reset=True)
            ts.append(total * 1e3)
        us = float(np.median(ts))
        return {"frames": n, "kernel": b.last_kernel(), "us_per_launch": round(us, 1), "us_per_frame": round(us / n, 2),
                "frac": round(b.algorithmic_bytes() / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}

    out = {}
    for name, (w, h, ri, sizes) in {"4K DRI=4": (3840, 2160, 4, (1, 2, 3, 4, 6, 8, 12, 16, 32, 64, 128, 256)),
                                   "960x720 DRI=10": (960, 720, 10, (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024))}.items():
        jpegs, images = frames_of(w, h, ri, 8)
        pts = [point(images, n) for n in sizes]
        worst = max(pts[i]["us_per_frame"] / min(pts[i - 1]["us_per_frame"] if i else 1e9, pts[i + 1]["us_per_frame"] if i + 1 < len(pts) else 1e9)
                    for i in range(len(pts)))
        out[name] = {"points": pts, "worst_step_vs_better_neighbour": round(worst, 2)}
    single = {}
    # (4K frames with longer restart intervals: the decoder's dispatch between the cooperative kernel and the walk +
    # lane-per-MCU route -- DRI = 16, and an interval per MCU row of 240 MCUs as many encoders write it)
    for name, (w, h, ri) in {"8K DRI=1 (configs[4])": (7680, 4320, 1), "1080p DRI=4": (1920, 1080, 4),
                             "4K DRI=16": (3840, 2160, 16), "4K DRI=240 (an interval per MCU row)": (3840, 2160, 240)}.items():
        jpegs, images = frames_of(w, h, ri, 1)
        pt = point(images, 1, reps=20)
        dec = compeg_amd.Decoder(gpu)
        for _ in range(3):
            dec.decode_blocking(images[0])
        ts = []
        for _ in range(15):
            t = time.perf_counter()
            dec.decode_blocking(images[0])
            ts.append(time.perf_counter() - t)
        pt["blocking_decode_ms"] = round(sorted(ts)[len(ts) // 2] * 1e3, 3)
        pt["decoder_kernel"] = dec.last_kernel()
        single[name] = pt
    out["single_frame_latency"] = single
    return out


def host_feed_child(spec, args):
    """One rank of `host_feed_scaling` (no device, no torch): rank/world/start-time/seconds in `spec`."""
    rank, world, t_start, seconds, bus_ids = (spec.split(",") + [""])[:5]
    rank, world, t_start, seconds = int(rank), int(world), float(t_start), float(seconds)
    import compeg_amd
    from compeg_amd.sharding import bind_rank_round_robin, bind_rank_to_its_cores
    from tools import synth
    bus_ids = [b for b in bus_ids.split(";") if b]
    if len(bus_ids) >= world:   # the GPUs are there: each rank next to its own
        cores, numa = bind_rank_to_its_cores(rank, world, bus_ids)
        binding = f"rank {rank} on the cores of GPU {bus_ids[rank]}'s NUMA node ({numa}), {len(cores)} cores"
    else:
        cores, numa, binding = bind_rank_round_robin(rank, world)
    # (threads: the rank's cores -- or its share of the CPU time the box's cgroup grants, where that is less: threads
    # beyond the quota only get the whole tree throttled)
    from compeg_amd.sharding import cpu_quota_cores
    quota = cpu_quota_cores()
    threads = max(1, min(64, len(cores), int(quota // world) if quota else 64))
    if world > 1 or quota:
        os.sched_setaffinity(0, cores[:threads])
    # The working set: ~1 GB of JPEG bytes per rank, every slot at an address of its own (copies of 16 distinct frames:
    # what the caches see is distinct bytes) -- far beyond the L3 a rank's cores share, as a stream of frames is.
    distinct = 16
    with ThreadPoolExecutor(threads) as ex:
        jpegs = list(ex.map(lambda i: synth.make_jpeg(args.width, args.height, seed=0xC0FFEE + rank * distinct + i,
                                                      quality=args.quality, ri=args.ri), range(distinct)))
    slots = max(64, int(HOST_FEED_WORKING_SET_BYTES // (sum(len(j) for j in jpegs) / distinct)))
    frames = [bytes(bytearray(jpegs[i % distinct])) for i in range(slots)]
    jl = compeg_amd.JpegList(frames)
    nbytes = sum(len(j) for j in frames)
    out = {"rank": rank, "threads": threads, "numa": numa, "binding": binding, "working_set_mb": round(nbytes / 1e6, 1), "slots": slots}
    roads = ((0, "host_preprocess_road"), (1, "copy_free_road"), (2, "pageable_default_road"))
    for road, name in roads:
        compeg_amd.host_feed_work(jl, threads, road, 1)                       # warm-up (the arena's pages, the thread pool)
        while time.time() < t_start + road * (seconds + 3.0):                  # every rank starts a road at the same time
            time.sleep(0.001)
        reps, el = 0, 0.0
        t0 = time.time()
        while time.time() - t0 < seconds:
            k = 1 if road != 1 else 4
            el += compeg_amd.host_feed_work(jl, threads, road, k)
            reps += k
        # host memory traffic of the road: JPEG bytes read (+ as many written: the preprocessed scan is within 1 % of the
        # segment's size, the staged copy is the file); the copy-free road reads the headers
        traffic = {0: 2.0, 1: 0.0, 2: 2.0}[road] * reps * nbytes / el / 1e9
        out[name] = {"frames_per_s": reps * len(frames) / el, "jpeg_gbs": reps * nbytes / el / 1e9, "host_memory_gbs": traffic}
    print(json.dumps(out))


HOST_FEED_WORKING_SET_BYTES = 1.0e9


def bench_host_feed_scaling(args, rank_counts, link_gbs_measured, seconds=2.5):
    """What one host can prepare for N GPUs at once: N processes, each bound to the cores a rank of an N-GPU run would
    get -- next to its GPU where the GPUs are there, round robin over the NUMA nodes on a one-card box
    (compeg_amd/sharding.py) --, each with ~1 GB of JPEG bytes of its own, do the host's share of feeding a GPU at the
    same time with no device involved: three roads, see compeg_host_feed_work.  The copy-free road leaves the host the
    headers; what it then needs per GPU is DMA reads of the JPEG bytes at the link's rate, which this cannot measure."""
    import subprocess

    pix = args.width * args.height
    from compeg_amd.sharding import cpu_quota_cores
    quota = cpu_quota_cores()
    rec = {"what": "N processes at once, each on its rank's share of the host's cores with ~1 GB of JPEG bytes of its own, each "
                   "doing the host's share of feeding one GPU (no device): frames/s per rank, the whole host's rate and the "
                   "host memory traffic that goes with it",
           "frame": f"{args.width}x{args.height} DRI={args.ri} q{args.quality}",
           "host_cores_usable": len(os.sched_getaffinity(0)), "cgroup_cpu_quota_cores": quota, "ranks": {}}
    bus_ids = ""
    try:
        import torch
        if torch.cuda.device_count() >= max(rank_counts):
            bus_ids = ";".join(torch.cuda.get_device_properties(i).pci_bus_id if hasattr(torch.cuda.get_device_properties(i), "pci_bus_id") else ""
                               for i in range(max(rank_counts)))
    except Exception:
        bus_ids = ""
    roads = ("host_preprocess_road", "copy_free_road", "pageable_default_road")
    for n in rank_counts:
        t_start = time.time() + 14.0 + 2.0 * n   # (imports, synthesis of the ranks' frames, the 1 GB of slots)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--host-feed-child", f"{r},{n},{t_start},{seconds},{bus_ids}",
                                   "--width", str(args.width), "--height", str(args.height), "--ri", str(args.ri),
                                   "--quality", str(args.quality)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for r in range(n)]
        outs = []
        for p_ in procs:
            o, _ = p_.communicate(timeout=400)
            outs.append(json.loads(o.strip().splitlines()[-1]))
        row = {"threads_per_rank": [o["threads"] for o in outs], "working_set_mb_per_rank": outs[0]["working_set_mb"],
               "numa_binding": [o["binding"] for o in outs]}
        for road in roads:
            per = [o[road]["frames_per_s"] for o in outs]
            row[road] = {"gpix_s_per_rank_min": round(min(per) * pix / 1e9, 1), "gpix_s_whole_host": round(sum(per) * pix / 1e9, 1),
                         "jpeg_gbs_whole_host": round(sum(o[road]["jpeg_gbs"] for o in outs), 1),
                         "host_memory_gbs_whole_host": round(sum(o[road]["host_memory_gbs"] for o in outs), 1)}
        rec["ranks"][str(n)] = row
    total_threads = {n: sum(row["threads_per_rank"]) for n, row in rec["ranks"].items()}
    rec["saturating_resource"] = (
        f"CPU time: this box's cgroup grants {quota} cores' worth (cpu.max) of its {len(os.sched_getaffinity(0))} visible ones, and every row runs "
        f"{sorted(set(total_threads.values()))} threads in all -- the whole-host rate is what that many cores preprocess, whatever the number "
        "of ranks (round 3 ran 32-64 threads per rank against the same quota: the tree was throttled, 431 -> 224 Gpixel/s from 1 to 8 ranks)"
        if quota else "none found: no CPU quota on this host")
    # how many GPUs this host keeps at the rate one GPU's link sustains (measured above by end_to_end)
    need = link_gbs_measured
    rec["per_gpu_link_gbs_measured"] = round(need, 1)
    for road in roads:
        ok = [int(n) for n, row in rec["ranks"].items() if row[road]["jpeg_gbs_whole_host"] / int(n) >= need]
        rec[road + "_gpus_kept_at_link_rate"] = max(ok) if ok else 0
    return rec


PCIE_LINK_GBS = 63.0  # MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s (spec)


def bench_end_to_end(compeg_amd, device, jpegs, images, threads, reps, ext):
    """The host-fed path as a pipeline: two batches on two streams, the upload of one (ImageData::new +
    ScanBuffer::process on `threads` host threads, staging in pinned memory, H2D over PCIe) runs under the decode
    of the other.  Steady state, `reps` batches per arm, median batch period.  Two arms: from JPEG bytes (parse
    included) and from parsed images (parse excluded).  Never `value`: that one starts with inputs in HBM."""
    import statistics

    import numpy as np
    from oracle import oracle as orc

    gpus = [compeg_amd.Gpu.open(device) for _ in range(2)]      # a stream each
    batches = [compeg_amd.Batch(g) for g in gpus]
    jpeg_bytes = sum(len(j) for j in jpegs)

    def arm(upload, batches=batches):
        for b in batches:                                        # warm-up: allocations, pinning, first launches
            upload(b)
            b.decode()
        for b in batches:
            b.wait()
        periods, uploads = [], []
        t_prev = time.perf_counter()
        for r in range(2 * reps):
            b = batches[r & 1]
            b.wait()                                             # its previous decode (two batches ago)
            t_u = time.perf_counter()
            upload(b)                                            # the other batch is decoding meanwhile
            uploads.append(time.perf_counter() - t_u)
            b.decode()
            t = time.perf_counter()
            periods.append(t - t_prev)
            t_prev = t
        for b in batches:
            b.wait()
        return statistics.median(periods), statistics.median(uploads), min(periods), max(periods)

    pix = len(jpegs) * 0  # (set below from the batch itself)
    res = {}
    for name, upload in (("from_jpeg_bytes_parse_included", lambda b: b.upload_jpegs(jpegs, host_threads=threads, allow_sampling=ext)),
                         ("from_parsed_images_parse_excluded", lambda b: b.upload(images, host_threads=threads))):
        period, up, lo, hi = arm(upload)
        pix = batches[0].pixels()
        h2d = batches[0].algorithmic_bytes() - 4 * pix           # what crosses PCIe: preprocessed scans, start positions, tables
        res[name] = {"ms_per_batch": round(period * 1e3, 3), "mpix_s": round(pix / period / 1e6, 1),
                     "upload_ms": round(up * 1e3, 3), "min_max_ms": [round(lo * 1e3, 3), round(hi * 1e3, 3)],
                     "pcie_gbs_during_upload": round(h2d / up / 1e9, 2),
                     "pcie_fraction_of_link": round(h2d / up / 1e9 / PCIE_LINK_GBS, 3),
                     "jpeg_gbs_consumed": round(jpeg_bytes / period / 1e9, 2)}
    ok = all(bool(np.array_equal(batches[k].read_output(i), orc.ImageData(jpegs[i], allow_sampling=ext).decode()))
             for k in (0, 1) for i in (0, len(jpegs) - 1))
    if not ext:
        # the copy-free road: JPEG bytes in page-locked memory (a capture ring), headers parsed on the host, segments
        # fetched by the DMA engines where they lie, scan preprocessing by the scan kernels on the card
        pinned = compeg_amd.HostBuffer(sum(len(j) + 64 for j in jpegs))
        views = compeg_amd.JpegList(pinned.place(jpegs))
        dev_batches = [compeg_amd.Batch(g) for g in gpus]
        for b in dev_batches:
            b.set_device_preprocess(1)
        period, up, lo, hi = arm(lambda b: b.upload_jpegs(views, host_threads=threads), dev_batches)
        res["from_pinned_jpeg_bytes_device_scan"] = {
            "ms_per_batch": round(period * 1e3, 3), "mpix_s": round(pix / period / 1e6, 1),
            "upload_ms": round(up * 1e3, 3), "min_max_ms": [round(lo * 1e3, 3), round(hi * 1e3, 3)],
            "pcie_gbs_during_upload": round(jpeg_bytes / up / 1e9, 2),
            "pcie_fraction_of_link": round(jpeg_bytes / up / 1e9 / PCIE_LINK_GBS, 3),
            "jpeg_gbs_consumed": round(jpeg_bytes / period / 1e9, 2),
            "host_fallbacks": dev_batches[0].host_fallbacks(),
            "what": "compeg_batch_upload_jpegs with device preprocessing from compeg_host_alloc'ed bytes: the host reads "
                    "headers only (what crosses PCIe: the raw entropy-coded segments + tables)"}
        # ... and the same with the upload in two steps (compeg_batch_upload_jpegs_begin / _end): the next batch's
        # transfers are queued before this one's results are read and its descriptors made -- the link never idles
        for b in dev_batches:
            b.wait()
        periods = []
        dev_batches[0].upload_jpegs_begin(views, host_threads=threads)
        t_prev = time.perf_counter()
        n_iter = 2 * reps + 8                                    # (the first few periods are the pipeline filling)
        for r in range(1, n_iter):
            b, o = dev_batches[r & 1], dev_batches[(r + 1) & 1]
            b.wait()                                             # its previous decode
            b.upload_jpegs_begin(views, host_threads=threads)    # queued behind the other batch's transfers
            o.upload_end()
            o.decode()
            t = time.perf_counter()
            periods.append(t - t_prev)
            t_prev = t
        last = dev_batches[(n_iter - 1) & 1]
        last.upload_end()
        last.decode()
        for b in dev_batches:
            b.wait()
        periods = periods[4:]
        period = statistics.median(periods)
        res["from_pinned_jpeg_bytes_device_scan_uploads_queued_ahead"] = {
            "ms_per_batch": round(period * 1e3, 3), "mpix_s": round(pix / period / 1e6, 1),
            "mean_ms_per_batch": round(statistics.mean(periods) * 1e3, 3),
            "min_max_ms": [round(min(periods) * 1e3, 3), round(max(periods) * 1e3, 3)],
            "pcie_gbs": round(jpeg_bytes / period / 1e9, 2), "pcie_fraction_of_link": round(jpeg_bytes / period / 1e9 / PCIE_LINK_GBS, 3),
            "batches_timed": len(periods),
            "what": "the copy-free road with compeg_batch_upload_jpegs_begin / _end: one feeder thread, two batches, the "
                    "upload of one queued while the other's arrives, is finished and decoded"}
        ok = ok and all(bool(np.array_equal(dev_batches[k].read_output(i), orc.ImageData(jpegs[i]).decode()))
                        for k in (0, 1) for i in (0, len(jpegs) // 2, len(jpegs) - 1))
        for b in dev_batches:
            b.wait()
        del dev_batches
        pinned.close()
    if not ok:
        raise SystemExit("bench: end_to_end output differs from the oracle")
    res.update({"frames_per_batch": len(jpegs), "batches_timed_per_arm": 2 * reps, "host_threads": threads,
                "verified_bit_exact_vs_oracle": ok, "pcie_link_gbs": PCIE_LINK_GBS,
                "what": "two compeg_batch objects on two streams; upload(k+1) = ImageData::new + ScanBuffer::process on the "
                        "host threads + H2D from pinned staging, under decode(k); median period of a batch in steady state"})
    return res


def main():
    args = parse_args()
    if args.host_feed_child:
        return host_feed_child(args.host_feed_child, args)
    if args.host_feed_only:
        rec = bench_host_feed_scaling(args, [int(v) for v in args.host_feed_ranks.split(",") if v], 54.0)
        print(json.dumps(rec))
        for n, row in rec["ranks"].items():
            print("#", n, "ranks", {k: v for k, v in row.items() if k != "numa_binding"}, file=sys.stderr)
            print("#   ", row["numa_binding"][-1], file=sys.stderr)
        return
    args.sampling_hv = tuple(int(v) for v in args.sampling.lower().split("x"))
    ext = args.sampling_hv != (2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch through torch.distributed.run (one rank per GPU)")

    import torch
    import torch.distributed as dist

    import compeg_amd  # fails loudly when the HIP library is missing

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.sweep_only:
        sweep = bench_launch_size_sweep(compeg_amd, compeg_amd.Gpu.open(local_rank), args.quality, 16)
        print(json.dumps(sweep))
        for name, rec in sweep.items():
            for pt in rec.get("points", rec.values() if name == "single_frame_latency" else []):
                print("#", name, pt, file=sys.stderr)
        return
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()

    # each rank keeps to its share of the cores of its own GPU's NUMA node (compeg_amd/sharding.py)
    from compeg_amd.sharding import bind_rank_to_its_cores
    def bus_id(i):
        pr = torch.cuda.get_device_properties(i)
        b = getattr(pr, "pci_bus_id", None)
        if isinstance(b, str):
            return b
        if b is None:
            return ""
        return "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), b, getattr(pr, "pci_device_id", 0))
    try:
        bus_ids = [bus_id(i) for i in range(torch.cuda.device_count())]
    except Exception:
        bus_ids = []
    cores, numa = bind_rank_to_its_cores(local_rank, world if not args.rehearse_on_one_gpu else 1, bus_ids)
    threads = max(1, min(int(os.environ.get("COMPEG_BENCH_HOST_THREADS", "32")), len(cores)))   # (the knob: thread-count experiments)
    t_gen = time.perf_counter()
    jpegs, distinct = make_inputs(args, rank, world, threads)
    t_gen = time.perf_counter() - t_gen

    gpu = compeg_amd.Gpu.open(local_rank)
    images = [compeg_amd.ImageData(j, copy=False, allow_sampling=ext) for j in jpegs[:distinct]]
    images = [images[i % distinct] for i in range(args.batch)]
    batch = compeg_amd.Batch(gpu)
    batch.set_device_preprocess({"host": 0, "device": 1, "device-per-step": 2}[args.preprocess])
    t_up = time.perf_counter()
    batch.upload(images, host_threads=threads)   # host preprocess + H2D: outside the timed region
    t_up = time.perf_counter() - t_up
    if args.chunk:
        batch.set_chunk(args.chunk)
    pixels = batch.pixels()
    alg_bytes = batch.algorithmic_bytes()

    # The card comes out of the idle time the host spent making the frames, and its launches get shorter for 30-40 ms
    # (under rocprofv3: 3041, 3335, 3032, 2843, 2809, 2797, 2774, 2759, 2738 ... 2720 us, profiles/r03/pmc_summary.md):
    # the same decodes, untimed, for PRIME_SECONDS in front of the --warmup steps, so that the timed steps are those
    # of a card that is being fed -- reported as "clock_prime" in the line.
    prime_decodes, t_prime = 0, time.perf_counter()
    while time.perf_counter() - t_prime < args.prime_seconds:
        batch.decode()
        batch.wait()
        prime_decodes += 1
    for _ in range(args.warmup):
        batch.decode()
    batch.wait()
    batch.timing(reset=True)

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.decode()
    batch.wait()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()

    n_timed, ev_total_ms, ev_huff_ms, ev_idct_ms = batch.timing(reset=True)

    from compeg_amd.sharding import max_over_ranks
    elapsed = max_over_ranks(elapsed, device="cpu" if args.rehearse_on_one_gpu else "cuda")

    # ---- everything below is outside the timed region -----------------------------------
    verified = None
    if not args.no_verify and rank == 0:
        import numpy as np
        from oracle import oracle as orc
        # a spread of output slots (first, last, middle ones) against the oracle output of each slot's source;
        # tests/test_gpu_parity.py::test_full_size_batch_* check every slot of this workload
        slots = sorted({0, args.batch - 1, args.batch // 2, args.batch // 3, (2 * args.batch) // 3})
        wants = {}
        verified = True
        for i in slots:
            srci = i % distinct
            if srci not in wants:
                wants[srci] = orc.ImageData(jpegs[srci], allow_sampling=ext).decode()
            verified = verified and bool(np.array_equal(batch.read_output(i), wants[srci]))
        verified_slots = slots
        if not verified:
            raise SystemExit("bench: GPU output differs from the oracle -- refusing to report a number")

    # the host-fed path (JPEG bytes in host memory -> RGBA in HBM) as a two-deep pipeline, every rank its own --
    # reported beside `value`, never as it
    end_to_end = None
    if args.e2e_reps > 0:
        barrier()
        e2e = bench_end_to_end(compeg_amd, local_rank, jpegs, images, threads, args.e2e_reps, ext)
        barrier()
        if world > 1:
            # whole-job rate: every rank's frames over the slowest rank's period
            for arm_name in ("from_jpeg_bytes_parse_included", "from_parsed_images_parse_excluded"):
                slowest = max_over_ranks(e2e[arm_name]["ms_per_batch"], device="cpu" if args.rehearse_on_one_gpu else "cuda")
                e2e[arm_name]["whole_job_mpix_s"] = round(pixels * world / (slowest * 1e-3) / 1e6, 1)
        if rank == 0:
            end_to_end = e2e
            end_to_end["host_cores_of_this_rank"] = len(cores)
            end_to_end["numa_node_of_this_rank"] = numa

    # single-frame latency (BASELINE config 2: one 4K frame), device-only and end-to-end
    single = None
    if rank == 0:
        one = compeg_amd.Batch(gpu)
        one.upload(images[:1], host_threads=1)
        for _ in range(5):
            one.decode()
        one.wait()
        one.timing(reset=True)
        reps = 50
        t1 = time.perf_counter()
        for _ in range(reps):
            one.decode()
        one.wait()
        dev_wall = (time.perf_counter() - t1) / reps * 1e3
        n1, tot1, h1, i1 = one.timing(reset=True)
        # (the same without timing events: every event is a packet the card works through between two kernels)
        one.set_timing(False)
        for _ in range(5):
            one.decode()
        one.wait()
        t1 = time.perf_counter()
        for _ in range(reps):
            one.decode()
        one.wait()
        dev_wall_untimed = (time.perf_counter() - t1) / reps * 1e3
        one.set_timing(True)
        def blocking_ms(dec, reps=20):
            for _ in range(3):
                dec.decode_blocking(images[0])
            ts = []
            for _ in range(reps):
                t2 = time.perf_counter()
                dec.decode_blocking(images[0])
                ts.append(time.perf_counter() - t2)
            ts.sort()
            return ts[len(ts) // 2] * 1e3

        dec = compeg_amd.Decoder(gpu)           # default: host scan preprocessor (several threads) + pulls + kernel + wait
        e2e = blocking_ms(dec)
        stages = dec.last_stage_times()          # the reference's three trace timers, through the C ABI
        dec.set_scan_threads(1)                  # the reference's data flow to the letter: one host thread
        e2e_1t = blocking_ms(dec)
        stages_1t = dec.last_stage_times()
        dec.set_device_preprocess(True)          # raw segment pulled to HBM + scan kernels + decode kernel + wait
        e2e_dev = blocking_ms(dec)
        stages_dev = dec.last_stage_times()
        t_parse = time.perf_counter()
        for _ in range(20):
            compeg_amd.ImageData(jpegs[0], copy=False, allow_sampling=ext)
        t_parse = (time.perf_counter() - t_parse) / 20 * 1e3
        single = {"frames": 1, "kernel": one.last_kernel(), "device_ms_per_frame": round(dev_wall_untimed, 4),
                  "device_ms_per_frame_with_timing_events": round(dev_wall, 4),
                  "kernel_ms": round(tot1 / n1, 4),
                  "device_mpix_s": round(one.pixels() / dev_wall_untimed / 1e3, 1),
                  "host_end_to_end_ms": round(e2e, 3),
                  "host_end_to_end_mpix_s": round(one.pixels() / e2e / 1e3, 1),
                  "host_end_to_end_one_thread_ms": round(e2e_1t, 3),
                  "end_to_end_device_scan_ms": round(e2e_dev, 3),
                  "end_to_end_device_scan_mpix_s": round(one.pixels() / e2e_dev / 1e3, 1),
                  "image_parse_ms": round(t_parse, 4),
                  "stage_times_us": {"what": "compeg_decoder_last_stage_times of the last blocking decode (t_preprocess, "
                                             "t_enqueue_writes, t_poll of src/lib.rs:391-396,452-475,516-522)",
                                     "default": {k: round(v, 1) for k, v in stages.items()},
                                     "one_host_thread": {k: round(v, 1) for k, v in stages_1t.items()},
                                     "device_scan": {k: round(v, 1) for k, v in stages_dev.items()}}}

    base = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        base = cpu_baseline(jpegs, args.cpu_seconds, args.width * args.height)
        base["host_nproc"] = os.cpu_count()
        base["all_cores"] = cpu_baseline_all_cores(jpegs, min(8.0, args.cpu_seconds), args.width * args.height)

    extra = None
    if rank == 0 and world == 1 and not args.no_extra_configs and not ext:
        extra = {"configs[0] scan.dat": bench_scan_dat(compeg_amd, gpu, jpegs[0], args.width * args.height)}
        if base is not None:
            base["a1_scan_preprocess_only"] = extra["configs[0] scan.dat"]["a1_on_one_benchmark_frame"]
        extra["configs[2] 256 x 1080p"] = bench_config(
            compeg_amd, gpu, 1920, 1080, 4, args.quality, 256, args.steps, args.warmup, threads, 64,
            "256 x 1920x1080 YUV 4:2:2 baseline JPEG, DRI=4 (BASELINE configs[2]), 64 distinct frames")
        extra["configs[4] 8K DRI=1"] = bench_config(
            compeg_amd, gpu, 7680, 4320, 1, args.quality, 8, args.steps, args.warmup, threads, 4,
            "8 x 7680x4320 YUV 4:2:2 baseline JPEG, DRI=1 per step (BASELINE configs[4] frame), 4 distinct frames")
        extra["mjpeg stream 960x720 DRI=10"] = bench_mjpeg_stream(compeg_amd, gpu, args.quality, args.steps, args.warmup, threads)
        if not args.no_sweep:
            extra["launch_size_sweep"] = bench_launch_size_sweep(compeg_amd, gpu, args.quality, threads)
        # the extension layouts (opt-in, not what the reference accepts), fused kernels
        for name, smp in (("4:4:4", (1, 1)), ("4:2:0", (2, 2))):
            extra[f"extension {name}, 64 x 4K"] = bench_config(
                compeg_amd, gpu, 3840, 2160, 4, args.quality, 64, args.steps, args.warmup, threads, 16,
                f"64 x 3840x2160 YUV {name} baseline JPEG, DRI=4, 16 distinct frames (extension layout)", sampling=smp)
        # (odd restart intervals: an interval's 8-pixel MCUs in pairs, its last one alone; every second interval's pairs across two
        # 64-byte segments -- decode_fused_444_kernel's cut branch and ordinary stores)
        extra["extension 4:4:4 odd DRI, 64 x 4K"] = bench_config(
            compeg_amd, gpu, 3840, 2160, 3, args.quality, 64, max(3, args.steps // 2), args.warmup, threads, 16,
            "64 x 3840x2160 YUV 4:4:4 baseline JPEG, DRI=3, 16 distinct frames (extension layout, MCUs in pairs, an interval's last one alone)", sampling=(1, 1))

    feed_scaling = None
    if rank == 0 and world == 1 and args.host_feed_ranks and end_to_end is not None and not ext:
        best = max(v.get("jpeg_gbs_consumed", v.get("pcie_gbs", 0.0)) for v in end_to_end.values() if isinstance(v, dict))
        feed_scaling = bench_host_feed_scaling(args, [int(v) for v in args.host_feed_ranks.split(",") if v], best)
        end_to_end["host_feed_scaling"] = feed_scaling

    if rank == 0:
        total_pixels = pixels * world * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        huff_ms, idct_ms = ev_huff_ms / max(n_timed, 1), ev_idct_ms / max(n_timed, 1)
        which = batch.last_kernel()   # (compeg_batch_last_kernel: where the dispatch sent the timed launches)
        fused = which not in ("split", "generic")
        if fused:
            # one kernel does the whole path; the event pair brackets exactly its launches
            name = {"fused": "decode_fused_422_mcu_kernel" if args.ri == 1 else "decode_fused_422_kernel",
                    "fused_stream": {(1, 1): "decode_fused_444_stream_kernel", (1, 2): "decode_fused_440_stream_kernel",
                                     (2, 2): "decode_fused_420_stream_kernel"}.get(args.sampling_hv, "decode_fused_422_stream_kernel"),
                    "pair": "decode_pair_422_kernel", "coop_team": "decode_coop_team_422_kernel",
                    "walk_mcu": "walk_mcus_422_kernel + decode_fused_422_mcu_rec_kernel",
                    "fused_layout": {(1, 1): "decode_fused_444_kernel", (1, 2): "decode_fused_440_kernel",
                                     (2, 2): "decode_fused_420_kernel"}.get(args.sampling_hv, "decode_fused_layout_kernel")}[which]
            kernels_ms = {name: round(ev_total_ms / max(n_timed, 1), 4)}
            dominant = (name, ev_total_ms / max(n_timed, 1))
        else:
            first, second = ("entropy_kernel", "idct_composite_kernel") if which == "split" else ("entropy_samples_kernel", "composite_generic_kernel")
            kernels_ms = {first: round(huff_ms, 4), second: round(idct_ms, 4)}
            dominant = (first, huff_ms) if huff_ms >= idct_ms else (second, idct_ms)
        achieved = alg_bytes / (dominant[1] * 1e-3) / 1e9 if dominant[1] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.width}x{args.height}_ri{args.ri}_q{args.quality}_k{args.kind}"
                if key in tj and dominant[0] in tj[key]:
                    traffic = tj[key][dominant[0]]["hbm_bytes_per_image"] * args.batch
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/s decoded (4K 4:2:2 restart-interval baseline JPEG -> RGBA8 in HBM)" if not ext else
                      f"Mpixels/s decoded (extension layout, luma {args.sampling}; not the headline metric)",
            "value": round(total_pixels / elapsed / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_prime": {"seconds": args.prime_seconds, "decodes": prime_decodes,
                            "what": "untimed decodes in front of the warmup steps: the card's launches get shorter for 30-40 ms after idle"},
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_frame": round(ms_per_step / args.batch, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-reader / i16 levels / f32 IDCT / u8 RGBA",
            "data": "synthetic",
            "config": {
                "workload": f"{args.batch} x {args.width}x{args.height} YUV {'4:2:2' if not ext else 'luma ' + args.sampling} baseline JPEG, DRI={args.ri}, "
                            f"q{args.quality} Annex-K tables, per GPU per step (BASELINE configs[1] frame; "
                            f"256/GPU = configs[3]'s 2048-frame batch over 8 GPUs)",
                "images_per_gpu": args.batch, "distinct_images": distinct,
                "width": args.width, "height": args.height, "restart_interval": args.ri,
                "bits_per_pixel": round(8 * sum(len(j) for j in jpegs[:distinct]) / distinct / (args.width * args.height), 3),
                "parallelism": f"image-sharded replicas x{world}, no collective",
                "inputs": "preprocessed scans + LUTs resident in HBM before the timed region",
                "chunk": args.chunk,
                "scan_preprocess": args.preprocess,
            },
            "roofline": {
                "bound": "hbm", "kernel": dominant[0],
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": ("profiles/traffic.json: FETCH_SIZE (doubled, MI355X_MICROARCH.md) + WRITE_SIZE of this kernel on this "
                                   "workload from committed rocprofv3 --pmc passes, per image x the batch -- a look-up, not a "
                                   "measurement of this run") if traffic is not None else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": round(dominant[1], 4),
                "kernels_ms": kernels_ms,
                "timing": "HIP events on the decode stream, averaged over the timed steps",
            },
            "cpu_baseline": base,
            "other_configs": extra,
            "single_frame": single,
            "end_to_end": end_to_end,
            "verified_bit_exact_vs_oracle": verified,
            "verified_slots": None if verified is None else verified_slots,
            "setup_s": {"synthesize": round(t_gen, 2), "host_preprocess_and_upload": round(t_up, 2)},
            "device": gpu.name(),
        }
        print(json.dumps(out))

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
