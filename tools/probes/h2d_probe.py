"""Probe (development aid): host-to-device rates of pinned memory on this box -- one large copy, many 1.6 MB
copies on one stream, and the same spread over several streams (what a batch upload does)."""
import time
import torch

torch.cuda.set_device(0)
total, piece = 420 << 20, 1640 << 10
host = torch.empty(total, dtype=torch.uint8).pin_memory()
host.random_(0, 255)
dev = torch.empty(total, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def rate(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return total / best / 1e9


print("one copy of 420 MiB: %.1f GB/s" % rate(lambda: dev.copy_(host, non_blocking=True)))
n = total // piece
for streams in (1, 2, 4, 8):
    ss = [torch.cuda.Stream() for _ in range(streams)]

    def many():
        for i in range(n):
            with torch.cuda.stream(ss[i % streams]):
                dev[i * piece:(i + 1) * piece].copy_(host[i * piece:(i + 1) * piece], non_blocking=True)
    print("%d copies of 1.6 MiB on %d stream(s): %.1f GB/s" % (n, streams, rate(many)))
# device-side pull: a kernel reading the pinned host buffer directly (what launch_pull does)
hmap = host  # pinned memory is mapped into the device address space on ROCm
print("note: the kernel-pull path is measured by the library itself (COMPEG_PULL)")
