"""Development aid: the corrupt-stream / hostile-table inputs of the GPU test, one by one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import compeg_amd as ca
import oracle.oracle as orc
import test_gpu_parity as T
gpu = ca.Gpu.open()
for idx, j in enumerate(T._corrupt_variants(12) + T._hostile_table_variants(10)):
    try:
        want = orc.ImageData(j).decode()
    except orc.OracleError:
        continue
    dec = ca.Decoder(gpu)
    data = ca.ImageData(j)
    dec.decode_blocking(data)
    got = dec.read_texture(data.width(), data.height())
    diff = (got != want).any(axis=2)
    ys, xs = np.nonzero(diff)
    print(idx, "ri-intervals", data.parallelism(), "bad", int(diff.sum()), (int(xs[0]), int(ys[0])) if len(xs) else "")
