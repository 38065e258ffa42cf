import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import compeg_amd as ca
from compeg_amd._lib import lib
from tools import synth
lib.compeg_debug_read_dc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
gpu = ca.Gpu.open(0)
for (w,h,ri) in ((1920,1080,96),(1920,1080,100)):
    jpeg = synth.make_jpeg(w, h, seed=4000+ri, ri=ri, quality=85)
    img = ca.ImageData(jpeg)
    dec = ca.Decoder(gpu)
    for _ in range(3):
        dec.decode_blocking(img)
    waves = img.parallelism() * 4
    full = np.zeros((waves, 16), dtype=np.uint64)
    assert lib.compeg_debug_read_dc(dec._h, full.ctypes.data, full.nbytes) == 0
    buf = full[:, :8].astype(np.int64); wall = full[:, 8:10].astype(np.int64)
    ok = wall[:,0] > 0; t0 = wall[ok,0].min(); end = wall[:,1]-t0
    print(w,h,ri,"kernel",dec.last_kernel(),"intervals",img.parallelism(),"ends (10 ns ticks): median",int(np.median(end[ok])),"max",int(end[ok].max()))
    for i in np.argsort(end)[-6:]:
        print("  wave",i,"team",i//4,"member",i%4,"start",int(wall[i,0]-t0),"end",int(end[i]),"phases setup/chase/validate/decode/fixups/dc+idct/composite/wait:",list(buf[i]))
