"""One-off randomized check (not part of the test suite): random geometry, quality, restart interval, bit flips in
the scan, both entropy modes, cut-short windows -- the emulated fused and two-kernel pipelines and the cooperative
kernel (both forms: one-wave with speculative walks; team geometry with walk tables) against the oracle.
    python tools/fuzz_emulation.py [seed] [iterations]   (tests/emul/emul_runner must be built)"""
import os, subprocess, sys, numpy as np, tempfile
sys.path.insert(0,'/root/repo')
from tools import synth
import oracle.oracle as orc
RUN='/root/repo/tests/emul/emul_runner'
tmp=tempfile.mkdtemp()
def run(jpeg, fused, window=2048, standard=False, coop_passes=1, chunk=6):
    p=os.path.join(tmp,'in.jpg'); open(p,'wb').write(jpeg)
    env=dict(os.environ); env['EMUL_FUSED']=str(fused); env['EMUL_COOP_PASSES']=str(coop_passes); env['EMUL_WALK_CHUNK']=str(chunk)
    if standard: env['EMUL_STANDARD']='1'
    r=subprocess.run([RUN,p,tmp+'/rgba',tmp+'/ac',tmp+'/dc','1',str(window),'12288'],capture_output=True,text=True,env=env,timeout=600)
    if fused in (5,8) and 'does not qualify' in r.stdout: return 'skip', ''
    if r.returncode!=0: return None, r.stdout+r.stderr[-500:]
    _,w,h,_=r.stdout.split()
    return np.fromfile(tmp+'/rgba',dtype=np.uint8).reshape(int(h),int(w),4), ''
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1234)
bad=0; n=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 150):
    w=int(rng.integers(16,300)); h=int(rng.integers(8,120)); kind=int(rng.integers(0,3)); q=int(rng.choice([30,60,85,95,100])); ri=int(rng.integers(0,9))
    if it%4==3:   # long restart intervals that seldom divide the image: the cooperative kernel's speculative walks, the last interval's end
        w=int(rng.integers(200,700)); h=int(rng.integers(64,260)); ri=int(rng.choice([10,17,30,45,60,100,130]))
    if os.environ.get('FUZZ_NARROW'):   # (one to three MCUs across)
        w=int(rng.integers(8,50))
    base=synth.make_jpeg(w,h,seed=int(rng.integers(1,1<<30)),kind=kind,quality=q,ri=ri)
    j=bytearray(base)
    if it%3!=0:
        scan_at=j.find(b"\xff\xda")+14
        for _ in range(int(rng.integers(1,40))):
            pos=int(rng.integers(scan_at,len(j)-2))
            if j[pos]!=0xFF and j[pos-1]!=0xFF:
                j[pos]^=1<<int(rng.integers(0,8))
                if j[pos]==0xFF: j[pos]=0xFE
    if os.environ.get('FUZZ_ONES') and it%3!=0:
        # runs of one bits in the scan (0xFF 0x00 is eight of them, 0xFE seven): bits that are no Huffman code at all,
        # in front of DC codes and AC codes alike -- what a reader with a few bits left makes of them is its own affair
        scan_at=j.find(b"\xff\xda")+14
        for _ in range(int(rng.integers(1,7))):
            pos=int(rng.integers(scan_at,max(scan_at+1,len(j)-12)))
            if j[pos-1]==0xFF or j[pos]==0xFF: continue
            run_=bytes([0xFF,0x00]*int(rng.integers(1,3))+[0xFE]*int(rng.integers(0,2)))
            if 0xFF in j[pos+len(run_):pos+len(run_)+1]: continue
            j[pos:pos+len(run_)]=run_
    j=bytes(j)
    std=bool(it%2)
    try:
        want=orc.ImageData(j,standard_entropy=std).decode()
    except orc.OracleError:
        continue
    for fused, passes in ((1,1),(3,1),(5,1),(5,4),(8,1)):
        got,err=run(j,fused,window=(int(rng.choice([64,200,2048])) if fused not in (5,8) else int(rng.choice([0,0,64,200])) if fused==5 else int(rng.choice([16,64,192]))),standard=std,coop_passes=passes,chunk=int(rng.choice([1,6,16])))
        if isinstance(got,str): continue
        n+=1
        if got is None or not np.array_equal(got,want):
            bad+=1; print('MISMATCH',it,w,h,kind,q,ri,fused,passes,std,err[:200]); open('/tmp/bad_%d.jpg'%it,'wb').write(j)
print('runs',n,'bad',bad)
