// Test infrastructure: the host scan preprocessor with 2, 3, 4, 5 and 8 threads against the one-thread loop on
// random segments, built with -fsanitize=thread (tests/emul/Makefile) and run by tests/test_kernel_emulation.py.
#include "scan.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace compeg;
int main(){
  std::mt19937 rng(5);
  int bad=0;
  for(int it=0;it<16;it++){
    size_t n=300000+rng()%900000; std::vector<uint8_t> d(n);
    // iterations 12..15: skewed segments -- markers crowd into the last sixteenth (12, 13) or the second half
    // (14, 15) and are rare elsewhere, so that one piece of one round holds almost all of them; every
    // callback still has to find the whole reported prefix final
    const size_t dense_from = it<12 ? 0 : (it<14 ? n-n/16 : n/2);
    for(size_t i=0;i<n;i++){uint8_t&b=d[i]; uint32_t r=rng(); b=(r&0xff); if(b==0xff)b=0x7f;
      const uint32_t every = it<12 ? 50 : (i>=dense_from ? 3 : 20000);
      if((r>>8)%every==0)b=0xff; if((r>>16)%90==0)b=0;}
    uint32_t exp=1+rng()%20000;
    ScanBuffer ref; ref.process(d.data(),n,exp);
    for(unsigned T:{2u,3u,4u,5u,8u}){ ScanBuffer sb; sb.set_threads(T); sb.process(d.data(),n,exp); sb.process(d.data(),n,exp);
      if(sb.nwords()!=ref.nwords()||sb.nstarts()!=ref.nstarts()||memcmp(sb.data(),ref.data(),ref.nwords()*4)||memcmp(sb.starts(),ref.starts(),ref.nstarts()*4)){bad++; printf("DIFF it %d T %u\n",it,T);}
      // with a progress callback (rounds): what is reported as final must already equal the final bytes
      size_t reported=0, calls=0; bool early_ok=true;
      sb.process(d.data(),n,exp,[&](size_t fin){ calls++; if(fin<reported||fin%16||fin>ref.nwords()*4||memcmp(sb.data(),ref.data(),fin)) early_ok=false; reported=fin; },64u<<10);
      if(!early_ok||sb.nwords()!=ref.nwords()||memcmp(sb.data(),ref.data(),ref.nwords()*4)||memcmp(sb.starts(),ref.starts(),ref.nstarts()*4)){bad++; printf("DIFF with progress it %d T %u calls %zu\n",it,T,calls);}
      if(it==0) printf("T %u: %zu progress calls, %zu of %zu bytes reported early\n",T,calls,reported,ref.nwords()*4); }
  }
  // ScanBuffer::copy (the decoder stages raw segments with it): every byte, whatever the length and the thread
  // count -- lengths around multiples of 64 x threads, where a piece size rounded down first once lost the tail
  for(unsigned T:{2u,3u,4u,5u}){ ScanBuffer sb; sb.set_threads(T);
    for(size_t base:{size_t(64u<<10)*T, size_t(85248)*4, size_t(300000)}) for(size_t extra=0;extra<70;extra+=(extra<6?1:13)){
      const size_t n=base+extra; std::vector<uint8_t> src(n), dst(n+64,0xee);
      for(size_t i=0;i<n;i++) src[i]=uint8_t(rng());
      sb.copy(dst.data(),src.data(),n);
      if(memcmp(dst.data(),src.data(),n)||dst[n]!=0xee){bad++; printf("COPY DIFF T %u n %zu\n",T,n);} } }
  printf("bad %d\n",bad); return bad;
}
