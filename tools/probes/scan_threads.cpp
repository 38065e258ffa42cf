// Probe: wall time of ScanBuffer::process on a synthetic 1.6 MB segment with N threads (argv[1]), 12 calls.
//   g++ -O3 -std=c++17 -pthread -Icompeg_amd/csrc -Iinclude tools/probes/scan_threads.cpp \
//       compeg_amd/csrc/scan.cpp compeg_amd/csrc/front.cpp -o /tmp/scan_threads
// On the GPU box's host CPU: 166 us (1 thread), 80 (2), 38 (4), 24 (8).
#include "scan.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <cstdlib>
using namespace compeg;
int main(int argc,char**argv){
  unsigned T = argc>1?atoi(argv[1]):2;
  std::mt19937 rng(1); std::vector<uint8_t> d; uint32_t markers=0;
  while(d.size()<1626961){ int run=60+rng()%80; for(int i=0;i<run;i++){uint8_t b=rng()&0xff; d.push_back(b); if(b==0xff)d.push_back(0);} d.push_back(0xff); d.push_back(0xd0+(markers&7)); markers++; }
  ScanBuffer sb; sb.set_threads(T);
  for(int it=0;it<12;it++){auto t=std::chrono::steady_clock::now(); sb.process(d.data(),d.size(),markers+1); double us=std::chrono::duration<double,std::micro>(std::chrono::steady_clock::now()-t).count(); printf("%.0f ",us);} printf("\n");
}
