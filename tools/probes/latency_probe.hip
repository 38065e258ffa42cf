// Probe (development aid, not part of the library): what a dependent chain costs on gfx950 --
// LDS pointer chase, dependent VALU chain, and the two mixed like a decoder step -- for one wave per SIMD and four.
//   hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip && ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned *out, int iters, int mode)
{
    __shared__ unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x)
        lds[i] = (i * 37 + 11) & 4095;
    __syncthreads();
    unsigned v = threadIdx.x & 4095, acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    if (mode == 0) { // LDS pointer chase
        for (int i = 0; i < iters; i++)
            v = lds[v];
    } else if (mode == 1) { // dependent VALU chain, 8 per iteration
        for (int i = 0; i < iters; i++) {
            v = v * 3 + 1; v ^= v >> 3; v = v * 5 + 7; v ^= v >> 5; v = v * 9 + 3; v ^= v >> 7; v = v * 11 + 1; v ^= v >> 2;
        }
    } else if (mode == 2) { // chase with 10 dependent VALU between the reads
        for (int i = 0; i < iters; i++) {
            unsigned w = lds[v & 4095];
            w = w * 3 + 1; w ^= w >> 3; w += v; w ^= w >> 5; w = w * 9 + 3;
            v = w & 4095;
        }
    } else { // two dependent LDS reads + 20 VALU: the shape of a bit-position reader step
        for (int i = 0; i < iters; i++) {
            unsigned a = lds[v & 4095];
            a = a * 3 + v; a ^= a >> 3; a += 5; a ^= a >> 7; a = a * 5 + 1;
            unsigned b = lds[a & 4095];
            b = b * 3 + 1; b ^= b >> 3; b += a; b ^= b >> 5; b = b * 9 + 3; b ^= b >> 2; b += 7; b ^= b >> 4; b = b * 3 + 2; b ^= v;
            v = b; acc += b;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0)
        out[blockIdx.x * 2] = unsigned(t1 - t0);
    out[blockIdx.x * 2 + 1] = v + acc;
}

int main()
{
    unsigned *d;
    hipMalloc(&d, 1 << 20);
    const int iters = 2000;
    for (int mode = 0; mode < 4; mode++)
        for (int wpb : {64, 256, 512, 1024}) { // 1, 4, 8, 16 waves per workgroup (one workgroup per CU): 0.25, 1, 2, 4 per SIMD
            hipLaunchKernelGGL(probe, dim3(256), dim3(wpb), 0, 0, d, iters, mode);
            hipDeviceSynchronize();
            std::vector<unsigned> h(512);
            hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
            double s = 0;
            for (int b = 0; b < 256; b++) s += h[2 * b];
            printf("mode %d waves/CU %2d: %.1f cycles per iteration\n", mode, wpb / 64, s / 256 / iters);
        }
    return 0;
}
