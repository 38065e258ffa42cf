// Probe: how fast does 1.6 MB travel from pinned host memory to HBM?  One pull kernel, two pull kernels on
// two streams, copy engine, copy engine + pull kernel side by side.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned V4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) pull(V4 *dst, const V4 *src, unsigned n16)
{
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256)
        dst[i] = __builtin_nontemporal_load(src + i);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 1632 * 1024;
    void *h, *d;
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d, bytes));
    memset(h, 0x5a, bytes);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const unsigned n16 = bytes / 16, half = n16 / 2;
    auto blocks = [](unsigned n) { unsigned b = (n + 255) / 256; return b > 2048 ? 2048u : b; };
    for (int mode = 0; mode < 11; mode++) {
        double best = 1e9;
        for (int it = 0; it < 30; it++) {
            CK(hipDeviceSynchronize());
            const double t0 = now_us();
            switch (mode) {
            case 0: pull<<<blocks(n16), 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16); break;
            case 1: pull<<<blocks(half), 256, 0, s1>>>((V4 *)d, (const V4 *)h, half);
                    pull<<<blocks(half), 256, 0, s2>>>((V4 *)d + half, (const V4 *)h + half, half); break;
            case 2: CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s1)); break;
            case 3: CK(hipMemcpyAsync(d, h, bytes / 2, hipMemcpyHostToDevice, s1));
                    pull<<<blocks(half), 256, 0, s2>>>((V4 *)d + half, (const V4 *)h + half, half); break;
            case 4: pull<<<256, 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16); break;       // one workgroup per CU
            case 5: pull<<<blocks(n16) , 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16 / 4); break; // a quarter
            case 6: pull<<<64, 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16); break;
            case 7: pull<<<128, 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16); break;
            case 8: pull<<<32, 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16); break;
            case 9: pull<<<16, 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16); break;
            case 10: pull<<<8, 256, 0, s1>>>((V4 *)d, (const V4 *)h, n16); break;
            }
            CK(hipStreamSynchronize(s1));
            CK(hipStreamSynchronize(s2));
            const double t = now_us() - t0;
            if (t < best) best = t;
        }
        const char *names[] = {"one pull kernel", "two pull kernels, two streams", "copy engine", "copy engine + pull kernel",
                               "pull kernel, 256 workgroups", "pull kernel, a quarter of the bytes", "pull, 64 wg", "pull, 128 wg",
                               "pull, 32 wg", "pull, 16 wg", "pull, 8 wg"};
        printf("%-38s %.1f us wall (launch + sync included)\n", names[mode], best);
    }
    return 0;
}
