// Write-bandwidth probe for the composite stage's store patterns (diagnostic;
// not part of the library).  A frame is 3840x2160 RGBA8 (pitch 15360); a lane
// owns 4 consecutive MCUs (16x8 pixels each = 64 B x 8 rows), like a restart
// interval of the bench workload.
//   pattern 0: MCU by MCU (8 rows x 4 x 16 B), `delay` busy iterations between MCUs  (what the kernel does)
//   pattern 1: all 4 MCUs row by row (16 x 16 B contiguous per row)
//   pattern 2: two MCUs at a time (128-B lines completed at once)
//   pattern 3: fully coalesced (each wave instruction writes 1 KiB contiguous)   (upper bound)
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct alignas(16) V4 { unsigned x, y, z, w; };

__device__ __forceinline__ unsigned spin(unsigned v, int n)
{
    for (int i = 0; i < n; i++)
        asm volatile("v_mad_u32_u24 %0, %0, 3, 1" : "+v"(v));
    return v;
}

__global__ void __launch_bounds__(768) store_kernel(unsigned char *out, int frames, int pattern, int delay)
{
    const unsigned pitch = 15360, width_mcus = 240, intervals = 16200;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned frame = gid / 16384, interval = gid % 16384; // 16384 >= 16200: padded lanes idle
    if (frame >= (unsigned)frames || interval >= intervals)
        return;
    unsigned char *img = out + size_t(frame) * pitch * 2160;
    unsigned v = gid;
    if (pattern == 3) {
        // same bytes per lane (2 KiB), but lane-contiguous: chunk c of the wave's 128 KiB
        const unsigned wave = gid / 64, lane = gid % 64;
        V4 *base = reinterpret_cast<V4 *>(out + size_t(wave) * 131072);
        for (int m = 0; m < 4; m++) {
            v = spin(v, delay);
            for (int i = 0; i < 32; i++)
                base[(m * 32 + i) * 64 + lane] = V4{v, v, v, v};
        }
        return;
    }
    const unsigned mcu0 = interval * 4;
    if (pattern == 0 || pattern == 2) {
        const int group = pattern == 0 ? 1 : 2;
        for (int m = 0; m < 4; m += group) {
            v = spin(v, delay * group);
            for (int row = 0; row < 8; row++)
                for (int g = 0; g < group; g++) {
                    const unsigned mcu = mcu0 + m + g, mx = mcu % width_mcus, my = mcu / width_mcus;
                    V4 *line = reinterpret_cast<V4 *>(img + size_t(my * 8 + row) * pitch + mx * 64);
                    for (int q = 0; q < 4; q++)
                        line[q] = V4{v, v, v, v};
                }
        }
    } else {
        v = spin(v, delay * 4);
        for (int row = 0; row < 8; row++)
            for (int m = 0; m < 4; m++) {
                const unsigned mcu = mcu0 + m, mx = mcu % width_mcus, my = mcu / width_mcus;
                V4 *line = reinterpret_cast<V4 *>(img + size_t(my * 8 + row) * pitch + mx * 64);
                for (int q = 0; q < 4; q++)
                    line[q] = V4{v, v, v, v};
            }
    }
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int frames = argc > 1 ? atoi(argv[1]) : 128;
    const size_t bytes = size_t(frames) * 15360 * 2160;
    unsigned char *out;
    // pattern 3 addresses by padded lane id: 16384 lanes x 2 KiB per frame
    if (hipMalloc(&out, size_t(frames) * 16384 * 2048 + (1 << 20)) != hipSuccess)
        return 1;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int threads = 768, blocks = frames * 16384 / threads + 1;
    printf("%d frames, %.2f GB per pass\n", frames, bytes / 1e9);
    for (int delay : {0, 2000, 8000})
        for (int pattern = 0; pattern < 4; pattern++) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(threads), 0, 0, out, frames, pattern, delay);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (rep && ms < best)
                    best = ms;
            }
            printf("delay %5d pattern %d: %.3f ms  %.0f GB/s  (%.2f us/frame)\n", delay, pattern, best,
                   bytes / best / 1e6, best * 1e3 / frames);
        }
    return 0;
}
