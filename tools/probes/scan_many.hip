// Probe: T threads, each running ScanBuffer::process_to over its own share of 256 synthetic 1.6 MB segments (the host
// side of a batch upload without the copies), into (a) ordinary memory, (b) pinned memory, (c) pinned memory while
// copies of the finished segments run.  Wall time and the threads' summed time.
//   hipcc -O3 -std=c++17 -pthread -Icompeg_amd/csrc -Iinclude tools/probes/scan_many.hip \
//       compeg_amd/csrc/scan.cpp compeg_amd/csrc/front.cpp -o gpurun_ab/probes/scan_many
#include "scan.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
using namespace compeg;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const unsigned T = argc > 1 ? atoi(argv[1]) : 32, N = 256, DISTINCT = 64;
    std::vector<std::vector<uint8_t>> segs(DISTINCT);
    std::vector<uint32_t> markers(DISTINCT, 0);
    for (unsigned s = 0; s < DISTINCT; s++) {
        std::mt19937 rng(s + 1);
        auto &d = segs[s];
        while (d.size() < 1626961) {
            int run = 60 + rng() % 80;
            for (int i = 0; i < run; i++) { uint8_t b = rng() & 0xff; d.push_back(b); if (b == 0xff) d.push_back(0); }
            d.push_back(0xff); d.push_back(0xd0 + (markers[s] & 7)); markers[s]++;
        }
    }
    const size_t per = 2u << 20;
    uint8_t *plain = static_cast<uint8_t *>(aligned_alloc(4096, per * N)), *pinned = nullptr, *dev = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&pinned), per * N, hipHostMallocDefault) != hipSuccess) return 1;
    if (hipMalloc(reinterpret_cast<void **>(&dev), per * N) != hipSuccess) return 1;
    memset(plain, 1, per * N); memset(pinned, 1, per * N);
    hipStream_t cs[4];
    for (auto &c : cs) (void)hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; mode++) for (int rep = 0; rep < 3; rep++) {
        uint8_t *dst = mode == 0 ? plain : pinned;
        std::atomic<size_t> next{0};
        std::atomic<uint64_t> busy_us{0};
        const double t0 = now_ms();
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) th.emplace_back([&, t] {
            (void)hipSetDevice(0);
            for (size_t i; (i = next.fetch_add(1)) < N;) {
                const auto &d = segs[i % DISTINCT];
                const size_t slots = ScanBuffer::start_slots(markers[i % DISTINCT] + 1);
                uint32_t *starts = reinterpret_cast<uint32_t *>(dst + per * i);
                size_t nw = 0, ns = 0;
                const double a = now_ms();
                (void)ScanBuffer::process_to(d.data(), d.size(), markers[i % DISTINCT] + 1, dst + per * i + slots * 4, starts, nw, ns);
                busy_us += uint64_t((now_ms() - a) * 1e3);
                if (mode == 2) (void)hipMemcpyAsync(dev + per * i, dst + per * i, slots * 4 + nw * 4, hipMemcpyHostToDevice, cs[t % 4]);
            }
        });
        for (auto &x : th) x.join();
        const double t1 = now_ms();
        for (auto &c : cs) (void)hipStreamSynchronize(c);
        printf("%s, %u threads: host %.2f ms (+ %.2f ms until the copies are done), threads' time %.1f ms = %.2f ms per segment\n",
               mode == 0 ? "ordinary memory" : mode == 1 ? "pinned memory" : "pinned memory + copies", T, t1 - t0, now_ms() - t1,
               busy_us.load() / 1e3, busy_us.load() / 1e3 / N);
    }
    return 0;
}
