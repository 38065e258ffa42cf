// Probe (development aid): the lean walk loop -- 32-bit walk-table entries (adv << 16 | -size), position and zig-zag
// state in one register (packed 16-bit add), 16-byte list entries {word address, state, AC table, next DC table}.
// -DV: 1: stream words by one ds_read_b96 (4-byte aligned addresses), 2: 16 lanes of 64 walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef V
#define V 0
#endif
#if V & 1
#define RELOAD "ds_read_b96 v[42:44], v50\n\t"
#else
#define RELOAD "ds_read2_b32 v[42:43], v50 offset1:1\n\tds_read_b32 v44, v50 offset:8\n\t"
#endif

__global__ void probe(unsigned *out, int steps)
{
    __shared__ unsigned win[2048];
    __shared__ __attribute__((aligned(4096))) unsigned tab[4096];
    __shared__ __attribute__((aligned(16))) unsigned lists[1024 * 8];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        win[i] = i * 2654435761u ^ (i << 7);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        const unsigned tot = 2 + (i * 7) % 9, adv = 1 + (i % 37 == 0 ? 63 : (i % 3));
        tab[i] = (adv << 16) | ((0u - tot) & 0xffffu);
    }
    const unsigned winaddr = (unsigned)(size_t)win, tabaddr = (unsigned)(size_t)tab;
    // list entries: two per lane, used round robin by the probe (the real walk has one per data unit)
    unsigned lp = (unsigned)(size_t)(lists + threadIdx.x * 8);
    lists[threadIdx.x * 8 + 2] = tabaddr; lists[threadIdx.x * 8 + 3] = tabaddr + 4096 * 2;
    lists[threadIdx.x * 8 + 6] = tabaddr + 4096; lists[threadIdx.x * 8 + 7] = tabaddr + 4096 * 3;
    __syncthreads();
    unsigned ent = tab[threadIdx.x];
    unsigned wa = winaddr + 4 * ((threadIdx.x * 2) & 0x3ff), T = (1u << 16) | 31u;
    const unsigned lpmax = lp + 0x100000, K = 0x3fffffu, M0 = 0xffff001fu;
    unsigned n = steps;
    unsigned alive = (V & 2) ? ((threadIdx.x & 3) == 0) : 1;
    const long long t0 = __builtin_readcyclecounter();
    asm volatile(
        "s_mov_b64 s[74:75], exec\n\t"
        "v_cmp_ne_u32 vcc, 0, %[alive]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "v_mov_b32 v50, %[wa]\n\t"
        "v_mov_b32 v51, %[T]\n\t"
        RELOAD
        "ds_read_b64 v[48:49], %[lp] offset:8\n"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_add_u16 v51, v51, %[ent]\n\t"                    // size off the shift, advance onto the zig-zag state
        "v_cmp_eq_u32 s[76:77], 0, %[ent]\n\t"                 // escape
        "v_alignbit_b32 v41, v42, v43, v51\n\t"
        "v_alignbit_b32 v45, v43, v44, v51\n\t"
        "v_cmp_gt_i16 vcc, 0, v51\n\t"                         // the position has left A
        "v_cmp_lt_u32 s[72:73], %[K], v51\n\t"                 // the data unit is complete
        "v_cndmask_b32 v41, v41, v45, vcc\n\t"
        "v_cndmask_b32_e64 v46, v48, v49, s[72:73]\n\t"
        "v_lshrrev_b32 v41, 20, v41\n\t"
        "v_and_or_b32 v46, v41, %[idxm], v46\n\t"
        "ds_read_b32 %[ent], v46\n\t"
        "v_cndmask_b32_e64 v40, 0, 4, vcc\n\t"
        "v_add_u32 v50, v50, v40\n\t"
        "v_and_b32 v50, 0x1fff, v50\n\t"                       // (probe only: stay inside the window)
        "v_cndmask_b32_e64 v40, %[M0], 31, s[72:73]\n\t"
        "v_and_b32 v51, v51, v40\n\t"
        "ds_write_b64 %[lp], v[50:51]\n\t"
        "v_cndmask_b32_e64 v40, 0, 16, s[72:73]\n\t"
        "v_add_u32 %[lp], %[lp], v40\n\t"
        "v_and_b32 %[lp], 0xffffffef, %[lp]\n\t"               // (probe only: two entries)
        RELOAD
        "ds_read_b64 v[48:49], %[lp] offset:8\n\t"
        "v_cmp_ge_u32 s[78:79], %[lp], %[lpmax]\n\t"
        "s_andn2_b64 exec, exec, s[78:79]\n\t"
        "s_and_b64 s[76:77], s[76:77], exec\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mov_b32 %[wa], v50\n\t"
        "v_mov_b32 %[T], v51\n\t"
        "s_mov_b64 exec, s[74:75]\n\t"
        : [lp] "+v"(lp), [ent] "+v"(ent), [n] "+s"(n), [T] "+v"(T), [wa] "+v"(wa)
        : [K] "s"(K), [M0] "v"(M0), [idxm] "v"(0xffcu), [lpmax] "v"(lpmax), [alive] "v"(alive)
        : "memory", "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v48", "v49", "v50", "v51",
          "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0)
        out[blockIdx.x * 2] = unsigned(t1 - t0);
    out[blockIdx.x * 2 + 1] = T + ent + n + wa;
}

int main()
{
    unsigned *d;
    (void)hipMalloc(&d, 1 << 20);
    const int steps = 2000;
    for (int wpb : {64, 256, 1024}) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(wpb), 0, 0, d, steps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<unsigned> h(512);
        (void)hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
        double s = 0;
        for (int b = 0; b < 256; b++) s += h[2 * b];
        printf("v5 variant %2d waves/CU %2d: %.1f cycles per step (check %u)\n", V, wpb / 64, s / 256 / steps, h[1]);
    }
    return 0;
}
