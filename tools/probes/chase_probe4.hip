// Probe (development aid): the cooperative kernel's walk loop (fourth version: stream words in registers) on
// synthetic data; -DV bits knock parts out.  1: no list store, 2: no escape mask/branch, 4: no end test,
// 8: no reload of the stream words, 16: no table recompute, 32: only 16 lanes of 64 walk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef V
#define V 0
#endif
#if V & 1
#define STORE ""
#else
#define STORE "ds_write_b32 %[lp], v40\n\t"
#endif
#if V & 2
#define ESCM ""
#define ESCB ""
#else
#define ESCM "v_cmp_eq_u32 s[76:77], 15, %[ent]\n\t"
#define ESCB "s_and_b64 s[76:77], s[76:77], exec\n\ts_cbranch_scc1 3f\n\t"
#endif
#if V & 4
#define ENDM ""
#define ENDB ""
#else
#define ENDM "v_cmp_ge_u32 s[78:79], %[p], %[stopp]\n\tv_cmp_ge_u32 s[80:81], %[lp], %[lpmax]\n\t"
#define ENDB "s_or_b64 s[78:79], s[78:79], s[80:81]\n\ts_and_b64 s[72:73], s[72:73], s[78:79]\n\ts_andn2_b64 exec, exec, s[72:73]\n\t"
#endif
#if V & 8
#define RELOAD ""
#else
#define RELOAD "ds_read2_b32 v[42:43], %[wa] offset1:1\n\tds_read_b32 v47, %[wa] offset:8\n\t"
#endif
#if V & 16
#define TABLES ""
#else
#define TABLES "v_lshl_add_u32 v44, %[lp], 1, %[kc]\n\tv_bfe_u32 v48, %[acsel], v44, 8\n\tv_bfe_u32 v49, %[dcseln], v44, 8\n\tv_lshl_add_u32 v48, v48, 10, %[tab]\n\tv_lshl_add_u32 v49, v49, 10, %[tab]\n\t"
#endif

#if V & 64
#define NOPS "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
#elif V & 128
#define NOPS "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
#else
#define NOPS ""
#endif
__global__ void probe(unsigned *out, int steps)
{
    __shared__ unsigned win[2048];
    __shared__ unsigned short tab[8192];
    __shared__ unsigned lists[1024 * 20];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        win[i] = i * 2654435761u ^ (i << 7);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) {
        const unsigned tot = 2 + (i * 7) % 9, adv = 1 + (i % 37 == 0 ? 63 : (i % 3));
        tab[i] = (unsigned short)((adv << 9) | (tot << 4) | 2);
    }
    __syncthreads();
    unsigned p = (threadIdx.x * 64) & 0x3fff, st = 1, ent = tab[threadIdx.x];
    unsigned lp = (unsigned)(size_t)(lists + threadIdx.x * 20);
    const unsigned lpmax = lp + 0x100000, stopp = 0x7fffffff;
    const unsigned winaddr = (unsigned)(size_t)win, tabaddr = (unsigned)(size_t)tab;
    const unsigned acsel = 0x04040000u, dcseln = 0x08090908u;
    unsigned wa = winaddr + 4 * (p >> 5), sn = 31, kc = 8 * (threadIdx.x & 3) - 2 * lp;
    unsigned n = steps;
    unsigned alive = (V & 32) ? ((threadIdx.x & 3) == 0) : 1;
    const long long t0 = __builtin_readcyclecounter();
    asm volatile(
        "s_mov_b64 s[74:75], exec\n\t"
        "v_cmp_ne_u32 vcc, 0, %[alive]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "ds_read2_b32 v[42:43], %[wa] offset1:1\n\t"
        "ds_read_b32 v47, %[wa] offset:8\n\t"
        "v_lshl_add_u32 v44, %[lp], 1, %[kc]\n\t"
        "v_bfe_u32 v48, %[acsel], v44, 8\n\t"
        "v_bfe_u32 v49, %[dcseln], v44, 8\n\t"
        "v_lshl_add_u32 v48, v48, 10, %[tab]\n\t"
        "v_lshl_add_u32 v49, v49, 10, %[tab]\n"
        "1:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 v40, %[ent], 4, 5\n\t"
        "v_lshrrev_b32 v44, 9, %[ent]\n\t"
        ESCM
        "v_sub_u32 %[sn], %[sn], v40\n\t"
        "v_add_u32 %[st], %[st], v44\n\t"
        "v_alignbit_b32 v41, v42, v43, %[sn]\n\t"
        "v_alignbit_b32 v45, v43, v47, %[sn]\n\t"
        "v_cmp_gt_i32 vcc, 0, %[sn]\n\t"
        "v_cmp_lt_u32 s[72:73], 63, %[st]\n\t"
        "v_cndmask_b32 v41, v41, v45, vcc\n\t"
        "v_cndmask_b32_e64 v46, v48, v49, s[72:73]\n\t"
        "v_cndmask_b32_e64 v45, 21, 23, s[72:73]\n\t"
        "v_lshrrev_b32 v45, v45, v41\n\t"
        "v_lshl_add_u32 v46, v45, 1, v46\n\t"
        "ds_read_u16 %[ent], v46\n\t"
        "v_add_u32 %[p], %[p], v40\n\t"
        "v_lshl_or_b32 v40, v40, 16, %[p]\n\t"
        STORE
        "v_cndmask_b32_e64 v44, 0, 4, s[72:73]\n\t"
        "v_add_u32 %[lp], %[lp], v44\n\t"
        "v_cndmask_b32_e64 %[st], %[st], 0, s[72:73]\n\t"
        "v_cndmask_b32_e64 v44, 0, 4, vcc\n\t"
        "v_add_u32 %[wa], %[wa], v44\n\t"
        "v_and_b32 %[wa], 0x1fff, %[wa]\n\t"   // (probe only: stay inside the window)
        "v_and_b32 %[sn], 31, %[sn]\n\t"
        RELOAD
        TABLES
        NOPS
        ENDM
        ENDB
        ESCB
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 exec, s[74:75]\n\t"
        : [p] "+v"(p), [st] "+v"(st), [lp] "+v"(lp), [ent] "+v"(ent), [n] "+s"(n), [sn] "+v"(sn), [wa] "+v"(wa)
        : [tab] "s"(tabaddr), [acsel] "v"(acsel), [dcseln] "v"(dcseln), [kc] "v"(kc), [stopp] "v"(stopp), [lpmax] "v"(lpmax), [alive] "v"(alive)
        : "memory", "vcc", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0)
        out[blockIdx.x * 2] = unsigned(t1 - t0);
    out[blockIdx.x * 2 + 1] = p + st + ent + n + wa + sn;
}

int main()
{
    unsigned *d;
    (void)hipMalloc(&d, 1 << 20);
    const int steps = 2000;
    for (int wpb : {64, 256, 1024}) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(wpb), 0, 0, d, steps);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(512);
        (void)hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
        double s = 0;
        for (int b = 0; b < 256; b++) s += h[2 * b];
        printf("v4 variant %2d waves/CU %2d: %.1f cycles per step\n", V, wpb / 64, s / 256 / steps);
    }
    return 0;
}
