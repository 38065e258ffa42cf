// Probe (development aid): cycles per iteration of the cooperative kernel's hand-written walk loop on synthetic
// data, with parts of it knocked out (-DV=bits), to see what a step is made of.
//   for v in 0 1 2 4 8 16; do hipcc --offload-arch=gfx950 -O3 -DV=$v -o chase_probe_$v chase_probe.hip; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#ifndef V
#define V 0
#endif
// bit 0: no list store; bit 1: no escape test/branch; bit 2: no 64-bit shift (two 32-bit ops instead);
// bit 3: exit test through a counter in an SGPR only (no per-lane end test); bit 4: no second LDS read (table)
#if V & 1
#define STORE ""
#else
#define STORE "ds_write_b32 %[lp], v40\n\t"
#endif
#if V & 2
#define ESC ""
#else
#define ESC "v_cmp_le_u32 vcc, 0xfe00, %[ent]\n\ts_cbranch_vccnz 3f\n\t"
#endif
#if V & 4
#define SHIFT "v_lshlrev_b32 v43, v44, v43\n\tv_lshrrev_b32 v42, 3, v42\n\tv_or_b32 v43, v43, v42\n\t"
#else
#define SHIFT "v_lshlrev_b64 v[42:43], v44, v[42:43]\n\t"
#endif
#if V & 8
#define ENDTEST ""
#else
#define ENDTEST                                                                                        \
    "v_cndmask_b32_e64 v45, 0, %[p], vcc\n\t"                                                          \
    "v_cmp_ge_u32 s[72:73], %[lp], %[lpmax]\n\t"                                                       \
    "v_cndmask_b32_e64 v46, %[stopp], 1, s[72:73]\n\t"                                                 \
    "v_cmp_ge_u32 vcc, v45, v46\n\t"                                                                   \
    "s_andn2_b64 exec, exec, vcc\n\t"
#endif
#if V & 32
#undef ESC
#define ESC "v_cmp_le_u32 vcc, 0xfe00, %[ent]\n\t"
#define ESCBR "s_cbranch_vccnz 3f\n\t"
#else
#define ESCBR ""
#endif
#if V & 16
#define TABLE "v_mov_b32 %[ent], 0x252\n\t"
#else
#define TABLE "ds_read_u16 %[ent], v46\n\ts_waitcnt lgkmcnt(0)\n\t"
#endif

__global__ void probe(unsigned *out, int steps)
{
    __shared__ unsigned win[2048];
    __shared__ unsigned short tab[8192]; // 2 AC x 2048 + 2 DC x 512 (KiB offsets 0, 4, 8, 9)
    __shared__ unsigned lists[1024 * 20];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x)
        win[i] = i * 2654435761u ^ (i << 7);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) {
        const unsigned tot = 2 + (i * 7) % 9, adv = 1 + (i % 37 == 0 ? 63 : (i % 3));
        tab[i] = (unsigned short)((adv << 9) | (tot << 4) | 2);
    }
    __syncthreads();
    unsigned p = (threadIdx.x * 64) & 0x7fff, st = 1, k8 = 8 * (threadIdx.x & 3), ent = 0;
    unsigned lp = (unsigned)(size_t)(lists + threadIdx.x * 20);
    const unsigned lpmax = lp + 19 * 4 + 0x100000, stopp = 0x7fffffff;
    const unsigned winaddr = (unsigned)(size_t)win, tabaddr = (unsigned)(size_t)tab;
    const unsigned acsel = 0x04040000u, dcsel = 0x09090808u;
    unsigned n = steps;
    const long long t0 = __builtin_readcyclecounter();
    asm volatile(
        "s_mov_b64 s[74:75], exec\n"
        "1:\n\t"
        "v_lshrrev_b32 v40, 5, %[p]\n\t"
        "v_and_b32 v40, 0x3ff, v40\n\t" // (synthetic: stay inside the window)
        "v_lshl_add_u32 v40, v40, 2, %[win]\n\t"
        "ds_read2_b32 v[42:43], v40 offset0:1 offset1:0\n\t"
        "v_and_b32 v44, 31, %[p]\n\t"
        "v_cmp_eq_u32 vcc, 0, %[st]\n\t"
        "v_cndmask_b32 v46, %[acsel], %[dcsel], vcc\n\t"
        "v_bfe_u32 v46, v46, %[k8], 8\n\t"
        "v_cndmask_b32_e64 v45, 21, 23, vcc\n\t"
        "v_lshl_add_u32 v46, v46, 10, %[tab]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        SHIFT
        "v_lshrrev_b32 v45, v45, v43\n\t"
        "v_lshl_add_u32 v46, v45, 1, v46\n\t"
        TABLE
        ESC
        "v_bfe_u32 v40, %[ent], 4, 5\n\t"
        "v_lshrrev_b32 v44, 9, %[ent]\n\t"
        "v_add_u32 v47, %[p], v40\n\t"
        "v_add_u32 v48, %[st], v44\n\t"
        "v_lshl_or_b32 v40, v40, 16, v47\n\t"
        ESCBR
        "v_mov_b32 %[p], v47\n\t"
        "v_mov_b32 %[st], v48\n\t"
        STORE
        "v_cmp_lt_u32 vcc, 63, %[st]\n\t"
        "v_cndmask_b32_e64 v44, 0, 1, vcc\n\t"
        "v_lshl_add_u32 %[k8], v44, 3, %[k8]\n\t"
        "v_cndmask_b32_e64 %[st], %[st], 0, vcc\n\t"
        ENDTEST
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "3:\n\t"
        "s_mov_b64 exec, s[74:75]\n\t"
        : [p] "+v"(p), [st] "+v"(st), [k8] "+v"(k8), [lp] "+v"(lp), [ent] "+v"(ent), [n] "+s"(n)
        : [win] "s"(winaddr), [tab] "s"(tabaddr), [acsel] "v"(acsel), [dcsel] "v"(dcsel), [stopp] "v"(stopp), [lpmax] "v"(lpmax)
        : "memory", "vcc", "scc", "v40", "v42", "v43", "v44", "v45", "v46", "s72", "s73", "s74", "s75", "v47", "v48");
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0)
        out[blockIdx.x * 2] = unsigned(t1 - t0);
    out[blockIdx.x * 2 + 1] = p + st + k8 + ent + n;
}

int main()
{
    unsigned *d;
    (void)hipMalloc(&d, 1 << 20);
    const int steps = 2000;
    for (int wpb : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(wpb), 0, 0, d, steps);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(512);
        (void)hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
        double s = 0;
        for (int b = 0; b < 256; b++) s += h[2 * b];
        printf("variant %2d waves/CU %2d: %.1f cycles per step\n", V, wpb / 64, s / 256 / steps);
    }
    return 0;
}
