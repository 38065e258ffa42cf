// Issue-rate probe for the VALU instructions the decode kernel is made of
// (diagnostic; not part of the library).  Each wave runs ITER x 32 independent
// instances of one instruction between two cycle-counter reads; with W waves
// per SIMD the per-SIMD cost is (wave cycles / instructions) / W.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

#define PROBE(name, body, clob)                                                              \
    __global__ void name(uint64_t *out, int iters)                                           \
    {                                                                                        \
        float a = __builtin_bit_cast(float, (threadIdx.x & 63u) * 8u), b = blockIdx.x + 1.5f;                                        \
        uint64_t t0 = __builtin_readcyclecounter();                                          \
        for (int i = 0; i < iters; i++) {                                                    \
            asm volatile(REP32(body) : "+v"(a), "+v"(b)::"v10", "v11", "v12", "v13", "v14", "v15", "vcc", "scc", "s20", "s21", "s22", "s23", clob);  \
        }                                                                                    \
        uint64_t t1 = __builtin_readcyclecounter();                                          \
        if ((threadIdx.x & 63) == 0)                                                         \
            out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;                     \
        if (a == 12345.f && b == 1.25f)                                                        \
            out[0] = 0;                                                                      \
    }

PROBE(k_add_f32, "v_add_f32 v10, %0, %1\n", "v16")
PROBE(k_mul_f32, "v_mul_f32 v10, %0, %1\n", "v16")
PROBE(k_fma_f32, "v_fma_f32 v10, %0, %1, %1\n", "v16")
PROBE(k_pk_add_f32, "v_pk_add_f32 v[10:11], v[12:13], v[14:15]\n", "v16")
PROBE(k_pk_mul_f32, "v_pk_mul_f32 v[10:11], v[12:13], v[14:15]\n", "v16")
PROBE(k_pk_fma_f32, "v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[14:15]\n", "v16")
PROBE(k_lshl_b64, "v_lshlrev_b64 v[10:11], %0, v[12:13]\n", "v16")
PROBE(k_lshr_b32, "v_lshrrev_b32 v10, %0, %1\n", "v16")
PROBE(k_perm, "v_perm_b32 v10, %0, %1, v12\n", "v16")
PROBE(k_bfe, "v_bfe_u32 v10, %0, %1, 5\n", "v16")
PROBE(k_cndmask, "v_cndmask_b32 v10, %0, %1, vcc\n", "v16")
PROBE(k_cmp, "v_cmp_lt_u32 vcc, %0, %1\n", "v16")
PROBE(k_mul_lo_u32, "v_mul_lo_u32 v10, %0, %1\n", "v16")
PROBE(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 v10, %0, %1\n", "v16")
PROBE(k_pk_add_u16, "v_pk_add_u16 v10, %0, %1\n", "v16")
PROBE(k_sat_pk, "v_sat_pk_u8_i16 v10, %0\n", "v16")
PROBE(k_cvt_pk_u8, "v_cvt_pk_u8_f32 v10, %0, 1, %1\n", "v16")
PROBE(k_cvt_sdwa, "v_cvt_f32_i32_sdwa v10, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n", "v16")
PROBE(k_add_u32, "v_add_u32 v10, %0, %1\n", "v16")
PROBE(k_add3, "v_add3_u32 v10, %0, %1, %1\n", "v16")
PROBE(k_lshl_or, "v_lshl_or_b32 v10, %0, 3, %1\n", "v16")
PROBE(k_and_or, "v_and_or_b32 v10, %0, %1, %1\n", "v16")
PROBE(k_mov_b64, "v_mov_b64 v[10:11], v[12:13]\n", "v16")
PROBE(k_snop, "s_nop 0\n", "v16")
PROBE(k_salu, "s_add_u32 s20, s21, s22\n", "v16")
PROBE(k_cnd_e64, "v_cndmask_b32_e64 v10, %0, %1, s[20:21]\n", "v16")
PROBE(k_cnd_k, "v_cndmask_b32 v10, 0, %1, vcc\n", "v16")
PROBE(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 v10, %0, %1, vcc\n", "v16")
PROBE(k_cmp_s_cnd, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 v10, %0, %1, s[20:21]\n", "v16")
PROBE(k_cmp_e64, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n", "v16")
PROBE(k_max_u32, "v_max_u32 v10, %0, %1\n", "v16")
PROBE(k_min_u32, "v_min_u32 v10, %0, %1\n", "v16")
PROBE(k_med3, "v_med3_i32 v10, %0, %1, %1\n", "v16")
PROBE(k_and, "v_and_b32 v10, %0, %1\n", "v16")
PROBE(k_and_k, "v_and_b32 v10, 0xffff, %1\n", "v16")
PROBE(k_lshl_b32, "v_lshlrev_b32 v10, %0, %1\n", "v16")
PROBE(k_ashr_i32, "v_ashrrev_i32 v10, 31, %1\n", "v16")
PROBE(k_lshr_b64, "v_lshrrev_b64 v[10:11], %0, v[12:13]\n", "v16")
PROBE(k_alignbit, "v_alignbit_b32 v10, %0, %1, %1\n", "v16")
PROBE(k_bfi, "v_bfi_b32 v10, %0, %1, %1\n", "v16")
PROBE(k_sub_u32, "v_sub_u32 v10, %0, %1\n", "v16")
PROBE(k_addc, "v_add_co_u32 v10, vcc, %0, %1\n", "v16")
PROBE(k_mad_u32_u24, "v_mad_u32_u24 v10, %0, %1, %1\n", "v16")
PROBE(k_mul_u32_u24, "v_mul_u32_u24 v10, %0, %1\n", "v16")
PROBE(k_mov, "v_mov_b32 v10, %0\n", "v16")
PROBE(k_mov_k, "v_mov_b32 v10, 0x12345\n", "v16")
PROBE(k_add_lit, "v_add_u32 v10, 0x12345, %0\n", "v16")
PROBE(k_add_f32_e64, "v_add_f32_e64 v10, %0, -%1\n", "v16")
PROBE(k_sub_f32, "v_sub_f32 v10, %0, %1\n", "v16")
PROBE(k_mul_f32_k, "v_mul_f32 v10, 0x3fb504f3, %1\n", "v16")
PROBE(k_readfirst, "v_readfirstlane_b32 s20, %0\n", "v16")
PROBE(k_ds_read, "ds_read_b32 v10, %0\n", "v16")
PROBE(k_ds_read_u16, "ds_read_u16 v10, %0\n", "v16")
PROBE(k_ds_write_b16, "ds_write_b16 %0, %1\n", "v16")
PROBE(k_setreg, "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n", "v16")
PROBE(k_setreg_pair, "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n v_add_f32 v10, %0, %1\n s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n v_add_f32 v11, %0, %1\n", "v16")
PROBE(k_bfe_i32, "v_bfe_i32 v10, %0, %1, %1\n", "v16")
PROBE(k_ds_read_b64, "ds_read_b64 v[10:11], %0\n", "v16")
PROBE(k_ds_read_b64_u1, "ds_read_b64 v[10:11], %0 offset:1\n", "v16")
PROBE(k_ds_read_b64_u5, "ds_read_b64 v[10:11], %0 offset:5\n", "v16")
PROBE(k_ds_read2_b32, "ds_read2_b32 v[10:11], %0 offset0:0 offset1:1\n", "v16")
PROBE(k_ds_read_b96, "ds_read_b96 v[12:14], %0\n", "v16")
PROBE(k_ds_write_b32, "ds_write_b32 %0, %1\n", "v16")
PROBE(k_ds_write_b64, "ds_write_b64 %0, v[12:13]\n", "v16")
PROBE(k_pk_mad_i16, "v_pk_mad_i16 v10, %0, %1, %1\n", "v16")
PROBE(k_pk_max_i16, "v_pk_max_i16 v10, %0, %1\n", "v16")
PROBE(k_cvt_f32_i32, "v_cvt_f32_i32 v10, %0\n", "v16")
#define REP8x4(a,b,c,d) a b c d a b c d a b c d a b c d a b c d a b c d a b c d a b c d
#define PROBE4(name, i0, i1, i2, i3)                                                          \
    __global__ void name(uint64_t *out, int iters)                                           \
    {                                                                                        \
        float a = __builtin_bit_cast(float, (threadIdx.x & 63u) * 8u), b = blockIdx.x + 1.5f; \
        uint64_t t0 = __builtin_readcyclecounter();                                          \
        for (int i = 0; i < iters; i++) {                                                    \
            asm volatile(REP8x4(i0, i1, i2, i3) : "+v"(a), "+v"(b)::"v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "vcc", "scc");  \
        }                                                                                    \
        uint64_t t1 = __builtin_readcyclecounter();                                          \
        if ((threadIdx.x & 63) == 0)                                                         \
            out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;                     \
        if (a == 12345.f && b == 1.25f)                                                      \
            out[0] = 0;                                                                      \
    }
#define PROBE_N(name, REPS, N)                                                                 \
    __global__ void name(uint64_t *out, int iters)                                           \
    {                                                                                        \
        float a = __builtin_bit_cast(float, (threadIdx.x & 63u) * 8u), b = blockIdx.x + 1.5f; \
        const int loops = iters * 32 / N;                                                    \
        uint64_t t0 = __builtin_readcyclecounter();                                          \
        for (int i = 0; i < loops; i++) {                                                    \
            asm volatile(REPS("v_add_f32 v10, %0, %1\n") : "+v"(a), "+v"(b)::"v10");        \
        }                                                                                    \
        uint64_t t1 = __builtin_readcyclecounter();                                          \
        if ((threadIdx.x & 63) == 0)                                                         \
            out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;                     \
        if (a == 12345.f && b == 1.25f)                                                      \
            out[0] = 0;                                                                      \
    }
#define REP1(x) x
#define REP2(x) x x
#define REP8(x) REP4(x) REP4(x)
#define REP128(x) REP32(x) REP32(x) REP32(x) REP32(x)
PROBE_N(k_body2, REP2, 2)
PROBE_N(k_body4, REP4, 4)
PROBE_N(k_body8, REP8, 8)
PROBE_N(k_body128, REP128, 128)
PROBE4(k_ilp_add_f32, "v_add_f32 v10, %0, %1\n", "v_add_f32 v11, %0, %1\n", "v_add_f32 v12, %0, %1\n", "v_add_f32 v13, %0, %1\n")
PROBE4(k_ilp_chain4, "v_add_f32 v10, v10, %1\n", "v_add_f32 v11, v11, %1\n", "v_add_f32 v12, v12, %1\n", "v_add_f32 v13, v13, %1\n")
PROBE4(k_ilp_chain2, "v_add_f32 v10, v10, %1\n", "v_add_f32 v11, v11, %1\n", "v_add_f32 v10, v10, %1\n", "v_add_f32 v11, v11, %1\n")
PROBE4(k_ilp_perm, "v_perm_b32 v10, %0, %1, v14\n", "v_perm_b32 v11, %0, %1, v14\n", "v_perm_b32 v12, %0, %1, v14\n", "v_perm_b32 v13, %0, %1, v14\n")
PROBE4(k_ilp_pk, "v_pk_add_f32 v[10:11], v[14:15], v[16:17]\n", "v_pk_add_f32 v[12:13], v[14:15], v[16:17]\n", "v_pk_add_f32 v[18:19], v[14:15], v[16:17]\n", "v_pk_add_f32 v[10:11], v[14:15], v[16:17]\n")
PROBE4(k_ilp_mix, "v_add_f32 v10, %0, %1\n", "v_lshlrev_b64 v[12:13], %0, v[14:15]\n", "v_sub_f32 v11, %0, %1\n", "v_perm_b32 v16, %0, %1, v14\n")
PROBE(k_dep_add, "v_add_u32 %0, %0, %1\n", "v16")
PROBE(k_dep_lshl64, "v_lshlrev_b64 v[12:13], %0, v[12:13]\n", "v16")
PROBE(k_dep_perm, "v_perm_b32 %0, %0, %1, v12\n", "v16")
PROBE(k_dep_cnd, "v_cndmask_b32 %0, %0, %1, vcc\n", "v16")

struct Probe { const char *name; void (*fn)(uint64_t *, int); };

int main()
{
    const Probe probes[] = {
        {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32},
        {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_lshlrev_b64", k_lshl_b64}, {"v_lshrrev_b32", k_lshr_b32}, {"v_perm_b32", k_perm},
        {"v_bfe_u32", k_bfe}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_u32", k_cmp},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16}, {"v_pk_add_u16", k_pk_add_u16},
        {"v_sat_pk_u8_i16", k_sat_pk}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8}, {"v_cvt_f32_i32_sdwa", k_cvt_sdwa},
        {"v_add_u32", k_add_u32}, {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshl_or}, {"v_and_or_b32", k_and_or},
        {"v_mov_b64", k_mov_b64}, {"s_nop 0", k_snop}, {"s_add_u32", k_salu},
        {"v_cndmask_e64 sgpr", k_cnd_e64}, {"v_cndmask 0,v,vcc", k_cnd_k}, {"v_cmp+v_cndmask vcc /2", k_cmp_cnd},
        {"v_cmp+v_cndmask sgpr /2", k_cmp_s_cnd}, {"v_cmp_e64 sgpr", k_cmp_e64}, {"v_max_u32", k_max_u32}, {"v_min_u32", k_min_u32},
        {"v_med3_i32", k_med3}, {"v_and_b32", k_and}, {"v_and_b32 lit", k_and_k}, {"v_lshlrev_b32", k_lshl_b32},
        {"v_ashrrev_i32", k_ashr_i32}, {"v_lshrrev_b64", k_lshr_b64}, {"v_alignbit_b32", k_alignbit}, {"v_bfi_b32", k_bfi},
        {"v_sub_u32", k_sub_u32}, {"v_add_co_u32", k_addc}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mov_b32", k_mov}, {"v_mov_b32 lit", k_mov_k}, {"v_add_u32 lit", k_add_lit}, {"v_add_f32_e64 neg", k_add_f32_e64},
        {"v_sub_f32", k_sub_f32}, {"v_mul_f32 lit", k_mul_f32_k}, {"v_readfirstlane", k_readfirst},
        {"ds_read_b32", k_ds_read}, {"ds_read_u16", k_ds_read_u16}, {"ds_write_b16", k_ds_write_b16},
        {"s_setreg_imm32", k_setreg}, {"setreg,add,setreg,add /4", k_setreg_pair}, {"v_bfe_i32", k_bfe_i32},
        {"ds_read_b64", k_ds_read_b64}, {"ds_read_b64 +1", k_ds_read_b64_u1}, {"ds_read_b64 +5", k_ds_read_b64_u5},
        {"ds_read2_b32", k_ds_read2_b32}, {"ds_read_b96", k_ds_read_b96}, {"ds_write_b32", k_ds_write_b32}, {"ds_write_b64", k_ds_write_b64},
        {"v_pk_mad_i16", k_pk_mad_i16}, {"v_pk_max_i16", k_pk_max_i16}, {"v_cvt_f32_i32", k_cvt_f32_i32},
        {"loop body 2 v_add", k_body2}, {"loop body 4 v_add", k_body4}, {"loop body 8 v_add", k_body8}, {"loop body 128 v_add", k_body128},
        {"ILP4 v_add_f32", k_ilp_add_f32}, {"4 chains v_add_f32", k_ilp_chain4}, {"2 chains v_add_f32", k_ilp_chain2},
        {"ILP4 v_perm_b32", k_ilp_perm}, {"ILP3 v_pk_add_f32", k_ilp_pk}, {"ILP4 mix", k_ilp_mix},
        {"dep v_add_u32", k_dep_add}, {"dep v_lshlrev_b64", k_dep_lshl64}, {"dep v_perm_b32", k_dep_perm}, {"dep v_cndmask", k_dep_cnd},
    };
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 2000;
    uint64_t *out;
    hipMalloc(&out, sizeof(uint64_t) * cus * 64);
    printf("%-22s %10s %10s %10s %10s   (cycles per instruction per SIMD)\n", "instruction", "1 wave", "2 waves", "3 waves", "4 waves");
    for (const Probe &p : probes) {
        printf("%-22s", p.name);
        for (int w = 1; w <= 4; w++) {
            // one block of 256*w threads per CU: w waves on each of the 4 SIMDs
            const int nwaves = cus * 4 * w;
            hipLaunchKernelGGL(p.fn, dim3(cus), dim3(256 * w), 16384, 0, out, iters);
            if (hipDeviceSynchronize() != hipSuccess) {
                printf(" launch failed: %s\n", hipGetErrorString(hipGetLastError()));
                return 1;
            }
            std::vector<uint64_t> h(nwaves);
            hipMemcpy(h.data(), out, sizeof(uint64_t) * nwaves, hipMemcpyDeviceToHost);
            double sum = 0;
            for (uint64_t v : h)
                sum += double(v);
            printf(" %10.2f", sum / nwaves / (double(iters) * 32) / w);
        }
        printf("\n");
    }
    hipFree(out);
    return 0;
}
