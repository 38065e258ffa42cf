// Probe: what does v_cvt_pk_u8_f32 do with fractions, negatives and values > 255?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i < n) {
        unsigned r;
        float x = in[i];
        float y = x * 1.0000001f; // a rounding-sensitive op after the block must still be RNE
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                     "v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(r) : "v"(x));
        float z = x * 1.0000001f + 0.0f;
        if (__float_as_uint(y) != __float_as_uint(x * 1.0000001f) || z != y) r |= 0x80000000u;
        out[i] = r;
    }
}
int main()
{
    const int n = 4096;
    float h[n];
    unsigned o[n];
    for (int i = 0; i < n; i++)
        h[i] = -20.0f + i * (300.0f / n) + (i % 7) * 0.013f;
    h[0] = -0.0f; h[1] = 254.5f; h[2] = 255.0f; h[3] = 255.5f; h[4] = 0.5f; h[5] = 0.999f; h[6] = 1.5f; h[7] = 2.5f; h[8]=1e9f; h[9]=-1e9f; h[10]=254.999f;
    float *di; unsigned *dout;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(di, dout, n);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad_trunc = 0, bad_rne = 0;
    for (int i = 0; i < n; i++) {
        float c = h[i] < 0 ? 0 : (h[i] > 255 ? 255 : h[i]);
        unsigned t = (unsigned)c, r = (unsigned)nearbyintf(c);
        if (o[i] != t) bad_trunc++;
        if (o[i] != r) bad_rne++;
        if (i < 11) printf("%g -> %u (trunc %u rne %u)\n", h[i], o[i], t, r);
    }
    printf("mismatch vs clamp+trunc: %d, vs clamp+rne: %d of %d\n", bad_trunc, bad_rne, n);
    return 0;
}
