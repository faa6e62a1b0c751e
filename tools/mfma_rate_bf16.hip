// dev probe: sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip (4 waves per SIMD, four independent accumulators per wave):
// what "100 % MFMA-busy" means in TFLOP/s at the clock the part holds under that load.
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_rate_bf16.hip -o tools/bin/mfma_rate_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
__global__ __launch_bounds__(256) void k(float *out, int iters, float a)
{
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    bf8 av, bv;
    for (int e = 0; e < 8; ++e) av[e] = (__bf16)(a + e + threadIdx.x), bv[e] = (__bf16)(a * 0.5f - e);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out;
    const int nwg = 256 * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipMalloc(&out, nwg * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = rep == 0 ? 100 : 4000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)nwg * 4 * iters * 32;   // MFMAs
        printf("iters %d: %.3f ms, %.2f G MFMA/s, %.0f TFLOP/s bf16, implied clock at 32 cycles per MFMA and SIMD: %.2f GHz\n", iters, ms,
               n / ms / 1e6, n * 32768.0 / ms / 1e9, n * 32.0 / 1024 / ms / 1e6);
    }
    return 0;
}
