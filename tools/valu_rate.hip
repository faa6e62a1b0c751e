// Microbenchmark: fp32 FMA issue rate on gfx950 (scalar v_fmac_f32 vs v_pk_fma_f32), by waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float *out, int iters, float a)
{
    float acc[32];
    f2 acc2[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = (f2){(float)threadIdx.x, (float)i};
    f2 a2 = {a, a * 0.5f};
    for (int it = 0; it < iters; ++it) {
        if (PK) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[i] = __builtin_elementwise_fma(acc2[i], a2, a2);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = fmaf(acc[i], a, a);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc2[i][0] + acc2[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out;
    hipMalloc(&out, 256 * 1024 * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pk = 0; pk < 2; ++pk)
        for (int wps = 1; wps <= 4; ++wps) {
            const int threads = 256 * wps;  // wps waves per SIMD, 1 block per CU
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f);
                else hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double inst = (double)iters * (pk ? 64 : 128);           // wave-instructions per wave
            const double fma = inst * 64 * (pk ? 2 : 1) * 256 * 4 * wps;   // total FMAs
            printf("pk=%d waves/SIMD=%d: %.3f ms, %.2f TFLOP/s, %.2f ns per wave-instr per SIMD\n", pk, wps, ms,
                   2 * fma / ms / 1e9, ms * 1e6 / (inst * wps));
        }
    return 0;
}
