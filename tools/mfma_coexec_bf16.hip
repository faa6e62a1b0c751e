// Microbenchmark: do bf16 MFMA (v_mfma_f32_32x32x16_bf16) and ordinary VALU work overlap on a gfx950 SIMD?
//  mfma   : every wave runs a chain of dependent MFMAs
//  valu   : every wave runs independent v_fma_f32
//  split  : on every SIMD one wave runs the MFMA chain and one wave the VALU stream (different waves)
//  inter  : every wave interleaves 1 MFMA with NV VALU instructions (same wave, independent registers)
// If the pipes co-execute, split/inter take ~max(mfma, valu); if they serialise, ~sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void valu_block(float (&x)[16], float a)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], a, a);
}

template <int MODE>  // 0 mfma, 1 valu, 2 split by wave, 3 interleaved
__global__ __launch_bounds__(512) void k(float *out, int iters, float a, int shift)
{
    const int wave = threadIdx.x >> 6;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && ((wave >> shift) & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && ((wave >> shift) & 1) == 1);
    typedef __attribute__((ext_vector_type(8))) __bf16 bf8; bf8 av8, bv8; for (int e = 0; e < 8; ++e) { av8[e] = (__bf16)(a + e); bv8[e] = (__bf16)(a * 0.5f - e); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av8, bv8, acc, 0, 0, 0);
                valu_block(x, a);  // 16 VALU = 64 issue cycles, the length of one MFMA
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av8, bv8, acc, 0, 0, 0);
            }
            if (do_valu) {
#pragma unroll
                for (int s = 0; s < 8; ++s) valu_block(x, a);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[6] = {"mfma only (8 waves/CU)", "valu only (8 waves/CU)", "split: role = wave bit 0", "interleaved in every wave",
                            "split: role = wave bit 1", "split: role = wave bit 2"};
    for (int m6 = 0; m6 < 6; ++m6) {
        const int mode = m6 < 4 ? m6 : 2;
        const int shift = m6 == 2 ? 0 : m6 == 4 ? 1 : 2;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, shift);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, shift);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, shift);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, shift);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-36s %8.3f ms\n", names[m6], ms);
    }
    // reference points: per wave 8 bf16 MFMA x 32 cycles = 256 cycles/iter; 128 VALU x 4 = 512 cycles/iter
    return 0;
}
