// dev probe: sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip (4 waves per SIMD, four independent accumulators per wave):
// what "100 % MFMA-busy" means in TFLOP/s at the clock the part holds under that load -- with 4, 2 and 1 (fully dependent) chains per wave.
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_rate_bf16.hip -o tools/bin/mfma_rate_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
// the same with RANDOM operand bits (8 different fragments per operand cycling): the switching activity of real data -- the
// constant operands above are the part's best case for power, and the clock it holds depends on it
__global__ __launch_bounds__(256) void krand(float *out, const unsigned *rnd, int iters)
{
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    bf8 av[8], bv[8];
    for (int j = 0; j < 8; ++j) {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        u4 x = *reinterpret_cast<const u4 *>(rnd + ((threadIdx.x * 8 + j) * 8));
        u4 y = *reinterpret_cast<const u4 *>(rnd + ((threadIdx.x * 8 + j) * 8 + 4));
        for (int e = 0; e < 4; ++e) x[e] = (x[e] & 0x3fff3fffu) | 0x3c003c00u, y[e] = (y[e] & 0x3fff3fffu) | 0x3c003c00u;   // finite, O(1)
        av[j] = __builtin_bit_cast(bf8, x);
        bv[j] = __builtin_bit_cast(bf8, y);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s & 7], bv[(s * 3) & 7], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= 1e-3f;   // keep the accumulator finite
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a)
{
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    bf8 av, bv;
    for (int e = 0; e < 8; ++e) av[e] = (__bf16)(a + e + threadIdx.x), bv[e] = (__bf16)(a * 0.5f - e);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32 / NACC; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
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
    for (int rep = 0; rep < 7; ++rep) {
        const int iters = rep == 0 ? 100 : 4000;
        const int nacc = rep < 3 ? 4 : (rep < 5 ? 2 : 1);   // independent accumulator chains per wave
        hipEventRecord(e0);
        if (nacc == 4) hipLaunchKernelGGL(k<4>, dim3(nwg), dim3(256), 0, 0, out, iters, 1.0f);
        else if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 0, 0, out, iters, 1.0f);
        else hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)nwg * 4 * iters * 32;   // MFMAs
        printf("chains/wave %d, iters %d: %.3f ms, %.2f G MFMA/s, %.0f TFLOP/s bf16, implied clock at 32 cycles per MFMA and SIMD: %.2f GHz\n", nacc, iters, ms,
               n / ms / 1e6, n * 32768.0 / ms / 1e9, n * 32.0 / 1024 / ms / 1e6);
    }
    unsigned *rnd, h[256 * 64];
    for (auto &v : h) v = (unsigned)rand() * 2654435761u + (unsigned)rand();
    hipMalloc(&rnd, sizeof h);
    hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 4000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(krand, dim3(nwg), dim3(256), 0, 0, out, rnd, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)nwg * 4 * iters * 32;
        printf("random operands, 1 chain/wave: %.3f ms, %.2f G MFMA/s, %.0f TFLOP/s bf16, implied clock %.2f GHz\n", ms, n / ms / 1e6,
               n * 32768.0 / ms / 1e9, n * 32.0 / 1024 / ms / 1e6);
    }
    return 0;
}
