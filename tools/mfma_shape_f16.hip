// dev probe: the same fp16 matrix work (random operand bits, registers only, 4 waves per SIMD) as v_mfma_f32_32x32x16_f16
// and as v_mfma_f32_16x16x32_f16 -- which shape does the chip hold the higher clock on (MI355X_MICROARCH.md, DVFS item 7)?
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_shape_f16.hip -o /tmp/mfma_shape_f16 && /tmp/mfma_shape_f16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) unsigned u4;

__device__ inline void load_frags(const unsigned *rnd, h8 (&av)[8], h8 (&bv)[8])
{
    for (int j = 0; j < 8; ++j) {
        u4 x = *reinterpret_cast<const u4 *>(rnd + ((threadIdx.x * 8 + j) * 8));
        u4 y = *reinterpret_cast<const u4 *>(rnd + ((threadIdx.x * 8 + j) * 8 + 4));
        for (int e = 0; e < 4; ++e) x[e] = (x[e] & 0x33ff33ffu) | 0x38003800u, y[e] = (y[e] & 0x33ff33ffu) | 0x38003800u;   // finite, O(1)
        av[j] = __builtin_bit_cast(h8, x);
        bv[j] = __builtin_bit_cast(h8, y);
    }
}
// 32 x 32 output tile per wave, K = 16 per step
__global__ __launch_bounds__(256) void k32(float *out, const unsigned *rnd, int iters)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    h8 av[8], bv[8];
    load_frags(rnd, av, bv);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[s & 7], bv[(s * 3) & 7], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= 1e-3f;
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same 32 x 32 tile as four 16 x 16 accumulators, K = 32 per step: 16 steps x 4 MFMAs = the same multiply-adds
__global__ __launch_bounds__(256) void k16(float *out, const unsigned *rnd, int iters)
{
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
    h8 av[8], bv[8];
    load_frags(rnd, av, bv);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[(s + (i >> 1)) & 7], bv[(s * 3 + (i & 1)) & 7], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] *= 1e-3f;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out;
    const int nwg = 256 * 4;
    hipMalloc(&out, nwg * 256 * 4);
    unsigned *rnd, *h = new unsigned[256 * 64];
    for (int i = 0; i < 256 * 64; ++i) h[i] = (unsigned)rand() * 2654435761u + (unsigned)rand();
    hipMalloc(&rnd, 256 * 64 * 4);
    hipMemcpy(rnd, h, 256 * 64 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int rep = 0; rep < 8; ++rep) {
        const bool s16 = rep & 1;
        hipEventRecord(e0);
        if (s16) hipLaunchKernelGGL(k16, dim3(nwg), dim3(256), 0, 0, out, rnd, iters);
        else hipLaunchKernelGGL(k32, dim3(nwg), dim3(256), 0, 0, out, rnd, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)nwg * 4 * iters * 32 * 32768.0;
        printf("%s: %.3f ms, %.0f TFLOP/s fp16 (random operand bits, registers only)\n", s16 ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9);
    }
    return 0;
}
