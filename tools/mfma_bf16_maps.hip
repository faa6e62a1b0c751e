// Dev tool: verifies on the device the operand / result lane maps of v_mfma_f32_32x32x16_bf16 that the split-bf16 kernels
// rely on, with exact small-integer data (every value is exact in bf16 and in f32):
//   A[i][k]: lane l holds row i = l & 31, k = 8 (l >> 5) + j in element j of its 8-element fragment
//   B[k][j]: lane l holds column j = l & 31, k = 8 (l >> 5) + e in element e
//   D[i][j]: lane l holds column j = l & 31, rows i = (r & 3) + 8 (r >> 2) + 4 (l >> 5) in register r
// and the accumulator-as-operand k permutation: registers 8s .. 8s+7 of a 32x32 result X, packed pairwise to bf16, are the B
// fragment of k-step s of A2 . X when A2's element e of lane half h is column 16 s + 8 (e >> 2) + 4 h + (e & 3) of A2.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_bf16_maps.hip -o tools/mfma_bf16_maps.bin && tools/mfma_bf16_maps.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const float *A, const float *B, const float *A2, float *D, float *D2)
{
    const int l = threadIdx.x, r31 = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[r31 * 16 + 8 * h + e];   // A [32][16]
        b[e] = (__bf16)B[(8 * h + e) * 32 + r31]; // B [16][32]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r31] = c[r];
    // second product: D2 [32][32] = A2 [32][32] . X, X = c (32 x 32), contraction over X's rows in two k-steps
    f32x16 o = {0};
    for (int s = 0; s < 2; ++s) {
        bf16x8 xb, a2;
        for (int e = 0; e < 8; ++e) {
            xb[e] = (__bf16)c[8 * s + e];
            a2[e] = (__bf16)A2[r31 * 32 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)];
        }
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, o, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D2[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r31] = o[r];
}

int main()
{
    float hA[32 * 16], hB[16 * 32], hA2[32 * 32], hD[32 * 32], hD2[32 * 32], X[32 * 32];
    srand(1);
    for (auto &v : hA) v = (float)(rand() % 7 - 3);
    for (auto &v : hB) v = (float)(rand() % 5 - 2);
    for (auto &v : hA2) v = (float)(rand() % 3 - 1);
    float *dA, *dB, *dA2, *dD, *dD2;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dA2, sizeof hA2); hipMalloc(&dD, sizeof hD); hipMalloc(&dD2, sizeof hD2);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dA2, hA2, sizeof hA2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dA2, dD, dD2);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    hipMemcpy(hD2, dD2, sizeof hD2, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j];
            X[i * 32 + j] = s;
            bad += s != hD[i * 32 + j];
        }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int kk = 0; kk < 32; ++kk) s += hA2[i * 32 + kk] * X[kk * 32 + j];
            bad2 += s != hD2[i * 32 + j];
        }
    printf("mfma_f32_32x32x16_bf16 operand/result maps: %s (%d mismatches); accumulator-as-B-operand k permutation: %s (%d)\n",
           bad ? "WRONG" : "ok", bad, bad2 ? "WRONG" : "ok", bad2);
    return bad || bad2;
}
