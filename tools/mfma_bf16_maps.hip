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
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Third check: the A operand (rows = keys, k = channels) gathered from a [channel][key] bf16 image in LDS with the transposed
// read ds_read_b64_tr_b16: per 16-lane group g, block rows (channels) R0 .. R0+3, columns (keys) K0 .. K0+15; lane 4q+p of the
// group supplies the address of row R0+q, columns K0+4p ..; lane i receives column K0+i, rows R0 .. R0+3.
//   S^T[key][query] = sum_c K[c][key] Q[c][query], 32 keys x 32 queries x 32 channels (two k-steps)
constexpr int RS = 72;  // image row stride in elements (64 keys + 8 pad)
__global__ void ktr(const float *K, const float *Q, float *D3, int keybase)
{
    __shared__ __attribute__((aligned(16))) __bf16 img[32 * RS];
    for (int i = threadIdx.x; i < 32 * 64; i += 64) img[(i / 64) * RS + (i % 64)] = (__bf16)K[i];   // K [32 ch][64 keys]
    __syncthreads();
    const int l = threadIdx.x, r31 = l & 31, h = l >> 5, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    f32x16 c = {0};
    for (int s = 0; s < 2; ++s) {
        bf16x8 a, b;
        for (int t = 0; t < 2; ++t) {
            const __bf16 *src = &img[(16 * s + 8 * h + 4 * t + q) * RS + keybase + 16 * (g & 1) + 4 * p];
            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
            const short tv[4] = {v[0], v[1], v[2], v[3]};   // (indexing the builtin's result vector directly miscompiled)
            for (int e = 0; e < 4; ++e) a[4 * t + e] = __builtin_bit_cast(__bf16, tv[e]);
        }
        for (int e = 0; e < 8; ++e) b[e] = (__bf16)Q[(16 * s + 8 * h + e) * 32 + r31];   // Q [32 ch][32 queries]
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D3[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r31] = c[r];
}

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
    {   // transposed-read check
        static float hK[32 * 64], hQ[32 * 32], hD3[32 * 32];
        for (auto &v : hK) v = (float)(rand() % 7 - 3);
        for (auto &v : hQ) v = (float)(rand() % 5 - 2);
        float *dK, *dQ, *dD3;
        hipMalloc(&dK, sizeof hK); hipMalloc(&dQ, sizeof hQ); hipMalloc(&dD3, sizeof hD3);
        hipMemcpy(dK, hK, sizeof hK, hipMemcpyHostToDevice);
        hipMemcpy(dQ, hQ, sizeof hQ, hipMemcpyHostToDevice);
        int bad3 = 0;
        for (int keybase = 0; keybase <= 32; keybase += 16) {
            hipLaunchKernelGGL(ktr, dim3(1), dim3(64), 0, 0, dK, dQ, dD3, keybase);
            hipMemcpy(hD3, dD3, sizeof hD3, hipMemcpyDeviceToHost);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    float sacc = 0;
                    for (int cc = 0; cc < 32; ++cc) sacc += hK[cc * 64 + keybase + i] * hQ[cc * 32 + j];
                    bad3 += sacc != hD3[i * 32 + j];
                }
        }
        printf("A operand from a [channel][key] image via ds_read_b64_tr_b16: %s (%d mismatches)\n", bad3 ? "WRONG" : "ok", bad3);
        if (bad3) return 1;
    }
    printf("mfma_f32_32x32x16_bf16 operand/result maps: %s (%d mismatches); accumulator-as-B-operand k permutation: %s (%d)\n",
           bad ? "WRONG" : "ok", bad, bad2 ? "WRONG" : "ok", bad2);
    return bad || bad2;
}
