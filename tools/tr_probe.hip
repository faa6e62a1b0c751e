// Dev probe: the lane permutation of ds_read_b64_tr_b16.  Lane j of the wave points at its own 8-byte chunk holding the four
// bf16 values 4 j .. 4 j + 3 (j < 64: exact); every lane prints which chunk elements it received.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(float *out)
{
    __shared__ __attribute__((aligned(16))) __bf16 img[64 * 4];
    for (int e = 0; e < 4; ++e) img[threadIdx.x * 4 + e] = (__bf16)(float)(threadIdx.x * 4 + e);
    __syncthreads();
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)&img[threadIdx.x * 4]);
    short ta[4] = {a[0], a[1], a[2], a[3]};
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (float)__builtin_bit_cast(__bf16, ta[e]);
}
int main()
{
    float *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  [lane %2d elem %d]", (int)h[l * 4 + e] / 4, (int)h[l * 4 + e] % 4);
        printf("\n");
    }
    return 0;
}
