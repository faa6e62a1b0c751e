// dev probe: which offsets does the raw-buffer range check see on gfx950?   hipcc --offload-arch=gfx950 tools/buf_probe.hip -o /tmp/buf_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *p, float *o, int nrec, int soff)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, (short)0, nrec, 0x00020000);
    const int t = threadIdx.x;
    o[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (t - 4) * 4, soff, 0));
}
int main()
{
    float *p, *o, h[64], src[256];
    for (int i = 0; i < 256; ++i) src[i] = 1000.f + i;
    hipMalloc(&p, sizeof src);
    hipMalloc(&o, sizeof h);
    hipMemcpy(p, src, sizeof src, hipMemcpyHostToDevice);
    for (int soff : {0, 64 * 4}) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, p, o, 16 * 4, soff);   // 16 records; lanes 4..19 are in range by voffset
        hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
        printf("soff %d:", soff);
        for (int i = 0; i < 24; ++i) printf(" %g", h[i]);
        printf("\n");
    }
    return 0;
}
