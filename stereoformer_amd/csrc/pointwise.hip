// ConvGRU gate math (A11/A14) around the host framework's convolutions.
// Attempts/update.py:41-48 and update/update.py:28-32: pure streaming
// elementwise kernels, 16 B/lane when the element count allows.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int V>
__global__ __launch_bounds__(256) void gru_reset_kernel(const float *__restrict__ h, const float *__restrict__ rc,
                                                        const float *__restrict__ cr, float *__restrict__ out,
                                                        long long nvec)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
         t += (long long)gridDim.x * blockDim.x) {
        if (V == 4) {  // 16 B per lane on every stream
            const f32x4 hv = ld4(h + t * 4), rv = ld4(rc + t * 4);
            const f32x4 bv = cr ? ld4(cr + t * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = sigmoidf_(rv[i] + bv[i]) * hv[i];
            st4(out + t * 4, o);
        } else {
            const float r = sigmoidf_(rc[t] + (cr ? cr[t] : 0.0f));
            out[t] = r * h[t];
        }
    }
}

template <int V, bool SOFTSIGN>
__global__ __launch_bounds__(256) void gru_blend_kernel(const float *__restrict__ h, const float *__restrict__ zc,
                                                        const float *__restrict__ qc, const float *__restrict__ cz,
                                                        const float *__restrict__ cq, float *__restrict__ out,
                                                        long long nvec)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
         t += (long long)gridDim.x * blockDim.x) {
        if (V == 4) {
            const f32x4 hv = ld4(h + t * 4), zv = ld4(zc + t * 4), qv = ld4(qc + t * 4);
            const f32x4 bz = cz ? ld4(cz + t * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 bq = cq ? ld4(cq + t * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float z = sigmoidf_(zv[i] + bz[i]);
                const float qx = qv[i] + bq[i];
                const float q = SOFTSIGN ? qx / (1.0f + fabsf(qx)) : tanhf(qx);
                o[i] = (1.0f - z) * hv[i] + z * q;
            }
            st4(out + t * 4, o);
        } else {
            const float z = sigmoidf_(zc[t] + (cz ? cz[t] : 0.0f));
            const float qx = qc[t] + (cq ? cq[t] : 0.0f);
            const float q = SOFTSIGN ? qx / (1.0f + fabsf(qx)) : tanhf(qx);
            out[t] = (1.0f - z) * h[t] + z * q;
        }
    }
}

inline unsigned grid_for(long long nthreads)
{
    long long blocks = (nthreads + 255) / 256;
    const long long cap = 256LL * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int stereo_gru_reset_f32(const float *h, const float *rc, const float *cr, float *out, long long n,
                                    void *stream)
{
    if (n < 0) return STEREO_ERR_BAD_SHAPE;
    if (n == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(h);
    STEREO_REQUIRE_PTR(rc);
    STEREO_REQUIRE_PTR(out);
    if (n % 4 == 0 && aligned16(h) && aligned16(rc) && aligned16(out) && (cr == nullptr || aligned16(cr)))
        hipLaunchKernelGGL(gru_reset_kernel<4>, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(stream), h, rc, cr, out,
                           n / 4);
    else
        hipLaunchKernelGGL(gru_reset_kernel<1>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), h, rc, cr, out, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_gru_blend_f32(const float *h, const float *zc, const float *qc, const float *cz, const float *cq,
                                    float *out, long long n, int softsign, void *stream)
{
    if (n < 0) return STEREO_ERR_BAD_SHAPE;
    if (n == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(h);
    STEREO_REQUIRE_PTR(zc);
    STEREO_REQUIRE_PTR(qc);
    STEREO_REQUIRE_PTR(out);
    hipStream_t st = as_stream(stream);
    const bool v4 = n % 4 == 0 && aligned16(h) && aligned16(zc) && aligned16(qc) && aligned16(out) &&
                    (cz == nullptr || aligned16(cz)) && (cq == nullptr || aligned16(cq));
    const long long nv = v4 ? n / 4 : n;
#define BLEND(V, S) hipLaunchKernelGGL((gru_blend_kernel<V, S>), dim3(grid_for(nv)), dim3(256), 0, st, h, zc, qc, cz, cq, out, nv)
    if (v4 && softsign) BLEND(4, true);
    else if (v4) BLEND(4, false);
    else if (softsign) BLEND(1, true);
    else BLEND(1, false);
#undef BLEND
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
