// ConvGRU gate math (A11/A14) around the host framework's convolutions.
// Attempts/update.py:41-48 and update/update.py:28-32: pure streaming
// elementwise kernels, 16 B/lane when the element count allows.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return gate_sigmoid(x); }

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
                const float q = SOFTSIGN ? gate_softsign(qx) : gate_tanh(qx);
                o[i] = gate_blend(z, hv[i], q);
            }
            st4(out + t * 4, o);
        } else {
            const float z = sigmoidf_(zc[t] + (cz ? cz[t] : 0.0f));
            const float qx = qc[t] + (cq ? cq[t] : 0.0f);
            const float q = SOFTSIGN ? gate_softsign(qx) : gate_tanh(qx);
            out[t] = gate_blend(z, h[t], q);
        }
    }
}

// fp16 in / fp16 out variants for the update block under the reference's mixed_precision flag (fp16 convolutions):
// the gate math itself stays fp32 in registers; 8 halves (16 B) per lane.

template <int V>
__global__ __launch_bounds__(256) void gru_reset_f16_kernel(const _Float16 *__restrict__ h, const _Float16 *__restrict__ rc,
                                                            const _Float16 *__restrict__ cr, _Float16 *__restrict__ out,
                                                            long long nvec)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
         t += (long long)gridDim.x * blockDim.x) {
        if (V == 8) {
            const f16x8 hv = *reinterpret_cast<const f16x8 *>(h + t * 8), rv = *reinterpret_cast<const f16x8 *>(rc + t * 8);
            f16x8 bv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (cr) bv = *reinterpret_cast<const f16x8 *>(cr + t * 8);
            f16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (_Float16)(sigmoidf_((float)rv[i] + (float)bv[i]) * (float)hv[i]);
            *reinterpret_cast<f16x8 *>(out + t * 8) = o;
        } else {
            out[t] = (_Float16)(sigmoidf_((float)rc[t] + (cr ? (float)cr[t] : 0.0f)) * (float)h[t]);
        }
    }
}

template <int V, bool SOFTSIGN>
__global__ __launch_bounds__(256) void gru_blend_f16_kernel(const _Float16 *__restrict__ h, const _Float16 *__restrict__ zc,
                                                            const _Float16 *__restrict__ qc, const _Float16 *__restrict__ cz,
                                                            const _Float16 *__restrict__ cq, _Float16 *__restrict__ out,
                                                            long long nvec)
{
    auto one = [](float hv, float zv, float qv) {
        const float z = sigmoidf_(zv);
        const float q = SOFTSIGN ? gate_softsign(qv) : gate_tanh(qv);
        return gate_blend(z, hv, q);
    };
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
         t += (long long)gridDim.x * blockDim.x) {
        if (V == 8) {
            const f16x8 hv = *reinterpret_cast<const f16x8 *>(h + t * 8), zv = *reinterpret_cast<const f16x8 *>(zc + t * 8),
                        qv = *reinterpret_cast<const f16x8 *>(qc + t * 8);
            f16x8 bz = {0, 0, 0, 0, 0, 0, 0, 0}, bq = {0, 0, 0, 0, 0, 0, 0, 0};
            if (cz) bz = *reinterpret_cast<const f16x8 *>(cz + t * 8);
            if (cq) bq = *reinterpret_cast<const f16x8 *>(cq + t * 8);
            f16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i] = (_Float16)one((float)hv[i], (float)zv[i] + (float)bz[i], (float)qv[i] + (float)bq[i]);
            *reinterpret_cast<f16x8 *>(out + t * 8) = o;
        } else {
            out[t] = (_Float16)one((float)h[t], (float)zc[t] + (cz ? (float)cz[t] : 0.0f),
                                   (float)qc[t] + (cq ? (float)cq[t] : 0.0f));
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

extern "C" int stereo_gru_reset_f16(const void *h, const void *rc, const void *cr, void *out, long long n, void *stream)
{
    if (n < 0) return STEREO_ERR_BAD_SHAPE;
    if (n == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(h);
    STEREO_REQUIRE_PTR(rc);
    STEREO_REQUIRE_PTR(out);
    const _Float16 *hh = static_cast<const _Float16 *>(h), *rr = static_cast<const _Float16 *>(rc),
                   *cc = static_cast<const _Float16 *>(cr);
    _Float16 *oo = static_cast<_Float16 *>(out);
    if (n % 8 == 0 && aligned16(h) && aligned16(rc) && aligned16(out) && (cr == nullptr || aligned16(cr)))
        hipLaunchKernelGGL(gru_reset_f16_kernel<8>, dim3(grid_for(n / 8)), dim3(256), 0, as_stream(stream), hh, rr, cc, oo,
                           n / 8);
    else
        hipLaunchKernelGGL(gru_reset_f16_kernel<1>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), hh, rr, cc, oo, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_gru_blend_f16(const void *h, const void *zc, const void *qc, const void *cz, const void *cq, void *out,
                                    long long n, int softsign, void *stream)
{
    if (n < 0) return STEREO_ERR_BAD_SHAPE;
    if (n == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(h);
    STEREO_REQUIRE_PTR(zc);
    STEREO_REQUIRE_PTR(qc);
    STEREO_REQUIRE_PTR(out);
    hipStream_t st = as_stream(stream);
    const bool v8 = n % 8 == 0 && aligned16(h) && aligned16(zc) && aligned16(qc) && aligned16(out) &&
                    (cz == nullptr || aligned16(cz)) && (cq == nullptr || aligned16(cq));
    const long long nv = v8 ? n / 8 : n;
    const _Float16 *hh = static_cast<const _Float16 *>(h), *zz = static_cast<const _Float16 *>(zc),
                   *qq = static_cast<const _Float16 *>(qc), *bz = static_cast<const _Float16 *>(cz),
                   *bq = static_cast<const _Float16 *>(cq);
    _Float16 *oo = static_cast<_Float16 *>(out);
#define BLEND16(V, S) \
    hipLaunchKernelGGL((gru_blend_f16_kernel<V, S>), dim3(grid_for(nv)), dim3(256), 0, st, hh, zz, qq, bz, bq, oo, nv)
    if (v8 && softsign) BLEND16(8, true);
    else if (v8) BLEND16(8, false);
    else if (softsign) BLEND16(1, true);
    else BLEND16(1, false);
#undef BLEND16
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
