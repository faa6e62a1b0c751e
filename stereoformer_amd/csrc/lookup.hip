// Local cost-volume re-sampling (A5 fixed radius, A6 variance based, A11
// learned bounds), the raw sampler, and the A11 uncertainty volume.
//
// One lane = one pixel (consecutive lanes = consecutive w, so every volume tap
// of a wave touches at most a few 128-B lines per D plane).  The S+1
// candidates are produced, tapped (two clamped planes + lerp) and folded into
// an online softmax-expectation in registers; nothing but the final disparity
// is written.
#include "common.h"

namespace {

struct Softmax1 {
    float m, s;
    __device__ __forceinline__ void init() { m = -INFINITY, s = 0.0f; }
    __device__ __forceinline__ float push(float x)  // returns the rescale applied to earlier terms
    {
        const float cm = fmaxf(m, x);
        const float sc = (m == cm) ? 1.0f : fast_exp(m - cm);
        s = s * sc + fast_exp(x - cm);
        m = cm;
        return sc;
    }
};

// candidates -> taps -> softmax -> expectation
// (fixed_local_cost_volume.py:33-56 == variance_local_cost_volume.py:56-72 == Attempts/update.py:297-313)
// invalid: -1 = consider_valid off, 0/1 = the pixel's invalid mask
__device__ __forceinline__ float lookup_pixel(const float *__restrict__ v, size_t HW, int D, float cur, float lower,
                                              float upper, int S, int invalid)
{
    const float interval = __fdiv_rn(__fmul_rn(__fsub_rn(upper, lower), 1.0f), (float)S);
    const float inv = invalid > 0 ? 1.0f : 0.0f;
    float m = -INFINITY, s = 0.0f, e = 0.0f;
    for (int k = 0; k <= S; ++k) {
        float c = __fadd_rn(lower, __fmul_rn((float)k, interval));
        if (invalid >= 0) c = __fadd_rn(__fmul_rn(c, 1.0f - inv), __fmul_rn(inv, cur));  // update.py:304
        const float val = sample_tap(v, HW, D, c);
        const float cm = fmaxf(m, val);
        const float sc = (m == cm) ? 1.0f : fast_exp(m - cm);
        const float p = fast_exp(val - cm);
        s = s * sc + p;
        e = e * sc + p * c;
        m = cm;
    }
    return e / s;
}

enum { MODE_FIXED = 0, MODE_VARIANCE = 1, MODE_BOUNDS = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void lookup_kernel(const float *__restrict__ vol, const float *__restrict__ disp,
                                                     const float *__restrict__ bounds, float *__restrict__ out, int D,
                                                     int W, long long HW, long long npix, float param, int S,
                                                     int consider_valid)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float *v = vol + (size_t)b * D * HW + p;
        const float cur = disp[t];
        float lower, upper;
        int invalid = -1;
        if (MODE == MODE_FIXED) {
            lower = cur - param;
            upper = cur + param;
        } else if (MODE == MODE_BOUNDS) {
            lower = cur - bounds[(size_t)b * 2 * HW + p];
            upper = cur + bounds[(size_t)b * 2 * HW + HW + p];
        } else {
            // GetVarince (variance_local_cost_volume.py:12-25): sigma of softmax(vol) around cur
            float m = -INFINITY, s = 0.0f, var = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float x = v[(size_t)d * HW];
                const float cm = fmaxf(m, x);
                const float sc = (m == cm) ? 1.0f : fast_exp(m - cm);
                const float pe = fast_exp(x - cm);
                const float err = (float)d - cur;
                s = s * sc + pe;
                var = var * sc + pe * (err * err);
                m = cm;
            }
            const float sigma = sqrtf(var / s);
            lower = cur - param * sigma;
            upper = cur + param * sigma;
        }
        if (consider_valid) {
            invalid = (lower < 0.0f) || (upper >= (float)(D - 1));
            if (MODE == MODE_VARIANCE) invalid = invalid || (upper > (float)(p % W));  // :44,48
        } else if (MODE == MODE_VARIANCE) {
            lower = clampf(lower, 0.0f, (float)(D - 1));  // :52-53
            upper = clampf(upper, 0.0f, (float)(D - 1));
        } else {
            lower = lower < 0.0f ? 0.0f : lower;       // fixed_local_cost_volume.py:29
            upper = clampf(upper, 0.0f, (float)D);     // :30 (max = D)
        }
        out[t] = lookup_pixel(v, (size_t)HW, D, cur, lower, upper, S, invalid);
    }
}

// A6 with 4 pixels per lane: the softmax-variance pass streams the volume with 16-byte loads
// (one wave load = 1 KiB along W) exactly like soft-argmin; the four lookups follow in registers.
__global__ __launch_bounds__(256) void variance_lookup_vec_kernel(const float *__restrict__ vol,
                                                                  const float *__restrict__ disp,
                                                                  float *__restrict__ out, int D, int W, long long HW,
                                                                  long long nquads, float gamma, int S,
                                                                  int consider_valid)
{
    const long long HWq = HW / 4;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nquads;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HWq, p = (t - b * HWq) * 4;
        const float *v = vol + (size_t)b * D * HW + p;
        const f32x4 cur = ld4(disp + (size_t)b * HW + p);
        float m[4], s[4], var[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = -INFINITY, s[i] = 0.0f, var[i] = 0.0f;
        constexpr int U = 8;  // planes in flight per lane; one running-max update (and one rescale exp) per batch
        for (int d0 = 0; d0 < D; d0 += U) {
            f32x4 x[U];
            const int nv = (D - d0) < U ? (D - d0) : U;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u < nv) x[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(v + (size_t)(d0 + u) * HW));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float cm = m[i];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u < nv) cm = fmaxf(cm, x[u][i]);
                const float sc = (m[i] == cm) ? 1.0f : fast_exp(m[i] - cm);
                float ss = s[i] * sc, vv = var[i] * sc;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u < nv) {
                        const float pe = fast_exp(x[u][i] - cm);
                        const float err = (float)(d0 + u) - cur[i];
                        ss += pe;
                        vv = fmaf(pe, err * err, vv);
                    }
                m[i] = cm, s[i] = ss, var[i] = vv;
            }
        }
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sigma = sqrtf(var[i] / s[i]);
            float lower = cur[i] - gamma * sigma, upper = cur[i] + gamma * sigma;
            int invalid = -1;
            if (consider_valid) {
                invalid = (lower < 0.0f) || (upper >= (float)(D - 1)) || (upper > (float)((p + i) % W));
            } else {
                lower = clampf(lower, 0.0f, (float)(D - 1));
                upper = clampf(upper, 0.0f, (float)(D - 1));
            }
            o[i] = lookup_pixel(v + i, (size_t)HW, D, cur[i], lower, upper, S, invalid);
        }
        st4(out + (size_t)b * HW + p, o);
    }
}

__global__ __launch_bounds__(256) void sample_volume_kernel(const float *__restrict__ vol,
                                                            const float *__restrict__ cand, float *__restrict__ out,
                                                            int D, int K, long long HW, long long n)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x) {
        const long long p = t % HW, b = t / (HW * K);
        out[t] = sample_tap(vol + (size_t)b * D * HW + p, (size_t)HW, D, cand[t]);
    }
}

// U[b,d,p] = softmax_d(vol)[d] * d * (d - disp)^2   (Attempts/update.py:71-72,86)
__global__ __launch_bounds__(256) void uncertainty_kernel(const float *__restrict__ vol,
                                                          const float *__restrict__ disp, float *__restrict__ out,
                                                          int D, long long HW, long long npix)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float *v = vol + (size_t)b * D * HW + p;
        float *o = out + (size_t)b * D * HW + p;
        const float cur = disp[t];
        Softmax1 sm;
        sm.init();
        for (int d = 0; d < D; ++d) sm.push(v[(size_t)d * HW]);
        const float inv = 1.0f / sm.s;
        for (int d = 0; d < D; ++d) {
            const float pd = fast_exp(v[(size_t)d * HW] - sm.m) * inv;
            const float e = (float)d - cur;
            o[(size_t)d * HW] = pd * (float)d * e * e;
        }
    }
}

// The inputs of BasicGuidanceNet's two encoders (Attempts/update.py:75-90) in ONE pass, channels-last and zero-padded to the
// channel counts the convolution kernel takes (multiples of 16): err[b,p,c] = disp_warp(right, disp)[c] - left[c] (border
// padding, :83-84) for c < Ci, unc[b,p,d] = softmax_d(vol)[d] * d * (d - disp)^2 (:86) for d < D; the padding channels are 0.
// Same arithmetic as disp_warp_kernel / uncertainty_kernel (shared device functions).
__global__ __launch_bounds__(256) void guidance_inputs_nhwc_kernel(const float *__restrict__ vol, const float *__restrict__ disp,
                                                                   const float *__restrict__ left, const float *__restrict__ right,
                                                                   float *__restrict__ err, float *__restrict__ unc, int D, int Ci,
                                                                   int H, int W, int CE, int CU, long long npix)
{
    const long long HW = (long long)H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix; t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float cur = disp[t];
        {   // ---- error map
            float ux, uy;
            disp_warp_coords(x, y, cur, W, H, ux, uy);
            const float sx = clampf(ux, 0.0f, (float)(W - 1)), sy = clampf(uy, 0.0f, (float)(H - 1));
            const float fx = floorf(sx), fy = floorf(sy);
            const float ax = sx - fx, ay = sy - fy;
            const int ix = (int)fx, iy = (int)fy;
            const bool x0 = ix >= 0 && ix <= W - 1, x1 = ix + 1 >= 0 && ix + 1 <= W - 1;
            const bool y0 = iy >= 0 && iy <= H - 1, y1 = iy + 1 >= 0 && iy + 1 <= H - 1;
            const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
            float *eo = err + (size_t)t * CE;
            auto err_c = [&](int c) {
                float v = 0.0f;
                if (c < Ci) {
                    const float *im = right + ((size_t)b * Ci + c) * HW;
                    float s = 0.0f;
                    if (y0 && x0) s += im[(size_t)iy * W + ix] * w00;
                    if (y0 && x1) s += im[(size_t)iy * W + ix + 1] * w01;
                    if (y1 && x0) s += im[(size_t)(iy + 1) * W + ix] * w10;
                    if (y1 && x1) s += im[(size_t)(iy + 1) * W + ix + 1] * w11;
                    v = s - left[((size_t)b * Ci + c) * HW + p];
                }
                return v;
            };
            if ((CE & 3) == 0) {
                for (int c = 0; c < CE; c += 4) st4(eo + c, (f32x4){err_c(c), err_c(c + 1), err_c(c + 2), err_c(c + 3)});
            } else {
                for (int c = 0; c < CE; ++c) eo[c] = err_c(c);
            }
        }
        {   // ---- uncertainty volume
            const float *v = vol + (size_t)b * D * HW + p;
            float *uo = unc + (size_t)t * CU;
            if (D <= 32 && CU <= 32 && (CU & 3) == 0) {
                // the column in registers: every plane loaded once, all loads in flight (the two-pass loop below is 2 D dependent
                // rounds), the 32 padded channels leave as eight 16-byte stores instead of 32 scattered dwords
                float xs[32];
#pragma unroll
                for (int d = 0; d < 32; ++d) xs[d] = d < D ? v[(size_t)d * HW] : -INFINITY;
                Softmax1 sm;   // (the same online recurrence as the generic path and uncertainty_kernel: bit-identical sums)
                sm.init();
#pragma unroll
                for (int d = 0; d < 32; ++d)
                    if (d < D) sm.push(xs[d]);
                const float inv = 1.0f / sm.s;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (4 * q < CU) {
                        f32x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int d = 4 * q + i;
                            const float e = (float)d - cur;
                            float val = 0.0f;
                            if (d < D) {
                                const float pd = fast_exp(xs[d] - sm.m) * inv;
                                val = pd * (float)d * e * e;
                            }
                            o[i] = val;
                        }
                        st4(uo + 4 * q, o);
                    }
                }
            } else {
                Softmax1 sm;
                sm.init();
                for (int d = 0; d < D; ++d) sm.push(v[(size_t)d * HW]);
                const float inv = 1.0f / sm.s;
                for (int d = 0; d < CU; ++d) {
                    float o = 0.0f;
                    if (d < D) {
                        const float pd = fast_exp(v[(size_t)d * HW] - sm.m) * inv;
                        const float e = (float)d - cur;
                        o = pd * (float)d * e * e;
                    }
                    uo[d] = o;
                }
            }
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

template <int MODE>
int lookup_dispatch(const float *vol, const float *disp, const float *bounds, float *out, int B, int D, int H, int W,
                    float param, int S, int consider_valid, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0 || S <= 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(out);
    if (MODE == MODE_BOUNDS) STEREO_REQUIRE_PTR(bounds);
    if (D == 0) return STEREO_ERR_BAD_SHAPE;
    const long long HW = (long long)H * W, n = (long long)B * HW;
    if (MODE == MODE_VARIANCE && HW % 4 == 0 && aligned16(vol) && aligned16(disp) && aligned16(out)) {
        hipLaunchKernelGGL(variance_lookup_vec_kernel, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(stream), vol, disp,
                           out, D, W, HW, n / 4, param, S, consider_valid);
        STEREO_LAUNCH_CHECK();
        return STEREO_OK;
    }
    hipLaunchKernelGGL(lookup_kernel<MODE>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, disp, bounds, out,
                       D, W, HW, n, param, S, consider_valid);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

extern "C" int stereo_local_lookup_f32(const float *vol, const float *disp, float *out, int B, int D, int H, int W,
                                       float radius, int S, int consider_valid, void *stream)
{
    return lookup_dispatch<MODE_FIXED>(vol, disp, nullptr, out, B, D, H, W, radius, S, consider_valid, stream);
}

extern "C" int stereo_variance_lookup_f32(const float *vol, const float *disp, float *out, int B, int D, int H, int W,
                                          float gamma, int S, int consider_valid, void *stream)
{
    return lookup_dispatch<MODE_VARIANCE>(vol, disp, nullptr, out, B, D, H, W, gamma, S, consider_valid, stream);
}

extern "C" int stereo_bounds_lookup_f32(const float *vol, const float *disp, const float *bounds, float *out, int B,
                                        int D, int H, int W, int S, int consider_valid, void *stream)
{
    return lookup_dispatch<MODE_BOUNDS>(vol, disp, bounds, out, B, D, H, W, 0.0f, S, consider_valid, stream);
}

extern "C" int stereo_sample_volume_f32(const float *vol, const float *cand, float *out, int B, int D, int H, int W,
                                        int K, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0 || K < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0 || K == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(cand);
    STEREO_REQUIRE_PTR(out);
    if (D == 0) return STEREO_ERR_BAD_SHAPE;
    const long long HW = (long long)H * W, n = (long long)B * K * HW;
    hipLaunchKernelGGL(sample_volume_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, cand, out, D, K,
                       HW, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_uncertainty_volume_f32(const float *vol, const float *disp, float *out, int B, int D, int H,
                                             int W, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0 || D == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(out);
    const long long HW = (long long)H * W, n = (long long)B * HW;
    hipLaunchKernelGGL(uncertainty_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, disp, out, D, HW,
                       n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_guidance_inputs_nhwc_f32(const float *vol, const float *disp, const float *left, const float *right, float *err,
                                               float *unc, int B, int D, int Ci, int H, int W, int err_channels, int unc_channels,
                                               void *stream)
{
    if (B < 0 || D < 0 || Ci < 0 || H < 0 || W < 0 || err_channels < Ci || unc_channels < D) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(left);
    STEREO_REQUIRE_PTR(right);
    STEREO_REQUIRE_PTR(err);
    STEREO_REQUIRE_PTR(unc);
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(guidance_inputs_nhwc_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, disp, left, right, err,
                       unc, D, Ci, H, W, err_channels, unc_channels, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
