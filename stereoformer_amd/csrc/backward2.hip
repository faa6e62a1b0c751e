// Backward kernels, second file (SURVEY 8(f) N3 completed): A6 variance-based lookup, A2 difference / concat volumes.
// (A7's backward is a composition of A1's forward / backward kernels on the host side -- cross_attention.py.)
#include "common.h"

namespace {

inline unsigned grid_for(long long nthreads)
{
    long long blocks = (nthreads + 255) / 256;
    const long long cap = 256LL * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// ------------------------------------------------------------------ A6 backward
// Variance_Based_Local_Cost_Volume.forward (models/LocalCostVolume/Attempts/variance_local_cost_volume.py:34-78) as autograd
// differentiates it:
//   P = softmax_d(vol);  V = sum_d P_d (d - cur)^2;  sigma = sqrt(V);  lower/upper = cur -/+ gamma sigma
//   consider_valid: invalid pixels take every candidate = cur, valid ones c_k = lower + k (upper - lower) / S
//   else:           lower, upper clamped to [0, D-1] (clamp passes the gradient inside the closed interval)
//   new_k = vol[floor_k] (ceil_k - c_k) + vol[ceil_k] (1 - (ceil_k - c_k))      (indices are .long(): no gradient; the
//           rates carry d new_k / d c_k = vol[ceil_k] - vol[floor_k])
//   out = sum_k softmax_k(new)_k c_k
// Gradients (g = d loss / d out):
//   a_k = g p_k (c_k - out)            -> vol[floor_k] += a_k (ceil_k - c_k),  vol[ceil_k] += a_k (1 - (ceil_k - c_k))
//   dc_k = g p_k + a_k (vol[ceil_k] - vol[floor_k])
//   d cur (direct) = sum_k dc_k dc_k/dcur,   G_sigma = sum_k dc_k dc_k/dsigma
//   sigma path: dV/dcur = -2 (E[d] - cur),  dV/dvol_d = P_d ((d - cur)^2 - V),  dsigma/dV = 1 / (2 sigma)
//   (sigma == 0: the reference's sqrt backward yields inf * 0 = NaN; here that path contributes 0.)
// grad_vol must be zero-initialised by the caller; every lane owns its pixel column (no cross-lane collisions).
__global__ __launch_bounds__(256) void variance_lookup_bwd_kernel(const float *__restrict__ vol, const float *__restrict__ disp,
                                                                  const float *__restrict__ gout, float *__restrict__ gvol,
                                                                  float *__restrict__ gdisp, int D, int W, long long HW,
                                                                  long long npix, float gamma, int S, int consider_valid)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix; t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float *v = vol + (size_t)b * D * HW + p;
        float *gvp = gvol + (size_t)b * D * HW + p;
        const float cur = disp[t], g = gout[t];
        // softmax statistics of the column: max, sum, variance and mean around / of d
        float m0 = -INFINITY, s0 = 0.0f, var = 0.0f, ed = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float x = v[(size_t)d * HW];
            const float cm = fmaxf(m0, x);
            const float sc = (m0 == cm) ? 1.0f : fast_exp(m0 - cm);
            const float pe = fast_exp(x - cm);
            const float err = (float)d - cur;
            s0 = s0 * sc + pe;
            var = var * sc + pe * (err * err);
            ed = ed * sc + pe * (float)d;
            m0 = cm;
        }
        const float V = var / s0, sigma = sqrtf(V), meand = ed / s0;
        float lower = cur - gamma * sigma, upper = cur + gamma * sigma;
        int invalid = -1;
        float dlo = 1.0f, dup = 1.0f;  // clamp pass-through of lower / upper (non-valid mode)
        if (consider_valid) {
            invalid = (lower < 0.0f) || (upper >= (float)(D - 1)) || (upper > (float)(p % W));
        } else {
            if (lower < 0.0f) lower = 0.0f, dlo = 0.0f;
            else if (lower > (float)(D - 1)) lower = (float)(D - 1), dlo = 0.0f;
            if (upper < 0.0f) upper = 0.0f, dup = 0.0f;
            else if (upper > (float)(D - 1)) upper = (float)(D - 1), dup = 0.0f;
        }
        const float interval = __fdiv_rn(__fmul_rn(__fsub_rn(upper, lower), 1.0f), (float)S);
        const float inv = invalid > 0 ? 1.0f : 0.0f;
        float m = -INFINITY, s = 0.0f, e = 0.0f;
        for (int k = 0; k <= S; ++k) {
            float c = __fadd_rn(lower, __fmul_rn((float)k, interval));
            if (invalid >= 0) c = __fadd_rn(__fmul_rn(c, 1.0f - inv), __fmul_rn(inv, cur));
            const float val = sample_tap(v, (size_t)HW, D, c);
            const float cm = fmaxf(m, val);
            const float sc = (m == cm) ? 1.0f : fast_exp(m - cm);
            const float pe = fast_exp(val - cm);
            s = s * sc + pe;
            e = e * sc + pe * c;
            m = cm;
        }
        const float out = e / s, invs = 1.0f / s;
        float gd = 0.0f, gsig = 0.0f;
        int pend_idx = -1;
        float pend_val = 0.0f;
        auto push = [&](int idx, float val) {
            if (idx == pend_idx) {
                pend_val += val;
            } else {
                if (pend_idx >= 0) gvp[(size_t)pend_idx * HW] += pend_val;
                pend_idx = idx;
                pend_val = val;
            }
        };
        for (int k = 0; k <= S; ++k) {
            float c = __fadd_rn(lower, __fmul_rn((float)k, interval));
            if (invalid >= 0) c = __fadd_rn(__fmul_rn(c, 1.0f - inv), __fmul_rn(inv, cur));
            const float cc = clampf(ste_ceil_f(c), 0.0f, (float)(D - 1));
            const float cf = clampf(ste_floor_f(c), 0.0f, (float)(D - 1));
            const int ic = (int)cc, ifl = (int)cf;
            const float fr = (float)ic - c, cr = 1.0f - fr;
            const float vf = v[(size_t)ifl * HW], vc = v[(size_t)ic * HW];
            const float pk = fast_exp(vf * fr + vc * cr - m) * invs;
            const float a = g * pk * (c - out);
            push(ifl, a * fr);
            push(ic, a * cr);
            const float dc = g * pk + a * (vc - vf);
            const float tk = (float)k / (float)S;
            float dcd, dcs;  // d c_k / d cur, d c_k / d sigma
            if (consider_valid) {
                dcd = 1.0f;                                          // valid: lower + k interval; invalid: cur itself
                dcs = invalid > 0 ? 0.0f : gamma * (2.0f * tk - 1.0f);
            } else {
                dcd = (1.0f - tk) * dlo + tk * dup;
                dcs = gamma * (tk * dup - (1.0f - tk) * dlo);
            }
            gd = fmaf(dc, dcd, gd);
            gsig = fmaf(dc, dcs, gsig);
        }
        if (pend_idx >= 0) gvp[(size_t)pend_idx * HW] += pend_val;
        // sigma path
        const float gV = sigma > 0.0f ? gsig / (2.0f * sigma) : 0.0f;
        gdisp[t] = gd + gV * (-2.0f) * (meand - cur);
        if (gV != 0.0f) {
            const float is0 = 1.0f / s0;
            for (int d = 0; d < D; ++d) {
                const float pd = fast_exp(v[(size_t)d * HW] - m0) * is0;
                const float err = (float)d - cur;
                gvp[(size_t)d * HW] += gV * pd * (err * err - V);
            }
        }
    }
}

// ------------------------------------------------------------------ A2 backward
// 'difference': vol[b,c,d,h,w] = L[b,c,h,w] - R[b,c,h,w-d] (w >= d), else 0     (build_cost_volume.py:21-28)
//   gL[b,c,h,w] = sum_{d <= w} g[b,c,d,h,w],   gR[b,c,h,v] = -sum_{d, v+d < W} g[b,c,d,h,v+d]
// 'concat': vol[b,c,d] = L (w >= d), vol[b,C+c,d] = R(w-d) (w >= d)             (:30-37, submodule.py:30-41)
//   gL = sum_{d <= w} g[b,c,d,h,w],   gR[b,c,h,v] = sum_d g[b,C+c,d,h,v+d]
// One thread per (b,c,h,w); lanes along w, so every plane read is W-contiguous.  The 5-D gradient is read once per output
// (twice in total): the op is a copy-bound utility nobody trains through at scale.
template <bool CONCAT>
__global__ __launch_bounds__(256) void shift_volume_bwd_kernel(const float *__restrict__ g, float *__restrict__ gL,
                                                               float *__restrict__ gR, int C, int H, int W, int D,
                                                               long long n)
{
    const long long HW = (long long)H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(t % W);
        const long long r = t / W;          // (b * C + c) * H + h
        const int h = (int)(r % H);
        const long long bc = r / H;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        const int Cout = CONCAT ? 2 * C : C;
        const float *gl = g + (((size_t)b * Cout + c) * D) * HW + (size_t)h * W;
        const float *gr = g + (((size_t)b * Cout + (CONCAT ? C + c : c)) * D) * HW + (size_t)h * W;
        // eight planes per round with their loads in flight together and four partial sums (a one-plane-at-a-time loop is a
        // chain of dependent adds behind one load each: 2.0 TB/s); planes beyond a lane's own limit re-read its last valid
        // one (a cache hit) and are masked out of the sum
        if (gL) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int dmax = w < D - 1 ? w : D - 1;
            for (int d0 = 0; d0 <= dmax; d0 += 8) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = gl[(size_t)(d0 + j <= dmax ? d0 + j : dmax) * HW + w];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 3] += d0 + j <= dmax ? x[j] : 0.0f;
            }
            gL[t] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        if (gR) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int dmax = (W - 1 - w) < D - 1 ? (W - 1 - w) : D - 1;
            for (int d0 = 0; d0 <= dmax; d0 += 8) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int d = d0 + j <= dmax ? d0 + j : dmax;
                    x[j] = gr[(size_t)d * HW + w + d];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j & 3] += d0 + j <= dmax ? x[j] : 0.0f;
            }
            const float a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            gR[t] = CONCAT ? a : -a;
        }
    }
}

}  // namespace

extern "C" int stereo_variance_lookup_bwd_f32(const float *vol, const float *disp, const float *grad_out,
                                              float *grad_vol_zeroed, float *grad_disp, int B, int D, int H, int W,
                                              float gamma, int S, int consider_valid, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0 || S <= 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    if (D == 0) return STEREO_ERR_BAD_SHAPE;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(grad_out);
    STEREO_REQUIRE_PTR(grad_vol_zeroed);
    STEREO_REQUIRE_PTR(grad_disp);
    const long long HW = (long long)H * W, n = (long long)B * HW;
    hipLaunchKernelGGL(variance_lookup_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, disp, grad_out,
                       grad_vol_zeroed, grad_disp, D, W, HW, n, gamma, S, consider_valid);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

static int shift_bwd(bool concat, const float *grad_vol, float *grad_left, float *grad_right, int B, int C, int H, int W, int D,
                     void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || D < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || C == 0 || H == 0 || W == 0) return STEREO_OK;
    if (grad_left == nullptr && grad_right == nullptr) return STEREO_OK;
    if (D > 0) STEREO_REQUIRE_PTR(grad_vol);
    const long long n = (long long)B * C * H * W;
    if (concat)
        hipLaunchKernelGGL(shift_volume_bwd_kernel<true>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), grad_vol,
                           grad_left, grad_right, C, H, W, D, n);
    else
        hipLaunchKernelGGL(shift_volume_bwd_kernel<false>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), grad_vol,
                           grad_left, grad_right, C, H, W, D, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_diff_volume_bwd_f32(const float *grad_vol, float *grad_left, float *grad_right, int B, int C, int H,
                                          int W, int D, void *stream)
{
    return shift_bwd(false, grad_vol, grad_left, grad_right, B, C, H, W, D, stream);
}

extern "C" int stereo_concat_volume_bwd_f32(const float *grad_vol, float *grad_left, float *grad_right, int B, int C, int H,
                                            int W, int D, void *stream)
{
    return shift_bwd(true, grad_vol, grad_left, grad_right, B, C, H, W, D, stream);
}
