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

// Dense single-launch form (end of round 3): the gradient of the volume is DENSE in d (the variance path reaches every plane),
// so the thread-per-pixel kernel above -- a serial walk down the column, a second one for the variance path, and `+=` into a
// zero-filled tensor -- moves 2.0 GB for 0.8 GB of algorithmic bytes at cfg-2 size.  Here a workgroup takes PIX = 64 pixels:
// the column of a pixel quad is split over NPH = 16 lanes' registers (J values each, soft_argmin_bwd_regs_kernel's layout),
// the softmax statistics are reduced through LDS, ONE wave then runs the candidate loops of the 64 pixels (all taps of a
// pixel loaded together: cache hits on planes the workgroup has just read) and pushes the tap gradients into a private
// [D][PIX] LDS image, and every thread writes its own planes once: variance term from its registers + the pushed taps.
// No zero-fill, no read-modify-write.  Envelope: D <= 16 J, H*W % 4 == 0, S <= SMAX.
template <int J, int SMAX>
__global__ __launch_bounds__(256, 3) void variance_lookup_bwd_dense_kernel(const float *__restrict__ vol, const float *__restrict__ disp,
                                                                        const float *__restrict__ gout, float *__restrict__ gvol,
                                                                        float *__restrict__ gdisp, int D, int W, long long HW,
                                                                        float gamma, int S, int consider_valid)
{
    constexpr int NWV = 4, SPLIT = 4, QPW = 64 / SPLIT, NPH = NWV * SPLIT, PIX = QPW * 4;
    extern __shared__ float vsm[];
    float *G = vsm;                          // [max(D, 4 NPH)][PIX] pushed tap gradients ...
    float *red = vsm;                        // ... and, before that, [4][NPH][PIX] partial statistics (50 KB: three workgroups per CU)
    float *par = vsm + (size_t)(D > 4 * NPH ? D : 4 * NPH) * PIX;   // [5][PIX] m0, s0, V, E[d], gV of a pixel
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ph = wave * SPLIT + lane / QPW, ql = lane % QPW;
    const long long p = ((long long)blockIdx.x * QPW + ql) * 4;
    const bool okp = p < HW;                 // HW % 4 == 0: a quad is entirely in or out
    const size_t boff = (size_t)blockIdx.y * D * HW + (okp ? p : 0);
    const float *v = vol + boff;
    f32x4 x[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = ph + NPH * j;
        x[j] = (okp && d < D) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(v + (size_t)d * HW))
                              : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    const f32x4 cur4 = okp ? ld4(disp + (size_t)blockIdx.y * HW + p) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float mm = -INFINITY;
#pragma unroll
        for (int j = 0; j < J; ++j) mm = fmaxf(mm, x[j][i]);
        red[(0 * NPH + ph) * PIX + ql * 4 + i] = mm;
    }
    __syncthreads();
    float m0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float mm = red[ql * 4 + i];
#pragma unroll
        for (int k = 1; k < NPH; ++k) mm = fmaxf(mm, red[k * PIX + ql * 4 + i]);
        m0[i] = (mm == -INFINITY) ? 0.0f : mm;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ss = 0.0f, vv = 0.0f, ee = 0.0f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float e = fast_exp(x[j][i] - m0[i]);   // -inf -> 0
            const float dd = (float)(ph + NPH * j), err = dd - cur4[i];
            x[j][i] = e;
            ss += e;
            vv = fmaf(e, err * err, vv);
            ee = fmaf(e, dd, ee);
        }
        red[(1 * NPH + ph) * PIX + ql * 4 + i] = ss;
        red[(2 * NPH + ph) * PIX + ql * 4 + i] = vv;
        red[(3 * NPH + ph) * PIX + ql * 4 + i] = ee;
    }
    __syncthreads();
    float s0[4], V[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ss = 0.0f, vv = 0.0f, ee = 0.0f;
#pragma unroll
        for (int k = 0; k < NPH; ++k) {
            ss += red[(1 * NPH + k) * PIX + ql * 4 + i];
            vv += red[(2 * NPH + k) * PIX + ql * 4 + i];
            ee += red[(3 * NPH + k) * PIX + ql * 4 + i];
        }
        s0[i] = ss, V[i] = vv / ss;
        if (ph == 0) {
            par[0 * PIX + ql * 4 + i] = m0[i];
            par[1 * PIX + ql * 4 + i] = ss;
            par[2 * PIX + ql * 4 + i] = V[i];
            par[3 * PIX + ql * 4 + i] = ee / ss;
        }
    }
    __syncthreads();                         // the statistics have been read: their memory becomes the (zeroed) push image
    for (int i = tid; i < D * PIX / 4; i += 256) reinterpret_cast<f32x4 *>(G)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // ---- candidates: wave 0, lane = pixel of the workgroup
    if (wave == 0) {
        const int pi = lane;                                   // pixel (ql = pi / 4, i = pi & 3)
        const long long pp = (long long)blockIdx.x * PIX + pi;
        if (pp < HW) {
            const size_t t = (size_t)blockIdx.y * HW + pp;
            const float *vp = vol + (size_t)blockIdx.y * D * HW + pp;
            const float cur = disp[t], g = gout[t];
            const float Vp = par[2 * PIX + pi], sigma = sqrtf(Vp), meand = par[3 * PIX + pi];
            float lower = cur - gamma * sigma, upper = cur + gamma * sigma;
            int invalid = -1;
            float dlo = 1.0f, dup = 1.0f;
            if (consider_valid) {
                invalid = (lower < 0.0f) || (upper >= (float)(D - 1)) || (upper > (float)(pp % W));
            } else {
                if (lower < 0.0f) lower = 0.0f, dlo = 0.0f;
                else if (lower > (float)(D - 1)) lower = (float)(D - 1), dlo = 0.0f;
                if (upper < 0.0f) upper = 0.0f, dup = 0.0f;
                else if (upper > (float)(D - 1)) upper = (float)(D - 1), dup = 0.0f;
            }
            const float interval = __fdiv_rn(__fmul_rn(__fsub_rn(upper, lower), 1.0f), (float)S);
            const float inv = invalid > 0 ? 1.0f : 0.0f;
            // candidate k: position, tap indices, floor rate -- recomputed wherever needed (a handful of instructions) instead
            // of kept: what stays in registers over the three passes are the two tap values per candidate
            auto cand = [&](int k, float &c, int &ic, int &ifl) {
                c = __fadd_rn(lower, __fmul_rn((float)(k <= S ? k : S), interval));
                if (invalid >= 0) c = __fadd_rn(__fmul_rn(c, 1.0f - inv), __fmul_rn(inv, cur));
                ic = (int)clampf(ste_ceil_f(c), 0.0f, (float)(D - 1));
                ifl = (int)clampf(ste_floor_f(c), 0.0f, (float)(D - 1));
            };
            // two passes over the candidates in batches of KB (the taps of a batch in flight together; the second pass re-reads
            // them: cache hits) -- keeping all 2 (S + 1) taps in registers next to the column slice spilled
            constexpr int KB = 7, NB = (SMAX + KB) / KB;
            float m = -INFINITY, ssum = 0.0f, e = 0.0f;
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                float va[KB], vb[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    float c;
                    int ic, ifl;
                    cand(b * KB + u, c, ic, ifl);
                    va[u] = vp[(size_t)ifl * HW];
                    vb[u] = vp[(size_t)ic * HW];
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int k = b * KB + u;
                    if (k <= S) {
                        float c;
                        int ic, ifl;
                        cand(k, c, ic, ifl);
                        const float fr = (float)ic - c;
                        const float val = va[u] * fr + vb[u] * (1.0f - fr);
                        const float cm = fmaxf(m, val);
                        const float sc = (m == cm) ? 1.0f : fast_exp(m - cm);
                        const float pe = fast_exp(val - cm);
                        ssum = ssum * sc + pe;
                        e = e * sc + pe * c;
                        m = cm;
                    }
                }
            }
            const float invs = 1.0f / ssum, out = e * invs;
            float gd = 0.0f, gsig = 0.0f;
            float *Gp = G + pi;
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                float va[KB], vb[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    float c;
                    int ic, ifl;
                    cand(b * KB + u, c, ic, ifl);
                    va[u] = vp[(size_t)ifl * HW];
                    vb[u] = vp[(size_t)ic * HW];
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int k = b * KB + u;
                    if (k <= S) {
                        float c;
                        int ic, ifl;
                        cand(k, c, ic, ifl);
                        const float fr = (float)ic - c, cr = 1.0f - fr;
                        const float pk = fast_exp(va[u] * fr + vb[u] * cr - m) * invs;
                        const float a = g * pk * (c - out);
                        Gp[ifl * PIX] += a * fr;               // (this lane's column: no other lane touches it)
                        Gp[ic * PIX] += a * cr;
                        const float dc = g * pk + a * (vb[u] - va[u]);
                        const float tk = (float)k / (float)S;
                        float dcd, dcs;
                        if (consider_valid) {
                            dcd = 1.0f;
                            dcs = invalid > 0 ? 0.0f : gamma * (2.0f * tk - 1.0f);
                        } else {
                            dcd = (1.0f - tk) * dlo + tk * dup;
                            dcs = gamma * (tk * dup - (1.0f - tk) * dlo);
                        }
                        gd = fmaf(dc, dcd, gd);
                        gsig = fmaf(dc, dcs, gsig);
                    }
                }
            }
            const float gV = sigma > 0.0f ? gsig / (2.0f * sigma) : 0.0f;
            gdisp[t] = gd + gV * (-2.0f) * (meand - cur);
            par[4 * PIX + pi] = gV;
        } else {
            par[4 * PIX + pi] = 0.0f;
        }
    }
    __syncthreads();
    if (!okp) return;
    float *o = gvol + boff;
    float gV[4], is0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gV[i] = par[4 * PIX + ql * 4 + i], is0[i] = 1.0f / s0[i];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = ph + NPH * j;
        if (d < D) {
            f32x4 q = *reinterpret_cast<const f32x4 *>(&G[(size_t)d * PIX + ql * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float err = (float)d - cur4[i];
                q[i] = fmaf(gV[i] * (x[j][i] * is0[i]), err * err - V[i], q[i]);
            }
            __builtin_nontemporal_store(q, reinterpret_cast<f32x4 *>(o + (size_t)d * HW));
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

// The dense form: grad_vol need NOT be zero-filled (every element is written once).  D <= 192, H*W % 4 == 0, S <= 20 and
// 16-byte aligned tensors, else STEREO_ERR_UNSUPPORTED (the caller zero-fills and takes stereo_variance_lookup_bwd_f32).
extern "C" int stereo_variance_lookup_bwd_dense_f32(const float *vol, const float *disp, const float *grad_out, float *grad_vol,
                                                    float *grad_disp, int B, int D, int H, int W, float gamma, int S,
                                                    int consider_valid, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0 || S <= 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    if (D == 0) return STEREO_ERR_BAD_SHAPE;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(grad_out);
    STEREO_REQUIRE_PTR(grad_vol);
    STEREO_REQUIRE_PTR(grad_disp);
    const long long HW = (long long)H * W;
    if (D > 192 || S > 20 || HW % 4 != 0 || !aligned16(vol) || !aligned16(disp) || !aligned16(grad_vol) || !fits_grid_yz(B))
        return STEREO_ERR_UNSUPPORTED;
    constexpr int PIX = 64, NPH = 16;
    const size_t lds = ((size_t)(D > 4 * NPH ? D : 4 * NPH) * PIX + 5 * PIX) * sizeof(float);
    const long long nblk = (HW / 4 + 15) / 16;
    if (nblk >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)nblk, (unsigned)B);
#define VLB(JV)                                                                                                         \
    do {                                                                                                                \
        auto kern = variance_lookup_bwd_dense_kernel<JV, 20>;                                                           \
        static std::atomic<unsigned long long> lds_done{0};                                                             \
        if (!lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, as_stream(stream), vol, disp, grad_out, grad_vol, grad_disp, D, W, HW,  \
                           gamma, S, consider_valid);                                                                   \
    } while (0)
    if (D <= 32) VLB(2);
    else if (D <= 64) VLB(4);
    else if (D <= 128) VLB(8);
    else VLB(12);
#undef VLB
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
