// Backward kernels for the cfg-2 chain: soft-argmin, correlation / group-wise volume,
// local cost-volume lookup.  Same conventions as the forward kernels: fp32, NCHW, lanes along W.

#include "common.h"

namespace {

inline unsigned grid_for(long long nthreads)
{
    long long blocks = (nthreads + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > (1LL << 30)) blocks = 1LL << 30;
    return (unsigned)blocks;
}

// ------------------------------------------------------------------ soft-argmin backward
// disp = sum_d p_d d, p = softmax(sgn*vol):  d disp / d vol_d = sgn * p_d * (d - disp).
// With a probability cotangent gp (DisparityEstimationWithProb): cot_d = g*d + gp_d,
//   d L / d vol_d = sgn * p_d * (cot_d - sum_k p_k cot_k).
// One lane = 4 pixels; pass 1 (online softmax) reads the column once, pass 2 re-reads it (cache-warm) and
// writes the gradient with 16-byte stores.
template <int V, bool HAS_GP>
__global__ __launch_bounds__(256) void soft_argmin_bwd_kernel(const float *__restrict__ vol,
                                                              const float *__restrict__ gdisp,
                                                              const float *__restrict__ gprob,
                                                              float *__restrict__ gvol, int D, long long HW,
                                                              long long ncols, float sgn)
{
    const long long HWv = HW / V;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < ncols;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HWv, p = (t - b * HWv) * V;
        const float *v = vol + (size_t)b * D * HW + p;
        const float *gp = HAS_GP ? gprob + (size_t)b * D * HW + p : nullptr;
        float m[V], s[V], e[V], g[V];
#pragma unroll
        for (int i = 0; i < V; ++i) m[i] = -INFINITY, s[i] = 0.0f, e[i] = 0.0f, g[i] = gdisp[(size_t)b * HW + p + i];
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float x = sgn * v[(size_t)d * HW + i];
                const float cot = g[i] * (float)d + (HAS_GP ? gp[(size_t)d * HW + i] : 0.0f);
                const float cm = fmaxf(m[i], x);
                const float sc = (m[i] == cm) ? 1.0f : fast_exp(m[i] - cm);
                const float pe = fast_exp(x - cm);
                s[i] = s[i] * sc + pe;
                e[i] = e[i] * sc + pe * cot;
                m[i] = cm;
            }
        }
        float inv[V], mean_cot[V];
#pragma unroll
        for (int i = 0; i < V; ++i) inv[i] = 1.0f / s[i], mean_cot[i] = e[i] * inv[i];
        float *o = gvol + (size_t)b * D * HW + p;
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float pd = fast_exp(sgn * v[(size_t)d * HW + i] - m[i]) * inv[i];
                const float cot = g[i] * (float)d + (HAS_GP ? gp[(size_t)d * HW + i] : 0.0f);
                o[(size_t)d * HW + i] = sgn * pd * (cot - mean_cot[i]);
            }
        }
    }
}

// The same in ONE pass over the volume (no probability cotangent, D <= NWV * SPLIT * J): regress.hip's register-resident
// softmax column (soft_argmin_prob_regs_kernel) with the gradient as the final write -- a workgroup of NWV waves owns
// 64 / SPLIT pixel quads, the D planes are dealt round-robin to NWV * SPLIT phases, every lane keeps its J planes of one quad
// with all J loads in flight, max and sums meet in LDS (two barriers):  the volume is read once and the gradient written
// once (the two-pass kernel above reads it twice: 1.93 ms at cfg-2 against 1.22 ms for the forward with probabilities).
// ADD: the gradient is ADDED to a second cotangent of the same volume (`addend`, e.g. what a lookup on the volume sent
// back) on its way out -- the sum autograd would otherwise form with a separate read-read-write pass over the volume.
template <int J, int NWV, int SPLIT, bool ADD>
__global__ __launch_bounds__(NWV * 64) void soft_argmin_bwd_regs_kernel(const float *__restrict__ vol,
                                                                        const float *__restrict__ gdisp,
                                                                        const float *__restrict__ addend,
                                                                        float *__restrict__ gvol, int D, long long HW, float sgn)
{
    constexpr int QPW = 64 / SPLIT, NPH = NWV * SPLIT;
    __shared__ float red[3][NPH][QPW * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ph = wave * SPLIT + lane / QPW, ql = lane % QPW;
    const long long p = ((long long)blockIdx.x * QPW + ql) * 4;
    const bool okp = p < HW;  // HW % 4 == 0: a quad is entirely in or out
    const size_t boff = (size_t)blockIdx.y * D * HW + (okp ? p : 0);
    const float *v = vol + boff;
    f32x4 x[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = ph + NPH * j;
        if (okp && d < D) {
            const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(v + (size_t)d * HW));
#pragma unroll
            for (int i = 0; i < 4; ++i) x[j][i] = sgn * r[i];
        } else {
            x[j] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
    }
    const f32x4 g = okp ? ld4(gdisp + (size_t)blockIdx.y * HW + p) : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float mm = -INFINITY;
#pragma unroll
        for (int j = 0; j < J; ++j) mm = fmaxf(mm, x[j][i]);
        red[0][ph][ql * 4 + i] = mm;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float mm = red[0][0][ql * 4 + i];
#pragma unroll
        for (int k = 1; k < NPH; ++k) mm = fmaxf(mm, red[0][k][ql * 4 + i]);
        m[i] = (mm == -INFINITY) ? 0.0f : mm;  // out-of-range quad: keep the arithmetic finite
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ss = 0.0f, ee = 0.0f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float e = fast_exp(x[j][i] - m[i]);  // -inf -> 0
            x[j][i] = e;
            ss += e;
            ee = fmaf(e, (float)(ph + NPH * j), ee);
        }
        red[1][ph][ql * 4 + i] = ss;
        red[2][ph][ql * 4 + i] = ee;
    }
    __syncthreads();
    float sc[4], dm[4];   // sgn * g / S and the disparity E / S
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float S = 0.0f, E = 0.0f;
#pragma unroll
        for (int k = 0; k < NPH; ++k) S += red[1][k][ql * 4 + i], E += red[2][k][ql * 4 + i];
        const float inv = 1.0f / S;
        dm[i] = E * inv;
        sc[i] = sgn * g[i] * inv;
    }
    if (!okp) return;
    float *o = gvol + boff;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = ph + NPH * j;
        if (d < D) {
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (ADD) q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(addend + boff + (size_t)d * HW));
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = fmaf(x[j][i] * sc[i], (float)d - dm[i], q[i]);   // sgn p_d g (d - disp)
            st4(o + (size_t)d * HW, q);
        }
    }
}

// ------------------------------------------------------------------ volume backward
// vol[b,g,d,h,w] = (1/CPG) sum_{c in g} L[c,w] R[c,w-d]  (w >= d)
//   dL[c,w] = (1/CPG) sum_{d <= w} gv[g,d,w]   * R[c,w-d]
//   dR[c,v] = (1/CPG) sum_{v+d < W} gv[g,d,v+d] * L[c,v+d]
// One workgroup = (row, 256-pixel tile, batch*group); the D disparities are split over the 8 waves in
// chunks of 8.  Each lane (4 pixels) keeps CB x 4 accumulators for ALL CB <= 32 channels of the block, so
// the gradient volume -- the big tensor -- is read exactly once per output (16-byte W-contiguous loads;
// the diagonal access of dR uses two aligned quads per plane with compile-time lane selection).  The other
// feature row sits in an LDS tile with halo (left halo for dL, right halo for dR) and is read as a sliding
// register window of ds_read_b128, like the forward kernel.  The 8 partial sums meet in LDS, 8 channels a round.
constexpr int BT = 256, BNW = 8, BDC = 8, BRC = 8;

template <bool WRT_LEFT, int CB>
__global__ __launch_bounds__(BNW * 64) void volume_bwd_kernel(const float *__restrict__ F,  // the OTHER feature map
                                                              const float *__restrict__ gv,
                                                              float *__restrict__ gF, int C, int H, int W, int D,
                                                              int G, int CPG, int Dpad, float inv_cpg, int exact_div)
{
    extern __shared__ f32x4 smem4[];
    float *Fs = reinterpret_cast<float *>(smem4);  // [CB][BT + Dpad]
    const int FS = BT + Dpad;
    const int x0 = blockIdx.x * BT, h = blockIdx.y;
    const int b = blockIdx.z / G, g = blockIdx.z - b * G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xl = lane * 4;
    const size_t HW = (size_t)H * W;
    const float *gvrow = gv + (((size_t)b * G + g) * D) * HW + (size_t)h * W;
    const int f0 = WRT_LEFT ? x0 - Dpad : x0;      // global w of LDS column 0
    const bool vec = (W % 4 == 0);

    for (int c0 = 0; c0 < CPG; c0 += CB) {
        const int ncc = (CPG - c0) < CB ? (CPG - c0) : CB;
        const float *Frow = F + ((size_t)b * C + (size_t)g * CPG + c0) * HW + (size_t)h * W;
        float *grow = gF + ((size_t)b * C + (size_t)g * CPG + c0) * HW + (size_t)h * W;
        __syncthreads();
        for (int idx = tid; idx < CB * (FS / 4); idx += BNW * 64) {
            const int c = idx / (FS / 4), x = (idx - c * (FS / 4)) * 4;
            const int w = f0 + x;
            f32x4 q;
            if (vec && c < ncc && w >= 0 && w + 3 < W) {
                q = ld4(Frow + (size_t)c * HW + w);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = (c < ncc && w + i >= 0 && w + i < W) ? Frow[(size_t)c * HW + w + i] : 0.0f;
            }
            *reinterpret_cast<f32x4 *>(Fs + c * FS + x) = q;
        }
        __syncthreads();
        float acc[CB][4];
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][i] = 0.0f;
        const int nchunks = (D + BDC - 1) / BDC;
        for (int ch = wave; ch < nchunks; ch += BNW) {
            const int d0 = ch * BDC;
            // ---- gradient quads for the 8 planes of this chunk: gq[dd][i] = gv[d0+dd][x + i (+ d for dR)]
            float gq[BDC][4];
#pragma unroll
            for (int dd = 0; dd < BDC; ++dd) {
                const int d = d0 + dd;
                const int x = x0 + xl;
                if (WRT_LEFT) {
                    f32x4 q = {0.f, 0.f, 0.f, 0.f};
                    if (d < D) {
                        if (vec && x + 3 < W) {
                            q = ld4(gvrow + (size_t)d * HW + x);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) q[i] = (x + i < W) ? gvrow[(size_t)d * HW + x + i] : 0.0f;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) gq[dd][i] = (x + i >= d) ? q[i] : 0.0f;  // zero-filled region: no grad
                } else {
                    // need gv[d][x + d + i], i = 0..3: aligned base x + d0 + 4*(dd/4), in-quad shift dd % 4
                    const int wb = x + d0 + 4 * (dd / 4);
                    f32x4 qa = {0.f, 0.f, 0.f, 0.f}, qb = {0.f, 0.f, 0.f, 0.f};
                    if (d < D) {
                        if (vec && wb + 7 < W) {
                            qa = ld4(gvrow + (size_t)d * HW + wb);
                            if (dd % 4) qb = ld4(gvrow + (size_t)d * HW + wb + 4);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                qa[i] = (wb + i < W) ? gvrow[(size_t)d * HW + wb + i] : 0.0f;
                                qb[i] = (wb + 4 + i < W) ? gvrow[(size_t)d * HW + wb + 4 + i] : 0.0f;
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = (dd % 4) + i;  // compile-time
                        gq[dd][i] = (x + i < W) ? (k < 4 ? qa[k] : qb[k - 4]) : 0.0f;
                    }
                }
            }
            // ---- feature window: dL taps R[x+i-d]: columns Dpad+xl-d0-8 .. +11 ; dR taps L[x+i+d]: xl+d0 .. +11
            const float *fwin = Fs + (WRT_LEFT ? (Dpad + xl - d0 - BDC) : (xl + d0));
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                if (c < ncc) {
                    float win[12];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const f32x4 t = *reinterpret_cast<const f32x4 *>(fwin + c * FS + 4 * q);
                        win[4 * q + 0] = t[0];
                        win[4 * q + 1] = t[1];
                        win[4 * q + 2] = t[2];
                        win[4 * q + 3] = t[3];
                    }
#pragma unroll
                    for (int dd = 0; dd < BDC; ++dd)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[c][i] = fmaf(gq[dd][i], win[WRT_LEFT ? (BDC + i - dd) : (i + dd)], acc[c][i]);
                }
            }
        }
        // ---- reduce the 8 waves' partial sums through LDS, BRC channels a round (tile no longer needed)
        float *red = Fs;  // [BNW][BRC][BT]
#pragma unroll
        for (int r0 = 0; r0 < CB; r0 += BRC) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < BRC; ++c)
                if (r0 + c < CB) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[(wave * BRC + c) * BT + xl + i] = acc[r0 + c][i];
                }
            __syncthreads();
            for (int idx = tid; idx < BRC * BT; idx += BNW * 64) {
                const int c = idx / BT, x = idx - c * BT;
                if (r0 + c < ncc && x0 + x < W) {
                    float s = 0.0f;
#pragma unroll
                    for (int k = 0; k < BNW; ++k) s += red[(k * BRC + c) * BT + x];
                    grow[(size_t)(r0 + c) * HW + x0 + x] = exact_div ? s / (float)CPG : s * inv_cpg;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ local lookup backward (A5)
// out = sum_k p_k c_k, p = softmax_k(val_k), val_k = v[fl_k]*fr_k + v[ce_k]*cr_k,
// fr_k = ce_k - c_k, cr_k = 1 - fr_k (fixed_local_cost_volume.py:61-80; .long() cuts the STE path, so the
// tap indices carry no gradient while fr/cr do).  With a_k = g * p_k * (c_k - out):
//   d/d vol[fl_k] += a_k * fr_k ;  d/d vol[ce_k] += a_k * cr_k
//   d/d c_k = g * p_k + a_k * (v[ce_k] - v[fl_k])
//   d c_k / d disp: consider_valid -> 1 (valid: both bounds move with disp; invalid: c_k = disp);
//                   otherwise (1-k/S)*[lower not clamped] + (k/S)*[upper not clamped].
// grad_vol must be zero-initialised by the caller; every lane owns its pixel column, so there are no
// cross-lane collisions and the candidates' taps (monotone in k) are merged in registers before the store.
__global__ __launch_bounds__(256) void local_lookup_bwd_kernel(const float *__restrict__ vol,
                                                               const float *__restrict__ disp,
                                                               const float *__restrict__ gout,
                                                               float *__restrict__ gvol, float *__restrict__ gdisp,
                                                               int D, long long HW, long long npix, float radius,
                                                               int S, int consider_valid)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float *v = vol + (size_t)b * D * HW + p;
        float *gvp = gvol + (size_t)b * D * HW + p;
        const float cur = disp[t], g = gout[t];
        float lower = cur - radius, upper = cur + radius;
        int invalid = -1;
        float dl = 1.0f, du = 1.0f;  // d lower / d disp, d upper / d disp
        if (consider_valid) {
            invalid = (lower < 0.0f) || (upper >= (float)(D - 1));
        } else {
            if (lower < 0.0f) lower = 0.0f, dl = 0.0f;
            if (upper < 0.0f) upper = 0.0f, du = 0.0f;
            else if (upper > (float)D) upper = (float)D, du = 0.0f;
        }
        const float interval = __fdiv_rn(__fmul_rn(__fsub_rn(upper, lower), 1.0f), (float)S);
        const float inv = invalid > 0 ? 1.0f : 0.0f;
        // pass 1: softmax statistics and the expectation
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
        // pass 2: gradients
        float gd = 0.0f;
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
            float dcd;
            if (consider_valid) {
                dcd = 1.0f;
            } else {
                const float tk = (float)k / (float)S;
                dcd = (1.0f - tk) * dl + tk * du;
            }
            gd = fmaf(dc, dcd, gd);
        }
        if (pend_idx >= 0) gvp[(size_t)pend_idx * HW] += pend_val;
        gdisp[t] = gd;
    }
}

// The same with the gradient volume written DENSELY by the kernel itself (no zero-fill pass, no read-modify-write): a pixel's
// candidates touch a window of at most 2 radius + 3 consecutive disparity planes, so a workgroup (256 pixels of one image)
// first does the per-pixel math as above with the pushes going to a private 16-entry LDS window (thread = pixel), then writes
// ALL D planes of its 256 columns -- the window values where a plane falls inside a pixel's window, zeros elsewhere -- as
// 16-byte stores (thread = pixel quad, one plane per wave and step): the 3.19 GB gradient volume of cfg-2 is written once
// at the store rate instead of memset + scattered updates.  radius <= 6 (window <= 16), HW % 4 == 0.
constexpr int LLB_WIN = 16;
__global__ __launch_bounds__(256) void local_lookup_bwd_dense_kernel(const float *__restrict__ vol,
                                                                     const float *__restrict__ disp,
                                                                     const float *__restrict__ gout,
                                                                     float *__restrict__ gvol, float *__restrict__ gdisp,
                                                                     int D, long long HW, float radius, int S, int consider_valid,
                                                                     int nplanes)
{
    __shared__ float win[256][LLB_WIN + 1];    // gradient window of a pixel
    __shared__ float vwin[256][LLB_WIN + 1];   // the volume's planes base .. base + 15 at the pixel
    __shared__ int wbase[256];
    const long long p0 = (long long)blockIdx.x * 256, b = blockIdx.y;
    const int tid = threadIdx.x;
    {
        const long long p = p0 + tid;
#pragma unroll
        for (int j = 0; j < LLB_WIN; ++j) win[tid][j] = 0.0f;
        int base = 0;
        if (p < HW) {
            const long long t = b * HW + p;
            const float *v = vol + (size_t)b * D * HW + p;
            const float cur = disp[t], g = gout[t];
            float lower = cur - radius, upper = cur + radius;
            int invalid = -1;
            float dl = 1.0f, du = 1.0f;  // d lower / d disp, d upper / d disp
            if (consider_valid) {
                invalid = (lower < 0.0f) || (upper >= (float)(D - 1));
            } else {
                if (lower < 0.0f) lower = 0.0f, dl = 0.0f;
                if (upper < 0.0f) upper = 0.0f, du = 0.0f;
                else if (upper > (float)D) upper = (float)D, du = 0.0f;
            }
            const float interval = __fdiv_rn(__fmul_rn(__fsub_rn(upper, lower), 1.0f), (float)S);
            const float inv = invalid > 0 ? 1.0f : 0.0f;
            auto cand = [&](int k) {
                float c = __fadd_rn(lower, __fmul_rn((float)k, interval));
                if (invalid >= 0) c = __fadd_rn(__fmul_rn(c, 1.0f - inv), __fmul_rn(inv, cur));
                return c;
            };
            {   // candidates (hence taps) are monotone in k: the first one's lower tap is the window's first plane
                const float c0 = cand(0);
                const int ic = (int)clampf(ste_ceil_f(c0), 0.0f, (float)(D - 1)), ifl = (int)clampf(ste_floor_f(c0), 0.0f, (float)(D - 1));
                base = (ifl < ic ? ifl : ic) - 1;   // one plane of slack: the straight-through floor, (floor(x) - x) + x, can land
                base = base < 0 ? 0 : base;         // one below floor(x) after the int cast
            }
            // every plane a tap of this pixel can touch, fetched ONCE with all loads in flight (the S + 1 candidates sample
            // 2 (S + 1) taps out of these <= 2 radius + 3 planes; fetched tap by tap, candidate after candidate, the rounds
            // of dependent loads held the workgroup's dense write back: 0.94 ms at cfg-2 size, 0.59 of it the write)
            {
                float pv[LLB_WIN];
#pragma unroll
                for (int j = 0; j < LLB_WIN; ++j) {
                    const int d = base + j < D ? base + j : D - 1;
                    pv[j] = j < nplanes ? v[(size_t)d * HW] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < LLB_WIN; ++j) vwin[tid][j] = pv[j];
            }
            auto tap = [&](int idx) {
                const int j = idx - base;
                return (j >= 0 && j < LLB_WIN) ? vwin[tid][j] : v[(size_t)idx * HW];   // inside the window by construction
            };
            float m = -INFINITY, s = 0.0f, e = 0.0f;
            for (int k = 0; k <= S; ++k) {
                const float c = cand(k);
                const float cc = clampf(ste_ceil_f(c), 0.0f, (float)(D - 1));
                const float cf = clampf(ste_floor_f(c), 0.0f, (float)(D - 1));
                const int ic = (int)cc, ifl = (int)cf;
                const float fr = (float)ic - c, cr = 1.0f - fr;
                const float val = tap(ifl) * fr + tap(ic) * cr;
                const float cm = fmaxf(m, val);
                const float sc = (m == cm) ? 1.0f : fast_exp(m - cm);
                const float pe = fast_exp(val - cm);
                s = s * sc + pe;
                e = e * sc + pe * c;
                m = cm;
            }
            const float out = e / s, invs = 1.0f / s;
            float gd = 0.0f;
            for (int k = 0; k <= S; ++k) {
                const float c = cand(k);
                const float cc = clampf(ste_ceil_f(c), 0.0f, (float)(D - 1));
                const float cf = clampf(ste_floor_f(c), 0.0f, (float)(D - 1));
                const int ic = (int)cc, ifl = (int)cf;
                const float fr = (float)ic - c, cr = 1.0f - fr;
                const float vf = tap(ifl), vc = tap(ic);
                const float pk = fast_exp(vf * fr + vc * cr - m) * invs;
                const float a = g * pk * (c - out);
                const int jf = ifl - base, jc = ic - base;
                if (jf >= 0 && jf < LLB_WIN) win[tid][jf] += a * fr;
                if (jc >= 0 && jc < LLB_WIN) win[tid][jc] += a * cr;
                const float dc = g * pk + a * (vc - vf);
                float dcd;
                if (consider_valid) {
                    dcd = 1.0f;
                } else {
                    const float tk = (float)k / (float)S;
                    dcd = (1.0f - tk) * dl + tk * du;
                }
                gd = fmaf(dc, dcd, gd);
            }
            gdisp[t] = gd;
        }
        wbase[tid] = base;
    }
    __syncthreads();
    // dense write: wave w writes planes w, w + 4, ...; lane l the pixel quad 4 l .. 4 l + 3 of the block
    const int wave = tid >> 6, lane = tid & 63;
    const long long pq = p0 + 4 * lane;
    if (pq < HW) {
        int bs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bs[i] = wbase[4 * lane + i];
        float *o = gvol + (size_t)b * D * HW + pq;
        for (int d = wave; d < D; d += 4) {
            f32x4 q;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = d - bs[i];
                q[i] = (j >= 0 && j < LLB_WIN) ? win[4 * lane + i][j] : 0.0f;
            }
            st4(o + (size_t)d * HW, q);
        }
    }
}

// ------------------------------------------------------------------ convex upsampling backward (A13)
// out[c, f*y+i, f*x+j] = sum_k p_k(i,j,y,x) * f * disp[c, y+k/3-1, x+k%3-1],  p = softmax_k(mask[k,i,j,y,x])
//   d mask[k,i,j,y,x] = sum_c go_c * p_k * (f*nb_k,c - out_c)
//   d disp[c, y', x'] = f * sum over the 9 coarse pixels (y,x) that see (y',x') as neighbour k, and all (i,j):
//                       go[c, f*y+i, f*x+j] * p_k(i,j,y,x)
// One workgroup = 64 coarse pixels of one coarse row: the f x 64f slab of the output gradient is staged in LDS
// with W-contiguous loads, mask planes are read and mask gradients written with lanes along x, and the
// per-pixel 3x3 neighbour contributions are combined in LDS before one float atomic per touched coarse pixel
// (grad_disp must be zero-filled by the caller).
template <int F, int CDT>
__global__ __launch_bounds__(256) void upsample_convex_bwd_kernel(const float *__restrict__ disp,
                                                                  const float *__restrict__ mask,
                                                                  const float *__restrict__ gout,
                                                                  float *__restrict__ gdisp, float *__restrict__ gmask,
                                                                  int Cd, int H, int W)
{
    constexpr int F2 = F * F, XT = 64;
    __shared__ float nb[CDT][3][XT + 2];
    __shared__ float gt[CDT][F][XT * F + 4];   // output-gradient slab
    __shared__ float dacc[CDT][3][XT + 2];     // d disp contributions of this workgroup
    const int x0 = blockIdx.x * XT, y = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x;
    const size_t HW = (size_t)H * W;
    const int xl = tid % XT, sub0 = tid / XT;
    const int x = x0 + xl;
    const int rowlen = ((W - x0) < XT ? (W - x0) : XT) * F;

    for (int c0 = 0; c0 < Cd; c0 += CDT) {
        const int nc = (Cd - c0) < CDT ? (Cd - c0) : CDT;
        __syncthreads();
        for (int idx = tid; idx < nc * 3 * (XT + 2); idx += 256) {
            const int c = idx / (3 * (XT + 2)), rem = idx - c * 3 * (XT + 2);
            const int r = rem / (XT + 2), xx = x0 - 1 + (rem - r * (XT + 2)), yy = y - 1 + r;
            nb[c][r][rem - r * (XT + 2)] = (yy >= 0 && yy < H && xx >= 0 && xx < W)
                                               ? (float)F * disp[((size_t)n * Cd + c0 + c) * HW + (size_t)yy * W + xx]
                                               : 0.0f;
            dacc[c][r][rem - r * (XT + 2)] = 0.0f;
        }
        for (int c = 0; c < nc; ++c) {
            const float *gp = gout + (((size_t)n * Cd + c0 + c) * (size_t)(F * H) + (size_t)F * y) * (size_t)(F * W) +
                              (size_t)F * x0;
            for (int idx = tid; idx < F * (XT * F); idx += 256) {
                const int i = idx / (XT * F), col = idx - i * (XT * F);
                gt[c][i][col] = (col < rowlen) ? gp[(size_t)i * (F * W) + col] : 0.0f;
            }
        }
        __syncthreads();
        float dn[CDT][9];  // this lane's neighbour contributions, summed over its (i,j) sub-positions
#pragma unroll
        for (int c = 0; c < CDT; ++c)
#pragma unroll
            for (int k = 0; k < 9; ++k) dn[c][k] = 0.0f;
        for (int ij = sub0; ij < F2; ij += 256 / XT) {
            const int i = ij / F, j = ij - i * F;
            float p[9], gm[9];
            if (x < W) {
                float mx = -INFINITY, sum = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    p[k] = mask[((size_t)n * 9 * F2 + (size_t)(k * F2 + ij)) * HW + (size_t)y * W + x];
                    mx = fmaxf(mx, p[k]);
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    p[k] = fast_exp(p[k] - mx);
                    sum += p[k];
                }
                const float inv = 1.0f / sum;
#pragma unroll
                for (int k = 0; k < 9; ++k) p[k] *= inv, gm[k] = 0.0f;
#pragma unroll
                for (int c = 0; c < CDT; ++c) {
                    if (c < nc) {
                        const float go = gt[c][i][xl * F + j];
                        float outc = 0.0f;
#pragma unroll
                        for (int k = 0; k < 9; ++k) outc = fmaf(p[k], nb[c][k / 3][xl + k % 3], outc);
#pragma unroll
                        for (int k = 0; k < 9; ++k) {
                            gm[k] = fmaf(go * p[k], nb[c][k / 3][xl + k % 3] - outc, gm[k]);
                            dn[c][k] = fmaf(go, p[k], dn[c][k]);
                        }
                    }
                }
                if (c0 == 0) {
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        gmask[((size_t)n * 9 * F2 + (size_t)(k * F2 + ij)) * HW + (size_t)y * W + x] = gm[k];
                } else {  // later channel blocks accumulate (Cd > CDT only)
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        gmask[((size_t)n * 9 * F2 + (size_t)(k * F2 + ij)) * HW + (size_t)y * W + x] += gm[k];
                }
            }
        }
        if (x < W) {
#pragma unroll
            for (int c = 0; c < CDT; ++c)
                if (c < nc) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) atomicAdd(&dacc[c][k / 3][xl + k % 3], (float)F * dn[c][k]);
                }
        }
        __syncthreads();
        for (int idx = tid; idx < nc * 3 * (XT + 2); idx += 256) {
            const int c = idx / (3 * (XT + 2)), rem = idx - c * 3 * (XT + 2);
            const int r = rem / (XT + 2), xx = x0 - 1 + (rem - r * (XT + 2)), yy = y - 1 + r;
            const float v = dacc[c][r][rem - r * (XT + 2)];
            if (yy >= 0 && yy < H && xx >= 0 && xx < W && v != 0.0f)
                atomicAdd(gdisp + ((size_t)n * Cd + c0 + c) * HW + (size_t)yy * W + xx, v);
        }
    }
}

// ------------------------------------------------------------------ GRU gate math backward
// reset: out = sigmoid(rc + cr) * h        -> gh = g*r ; grc = g*h*r*(1-r)            (gcr == grc)
// blend: out = (1-z)*h + z*q, z = sigmoid(zc+cz), q = act(qc+cq)
//        -> gh = g*(1-z) ; gzc = g*(q-h)*z*(1-z) ; gqc = g*z*act'(.)                   (gcz == gzc, gcq == gqc)
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_reset_bwd_kernel(const float *__restrict__ h, const float *__restrict__ rc,
                                                            const float *__restrict__ cr, const float *__restrict__ g,
                                                            float *__restrict__ gh, float *__restrict__ grc, long long n)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const float r = sigm(rc[t] + (cr ? cr[t] : 0.0f));
        gh[t] = g[t] * r;
        grc[t] = g[t] * h[t] * r * (1.0f - r);
    }
}

__global__ __launch_bounds__(256) void gru_blend_bwd_kernel(const float *__restrict__ h, const float *__restrict__ zc,
                                                            const float *__restrict__ qc, const float *__restrict__ cz,
                                                            const float *__restrict__ cq, const float *__restrict__ g,
                                                            float *__restrict__ gh, float *__restrict__ gzc,
                                                            float *__restrict__ gqc, long long n, int softsign)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const float z = sigm(zc[t] + (cz ? cz[t] : 0.0f));
        const float qx = qc[t] + (cq ? cq[t] : 0.0f);
        float q, dq;
        if (softsign) {
            const float den = 1.0f + fabsf(qx);
            q = qx / den;
            dq = 1.0f / (den * den);
        } else {
            q = tanhf(qx);
            dq = 1.0f - q * q;
        }
        gh[t] = g[t] * (1.0f - z);
        gzc[t] = g[t] * (q - h[t]) * z * (1.0f - z);
        gqc[t] = g[t] * z * dq;
    }
}

// ------------------------------------------------------------------ uncertainty volume backward (A11)
// U_d = p_d * d * (d - disp)^2, p = softmax(vol).  With a_d = gU_d * d * (d - disp)^2:
//   gvol_k = p_k * (a_k - sum_d p_d a_d) ;  gdisp = -2 * sum_d gU_d * p_d * d * (d - disp)
__global__ __launch_bounds__(256) void uncertainty_bwd_kernel(const float *__restrict__ vol,
                                                              const float *__restrict__ disp,
                                                              const float *__restrict__ gU, float *__restrict__ gvol,
                                                              float *__restrict__ gdisp, int D, long long HW,
                                                              long long npix)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float *v = vol + (size_t)b * D * HW + p;
        const float *gu = gU + (size_t)b * D * HW + p;
        float *gv = gvol + (size_t)b * D * HW + p;
        const float cur = disp[t];
        float m = -INFINITY, s = 0.0f, sa = 0.0f, sd = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float x = v[(size_t)d * HW];
            const float e = (float)d - cur;
            const float a = gu[(size_t)d * HW] * (float)d * e * e;
            const float cm = fmaxf(m, x);
            const float sc = (m == cm) ? 1.0f : fast_exp(m - cm);
            const float pe = fast_exp(x - cm);
            s = s * sc + pe;
            sa = sa * sc + pe * a;
            sd = sd * sc + pe * gu[(size_t)d * HW] * (float)d * e;
            m = cm;
        }
        const float inv = 1.0f / s, mean_a = sa * inv;
        for (int d = 0; d < D; ++d) {
            const float e = (float)d - cur;
            const float a = gu[(size_t)d * HW] * (float)d * e * e;
            gv[(size_t)d * HW] = fast_exp(v[(size_t)d * HW] - m) * inv * (a - mean_a);
        }
        gdisp[t] = -2.0f * sd * inv;
    }
}

// ------------------------------------------------------------------ disp_warp backward w.r.t. disp (A12)
// warped[c] = bilinear(img[c]; sx, sy), sx = clip(ux), ux = (x - disp) * W/(W-1) - 0.5  (align_corners=False
// un-normalisation of a grid normalised with (W-1), disparity_warper.py:55-56,100).
//   d warped[c] / d disp = -(W/(W-1)) * [ux not clipped] * d bilinear / d sx
// (grid_sample zeroes the coordinate gradient where border clipping is active).
__global__ __launch_bounds__(256) void disp_warp_bwd_kernel(const float *__restrict__ img, const float *__restrict__ disp,
                                                            const float *__restrict__ gw, float *__restrict__ gdisp,
                                                            int Ci, int H, int W, long long npix, int padding_border)
{
    const long long HW = (long long)H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        const float gx = (float)x + (-disp[t]);
        const float nx = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(gx, (float)(W - 1))), 1.0f);
        const float ny = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn((float)y, (float)(H - 1))), 1.0f);
        float sx = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(nx, 1.0f), (float)W), 1.0f), 2.0f);
        float sy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(ny, 1.0f), (float)H), 1.0f), 2.0f);
        float live = 1.0f;
        if (padding_border) {
            if (sx <= 0.0f || sx >= (float)(W - 1)) live = 0.0f;  // clip_coordinates_set_grad: zero outside (0, W-1)
            sx = clampf(sx, 0.0f, (float)(W - 1));
            sy = clampf(sy, 0.0f, (float)(H - 1));
        }
        const float fx = floorf(sx), fy = floorf(sy);
        const float ay = sy - fy;
        const int ix = (int)fx, iy = (int)fy;
        const bool x0 = ix >= 0 && ix <= W - 1, x1 = ix + 1 >= 0 && ix + 1 <= W - 1;
        const bool y0 = iy >= 0 && iy <= H - 1, y1 = iy + 1 >= 0 && iy + 1 <= H - 1;
        float acc = 0.0f;
        for (int c = 0; c < Ci; ++c) {
            const float *im = img + ((size_t)b * Ci + c) * HW;
            const float v00 = (y0 && x0) ? im[(size_t)iy * W + ix] : 0.0f;
            const float v01 = (y0 && x1) ? im[(size_t)iy * W + ix + 1] : 0.0f;
            const float v10 = (y1 && x0) ? im[(size_t)(iy + 1) * W + ix] : 0.0f;
            const float v11 = (y1 && x1) ? im[(size_t)(iy + 1) * W + ix + 1] : 0.0f;
            const float dsdx = (v01 - v00) * (1.0f - ay) + (v11 - v10) * ay;
            acc = fmaf(gw[((size_t)b * Ci + c) * HW + p], dsdx, acc);
        }
        // d ux / d disp = -(2/(W-1)) * (W/2)
        gdisp[t] = -acc * live * ((float)W / (float)(W - 1));
    }
}

// ------------------------------------------------------------------ learned-bounds lookup backward (A11 sampler)
// lower = cur - b0, upper = cur + b1, c_k = lower + k*(upper-lower)/S  (update.py:279-313), then as A5.
//   d c_k / d cur = 1 ; d c_k / d b0 = -(1 - k/S) ; d c_k / d b1 = k/S   (valid pixels, consider_valid)
//   invalid pixels: c_k = cur -> only d/d cur = 1.  consider_valid = 0: clamps cut the respective paths.
__global__ __launch_bounds__(256) void bounds_lookup_bwd_kernel(const float *__restrict__ vol,
                                                                const float *__restrict__ disp,
                                                                const float *__restrict__ bounds,
                                                                const float *__restrict__ gout, float *__restrict__ gvol,
                                                                float *__restrict__ gdisp, float *__restrict__ gbounds,
                                                                int D, long long HW, long long npix, int S,
                                                                int consider_valid)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float *v = vol + (size_t)b * D * HW + p;
        float *gvp = gvol + (size_t)b * D * HW + p;
        const float cur = disp[t], g = gout[t];
        float lower = cur - bounds[(size_t)b * 2 * HW + p], upper = cur + bounds[(size_t)b * 2 * HW + HW + p];
        int invalid = -1;
        float dl = 1.0f, du = 1.0f;  // are lower / upper still functions of their inputs?
        if (consider_valid) {
            invalid = (lower < 0.0f) || (upper >= (float)(D - 1));
        } else {
            if (lower < 0.0f) lower = 0.0f, dl = 0.0f;
            if (upper < 0.0f) upper = 0.0f, du = 0.0f;
            else if (upper > (float)D) upper = (float)D, du = 0.0f;
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
        float gd = 0.0f, gb0 = 0.0f, gb1 = 0.0f;
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
            if (invalid > 0) {
                gd += dc;                               // c_k = cur
            } else {
                const float wl = (1.0f - tk) * dl, wu = tk * du;   // c_k = (1-tk)*lower + tk*upper
                gd = fmaf(dc, wl + wu, gd);
                gb0 = fmaf(-dc, wl, gb0);
                gb1 = fmaf(dc, wu, gb1);
            }
        }
        if (pend_idx >= 0) gvp[(size_t)pend_idx * HW] += pend_val;
        gdisp[t] = gd;
        gbounds[(size_t)b * 2 * HW + p] = gb0;
        gbounds[(size_t)b * 2 * HW + HW + p] = gb1;
    }
}

}  // namespace

extern "C" int stereo_soft_argmin_bwd_f32(const float *vol, const float *grad_disp, const float *grad_prob,
                                          float *grad_vol, int B, int D, int H, int W, int match_similarity,
                                          void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0 || D == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(grad_disp);
    STEREO_REQUIRE_PTR(grad_vol);
    const long long HW = (long long)H * W;
    const float sgn = match_similarity ? 1.0f : -1.0f;
    hipStream_t st = as_stream(stream);
    const bool vec = (HW % 4 == 0);
    const long long n = vec ? (long long)B * (HW / 4) : (long long)B * HW;
#define SAB(V, GP) \
    hipLaunchKernelGGL((soft_argmin_bwd_kernel<V, GP>), dim3(grid_for(n)), dim3(256), 0, st, vol, grad_disp, grad_prob, \
                       grad_vol, D, HW, n, sgn)
    if (vec && !grad_prob && D <= 192 && fits_grid_yz(B) && aligned16(vol) && aligned16(grad_disp) && aligned16(grad_vol) &&
        stereo_opt(OPT_PROB_SINGLE_PASS) != 0) {   // one pass over the volume: the softmax column in registers
#define SAB_REGS(JV, NWVV, SPL)                                                                                             \
    hipLaunchKernelGGL((soft_argmin_bwd_regs_kernel<JV, NWVV, SPL, false>),                                                 \
                       dim3((unsigned)((HW / 4 + 64 / SPL - 1) / (64 / SPL)), (unsigned)B), dim3(NWVV * 64), 0, st, vol,    \
                       grad_disp, nullptr, grad_vol, D, HW, sgn)
        if (D <= 48) SAB_REGS(12, 4, 1);
        else if (D <= 96) SAB_REGS(24, 4, 1);
        else SAB_REGS(24, 4, 2);
#undef SAB_REGS
    } else if (vec && grad_prob) SAB(4, true);
    else if (vec) SAB(4, false);
    else if (grad_prob) SAB(1, true);
    else SAB(1, false);
#undef SAB
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_soft_argmin_bwd_add_f32(const float *vol, const float *grad_disp, const float *addend, float *grad_vol,
                                              int B, int D, int H, int W, int match_similarity, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0 || D == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(grad_disp);
    STEREO_REQUIRE_PTR(addend);
    STEREO_REQUIRE_PTR(grad_vol);
    const long long HW = (long long)H * W;
    if (HW % 4 != 0 || D > 192 || !fits_grid_yz(B) || !aligned16(vol) || !aligned16(grad_disp) || !aligned16(addend) ||
        !aligned16(grad_vol))
        return STEREO_ERR_UNSUPPORTED;  // the caller adds the two cotangents itself
    const float sgn = match_similarity ? 1.0f : -1.0f;
    hipStream_t st = as_stream(stream);
#define SAB_REGS(JV, NWVV, SPL)                                                                                             \
    hipLaunchKernelGGL((soft_argmin_bwd_regs_kernel<JV, NWVV, SPL, true>),                                                  \
                       dim3((unsigned)((HW / 4 + 64 / SPL - 1) / (64 / SPL)), (unsigned)B), dim3(NWVV * 64), 0, st, vol,    \
                       grad_disp, addend, grad_vol, D, HW, sgn)
    if (D <= 48) SAB_REGS(12, 4, 1);
    else if (D <= 96) SAB_REGS(24, 4, 1);
    else SAB_REGS(24, 4, 2);
#undef SAB_REGS
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_gwc_volume_bwd_f32(const float *left, const float *right, const float *grad_vol,
                                         float *grad_left, float *grad_right, int B, int C, int H, int W, int D, int G,
                                         void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || D < 0 || G <= 0) return STEREO_ERR_BAD_SHAPE;
    if (C % G != 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || C == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(left);
    STEREO_REQUIRE_PTR(right);
    if (D > 0) STEREO_REQUIRE_PTR(grad_vol);
    if (grad_left == nullptr && grad_right == nullptr) return STEREO_OK;
    {
        // Group g of batch b uses channels g CPG .. of the features and slab (b G + g) of the gradient volume: with the channels
        // of a group contiguous, the group-wise backward IS the correlation backward of B G "pairs" with CPG channels each
        // (mean over CPG) -- the same pointers, reinterpreted.  So the one-read kernel serves A3 as well.
        if (D > 0 && G >= 1 && C % G == 0 && (long long)B * G < (1LL << 30) && stereo_opt(OPT_CORR_MFMA) != 0) {
            const int rc = stereo_corr_volume_bwd_mfma(left, right, grad_vol, grad_left, grad_right, B * G, C / G, H, W, D,
                                                       as_stream(stream));
            if (rc != STEREO_ERR_UNSUPPORTED) return rc;
        }
    }
    if (!fits_grid_yz(H) || !fits_grid_yz((long long)B * G) || D > 4096) return STEREO_ERR_UNSUPPORTED;
    const int CPG = C / G;
    const int Dpad = (D + BDC - 1) / BDC * BDC;
    const int CB = CPG <= 4 ? 4 : (CPG <= 8 ? 8 : 32);
    size_t lds = (size_t)CB * (BT + Dpad) * sizeof(float);
    const size_t redb = (size_t)BNW * BRC * BT * sizeof(float);
    if (lds < redb) lds = redb;
    if (lds > 160 * 1024) return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const bool pow2 = (CPG & (CPG - 1)) == 0;
    dim3 grid((W + BT - 1) / BT, H, (unsigned)(B * G));
#define VB_LAUNCH(LEFT, CBV, OTHER, GOUT)                                                                               \
    {                                                                                                                  \
        auto k = volume_bwd_kernel<LEFT, CBV>;                                                                         \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k),                                  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return STEREO_ERR_UNSUPPORTED;                                                                             \
        hipLaunchKernelGGL(k, grid, dim3(BNW * 64), lds, st, OTHER, grad_vol, GOUT, C, H, W, D, G, CPG, Dpad,           \
                           1.0f / (float)CPG, pow2 ? 0 : 1);                                                           \
        STEREO_LAUNCH_CHECK();                                                                                         \
    }
    if (grad_left) {
        if (CB == 4) VB_LAUNCH(true, 4, right, grad_left)
        else if (CB == 8) VB_LAUNCH(true, 8, right, grad_left)
        else VB_LAUNCH(true, 32, right, grad_left)
    }
    if (grad_right) {
        if (CB == 4) VB_LAUNCH(false, 4, left, grad_right)
        else if (CB == 8) VB_LAUNCH(false, 8, left, grad_right)
        else VB_LAUNCH(false, 32, left, grad_right)
    }
#undef VB_LAUNCH
    return STEREO_OK;
}

extern "C" int stereo_local_lookup_bwd_f32(const float *vol, const float *disp, const float *grad_out,
                                           float *grad_vol_zeroed, float *grad_disp, int B, int D, int H, int W,
                                           float radius, int S, int consider_valid, void *stream)
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
    hipLaunchKernelGGL(local_lookup_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, disp, grad_out,
                       grad_vol_zeroed, grad_disp, D, HW, n, radius, S, consider_valid);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_local_lookup_bwd_dense_f32(const float *vol, const float *disp, const float *grad_out, float *grad_vol,
                                                 float *grad_disp, int B, int D, int H, int W, float radius, int S,
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
    // window of touched planes: floor(lower) .. ceil(upper) <= 2 radius + 3 planes, plus one of slack below
    if (!(radius >= 0.0f) || 2.0f * radius + 4.0f > (float)LLB_WIN || HW % 4 != 0 || !aligned16(grad_vol) || !fits_grid_yz(B))
        return STEREO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(local_lookup_bwd_dense_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)B), dim3(256), 0,
                       as_stream(stream), vol, disp, grad_out, grad_vol, grad_disp, D, HW, radius, S, consider_valid,
                       (int)fminf(ceilf(2.0f * radius) + 4.0f, (float)LLB_WIN));
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_upsample_convex_bwd_f32(const float *disp, const float *mask, const float *grad_out,
                                              float *grad_disp_zeroed, float *grad_mask, int N, int Cd, int H, int W,
                                              int f, void *stream)
{
    if (N < 0 || Cd < 0 || H < 0 || W < 0 || f <= 0) return STEREO_ERR_BAD_SHAPE;
    if (N == 0 || Cd == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(mask);
    STEREO_REQUIRE_PTR(grad_out);
    STEREO_REQUIRE_PTR(grad_disp_zeroed);
    STEREO_REQUIRE_PTR(grad_mask);
    if (!fits_grid_yz(H) || !fits_grid_yz(N)) return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    dim3 grid((W + 63) / 64, H, N);
    if (f == 8 && Cd == 1)
        hipLaunchKernelGGL((upsample_convex_bwd_kernel<8, 1>), grid, dim3(256), 0, st, disp, mask, grad_out,
                           grad_disp_zeroed, grad_mask, Cd, H, W);
    else if (f == 8)
        hipLaunchKernelGGL((upsample_convex_bwd_kernel<8, 2>), grid, dim3(256), 0, st, disp, mask, grad_out,
                           grad_disp_zeroed, grad_mask, Cd, H, W);
    else if (f == 4 && Cd == 1)
        hipLaunchKernelGGL((upsample_convex_bwd_kernel<4, 1>), grid, dim3(256), 0, st, disp, mask, grad_out,
                           grad_disp_zeroed, grad_mask, Cd, H, W);
    else if (f == 4)
        hipLaunchKernelGGL((upsample_convex_bwd_kernel<4, 2>), grid, dim3(256), 0, st, disp, mask, grad_out,
                           grad_disp_zeroed, grad_mask, Cd, H, W);
    else
        return STEREO_ERR_UNSUPPORTED;  // backward is built for the factors the reference uses (8 and 4)
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_gru_reset_bwd_f32(const float *h, const float *rc, const float *cr, const float *grad_out,
                                        float *grad_h, float *grad_rc, long long n, void *stream)
{
    if (n < 0) return STEREO_ERR_BAD_SHAPE;
    if (n == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(h);
    STEREO_REQUIRE_PTR(rc);
    STEREO_REQUIRE_PTR(grad_out);
    STEREO_REQUIRE_PTR(grad_h);
    STEREO_REQUIRE_PTR(grad_rc);
    hipLaunchKernelGGL(gru_reset_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), h, rc, cr, grad_out,
                       grad_h, grad_rc, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_gru_blend_bwd_f32(const float *h, const float *zc, const float *qc, const float *cz,
                                        const float *cq, const float *grad_out, float *grad_h, float *grad_zc,
                                        float *grad_qc, long long n, int softsign, void *stream)
{
    if (n < 0) return STEREO_ERR_BAD_SHAPE;
    if (n == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(h);
    STEREO_REQUIRE_PTR(zc);
    STEREO_REQUIRE_PTR(qc);
    STEREO_REQUIRE_PTR(grad_out);
    STEREO_REQUIRE_PTR(grad_h);
    STEREO_REQUIRE_PTR(grad_zc);
    STEREO_REQUIRE_PTR(grad_qc);
    hipLaunchKernelGGL(gru_blend_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), h, zc, qc, cz, cq,
                       grad_out, grad_h, grad_zc, grad_qc, n, softsign);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_uncertainty_volume_bwd_f32(const float *vol, const float *disp, const float *grad_u,
                                                 float *grad_vol, float *grad_disp, int B, int D, int H, int W,
                                                 void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0 || D == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(grad_u);
    STEREO_REQUIRE_PTR(grad_vol);
    STEREO_REQUIRE_PTR(grad_disp);
    const long long HW = (long long)H * W, n = (long long)B * HW;
    hipLaunchKernelGGL(uncertainty_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, disp, grad_u,
                       grad_vol, grad_disp, D, HW, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_disp_warp_bwd_disp_f32(const float *img, const float *disp, const float *grad_warped,
                                             float *grad_disp, int B, int Ci, int H, int W, int padding_border,
                                             void *stream)
{
    if (B < 0 || Ci < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(img);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(grad_warped);
    STEREO_REQUIRE_PTR(grad_disp);
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(disp_warp_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), img, disp, grad_warped,
                       grad_disp, Ci, H, W, n, padding_border);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_bounds_lookup_bwd_f32(const float *vol, const float *disp, const float *bounds,
                                            const float *grad_out, float *grad_vol_zeroed, float *grad_disp,
                                            float *grad_bounds, int B, int D, int H, int W, int S, int consider_valid,
                                            void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0 || S <= 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    if (D == 0) return STEREO_ERR_BAD_SHAPE;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(bounds);
    STEREO_REQUIRE_PTR(grad_out);
    STEREO_REQUIRE_PTR(grad_vol_zeroed);
    STEREO_REQUIRE_PTR(grad_disp);
    STEREO_REQUIRE_PTR(grad_bounds);
    const long long HW = (long long)H * W, n = (long long)B * HW;
    hipLaunchKernelGGL(bounds_lookup_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), vol, disp, bounds,
                       grad_out, grad_vol_zeroed, grad_disp, grad_bounds, D, HW, n, S, consider_valid);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
