// Soft-argmin disparity regression (A4) and its fused readers.
//
// One lane owns V (=4 when H*W % 4 == 0) consecutive pixels and walks the D
// planes of the volume with 16-byte loads (one wave load = 1 KiB contiguous
// along W), keeping an online-softmax state (running max, sum, d-weighted sum)
// per pixel: the volume is read exactly once.  The with-prob variant re-reads
// the (then L2/MALL-warm) column to write the normalised probabilities.

#include "common.h"

namespace {

template <int V>
struct Vec;
template <>
struct Vec<4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T load(const float *p) { return ld4(p); }
    static __device__ __forceinline__ void store(float *p, T v) { st4(p, v); }
};
template <>
struct Vec<1> {
    typedef float T __attribute__((ext_vector_type(1)));
    static __device__ __forceinline__ T load(const float *p) { return (T){*p}; }
    static __device__ __forceinline__ void store(float *p, T v) { *p = v[0]; }
};

// Online softmax-expectation state for V pixels.
template <int V>
struct Online {
    float m[V], s[V], e[V];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            m[i] = -INFINITY;
            s[i] = 0.0f;
            e[i] = 0.0f;
        }
    }
    // absorb U planes x[u][i] with candidate value d0+u
    template <int U>
    __device__ __forceinline__ void absorb(const float (&x)[U][V], int d0, int nvalid)
    {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float cm = m[i];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u < nvalid) cm = fmaxf(cm, x[u][i]);
            const float sc = (m[i] == cm) ? 1.0f : fast_exp(m[i] - cm);  // m=-inf,cm finite -> 0
            float ss = s[i] * sc, ee = e[i] * sc;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u < nvalid) {
                    const float p = fast_exp(x[u][i] - cm);
                    ss += p;
                    ee = fmaf(p, (float)(d0 + u), ee);
                }
            m[i] = cm;
            s[i] = ss;
            e[i] = ee;
        }
    }
};

// vol [B, D, HW] -> disp [B, HW] (+ prob)
template <int V, int U, bool PROB>
__global__ __launch_bounds__(256) void soft_argmin_kernel(const float *__restrict__ vol, float *__restrict__ disp,
                                                          float *__restrict__ prob, int D, long long HW,
                                                          long long ncols, float sgn)
{
    const long long HWv = HW / V;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < ncols;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HWv, p = (t - b * HWv) * V;
        const float *v = vol + (size_t)b * D * HW + p;
        Online<V> st;
        st.init();
        for (int d0 = 0; d0 < D; d0 += U) {
            float x[U][V];
            const int nv = (D - d0) < U ? (D - d0) : U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < nv) {
                    const typename Vec<V>::T r = Vec<V>::load(v + (size_t)(d0 + u) * HW);
#pragma unroll
                    for (int i = 0; i < V; ++i) x[u][i] = sgn * r[i];
                }
            }
            st.template absorb<U>(x, d0, nv);
        }
        typename Vec<V>::T o;
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = st.e[i] / st.s[i];
        Vec<V>::store(disp + (size_t)b * HW + p, o);
        if (PROB) {
            float inv[V];
#pragma unroll
            for (int i = 0; i < V; ++i) inv[i] = 1.0f / st.s[i];
            float *pr = prob + (size_t)b * D * HW + p;
            for (int d = 0; d < D; ++d) {
                const typename Vec<V>::T r = Vec<V>::load(v + (size_t)d * HW);
                typename Vec<V>::T q;
#pragma unroll
                for (int i = 0; i < V; ++i) q[i] = fast_exp(sgn * r[i] - st.m[i]) * inv[i];
                Vec<V>::store(pr + (size_t)d * HW, q);
            }
        }
    }
}

// A4 with probabilities, single pass: the whole softmax column lives in registers.  A workgroup of NWV waves owns
// 64/SPLIT pixel quads (16 B each); the D planes are dealt round-robin to NWV*SPLIT "phases" -- phase = wave * SPLIT +
// lane / (64/SPLIT) -- and every lane keeps its J planes of one quad (D <= NWV*SPLIT*J).  A wave load / store covers
// SPLIT segments of 1024/SPLIT contiguous bytes.  All J loads are issued before the first use, the running max / sums
// are combined across the phases through LDS (two barriers), and the normalised probabilities are written straight
// from the registers: the volume is read once and the probabilities written once.  Small workgroups (4 waves) so
// that several are resident per CU and one's load phase overlaps another's store phase.
template <int J, int NWV, int SPLIT>
__global__ __launch_bounds__(NWV * 64) void soft_argmin_prob_regs_kernel(const float *__restrict__ vol,
                                                                         float *__restrict__ disp,
                                                                         float *__restrict__ prob, int D, long long HW,
                                                                         float sgn)
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
    float inv[4];
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float S = 0.0f, E = 0.0f;
#pragma unroll
        for (int k = 0; k < NPH; ++k) S += red[1][k][ql * 4 + i], E += red[2][k][ql * 4 + i];
        inv[i] = 1.0f / S;
        o[i] = E / S;
    }
    if (!okp) return;
    if (ph == 0) st4(disp + (size_t)blockIdx.y * HW + p, o);
    float *pr = prob + boff;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int d = ph + NPH * j;
        if (d < D) {
            f32x4 q;
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = x[j][i] * inv[i];
            st4(pr + (size_t)d * HW, q);
        }
    }
}

// volg [B, G, D, HW] -> mean over G -> (vol_mean [B, D, HW]) -> disp [B, HW]
template <int V, int U, bool WRITE_MEAN, bool NT>
__global__ __launch_bounds__(256) void group_mean_soft_argmin_kernel(const float *__restrict__ volg,
                                                                     float *__restrict__ vol_mean,
                                                                     float *__restrict__ disp, int G, int D,
                                                                     long long HW, long long ncols, float sgn,
                                                                     float inv_g, int exact_div, int nt_store)
{
    const long long HWv = HW / V;
    const size_t gstride = (size_t)D * HW;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < ncols;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HWv, p = (t - b * HWv) * V;
        const float *v = volg + (size_t)b * G * gstride + p;
        Online<V> st;
        st.init();
        for (int d0 = 0; d0 < D; d0 += U) {
            float x[U][V];
            const int nv = (D - d0) < U ? (D - d0) : U;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < V; ++i) x[u][i] = 0.0f;
#pragma unroll 4
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u < nv) {
                        const float *src = v + (size_t)g * gstride + (size_t)(d0 + u) * HW;
                        const typename Vec<V>::T r =
                            NT ? __builtin_nontemporal_load(reinterpret_cast<const typename Vec<V>::T *>(src)) : Vec<V>::load(src);
#pragma unroll
                        for (int i = 0; i < V; ++i) x[u][i] += r[i];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < nv) {
                    typename Vec<V>::T mv;
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        mv[i] = exact_div ? x[u][i] / (float)G : x[u][i] * inv_g;
                        x[u][i] = sgn * mv[i];
                    }
                    if (WRITE_MEAN) {
                        float *dst = vol_mean + (size_t)b * gstride + (size_t)(d0 + u) * HW + p;
                        if (NT && nt_store)
                            __builtin_nontemporal_store(mv, reinterpret_cast<typename Vec<V>::T *>(dst));
                        else
                            Vec<V>::store(dst, mv);
                    }
                }
            }
            st.template absorb<U>(x, d0, nv);
        }
        typename Vec<V>::T o;
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = st.e[i] / st.s[i];
        Vec<V>::store(disp + (size_t)b * HW + p, o);
    }
}

// disparity_regression (models/utils/submodule.py:23-27): out[b,p] = sum_d x[b,d,p] * d
template <int V>
__global__ __launch_bounds__(256) void disparity_regression_kernel(const float *__restrict__ x,
                                                                   float *__restrict__ out, int D, long long HW,
                                                                   long long ncols)
{
    const long long HWv = HW / V;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < ncols;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HWv, p = (t - b * HWv) * V;
        const float *v = x + (size_t)b * D * HW + p;
        typename Vec<V>::T acc;
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.0f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const typename Vec<V>::T r = Vec<V>::load(v + (size_t)d * HW);
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = fmaf(r[i], (float)d, acc[i]);
        }
        Vec<V>::store(out + (size_t)b * HW + p, acc);
    }
}

inline unsigned grid_for(long long nthreads)
{
    long long blocks = (nthreads + 255) / 256;
    const long long cap = 256LL * 32;  // grid-stride above this
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

extern "C" int stereo_soft_argmin_f32(const float *vol, float *disp, float *prob, int B, int D, int H, int W,
                                      int match_similarity, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    if (D == 0) return STEREO_ERR_BAD_SHAPE;  // softmax over an empty axis
    const long long HW = (long long)H * W;
    const float sgn = match_similarity ? 1.0f : -1.0f;
    hipStream_t st = as_stream(stream);
    const bool vec = (HW % 4 == 0) && aligned16(vol) && aligned16(disp) && (prob == nullptr || aligned16(prob));
    if (vec) {
        const long long n = (long long)B * (HW / 4);
        if (prob && D <= 192 && fits_grid_yz(B) && stereo_opt(OPT_PROB_SINGLE_PASS) != 0) {
#define PROB_LAUNCH(JV, NWVV, SPL)                                                                                   \
    hipLaunchKernelGGL((soft_argmin_prob_regs_kernel<JV, NWVV, SPL>),                                                \
                       dim3((unsigned)((HW / 4 + 64 / SPL - 1) / (64 / SPL)), (unsigned)B), dim3(NWVV * 64), 0, st, \
                       vol, disp, prob, D, HW, sgn)
            if (D <= 48) PROB_LAUNCH(12, 4, 1);
            else if (D <= 96) PROB_LAUNCH(24, 4, 1);
            else PROB_LAUNCH(24, 4, 2);
#undef PROB_LAUNCH
        } else if (prob)
            hipLaunchKernelGGL((soft_argmin_kernel<4, 8, true>), dim3(grid_for(n)), dim3(256), 0, st, vol, disp, prob, D,
                               HW, n, sgn);
        else
            hipLaunchKernelGGL((soft_argmin_kernel<4, 8, false>), dim3(grid_for(n)), dim3(256), 0, st, vol, disp, prob,
                               D, HW, n, sgn);
    } else {
        const long long n = (long long)B * HW;
        if (prob)
            hipLaunchKernelGGL((soft_argmin_kernel<1, 8, true>), dim3(grid_for(n)), dim3(256), 0, st, vol, disp, prob, D,
                               HW, n, sgn);
        else
            hipLaunchKernelGGL((soft_argmin_kernel<1, 8, false>), dim3(grid_for(n)), dim3(256), 0, st, vol, disp, prob,
                               D, HW, n, sgn);
    }
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_group_mean_soft_argmin_f32(const float *volg, float *vol_mean, float *disp, int B, int G, int D,
                                                 int H, int W, int match_similarity, void *stream)
{
    if (B < 0 || G <= 0 || D < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(volg);
    STEREO_REQUIRE_PTR(disp);
    if (D == 0) return STEREO_ERR_BAD_SHAPE;
    const long long HW = (long long)H * W;
    const float sgn = match_similarity ? 1.0f : -1.0f;
    const bool pow2 = (G & (G - 1)) == 0;
    hipStream_t st = as_stream(stream);
    const bool vec =
        (HW % 4 == 0) && aligned16(volg) && aligned16(disp) && (vol_mean == nullptr || aligned16(vol_mean));
    // measured best on MI355X: 1 plane x G non-temporal loads per step, plain stores
    constexpr int cfg_cap = 1 << 20, cfg_nts = 0;
#define GM_LAUNCH(V, U, WM, NTV)                                                                                      \
    hipLaunchKernelGGL((group_mean_soft_argmin_kernel<V, U, WM, NTV>), dim3(gm_grid), dim3(256), 0, st, volg, vol_mean, \
                       disp, G, D, HW, n, sgn, 1.0f / (float)G, pow2 ? 0 : 1, cfg_nts)
    if (vec) {
        const long long n = (long long)B * (HW / 4);
        long long gb = (n + 255) / 256;
        if (gb > 256LL * cfg_cap) gb = 256LL * cfg_cap;
        const unsigned gm_grid = (unsigned)gb;
        if (vol_mean) {
            GM_LAUNCH(4, 1, true, true);
        } else {
            GM_LAUNCH(4, 1, false, true);
        }
    } else {
        const unsigned gm_grid = grid_for((long long)B * HW);
        const long long n = (long long)B * HW;
        if (vol_mean)
            GM_LAUNCH(1, 4, true, false);
        else
            GM_LAUNCH(1, 4, false, false);
    }
#undef GM_LAUNCH
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_disparity_regression_f32(const float *x, float *out, int B, int D, int H, int W, void *stream)
{
    if (B < 0 || D < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(x);
    STEREO_REQUIRE_PTR(out);
    const long long HW = (long long)H * W;
    hipStream_t st = as_stream(stream);
    if ((HW % 4 == 0) && aligned16(x) && aligned16(out)) {
        const long long n = (long long)B * (HW / 4);
        hipLaunchKernelGGL(disparity_regression_kernel<4>, dim3(grid_for(n)), dim3(256), 0, st, x, out, D, HW, n);
    } else {
        const long long n = (long long)B * HW;
        hipLaunchKernelGGL(disparity_regression_kernel<1>, dim3(grid_for(n)), dim3(256), 0, st, x, out, D, HW, n);
    }
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
