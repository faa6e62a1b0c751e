// RAFT-style epipolar correlation (A8 all-pairs + pyramid, A9 lookup, A10
// on-the-fly alternative).
//
// All-pairs: per image row (b,h) the volume is the dense product
//   out[w1,w2] = sum_c f1[c,w1] * f2[c,w2] / sqrt(C)
// computed on the matrix cores with the exact-f32 MFMA (v_mfma_f32_32x32x2_f32:
// bitwise an fmaf chain, so the 1e-3 disparity bar holds without any
// low-precision split).  A <-> fmap1 (rows = w1), B <-> fmap2 (cols = w2): both
// operands are read straight from the [C][W] layout, 32 consecutive pixels per
// half-wave, and every accumulator register stores a 128-B row segment.
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

// block = 4 waves (2 x 2), wave tile = 64 (w1) x 64 (w2) = 2 x 2 MFMA tiles
__global__ __launch_bounds__(256) void allpairs_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                       float *__restrict__ out, int C, int H, int W1, int W2,
                                                       float sqrt_c)
{
    const int bh = blockIdx.z;
    const int b = bh / H, h = bh - b * H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64;  // w1 origin of this wave
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64;   // w2 origin
    if (m0 >= W1 || n0 >= W2) return;                    // wave-uniform

    const size_t HW1 = (size_t)H * W1, HW2 = (size_t)H * W2;
    const float *a_base = f1 + (size_t)b * C * HW1 + (size_t)h * W1;
    const float *b_base = f2 + (size_t)b * C * HW2 + (size_t)h * W2;
    const int ia0 = m0 + l31, ia1 = m0 + 32 + l31;
    const int jb0 = n0 + l31, jb1 = n0 + 32 + l31;
    const bool va0 = ia0 < W1, va1 = ia1 < W1, vb0 = jb0 < W2, vb1 = jb1 < W2;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    for (int c0 = 0; c0 < C; c0 += 2) {
        const int c = c0 + half;
        const bool vc = c < C;
        const float a0 = (vc && va0) ? a_base[(size_t)c * HW1 + ia0] : 0.0f;
        const float a1 = (vc && va1) ? a_base[(size_t)c * HW1 + ia1] : 0.0f;
        const float b0 = (vc && vb0) ? b_base[(size_t)c * HW2 + jb0] : 0.0f;
        const float b1 = (vc && vb1) ? b_base[(size_t)c * HW2 + jb1] : 0.0f;
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
    }
    float *obase = out + (size_t)bh * W1 * W2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int w1a = m0 + row, w1b = m0 + 32 + row;
        if (w1a < W1) {
            if (vb0) obase[(size_t)w1a * W2 + jb0] = acc00[r] / sqrt_c;
            if (vb1) obase[(size_t)w1a * W2 + jb1] = acc01[r] / sqrt_c;
        }
        if (w1b < W1) {
            if (vb0) obase[(size_t)w1b * W2 + jb0] = acc10[r] / sqrt_c;
            if (vb1) obase[(size_t)w1b * W2 + jb1] = acc11[r] / sqrt_c;
        }
    }
}

// LDS-staged, software-pipelined version (fast path: W1 % 4 == 0, W2 % 4 == 0, 16-B aligned).
// WG tile 128 (w1) x 128 (w2); the K = C loop runs in chunks of KC = 16 channels: the [KC][128] slices
// of both feature rows are fetched with 16-byte loads (each lane 2 + 2 quads), held in registers while
// the previous chunk is on the matrix cores, then committed to the other LDS buffer: one barrier per
// chunk.  MFMA operands are conflict-free ds_read_b32 (lanes along w).
constexpr int AP_KC = 16;
struct PyrOut {
    float *p[STEREO_MAX_LEVELS];
    int w[STEREO_MAX_LEVELS];
};

// f32 MFMA and VALU instructions serialise on a gfx950 SIMD (tools/mfma_coexec.hip), so the epilogue's instruction
// count is paid in MFMA time: one multiply (or, when sqrt(C) is not a power of two, one true division) and one store
// per output, 32-bit offsets from a workgroup-uniform base, no bounds tests on interior tiles.  (Pooling the pyramid
// levels in this epilogue with cross-lane adds was bit-identical but slower than one streaming pass over level 0:
// 0.631 vs 0.583 ms at [8,256,136,240].)
// eight fp32 values -> MFMA operand pieces of 16 x (NP = 2: hi, lo unscaled -- two mixed-precision FMAs per value; NP = 1:
// the values are fp16 numbers already)
template <int NP>
__device__ __forceinline__ void ap_split(const float *x, f16x8 *p)
{
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    if constexpr (NP == 2) {
        float pre = 16.0f;
        asm volatile("" : "+v"(pre));
        unsigned h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[j]) : "v"(x[2 * j]), "v"(pre));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[j]) : "v"(x[2 * j + 1]), "v"(pre));
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l[j]) : "v"(x[2 * j]), "v"(pre), "v"(h[j]));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[j]) : "v"(x[2 * j + 1]), "v"(pre), "v"(h[j]));
        }
        p[0] = __builtin_bit_cast(f16x8, (u32x4v){h[0], h[1], h[2], h[3]});
        p[1] = __builtin_bit_cast(f16x8, (u32x4v){l[0], l[1], l[2], l[3]});
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) p[0][j] = (_Float16)x[j];
    }
}

// F16P (round 3, default): the matrix phase on the fp16 matrix cores with split operands.  A chunk of 16 channels is ONE
// k-step of v_mfma_f32_32x32x16_f16: a lane reads its 8 channels of a row / column block from the fp32 image (the same
// conflict-free ds_read_b32, lanes along w), splits them in registers into two fp16 pieces of 16 x -- hi = fp16(16 x),
// lo = fp16(16 x - hi), two v_fma_mix per value (attention_chunked16.hip's arithmetic: the low piece is a normal fp16 number
// for |x| >= 2^-7, below that its absolute error is <= 2^-29) -- and the three leading piece products of each of the four
// output blocks go to its ONE accumulator: 12 matrix instructions of 32 cycles per chunk instead of 32 of 64 (fp16 features:
// one piece, exactly, 4 instructions).  The 1 / 256 of the two pre-scales rides on the 1 / sqrt(C) of the epilogue (exact).
// Range: |feature| < 4 094 (fp32 features), inf / NaN beyond, never silently wrong; test option corr_f16p = 0: f32 MFMA.
template <typename TIN, bool F16P>
__global__ __launch_bounds__(256) void allpairs_lds_kernel(const TIN *__restrict__ f1, const TIN *__restrict__ f2,
                                                           float *__restrict__ out, int C, int H, int W1, int W2,
                                                           float sqrt_c, float inv_sqrt_c, int mul_exact)
{
    __shared__ float As[2][AP_KC][128];
    __shared__ float Bs[2][AP_KC][128];
    const int bh = blockIdx.z;
    const int b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int M0 = blockIdx.y * 128, N0 = blockIdx.x * 128;
    const int mw = (wave >> 1) * 64, nw = (wave & 1) * 64;  // wave origin inside the tile

    const size_t HW1 = (size_t)H * W1, HW2 = (size_t)H * W2;
    const TIN *a_base = f1 + (size_t)b * C * HW1 + (size_t)h * W1;
    const TIN *b_base = f2 + (size_t)b * C * HW2 + (size_t)h * W2;

    // staging slots: quad q = tid + k*256 of a [16][32 quads] slice
    int sc[2], sx[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = tid + k * 256;
        sc[k] = q >> 5;
        sx[k] = (q & 31) * 4;
    }
    typename Quad<TIN>::raw pa[2], pb[2];  // raw quads (16 B fp32 / 8 B fp16), widened at commit
    unsigned okmask = 0;  // applied at commit time so that the loads stay in flight during the MFMA block
    auto issue = [&](int c0) {
        okmask = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = c0 + sc[k];
            const bool oka = c < C && M0 + sx[k] + 3 < W1;  // W % 4 == 0: a quad is all-in or all-out
            const bool okb = c < C && N0 + sx[k] + 3 < W2;
            pa[k] = Quad<TIN>::ld(a_base + (oka ? (size_t)c * HW1 + M0 + sx[k] : 0));
            pb[k] = Quad<TIN>::ld(b_base + (okb ? (size_t)c * HW2 + N0 + sx[k] : 0));
            okmask |= (oka ? (1u << k) : 0u) | (okb ? (4u << k) : 0u);
        }
    };
    auto commit = [&](int buf) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            *reinterpret_cast<f32x4 *>(&As[buf][sc[k]][sx[k]]) = (okmask >> k) & 1u ? Quad<TIN>::widen(pa[k]) : zero;
            *reinterpret_cast<f32x4 *>(&Bs[buf][sc[k]][sx[k]]) = (okmask >> (2 + k)) & 1u ? Quad<TIN>::widen(pb[k]) : zero;
        }
    };

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const int nchunk = (C + AP_KC - 1) / AP_KC;
    issue(0);
    commit(0);
    if (nchunk > 1) issue(AP_KC);
    __syncthreads();
    for (int ck = 0; ck < nchunk; ++ck) {
        const int buf = ck & 1;
        if (ck + 1 < nchunk) commit(buf ^ 1);           // its last readers passed the previous barrier
        if (ck + 2 < nchunk) issue((ck + 2) * AP_KC);   // in flight during this chunk's MFMAs
        if constexpr (F16P) {
            constexpr int NP = sizeof(TIN) == 4 ? 2 : 1;
            f16x8 pa_[2][NP], pb_[2][NP];            // [32-wide block][piece]: element j = channel 8 half + j of the chunk
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float xa[8], xb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xa[j] = As[buf][8 * half + j][mw + 32 * blk + l31];
                    xb[j] = Bs[buf][8 * half + j][nw + 32 * blk + l31];
                }
                ap_split<NP>(xa, pa_[blk]);
                ap_split<NP>(xb, pb_[blk]);
            }
#pragma unroll
            for (int p = 0; p < (NP == 2 ? 3 : 1); ++p) {   // hi hi, hi lo, lo hi -- block-interleaved: four independent chains
                const int ia = p == 2 ? NP - 1 : 0, ib = p == 1 ? NP - 1 : 0;
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_[0][ia], pb_[0][ib], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_[0][ia], pb_[1][ib], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_[1][ia], pb_[0][ib], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_[1][ia], pb_[1][ib], acc11, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < AP_KC / 2; ++s) {
                const int c = 2 * s + half;
                const float a0 = As[buf][c][mw + l31], a1 = As[buf][c][mw + 32 + l31];
                const float b0 = Bs[buf][c][nw + l31], b1 = Bs[buf][c][nw + 32 + l31];
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int jb0 = N0 + nw + l31, jb1 = jb0 + 32;
    auto scaled = [&](float v) { return mul_exact ? v * inv_sqrt_c : v / sqrt_c; };  // corr.py:110 divides by sqrt(C)
    // launcher: W1 * W2 < 2^31, so a row block's offsets fit 32 bits next to the uniform 64-bit base
    float *ob = out + (size_t)bh * W1 * W2;
    const unsigned off0 = (unsigned)(M0 + mw + 4 * half) * (unsigned)W2 + (unsigned)jb0;
    const bool interior = M0 + 128 <= W1 && N0 + 128 <= W2;  // workgroup-uniform
    if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned o = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)W2;
            ob[o] = scaled(acc00[r]);
            ob[o + 32] = scaled(acc01[r]);
            ob[o + 32u * (unsigned)W2] = scaled(acc10[r]);
            ob[o + 32u * (unsigned)W2 + 32] = scaled(acc11[r]);
        }
    } else {
        const bool okj0 = jb0 < W2, okj1 = jb1 < W2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w1a = M0 + mw + 4 * half + (r & 3) + 8 * (r >> 2);
            const unsigned o = off0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)W2;
            if (w1a < W1 && okj0) ob[o] = scaled(acc00[r]);
            if (w1a < W1 && okj1) ob[o + 32] = scaled(acc01[r]);
            if (w1a + 32 < W1 && okj0) ob[o + 32u * (unsigned)W2] = scaled(acc10[r]);
            if (w1a + 32 < W1 && okj1) ob[o + 32u * (unsigned)W2 + 32] = scaled(acc11[r]);
        }
    }
}

// F.avg_pool2d(x,[1,2],stride=[1,2]) on rows: Win -> Win/2 (floor), corr.py:70-76
__global__ __launch_bounds__(256) void avgpool_w2_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         int Win, int Wout, long long nout)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nout;
         t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / Wout;
        const int x = (int)(t - r * Wout);
        const float *p = in + (size_t)r * Win + 2 * x;
        out[t] = (p[0] + p[1]) * 0.5f;
    }
}

// Pyramid levels 1..NL from level 0 in ONE pass (W0 % 2^NL == 0, 16-B aligned): a lane owns 2^NL consecutive level-0
// values (16-B loads), pools them NL times in registers -- each level from the already rounded previous one, exactly
// like NL successive avg_pool calls -- and writes every level.  Level 0 is read once instead of every level twice.
template <int NL>
__global__ __launch_bounds__(256) void avgpool_pyramid_kernel(const float *__restrict__ l0, PyrOut pyr, int W0,
                                                              long long ngroups)
{
    constexpr int G0 = 1 << NL;  // level-0 values per lane
    const int gpr = W0 / G0;     // groups per row
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < ngroups;
         t += (long long)gridDim.x * blockDim.x) {
        const long long row = t / gpr;
        const int gx = (int)(t - row * gpr);
        float v[G0];
        const float *src = l0 + (size_t)row * W0 + (size_t)gx * G0;
#pragma unroll
        for (int q = 0; q < G0 / 4; ++q) {
            const f32x4 x = ld4(src + 4 * q);
            v[4 * q + 0] = x[0], v[4 * q + 1] = x[1], v[4 * q + 2] = x[2], v[4 * q + 3] = x[3];
        }
#pragma unroll
        for (int l = 1; l <= NL; ++l) {
            const int n = G0 >> l;  // values of level l owned by this lane
#pragma unroll
            for (int k = 0; k < n; ++k) v[k] = (v[2 * k] + v[2 * k + 1]) * 0.5f;
            float *dst = pyr.p[l] + (size_t)row * pyr.w[l] + (size_t)gx * n;
            if (n >= 4) {
#pragma unroll
                for (int q = 0; q < n / 4; ++q) st4(dst + 4 * q, (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
            } else {
#pragma unroll
                for (int k = 0; k < n; ++k) dst[k] = v[k];
            }
        }
    }
}

struct LevelPtrs {
    const float *p[STEREO_MAX_LEVELS];
    int w[STEREO_MAX_LEVELS];
};

// 1-D linear tap, zeros padding, through bilinear_sampler's normalise
// (utils/utils.py:64-66) and grid_sample's align_corners=True un-normalise.
__device__ __forceinline__ float lerp_zero_pad_ac(const float *__restrict__ row, int Wl, float x)
{
    const float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, x), (float)(Wl - 1)), 1.0f);
    const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xg, 1.0f), 2.0f), (float)(Wl - 1));
    const float fx = floorf(ix);
    const float w1 = ix - fx, w0 = 1.0f - w1;
    float acc = 0.0f;
    if (fx >= 0.0f && fx <= (float)(Wl - 1)) acc += row[(int)fx] * w0;
    if (fx + 1.0f >= 0.0f && fx + 1.0f <= (float)(Wl - 1)) acc += row[(int)fx + 1] * w1;
    return acc;
}

// A9: CorrBlock1D.__call__ (corr.py:78-101).  One thread per (pixel, level) -- blockIdx.y is the level -- so that a
// pixel's 2r+1 taps (which fall into one or two cache lines of its correlation row) are fetched by one lane with all
// loads issued before the first use (clamped index + masked weight instead of a branch per tap), and consecutive
// lanes write consecutive pixels of each output plane.  R > 0: compile-time radius (taps unrolled).
template <int R>
__global__ __launch_bounds__(256) void corr_lookup_kernel(LevelPtrs lv, const float *__restrict__ coords,
                                                          float *__restrict__ out, int L, int r_rt, long long HW,
                                                          long long npix, long long bstride, long long cstride,
                                                          long long pstride, int cpad)
{
    // output element (b, channel c, pixel p) sits at b bstride + c cstride + p pstride: NCHW (L T HW, HW, 1) or channels-last
    // with `cpad` >= L T channels per pixel (HW cpad, 1, cpad; the padding channels are zeroed by the level-0 threads)
    const int r = R > 0 ? R : r_rt;
    const int T = 2 * r + 1;
    const int i = blockIdx.y;
    const float *__restrict__ base = lv.p[i];
    const int Wl = lv.w[i];
    const float wmax = (float)(Wl - 1);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float xl = coords[(size_t)b * 2 * HW + p] / (float)(1 << i);
        const float *row = base + (size_t)t * Wl;
        float *o = out + (size_t)b * bstride + (size_t)(i * T) * cstride + (size_t)p * pstride;
        if (i == 0)
            for (int c = L * T; c < cpad; ++c) out[(size_t)b * bstride + (size_t)c * cstride + (size_t)p * pstride] = 0.0f;
        if (R > 0) {
            float v0[2 * R + 1], v1[2 * R + 1], w0[2 * R + 1], w1[2 * R + 1];
#pragma unroll
            for (int k = 0; k < 2 * R + 1; ++k) {
                // bilinear_sampler's normalise (utils/utils.py:64-66) + grid_sample's align_corners=True un-normalise
                const float x = __fadd_rn((float)(k - R), xl);
                const float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, x), wmax), 1.0f);
                const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xg, 1.0f), 2.0f), wmax);
                const float fx = floorf(ix);
                const float a = ix - fx;
                const bool ok0 = fx >= 0.0f && fx <= wmax, ok1 = fx + 1.0f >= 0.0f && fx + 1.0f <= wmax;
                const int i0 = (int)fminf(fmaxf(fx, 0.0f), wmax), i1 = (int)fminf(fmaxf(fx + 1.0f, 0.0f), wmax);
                v0[k] = row[i0];
                v1[k] = row[i1];
                w0[k] = ok0 ? 1.0f - a : 0.0f;
                w1[k] = ok1 ? a : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 2 * R + 1; ++k) o[(size_t)k * cstride] = v0[k] * w0[k] + v1[k] * w1[k];
        } else {
            for (int k = 0; k < T; ++k) o[(size_t)k * cstride] = lerp_zero_pad_ac(row, Wl, __fadd_rn((float)(k - r), xl));
        }
    }
}

// The same lookup written CHANNELS-LAST with `cpad` channels per pixel (the motion encoder's 1x1 convolution takes its input that
// way).  A workgroup = 64 pixels x 4 levels (thread = (level, pixel): the same per-thread work and arithmetic as above); the
// 64 x cpad tile is assembled in LDS and leaves as ONE contiguous block of 16-B stores -- nine 4-byte stores per thread at a
// 192-B pixel stride measured 137 us for [8,36,136,240] against 51 us for the planar kernel.
template <int R>
__global__ __launch_bounds__(256) void corr_lookup_nhwc_kernel(LevelPtrs lv, const float *__restrict__ coords,
                                                               float *__restrict__ out, long long HW, long long npix, int cpad)
{
    constexpr int T = 2 * R + 1, L = 4, PX = 64;
    extern __shared__ float tile[];                 // [PX][cpad]
    const int i = threadIdx.x >> 6, lp = threadIdx.x & 63;
    const long long t0 = (long long)blockIdx.x * PX, t = t0 + lp;
    for (int k = threadIdx.x; k < PX * (cpad - L * T); k += 256) {     // the padding channels
        const int px = k / (cpad - L * T), c = L * T + k % (cpad - L * T);
        tile[px * cpad + c] = 0.0f;
    }
    if (t < npix) {
        const float *__restrict__ base = lv.p[i];
        const int Wl = lv.w[i];
        const float wmax = (float)(Wl - 1);
        const long long b = t / HW, p = t - b * HW;
        const float xl = coords[(size_t)b * 2 * HW + p] / (float)(1 << i);
        const float *row = base + (size_t)t * Wl;
        float v0[T], v1[T], w0[T], w1[T];
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const float x = __fadd_rn((float)(k - R), xl);
            const float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, x), wmax), 1.0f);
            const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xg, 1.0f), 2.0f), wmax);
            const float fx = floorf(ix);
            const float a = ix - fx;
            const bool ok0 = fx >= 0.0f && fx <= wmax, ok1 = fx + 1.0f >= 0.0f && fx + 1.0f <= wmax;
            const int i0 = (int)fminf(fmaxf(fx, 0.0f), wmax), i1 = (int)fminf(fmaxf(fx + 1.0f, 0.0f), wmax);
            v0[k] = row[i0];
            v1[k] = row[i1];
            w0[k] = ok0 ? 1.0f - a : 0.0f;
            w1[k] = ok1 ? a : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < T; ++k) tile[lp * cpad + i * T + k] = v0[k] * w0[k] + v1[k] * w1[k];
    }
    __syncthreads();
    const long long nvalid = npix - t0 < PX ? npix - t0 : PX;
    float *ob = out + (size_t)t0 * cpad;
    for (int q = threadIdx.x; q < (int)(nvalid * cpad / 4); q += 256) st4(ob + 4 * q, ld4(tile + 4 * q));
}

// A10: PytorchAlternateCorrBlock1D.__call__ (corr.py:8-51).  One thread per
// (b, level*tap, h, w); loops over channels.  lv.p[i] = fmap2 pooled i times.
__global__ __launch_bounds__(256) void alt_corr_lookup_kernel(const float *__restrict__ f1, LevelPtrs lv,
                                                              const float *__restrict__ coords,
                                                              float *__restrict__ out, int C, int H, int W, int L, int r,
                                                              float sqrt_c, long long nout)
{
    const int T = 2 * r + 1;
    const long long HW = (long long)H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nout;
         t += (long long)gridDim.x * blockDim.x) {
        const long long p = t % HW;
        long long q = t / HW;
        const int lt = (int)(q % (L * T));
        const long long b = q / (L * T);
        const int i = lt / T, k = lt - i * T;
        const int Wl = lv.w[i];
        const float x = coords[(size_t)b * 2 * HW + p], y = coords[(size_t)b * 2 * HW + HW + p];
        const float px = __fadd_rn(x / (float)(1 << i), (float)(k - r));
        const float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, px), (float)(Wl - 1)), 1.0f);
        const float yg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, y), (float)(H - 1)), 1.0f);
        const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xg, 1.0f), 2.0f), (float)(Wl - 1));
        const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(yg, 1.0f), 2.0f), (float)(H - 1));
        const float fx = floorf(ix), fy = floorf(iy);
        const float ax = ix - fx, ay = iy - fy;
        const int x0 = (int)fx, y0 = (int)fy;
        const bool vx0 = fx >= 0.0f && fx <= (float)(Wl - 1), vx1 = fx + 1.0f >= 0.0f && fx + 1.0f <= (float)(Wl - 1);
        const bool vy0 = fy >= 0.0f && fy <= (float)(H - 1), vy1 = fy + 1.0f >= 0.0f && fy + 1.0f <= (float)(H - 1);
        const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
        const float *img = lv.p[i] + (size_t)b * C * H * (size_t)Wl;
        const float *q1 = f1 + (size_t)b * C * HW + p;
        float dot = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float *im = img + (size_t)c * H * (size_t)Wl;
            float s = 0.0f;
            if (vy0 && vx0) s += im[(size_t)y0 * Wl + x0] * w00;
            if (vy0 && vx1) s += im[(size_t)y0 * Wl + x0 + 1] * w01;
            if (vy1 && vx0) s += im[(size_t)(y0 + 1) * Wl + x0] * w10;
            if (vy1 && vx1) s += im[(size_t)(y0 + 1) * Wl + x0 + 1] * w11;
            dot = fmaf(s, q1[(size_t)c * HW], dot);
        }
        out[t] = dot / sqrt_c;
    }
}

}  // namespace

// levels 1..num_levels from level 0 (shared by the fp32 and fp16 entry points)
static int pool_pyramid(float *const *levels, int B, int H, int W1, int W2, int num_levels, hipStream_t st)
{
    int first_unpooled = 1;  // first level still to be produced by the single-level pooling kernel
    const long long rows = (long long)B * H * W1;
    if (num_levels >= 2) {  // levels 1..NL in one pass over level 0
        const int NL = num_levels >= 4 ? 4 : num_levels >= 3 ? 3 : 2;
        bool ok = W2 % (1 << NL) == 0 && aligned16(levels[0]);
        for (int i = 1; i <= NL; ++i) ok = ok && aligned16(levels[i]);
        // 16-B stores of level l need (W2 >> l) * 4 B rows to keep 16-B alignment per lane group: W2 % 2^NL == 0 gives
        // a lane (2^NL >> l) values at offset gx * (2^NL >> l); st4 is used only when that count is a multiple of 4
        for (int i = 1; i <= NL; ++i)
            if (((1 << NL) >> i) >= 4 && (W2 >> i) % 4 != 0) ok = false;
        if (ok) {
            PyrOut pyr;
            for (int i = 0, w = W2; i <= NL; ++i, w /= 2) pyr.p[i] = levels[i], pyr.w[i] = w;
            const long long ng = rows * (W2 >> NL);
            if (NL == 4) hipLaunchKernelGGL(avgpool_pyramid_kernel<4>, dim3(grid_for(ng)), dim3(256), 0, st, levels[0], pyr, W2, ng);
            else if (NL == 3) hipLaunchKernelGGL(avgpool_pyramid_kernel<3>, dim3(grid_for(ng)), dim3(256), 0, st, levels[0], pyr, W2, ng);
            else hipLaunchKernelGGL(avgpool_pyramid_kernel<2>, dim3(grid_for(ng)), dim3(256), 0, st, levels[0], pyr, W2, ng);
            STEREO_LAUNCH_CHECK();
            first_unpooled = NL + 1;
        }
    }
    for (int i = 1, w = W2; i <= num_levels; ++i) {
        const int wo = w / 2;
        if (wo == 0) break;
        if (i >= first_unpooled) {
            const long long nout = rows * wo;
            hipLaunchKernelGGL(avgpool_w2_kernel, dim3(grid_for(nout)), dim3(256), 0, st, levels[i - 1], levels[i], w, wo,
                               nout);
            STEREO_LAUNCH_CHECK();
        }
        w = wo;
    }
    return STEREO_OK;
}

static int allpairs_args_check(const void *fmap1, const void *fmap2, float *const *levels, int B, int C, int H, int W1,
                               int W2, int num_levels)
{
    if (B < 0 || C < 0 || H < 0 || W1 < 0 || W2 < 0 || num_levels < 0) return STEREO_ERR_BAD_SHAPE;
    if (num_levels + 1 > STEREO_MAX_LEVELS) return STEREO_ERR_UNSUPPORTED;
    if (B == 0 || H == 0 || W1 == 0 || W2 == 0) return 1;  // nothing to do
    STEREO_REQUIRE_PTR(fmap1);
    STEREO_REQUIRE_PTR(fmap2);
    STEREO_REQUIRE_PTR(levels);
    for (int i = 0, w = W2; i <= num_levels; ++i, w /= 2)
        if (w > 0) STEREO_REQUIRE_PTR(levels[i]);
    if (!fits_grid_yz((long long)B * H) || !fits_grid_yz((W1 + 127) / 128)) return STEREO_ERR_UNSUPPORTED;
    return STEREO_OK;
}

template <typename TIN>
static int allpairs_fast(const TIN *fmap1, const TIN *fmap2, float *level0, int B, int C, int H, int W1, int W2,
                         hipStream_t st)
{
    if (!(W1 % 4 == 0 && W2 % 4 == 0 && aligned_quad<TIN>(fmap1) && aligned_quad<TIN>(fmap2) &&
          (long long)W1 * W2 < (1LL << 31)))
        return STEREO_ERR_UNSUPPORTED;
    const float sq = sqrtf((float)C);
    int e2 = 0;
    const int mul_exact = frexpf(sq, &e2) == 0.5f ? 1 : 0;  // sqrt(C) a power of two: x * (1/sqrt) == x / sqrt
    dim3 grid((W2 + 127) / 128, (W1 + 127) / 128, (unsigned)(B * H));
    if (stereo_opt(OPT_CORR_F16P)) {   // (the two pre-scales of 16: powers of two, folded into the epilogue's scale exactly)
        constexpr float U = sizeof(TIN) == 4 ? 256.0f : 1.0f;
        hipLaunchKernelGGL((allpairs_lds_kernel<TIN, true>), grid, dim3(256), 0, st, fmap1, fmap2, level0, C, H, W1, W2, sq * U,
                           1.0f / (sq * U), mul_exact);
    } else {
        hipLaunchKernelGGL((allpairs_lds_kernel<TIN, false>), grid, dim3(256), 0, st, fmap1, fmap2, level0, C, H, W1, W2, sq,
                           1.0f / sq, mul_exact);
    }
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_allpairs_pyramid_f32(const float *fmap1, const float *fmap2, float *const *levels, int B, int C,
                                           int H, int W1, int W2, int num_levels, void *stream)
{
    const int chk = allpairs_args_check(fmap1, fmap2, levels, B, C, H, W1, W2, num_levels);
    if (chk != STEREO_OK) return chk > 0 ? STEREO_OK : chk;
    hipStream_t st = as_stream(stream);
    int rc = allpairs_fast<float>(fmap1, fmap2, levels[0], B, C, H, W1, W2, st);
    if (rc == STEREO_ERR_UNSUPPORTED) {
        dim3 grid((W2 + 127) / 128, (W1 + 127) / 128, (unsigned)(B * H));
        hipLaunchKernelGGL(allpairs_kernel, grid, dim3(256), 0, st, fmap1, fmap2, levels[0], C, H, W1, W2,
                           sqrtf((float)C));
        STEREO_LAUNCH_CHECK();
        rc = STEREO_OK;
    }
    if (rc != STEREO_OK) return rc;
    return pool_pyramid(levels, B, H, W1, W2, num_levels, st);
}

// A8 with fp16 features: fmap1 / fmap2 [B,C,H,W] fp16, pyramid fp32.  W1 % 4 == 0 && W2 % 4 == 0 and
// 8-B aligned bases, else STEREO_ERR_UNSUPPORTED (the host layer then widens the features itself).
extern "C" int stereo_allpairs_pyramid_f16(const void *fmap1, const void *fmap2, float *const *levels, int B, int C,
                                           int H, int W1, int W2, int num_levels, void *stream)
{
    const int chk = allpairs_args_check(fmap1, fmap2, levels, B, C, H, W1, W2, num_levels);
    if (chk != STEREO_OK) return chk > 0 ? STEREO_OK : chk;
    hipStream_t st = as_stream(stream);
    const int rc = allpairs_fast<f16>(static_cast<const f16 *>(fmap1), static_cast<const f16 *>(fmap2), levels[0], B, C, H,
                                      W1, W2, st);
    if (rc != STEREO_OK) return rc;
    return pool_pyramid(levels, B, H, W1, W2, num_levels, st);
}

static int corr_lookup(const float *const *levels, const float *coords, float *out, int B, int H, int W1, int W2, int num_levels,
                       int radius, int nhwc_channels, void *stream)
{
    if (B < 0 || H < 0 || W1 < 0 || W2 < 0 || num_levels < 0 || radius < 0) return STEREO_ERR_BAD_SHAPE;
    if (num_levels > STEREO_MAX_LEVELS) return STEREO_ERR_UNSUPPORTED;
    if (nhwc_channels != 0 && nhwc_channels < num_levels * (2 * radius + 1)) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W1 == 0 || num_levels == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(levels);
    STEREO_REQUIRE_PTR(coords);
    STEREO_REQUIRE_PTR(out);
    LevelPtrs lv;
    for (int i = 0, w = W2; i < num_levels; ++i, w /= 2) {
        if (w < 2) return STEREO_ERR_UNSUPPORTED;  // 2x/(W-1) is undefined (reference yields NaN/inf)
        STEREO_REQUIRE_PTR(levels[i]);
        lv.p[i] = levels[i];
        lv.w[i] = w;
    }
    const long long HW = (long long)H * W1, n = (long long)B * HW;
    const long long LT = (long long)num_levels * (2 * radius + 1);
    const long long bs = nhwc_channels ? HW * nhwc_channels : LT * HW, cs = nhwc_channels ? 1 : HW, ps = nhwc_channels ? nhwc_channels : 1;
    const dim3 grid(grid_for(n), (unsigned)num_levels);
    if (nhwc_channels && radius == 4 && num_levels == 4 && nhwc_channels % 4 == 0 && nhwc_channels <= 64 && aligned16(out) &&
        (n + 63) / 64 < (1LL << 31)) {
        hipLaunchKernelGGL(corr_lookup_nhwc_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), (size_t)64 * nhwc_channels * 4,
                           as_stream(stream), lv, coords, out, HW, n, nhwc_channels);
        STEREO_LAUNCH_CHECK();
        return STEREO_OK;
    }
    if (radius == 4)
        hipLaunchKernelGGL(corr_lookup_kernel<4>, grid, dim3(256), 0, as_stream(stream), lv, coords, out, num_levels,
                           radius, HW, n, bs, cs, ps, nhwc_channels);
    else
        hipLaunchKernelGGL(corr_lookup_kernel<0>, grid, dim3(256), 0, as_stream(stream), lv, coords, out, num_levels,
                           radius, HW, n, bs, cs, ps, nhwc_channels);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_corr_lookup_f32(const float *const *levels, const float *coords, float *out, int B, int H, int W1,
                                      int W2, int num_levels, int radius, void *stream)
{
    return corr_lookup(levels, coords, out, B, H, W1, W2, num_levels, radius, 0, stream);
}

extern "C" int stereo_corr_lookup_nhwc_f32(const float *const *levels, const float *coords, float *out, int B, int H, int W1,
                                           int W2, int num_levels, int radius, int out_channels, void *stream)
{
    if (out_channels < 1) return STEREO_ERR_BAD_SHAPE;
    return corr_lookup(levels, coords, out, B, H, W1, W2, num_levels, radius, out_channels, stream);
}

// The same with one thread per (b, level, pixel) computing ALL 2 r + 1 taps in one pass over the channels: the taps of a pixel
// sit on consecutive columns of the pooled feature row (px differs by exactly 1 between taps), so a channel costs the thread one
// fmap1 value and 2 r + 2 consecutive fmap2 values per sampled row instead of 4 gathers and one fmap1 value PER TAP -- lanes are
// consecutive pixels, so the column reads of a wave are contiguous wherever the disparity is locally smooth.  Every per-tap
// coordinate / weight is computed with the reference's rounding sequence exactly as above; a thread whose taps are not on
// consecutive columns (possible only through rounding at huge coordinates) takes the per-tap gathers.
template <int R>
__global__ __launch_bounds__(256) void alt_corr_lookup_taps_kernel(const float *__restrict__ f1, LevelPtrs lv,
                                                                   const float *__restrict__ coords, float *__restrict__ out,
                                                                   int C, int H, int W, int L, float sqrt_c, long long nitems)
{
    // workgroup = 64 (b, level, pixel) items x 4 channel slices (one wave per slice: lanes stay consecutive pixels); the four
    // partial dot products of an item meet in LDS
    constexpr int T = 2 * R + 1;
    __shared__ float part[3][T][64];
    const long long HW = (long long)H * W;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    for (long long t0 = (long long)blockIdx.x * 64; t0 < nitems; t0 += (long long)gridDim.x * 64) {
        const long long t = t0 + lane;
        const bool live = t < nitems;
        const long long tt = live ? t : nitems - 1;
        const long long p = tt % HW;
        long long q = tt / HW;
        const int i = (int)(q % L);
        const long long b = q / L;
        const int Wl = lv.w[i];
        const float x = coords[(size_t)b * 2 * HW + p], y = coords[(size_t)b * 2 * HW + HW + p];
        const float yg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, y), (float)(H - 1)), 1.0f);
        const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(yg, 1.0f), 2.0f), (float)(H - 1));
        const float fy = floorf(iy), ay = iy - fy;
        const int y0 = (int)fminf(fmaxf(fy, -4.0f), (float)H + 4.0f);
        const bool vy0 = fy >= 0.0f && fy <= (float)(H - 1), vy1 = fy + 1.0f >= 0.0f && fy + 1.0f <= (float)(H - 1);
        int x0[T];
        float w00[T], w01[T], w10[T], w11[T];
        bool consec = true;
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const float px = __fadd_rn(x / (float)(1 << i), (float)(k - R));
            const float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, px), (float)(Wl - 1)), 1.0f);
            const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xg, 1.0f), 2.0f), (float)(Wl - 1));
            const float fx = floorf(ix), ax = ix - fx;
            // clamp before the int conversion (coordinates far outside the row): such taps are invalid anyway
            x0[k] = (int)fminf(fmaxf(fx, -4.0f), (float)Wl + 4.0f);
            w00[k] = (1.0f - ax) * (1.0f - ay);
            w01[k] = ax * (1.0f - ay);
            w10[k] = (1.0f - ax) * ay;
            w11[k] = ax * ay;
            consec = consec && x0[k] == x0[0] + k;
        }
        const int cper = (C + 3) / 4, c_lo = slice * cper, c_hi = c_lo + cper < C ? c_lo + cper : C;
        const size_t cstride = (size_t)H * (size_t)Wl;
        const float *img = lv.p[i] + (size_t)b * C * cstride + (size_t)c_lo * cstride;
        const float *q1 = f1 + (size_t)b * C * HW + (size_t)c_lo * HW + p;
        float dot[T];
#pragma unroll
        for (int k = 0; k < T; ++k) dot[k] = 0.0f;
        if (consec) {
            const int xb = x0[0];
            bool vcol[T + 1];
#pragma unroll
            for (int j = 0; j <= T; ++j) vcol[j] = xb + j >= 0 && xb + j <= Wl - 1;
            const float *r0 = img + (size_t)(vy0 ? y0 : 0) * Wl, *r1 = img + (size_t)(vy1 ? y0 + 1 : 0) * Wl;
#pragma unroll 2
            for (int c = c_lo; c < c_hi; ++c) {
                const float qv = *q1;
                float a[T + 1], d[T + 1];
#pragma unroll
                for (int j = 0; j <= T; ++j) {
                    a[j] = (vy0 && vcol[j]) ? r0[xb + j] : 0.0f;
                    d[j] = (vy1 && vcol[j]) ? r1[xb + j] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    const float sv = a[k] * w00[k] + a[k + 1] * w01[k] + d[k] * w10[k] + d[k + 1] * w11[k];
                    dot[k] = fmaf(sv, qv, dot[k]);
                }
                r0 += cstride;
                r1 += cstride;
                q1 += HW;
            }
        } else {
            for (int c = c_lo; c < c_hi; ++c) {
                const float *im = img + (size_t)(c - c_lo) * cstride;
                const float qv = q1[(size_t)(c - c_lo) * HW];
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    const bool vx0 = x0[k] >= 0 && x0[k] <= Wl - 1, vx1 = x0[k] + 1 >= 0 && x0[k] + 1 <= Wl - 1;
                    float sv = 0.0f;
                    if (vy0 && vx0) sv += im[(size_t)y0 * Wl + x0[k]] * w00[k];
                    if (vy0 && vx1) sv += im[(size_t)y0 * Wl + x0[k] + 1] * w01[k];
                    if (vy1 && vx0) sv += im[(size_t)(y0 + 1) * Wl + x0[k]] * w10[k];
                    if (vy1 && vx1) sv += im[(size_t)(y0 + 1) * Wl + x0[k] + 1] * w11[k];
                    dot[k] = fmaf(sv, qv, dot[k]);
                }
            }
        }
        __syncthreads();   // (the previous item's partials have been consumed)
        if (slice > 0) {
#pragma unroll
            for (int k = 0; k < T; ++k) part[slice - 1][k][lane] = dot[k];
        }
        __syncthreads();
        if (slice == 0 && live) {
            float *o = out + ((size_t)b * L * T + (size_t)i * T) * HW + p;
#pragma unroll
            for (int k = 0; k < T; ++k)
                o[(size_t)k * HW] = (((dot[k] + part[0][k][lane]) + part[1][k][lane]) + part[2][k][lane]) / sqrt_c;
        }
    }
}

extern "C" long long stereo_alt_corr_workspace_floats(int B, int C, int H, int W, int num_levels)
{
    long long total = 0;
    for (int i = 1, w = W; i < num_levels; ++i) {
        w /= 2;
        total += (long long)B * C * H * w;
    }
    return total;
}

extern "C" int stereo_alt_corr_lookup_f32(const float *fmap1, const float *fmap2, const float *coords, float *out,
                                          float *workspace, int B, int C, int H, int W, int num_levels, int radius,
                                          void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || num_levels < 0 || radius < 0) return STEREO_ERR_BAD_SHAPE;
    if (num_levels > STEREO_MAX_LEVELS) return STEREO_ERR_UNSUPPORTED;
    if (B == 0 || H == 0 || W == 0 || num_levels == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(fmap1);
    STEREO_REQUIRE_PTR(fmap2);
    STEREO_REQUIRE_PTR(coords);
    STEREO_REQUIRE_PTR(out);
    if (num_levels > 1 && stereo_alt_corr_workspace_floats(B, C, H, W, num_levels) > 0) STEREO_REQUIRE_PTR(workspace);
    hipStream_t st = as_stream(stream);
    LevelPtrs lv;
    lv.p[0] = fmap2;
    lv.w[0] = W;
    float *ws = workspace;
    const long long rows = (long long)B * C * H;
    for (int i = 1; i < num_levels; ++i) {
        const int wo = lv.w[i - 1] / 2;
        if (wo < 2) return STEREO_ERR_UNSUPPORTED;
        const long long nout = rows * wo;
        if (nout > 0) {
            hipLaunchKernelGGL(avgpool_w2_kernel, dim3(grid_for(nout)), dim3(256), 0, st, lv.p[i - 1], ws, lv.w[i - 1],
                               wo, nout);
            STEREO_LAUNCH_CHECK();
        }
        lv.p[i] = ws;
        lv.w[i] = wo;
        ws += nout;
    }
    if (W < 2 || H < 2) return STEREO_ERR_UNSUPPORTED;  // corr.py:20-21 divides by (W-1),(H-1)
    const int T = 2 * radius + 1;
    const long long nthr = (long long)B * num_levels * H * W;
#define ALT_TAPS(RV)                                                                                                          \
    hipLaunchKernelGGL(alt_corr_lookup_taps_kernel<RV>, dim3(grid_for(nthr * 4)), dim3(256), 0, st, fmap1, lv, coords, out, C, H, \
                       W, num_levels, sqrtf((float)C), nthr)
    if (radius == 4) ALT_TAPS(4);        // the reference's corr_radius (raft_stereo.py: corr_radius = 4)
    else if (radius == 3) ALT_TAPS(3);
    else if (radius == 2) ALT_TAPS(2);
    else if (radius == 1) ALT_TAPS(1);
    else {
        const long long nout = nthr * T;
        hipLaunchKernelGGL(alt_corr_lookup_kernel, dim3(grid_for(nout)), dim3(256), 0, st, fmap1, lv, coords, out, C, H, W,
                           num_levels, radius, sqrtf((float)C), nout);
    }
#undef ALT_TAPS
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
