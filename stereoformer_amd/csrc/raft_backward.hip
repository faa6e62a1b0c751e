// Backward of the RAFT-style correlation path: pyramid lookup (A9), pooling and the
// all-pairs product (A8), plus bilinear resize (`interp`).  Gradients w.r.t. the lookup coordinates are
// not produced: the reference detaches coords1 at the top of every iteration (raft_stereo.py:171).
#include "common.h"

namespace {

inline unsigned grid_for(long long nthreads)
{
    long long blocks = (nthreads + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > (1LL << 30)) blocks = 1LL << 30;
    return (unsigned)blocks;
}

struct LevelGrads {
    float *p[STEREO_MAX_LEVELS];
    int w[STEREO_MAX_LEVELS];
};

// A9 backward: out[b, i*T+k, h, w] = lerp(level_i[pixel row], x/2^i + k - r) -> scatter the cotangent onto the
// two taps.  One lane = one pixel = one private row of every level: no cross-lane collisions.
__global__ __launch_bounds__(256) void corr_lookup_bwd_kernel(LevelGrads lv, const float *__restrict__ coords,
                                                              const float *__restrict__ gout, int L, int r,
                                                              long long HW, long long npix)
{
    const int T = 2 * r + 1;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const float x = coords[(size_t)b * 2 * HW + p];
        const float *go = gout + (size_t)b * (size_t)(L * T) * HW + p;
        for (int i = 0; i < L; ++i) {
            const int Wl = lv.w[i];
            float *row = lv.p[i] + (size_t)t * Wl;
            const float xl = x / (float)(1 << i);
            for (int k = 0; k < T; ++k) {
                const float px = __fadd_rn((float)(k - r), xl);
                const float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, px), (float)(Wl - 1)), 1.0f);
                const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xg, 1.0f), 2.0f), (float)(Wl - 1));
                const float fx = floorf(ix);
                const float w1 = ix - fx, w0 = 1.0f - w1;
                const float g = go[(size_t)(i * T + k) * HW];
                if (fx >= 0.0f && fx <= (float)(Wl - 1)) row[(int)fx] += g * w0;
                if (fx + 1.0f >= 0.0f && fx + 1.0f <= (float)(Wl - 1)) row[(int)fx + 1] += g * w1;
            }
        }
    }
}

// pooling backward, accumulated into the finer level: g_fine[r][x] += 0.5 * g_coarse[r][x/2]  (x/2 < Wc)
__global__ __launch_bounds__(256) void avgpool_w2_bwd_kernel(const float *__restrict__ gc, float *__restrict__ gf,
                                                             int Wf, int Wc, long long nfine)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nfine;
         t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / Wf;
        const int x = (int)(t - r * Wf);
        if ((x >> 1) < Wc) gf[t] += 0.5f * gc[(size_t)r * Wc + (x >> 1)];
    }
}

// Strided, batched f32 MFMA GEMM:  C[m][n] = alpha * sum_k A[m][k] * B[k][n], batch z = (b, h).
// WG tile 64 x 64 (4 waves of 32 x 32, v_mfma_f32_32x32x2_f32), K in chunks of 16 through LDS; the tile
// loaders walk whichever index of each operand is contiguous so the global reads stay coalesced.
constexpr int GK = 16;
struct GemmDesc {
    long long a_m, a_k, a_b, a_h;  // element strides of A for m, k, batch b, batch h
    long long b_k, b_n, b_b, b_h;
    long long c_m, c_n, c_b, c_h;
    int M, N, K, H;
    float alpha;
};

__global__ __launch_bounds__(256) void sgemm_strided_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                            float *__restrict__ Cm, GemmDesc d)
{
    __shared__ float As[GK][64 + 1];
    __shared__ float Bs[GK][64 + 1];
    const int z = blockIdx.z, bb = z / d.H, hh = z - bb * d.H;
    const float *a = A + bb * d.a_b + hh * d.a_h;
    const float *b = Bm + bb * d.b_b + hh * d.b_h;
    float *c = Cm + bb * d.c_b + hh * d.c_h;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int mw = (wave >> 1) * 32, nw = (wave & 1) * 32;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < d.K; k0 += GK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;  // 1024 elements per operand tile
            int m, k;
            if (d.a_k == 1) k = idx & 15, m = idx >> 4; else m = idx & 63, k = idx >> 6;
            As[k][m] = (m0 + m < d.M && k0 + k < d.K) ? a[(m0 + m) * d.a_m + (k0 + k) * d.a_k] : 0.0f;
            int n, kb;
            if (d.b_k == 1) kb = idx & 15, n = idx >> 4; else n = idx & 63, kb = idx >> 6;
            Bs[kb][n] = (n0 + n < d.N && k0 + kb < d.K) ? b[(k0 + kb) * d.b_k + (n0 + n) * d.b_n] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * s + half][mw + l31], Bs[2 * s + half][nw + l31], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + mw + (r & 3) + 8 * (r >> 2) + 4 * half, n = n0 + nw + l31;
        if (m < d.M && n < d.N) c[m * d.c_m + n * d.c_n] = d.alpha * acc[r];
    }
}

// bilinear resize backward (scatter with float atomics; the maps are small): gin must be zero-filled
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gin,
                                                                  int Hi, int Wi, int Ho, int Wo, int align_corners,
                                                                  float scale, float rh, float rw, long long nout)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nout;
         t += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(t % Wo);
        long long r = t / Wo;
        const int y = (int)(r % Ho);
        const long long nc = r / Ho;
        float sy, sx;
        if (align_corners) {
            sy = rh * (float)y;
            sx = rw * (float)x;
        } else {
            sy = __fsub_rn(__fmul_rn(rh, (float)y + 0.5f), 0.5f);
            sx = __fsub_rn(__fmul_rn(rw, (float)x + 0.5f), 0.5f);
            sy = sy < 0.0f ? 0.0f : sy;
            sx = sx < 0.0f ? 0.0f : sx;
        }
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 > Hi - 1 ? Hi - 1 : y0;
        x0 = x0 > Wi - 1 ? Wi - 1 : x0;
        const int y1 = y0 < Hi - 1 ? y0 + 1 : y0, x1 = x0 < Wi - 1 ? x0 + 1 : x0;
        const float ay = sy - (float)y0, ax = sx - (float)x0;
        const float g = scale * gout[t];
        float *im = gin + (size_t)nc * Hi * Wi;
        atomicAdd(im + (size_t)y0 * Wi + x0, g * (1.0f - ax) * (1.0f - ay));
        atomicAdd(im + (size_t)y0 * Wi + x1, g * ax * (1.0f - ay));
        atomicAdd(im + (size_t)y1 * Wi + x0, g * (1.0f - ax) * ay);
        atomicAdd(im + (size_t)y1 * Wi + x1, g * ax * ay);
    }
}

}  // namespace

extern "C" int stereo_corr_lookup_bwd_f32(float *const *grad_levels_zeroed, const float *coords, const float *grad_out,
                                          int B, int H, int W1, int W2, int num_levels, int radius, void *stream)
{
    if (B < 0 || H < 0 || W1 < 0 || W2 < 0 || num_levels < 0 || radius < 0) return STEREO_ERR_BAD_SHAPE;
    if (num_levels > STEREO_MAX_LEVELS) return STEREO_ERR_UNSUPPORTED;
    if (B == 0 || H == 0 || W1 == 0 || num_levels == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(grad_levels_zeroed);
    STEREO_REQUIRE_PTR(coords);
    STEREO_REQUIRE_PTR(grad_out);
    LevelGrads lv;
    for (int i = 0, w = W2; i < num_levels; ++i, w /= 2) {
        if (w < 2) return STEREO_ERR_UNSUPPORTED;
        STEREO_REQUIRE_PTR(grad_levels_zeroed[i]);
        lv.p[i] = grad_levels_zeroed[i];
        lv.w[i] = w;
    }
    const long long HW = (long long)H * W1, n = (long long)B * HW;
    hipLaunchKernelGGL(corr_lookup_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), lv, coords, grad_out,
                       num_levels, radius, HW, n);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_allpairs_pyramid_bwd_f32(const float *fmap1, const float *fmap2, float *const *grad_levels,
                                               float *grad_fmap1, float *grad_fmap2, int B, int C, int H, int W1,
                                               int W2, int num_levels, void *stream)
{
    // grad_levels[i] (i = 0..num_levels) are the cotangents of the pyramid levels; they are folded, coarse to
    // fine, INTO grad_levels[0] (which is therefore modified), then dF1 = G0 F2^T / sqrt(C), dF2 = G0^T F1 / sqrt(C).
    if (B < 0 || C < 0 || H < 0 || W1 < 0 || W2 < 0 || num_levels < 0) return STEREO_ERR_BAD_SHAPE;
    if (num_levels + 1 > STEREO_MAX_LEVELS) return STEREO_ERR_UNSUPPORTED;
    if (B == 0 || C == 0 || H == 0 || W1 == 0 || W2 == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(fmap1);
    STEREO_REQUIRE_PTR(fmap2);
    STEREO_REQUIRE_PTR(grad_levels);
    STEREO_REQUIRE_PTR(grad_levels[0]);
    if (!fits_grid_yz((long long)B * H) || !fits_grid_yz((C + 63) / 64)) return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    int widths[STEREO_MAX_LEVELS];
    widths[0] = W2;
    for (int i = 1; i <= num_levels; ++i) widths[i] = widths[i - 1] / 2;
    const long long rows = (long long)B * H * W1;
    for (int i = num_levels; i >= 1; --i) {
        if (widths[i] == 0 || grad_levels[i] == nullptr) continue;
        // fold level i into level i-1 (a missing intermediate cotangent is treated as zero by the caller passing
        // a zero buffer; NULL levels above the first non-NULL one are simply skipped)
        if (grad_levels[i - 1] == nullptr) return STEREO_ERR_NULL_POINTER;
        const long long nfine = rows * widths[i - 1];
        hipLaunchKernelGGL(avgpool_w2_bwd_kernel, dim3(grid_for(nfine)), dim3(256), 0, st, grad_levels[i],
                           grad_levels[i - 1], widths[i - 1], widths[i], nfine);
        STEREO_LAUNCH_CHECK();
    }
    const float alpha = 1.0f / sqrtf((float)C);
    const long long HW1 = (long long)H * W1, HW2 = (long long)H * W2;
    if (grad_fmap1) {  // dF1[c][w1] = alpha * sum_w2 F2[c][w2] * G0[w1][w2]
        GemmDesc d;
        d.a_m = HW2, d.a_k = 1, d.a_b = (long long)C * HW2, d.a_h = W2;
        d.b_k = 1, d.b_n = W2, d.b_b = (long long)H * W1 * W2, d.b_h = (long long)W1 * W2;
        d.c_m = HW1, d.c_n = 1, d.c_b = (long long)C * HW1, d.c_h = W1;
        d.M = C, d.N = W1, d.K = W2, d.H = H, d.alpha = alpha;
        dim3 grid((W1 + 63) / 64, (C + 63) / 64, (unsigned)(B * H));
        hipLaunchKernelGGL(sgemm_strided_kernel, grid, dim3(256), 0, st, fmap2, grad_levels[0], grad_fmap1, d);
        STEREO_LAUNCH_CHECK();
    }
    if (grad_fmap2) {  // dF2[c][w2] = alpha * sum_w1 F1[c][w1] * G0[w1][w2]
        GemmDesc d;
        d.a_m = HW1, d.a_k = 1, d.a_b = (long long)C * HW1, d.a_h = W1;
        d.b_k = W2, d.b_n = 1, d.b_b = (long long)H * W1 * W2, d.b_h = (long long)W1 * W2;
        d.c_m = HW2, d.c_n = 1, d.c_b = (long long)C * HW2, d.c_h = W2;
        d.M = C, d.N = W2, d.K = W1, d.H = H, d.alpha = alpha;
        dim3 grid((W2 + 63) / 64, (C + 63) / 64, (unsigned)(B * H));
        hipLaunchKernelGGL(sgemm_strided_kernel, grid, dim3(256), 0, st, fmap1, grad_levels[0], grad_fmap2, d);
        STEREO_LAUNCH_CHECK();
    }
    return STEREO_OK;
}

extern "C" int stereo_resize_bilinear_bwd_f32(const float *grad_out, float *grad_in_zeroed, int NC, int Hi, int Wi,
                                              int Ho, int Wo, int align_corners, float scale, void *stream)
{
    if (NC < 0 || Hi < 0 || Wi < 0 || Ho < 0 || Wo < 0) return STEREO_ERR_BAD_SHAPE;
    if (NC == 0 || Ho == 0 || Wo == 0) return STEREO_OK;
    if (Hi == 0 || Wi == 0) return STEREO_ERR_BAD_SHAPE;
    STEREO_REQUIRE_PTR(grad_out);
    STEREO_REQUIRE_PTR(grad_in_zeroed);
    float rh, rw;
    if (align_corners) {
        rh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f;
        rw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f;
    } else {
        rh = (float)Hi / (float)Ho;
        rw = (float)Wi / (float)Wo;
    }
    const long long nout = (long long)NC * Ho * Wo;
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(grid_for(nout)), dim3(256), 0, as_stream(stream), grad_out,
                       grad_in_zeroed, Hi, Wi, Ho, Wo, align_corners, scale, rh, rw, nout);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
