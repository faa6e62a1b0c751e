// Epipolar cross-attention (A7).  BUILD-DEFINED: the reference body is an empty
// file (models/CrossAttentionCostVolume/cross_attention_cost_volume.py); the
// semantics are fixed by identities against A1/A4/A8:
//   S[w,v] = <L[:,h,w], R[:,h,v]> * scale  on the band 0 <= w-v < D
//   P = softmax over the band ('zero': out-of-image entries keep logit 0 and
//   take part, exactly like the zero-filled volume of A1 fed to A4; 'neginf':
//   they are excluded);  disp[w] = sum_d P[d]*d;  out[:,w] = sum_d P[d]*V[:,w-d].
//
// v1 (this file): one workgroup per (row, 64 queries); the band logits live in
// LDS ([D][64] floats), features are staged in 32-channel chunks, and the
// softmax never leaves the CU (the [B,H,W,D] score tensor is not materialised).
#include "common.h"

namespace {

constexpr int TQ = 64;   // queries per workgroup
constexpr int CK = 32;   // channels per staging round

__global__ __launch_bounds__(256) void cross_attention_kernel(const float *__restrict__ L, const float *__restrict__ R,
                                                              const float *__restrict__ V, float *__restrict__ out,
                                                              float *__restrict__ disp, int C, int Cv, int H, int W,
                                                              int D, int Dpad, int mask_mode, float scale)
{
    extern __shared__ f32x4 smem4[];
    float *Ls = reinterpret_cast<float *>(smem4);  // [CK][TQ]
    const int RS = TQ + Dpad;
    float *Rs = Ls + CK * TQ;   // [CK][RS]   (re-used for the value tile)
    float *P = Rs + CK * RS;    // [Dpad][TQ] logits, then probabilities

    const int w0 = blockIdx.x * TQ, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int wq = tid % TQ, sub = tid / TQ;  // 4 sub-lanes per query
    const size_t HW = (size_t)H * W;

    for (int idx = tid; idx < Dpad * TQ; idx += 256) P[idx] = 0.0f;

    // ---- logits: P[d][w] = sum_c L[c][w0+w] * R[c][w0+w-d]
    for (int c0 = 0; c0 < C; c0 += CK) {
        const int ncc = (C - c0) < CK ? (C - c0) : CK;
        __syncthreads();
        for (int idx = tid; idx < ncc * TQ; idx += 256) {
            const int c = idx / TQ, x = idx - c * TQ;
            Ls[c * TQ + x] = (w0 + x < W) ? L[((size_t)b * C + c0 + c) * HW + (size_t)h * W + w0 + x] : 0.0f;
        }
        for (int idx = tid; idx < ncc * RS; idx += 256) {
            const int c = idx / RS, x = idx - c * RS;
            const int w = w0 - Dpad + x;
            Rs[c * RS + x] = (w >= 0 && w < W) ? R[((size_t)b * C + c0 + c) * HW + (size_t)h * W + w] : 0.0f;
        }
        __syncthreads();
        for (int d = sub; d < D; d += 256 / TQ) {
            float acc = 0.0f;
            const float *rp = Rs + Dpad + wq - d;
            for (int c = 0; c < ncc; ++c) acc = fmaf(Ls[c * TQ + wq], rp[c * RS], acc);
            P[d * TQ + wq] += acc;
        }
    }
    __syncthreads();
    // ---- masked softmax over the band, one query per lane of the first wave
    if (tid < TQ) {
        const int w = w0 + tid;
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) {
            float x = P[d * TQ + tid] * scale;
            if (w - d < 0) x = mask_mode == STEREO_MASK_ZERO ? 0.0f : -INFINITY;
            P[d * TQ + tid] = x;
            m = fmaxf(m, x);
        }
        float s = 0.0f;
        for (int d = 0; d < D; ++d) s += fast_exp(P[d * TQ + tid] - m);
        const float inv = 1.0f / s;
        float e = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float p = fast_exp(P[d * TQ + tid] - m) * inv;
            P[d * TQ + tid] = p;
            e = fmaf(p, (float)d, e);
        }
        if (disp && w < W) disp[((size_t)b * H + h) * W + w] = e;
    }
    if (out == nullptr) return;
    // ---- out[c][w] = sum_d P[d][w] * V[c][w-d]
    for (int c0 = 0; c0 < Cv; c0 += CK) {
        const int ncc = (Cv - c0) < CK ? (Cv - c0) : CK;
        __syncthreads();
        for (int idx = tid; idx < ncc * RS; idx += 256) {
            const int c = idx / RS, x = idx - c * RS;
            const int w = w0 - Dpad + x;
            Rs[c * RS + x] = (w >= 0 && w < W) ? V[((size_t)b * Cv + c0 + c) * HW + (size_t)h * W + w] : 0.0f;
        }
        __syncthreads();
        for (int c = sub; c < ncc; c += 256 / TQ) {
            float acc = 0.0f;
            const float *vp = Rs + c * RS + Dpad + wq;
            for (int d = 0; d < D; ++d) acc = fmaf(P[d * TQ + wq], vp[-d], acc);
            if (w0 + wq < W) out[((size_t)b * Cv + c0 + c) * HW + (size_t)h * W + w0 + wq] = acc;
        }
    }
}

}  // namespace

extern "C" int stereo_cross_attention_f32(const float *left, const float *right, const float *value, float *out,
                                          float *disp, int B, int C, int Cv, int H, int W, int D, int mask_mode,
                                          int scale_mode, void *stream)
{
    if (B < 0 || C < 0 || Cv < 0 || H < 0 || W < 0 || D < 0) return STEREO_ERR_BAD_SHAPE;
    if (mask_mode != STEREO_MASK_ZERO && mask_mode != STEREO_MASK_NEGINF) return STEREO_ERR_BAD_SHAPE;
    if (scale_mode != STEREO_SCALE_SQRT_C && scale_mode != STEREO_SCALE_MEAN) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(left);
    STEREO_REQUIRE_PTR(right);
    if (out) STEREO_REQUIRE_PTR(value);
    if (D == 0 || C == 0) return STEREO_ERR_BAD_SHAPE;
    if (!fits_grid_yz(H) || !fits_grid_yz(B)) return STEREO_ERR_UNSUPPORTED;
    const int Dpad = (D + 3) / 4 * 4;
    const size_t lds = ((size_t)CK * TQ + (size_t)CK * (TQ + Dpad) + (size_t)Dpad * TQ) * sizeof(float);
    if (lds > 160 * 1024) return STEREO_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(cross_attention_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return STEREO_ERR_UNSUPPORTED;
    const float scale = scale_mode == STEREO_SCALE_MEAN ? 1.0f / (float)C : 1.0f / sqrtf((float)C);
    dim3 grid((W + TQ - 1) / TQ, H, B);
    hipLaunchKernelGGL(cross_attention_kernel, grid, dim3(256), lds, as_stream(stream), left, right, value, out, disp, C,
                       Cv, H, W, D, Dpad, mask_mode, scale);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
