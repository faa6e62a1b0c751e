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


// ---------------------------------------------------------------------------
// v2: both contractions on the matrix cores (exact-f32 v_mfma_f32_32x32x2_f32).
// WG = (row, 64 queries), 4 waves.
//  phase 1  S^T[key][query] = K Q^T: A <- right-feature tile (rows = keys), B <- left tile
//           (cols = queries), accumulated over 32-channel chunks staged in LDS; each wave owns
//           one 32-query tile x four 32-key tiles (64 accumulator registers).  The band
//           entries are scattered into the LDS score sheet P[d][query], d = query - key.
//  phase 2  masked softmax over d per query on the score sheet (never leaves the CU) + disp.
//  phase 3  O^T[c][query] = V^T P: A <- value tile (rows = channels), B <- P read back along
//           the band (address 65*q - 64*key: conflict-free); waves split the key range and
//           meet in LDS; stores are 128-B W-contiguous row segments.
constexpr int A2_TQ = 64, A2_CK = 32;

// Stage ROWS x ncols floats: dst[r*dstride + x] = (r < nrows && 0 <= wstart+x < W) ? src[r*cstride + wstart + x] : 0.
// Branch-free (clamped address + select) and unrolled so that 16 global loads per lane are in flight;
// (r, x) advance incrementally: one integer division per call, none per element.
template <int ROWS>
__device__ __forceinline__ void stage_rows(float *__restrict__ dst, int dstride, const float *__restrict__ src,
                                           size_t cstride, int nrows, int wstart, int ncols, int W, int tid)
{
    const int total = ROWS * ncols;
    const int dr = 256 / ncols, dx = 256 - dr * ncols;  // advance of (r, x) per 256-element step
    int r = tid / ncols, x = tid - r * ncols;
    constexpr int U = 16;
    for (int base = tid; base < total; base += 256 * U) {
        float v[U];
        int rr[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rr[u] = r;
            xx[u] = x;
            const int w = wstart + x;
            const int wc = w < 0 ? 0 : (w >= W ? W - 1 : w);
            const int rc = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
            v[u] = (base + u * 256 < total) ? src[(size_t)rc * cstride + wc] : 0.0f;
            x += dx;
            r += dr;
            if (x >= ncols) {
                x -= ncols;
                ++r;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int w = wstart + xx[u];
            if (base + u * 256 < total) dst[rr[u] * dstride + xx[u]] = (rr[u] < nrows && w >= 0 && w < W) ? v[u] : 0.0f;
        }
    }
}

template <int NPT>  // scores per lane in the softmax phase: NPT >= ceil(D / 4)
__global__ __launch_bounds__(256) void cross_attention_mfma_kernel(const float *__restrict__ L,
                                                                   const float *__restrict__ R,
                                                                   const float *__restrict__ V,
                                                                   float *__restrict__ out, float *__restrict__ disp,
                                                                   int C, int Cv, int H, int W, int D, int Dk,
                                                                   int mask_mode, float scale)
{
    extern __shared__ f32x4 smem4[];
    const int KS = A2_TQ + Dk + 1;                 // odd row stride: row- and column-wise reads conflict-free
    const int NKT = (A2_TQ + Dk) / 32;             // key tiles covered by this WG
    float *P = reinterpret_cast<float *>(smem4);   // [D][64]
    float *Ks = P + (size_t)D * A2_TQ;             // [CK][KS]  (K chunk, later V chunk)
    float *Qs = Ks + A2_CK * KS;                   // [CK][64]  (Q chunk; later reduction scratch)
    float *red = Qs;                               // [4][64] x 2 in phase 2; [2][16][64] in phase 3

    const int w0 = blockIdx.x * A2_TQ, h = blockIdx.y, b = blockIdx.z;
    const int kb = w0 - Dk;                        // global w of key column 0
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const size_t HW = (size_t)H * W;
    const int qt = wave & 1;                       // query tile of this wave
    const int q_lo = w0 + qt * 32;                 // first query (global w) of the tile

    // ---------------- phase 1: band logits
    {
        const int kg = wave >> 1;                  // key-tile group: tiles kg*NKH .. (kg+1)*NKH-1
        const int NKH = (NKT + 1) / 2;
        f32x16 acc[4];                             // up to 4 key tiles per wave per pass
        const int kt_end = ((kg + 1) * NKH < NKT) ? (kg + 1) * NKH : NKT;  // this wave's tiles: [kg*NKH, kt_end)
        const int npass = (NKH + 3) / 4;           // identical for every wave: the loop body has barriers
        for (int pass = 0; pass < npass; ++pass) {
            const int kt0 = kg * NKH + pass * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            for (int c0 = 0; c0 < C; c0 += A2_CK) {
                const int ncc = (C - c0) < A2_CK ? (C - c0) : A2_CK;
                __syncthreads();
                stage_rows<A2_CK>(Ks, KS, R + ((size_t)b * C + c0) * HW + (size_t)h * W, HW, ncc, kb, A2_TQ + Dk, W, tid);
                stage_rows<A2_CK>(Qs, A2_TQ, L + ((size_t)b * C + c0) * HW + (size_t)h * W, HW, ncc, w0, A2_TQ, W, tid);
                __syncthreads();
#pragma unroll 4
                for (int s = 0; s < A2_CK / 2; ++s) {
                    const int c = 2 * s + half;
                    const float qv = Qs[c * A2_TQ + qt * 32 + l31];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kt = kt0 + t;
                        if (kt < kt_end) {   // wave-uniform
                            const float kv = Ks[c * KS + kt * 32 + l31];
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qv, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            // scatter the band part of each tile into the score sheet
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kt = kt0 + t;
                if (kt < kt_end) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // global w of the key
                        const int d = (q_lo + l31) - key;
                        if (d >= 0 && d < D) P[d * A2_TQ + qt * 32 + l31] = acc[t][r];
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---------------- phase 2: masked softmax over d (4 partial lanes per query), scores held in registers
    {
        const int q = tid & 63, part = tid >> 6;
        const int w = w0 + q;
        float x[NPT];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int d = part + 4 * i;
            float v = -INFINITY;
            if (d < D) {
                v = P[d * A2_TQ + q] * scale;
                if (w - d < 0) v = mask_mode == STEREO_MASK_ZERO ? 0.0f : -INFINITY;
            }
            x[i] = v;
            m = fmaxf(m, v);
        }
        red[part * 64 + q] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[q], red[64 + q]), fmaxf(red[128 + q], red[192 + q]));
        float s = 0.0f, e = 0.0f;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const float p = fast_exp(x[i] - m);   // x = -inf (d >= D or masked) -> 0
            x[i] = p;
            s += p;
            e = fmaf(p, (float)(part + 4 * i), e);
        }
        red[256 + part * 64 + q] = s;
        red[512 + part * 64 + q] = e;
        __syncthreads();
        s = (red[256 + q] + red[320 + q]) + (red[384 + q] + red[448 + q]);
        e = (red[512 + q] + red[576 + q]) + (red[640 + q] + red[704 + q]);
        const float inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int d = part + 4 * i;
            if (d < D) P[d * A2_TQ + q] = x[i] * inv;
        }
        if (part == 0 && disp && w < W) disp[((size_t)b * H + h) * W + w] = e * inv;
    }
    if (out == nullptr) return;
    // ---------------- phase 3: O^T = V^T P, 32 value channels per round
    const int kh = wave >> 1;                      // key half of this wave
    const int NKH3 = (NKT + 1) / 2;
    for (int c0 = 0; c0 < Cv; c0 += 32) {
        const int ncc = (Cv - c0) < 32 ? (Cv - c0) : 32;
        __syncthreads();                           // P final / previous round's readers done
        stage_rows<32>(Ks, KS, V + ((size_t)b * Cv + c0) * HW + (size_t)h * W, HW, ncc, kb, A2_TQ + Dk, W, tid);
        __syncthreads();
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        const int qq = qt * 32 + l31;              // query column of this lane (B operand)
        for (int kt = kh * NKH3; kt < (kh + 1) * NKH3 && kt < NKT; ++kt) {
            // skip key tiles entirely outside the band of this query tile (wave-uniform)
            const int dmin = q_lo - (kb + kt * 32 + 31), dmax = q_lo + 31 - (kb + kt * 32);
            if (dmax < 0 || dmin >= D) continue;
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                const int x = kt * 32 + 2 * s + half;          // key column in the staged tile
                const float va = Ks[l31 * KS + x];             // A[i = channel l31][k = key]
                const int d = (w0 + qq) - (kb + x);
                const float pb = (d >= 0 && d < D) ? P[d * A2_TQ + qq] : 0.0f;  // B[k = key][j = query]
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(va, pb, o, 0, 0, 0);
            }
        }
        // reduce the two key halves through LDS (Qs region: 2 x 16 x 64 floats)
        __syncthreads();
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(qt * 16 + r) * 64 + lane] = o[r];
        }
        __syncthreads();
        if (kh == 0) {
            const int w = w0 + qq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = o[r] + red[(qt * 16 + r) * 64 + lane];
                if (c < ncc && w < W) out[((size_t)b * Cv + c0 + c) * HW + (size_t)h * W + w] = v;
            }
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
    {   // v2: MFMA path whenever its LDS footprint fits
        const int Dk = (D + 31) / 32 * 32;
        const size_t lds2 = ((size_t)D * A2_TQ + (size_t)A2_CK * (A2_TQ + Dk + 1) + (size_t)A2_CK * A2_TQ) * sizeof(float);
        if (lds2 <= 160 * 1024) {
            const int npt = (D + 3) / 4;
            auto kern = npt <= 16 ? cross_attention_mfma_kernel<16>
                        : npt <= 32 ? cross_attention_mfma_kernel<32>
                        : npt <= 48 ? cross_attention_mfma_kernel<48>
                        : npt <= 64 ? cross_attention_mfma_kernel<64>
                                    : cross_attention_mfma_kernel<104>;
            if (npt > 104) return STEREO_ERR_UNSUPPORTED;
            if (lds2 > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds2) != hipSuccess)
                return STEREO_ERR_UNSUPPORTED;
            const float scale2 = scale_mode == STEREO_SCALE_MEAN ? 1.0f / (float)C : 1.0f / sqrtf((float)C);
            dim3 grid2((W + A2_TQ - 1) / A2_TQ, H, B);
            hipLaunchKernelGGL(kern, grid2, dim3(256), lds2, as_stream(stream), left, right,
                               value, out, disp, C, Cv, H, W, D, Dk, mask_mode, scale2);
            STEREO_LAUNCH_CHECK();
            return STEREO_OK;
        }
    }
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
