// Epipolar cross-attention (A7).  BUILD-DEFINED: the reference body is an empty
// file (models/CrossAttentionCostVolume/cross_attention_cost_volume.py); the
// semantics are fixed by identities against A1/A4/A8:
//   S[w,v] = <L[:,h,w], R[:,h,v]> * scale  on the band 0 <= w-v < D
//   P = softmax over the band ('zero': out-of-image entries keep logit 0 and
//   take part, exactly like the zero-filled volume of A1 fed to A4; 'neginf':
//   they are excluded);  disp[w] = sum_d P[d]*d;  out[:,w] = sum_d P[d]*V[:,w-d].
//
// Kernels in this file, newest last (the dispatcher at the bottom picks the newest that applies):
//   v2  f32 MFMA for both contractions, probabilities through an LDS score sheet (ragged W, D > 256).
//   v4  scores stay in the MFMA accumulators, two-pass softmax (any C / Cv / value tensor).
//   v5  v4 + online softmax + persistent prefetching workgroups (C <= 32, value = right features).
//   v6  (attention_chunked.hip) v4's arithmetic as a persistent item pipeline over channel chunks (any C / Cv / value).
//   v7  (attention_bf16.hip) v5 on the bf16 matrix cores with split operands (C <= 32, value = right features, D <= 192):
//       the default where it applies -- 1.0 vs 1.4 ms at cfg-3 at the same parity tolerances.
// The [B,H,W,D] score tensor is never materialised in HBM by any of them.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// staging helper of the LDS-score-sheet kernel (v2) below

// Stage ROWS x ncols floats (ncols % 4 == 0): dst[r*dstride + x] = (r < nrows && 0 <= wstart+x < W) ?
// src[r*cstride + wstart + x] : 0.  VEC: 16-byte global loads (W % 4 == 0, wstart % 4 == 0, aligned
// base: a quad is then entirely inside or outside the row).  Branch-free and unrolled so that U loads
// per lane are in flight; (r, x) advance incrementally: one integer division per call.
template <int ROWS, bool VEC, int NTH>
__device__ __forceinline__ void stage_rows(float *__restrict__ dst, int dstride, const float *__restrict__ src,
                                           size_t cstride, int nrows, int wstart, int ncols, int W, int tid)
{
    if (VEC) {
        const int nq = ncols >> 2;                 // quads per row
        const int total = ROWS * nq;
        const int dr = NTH / nq, dx = NTH - dr * nq;
        int r = tid / nq, x = tid - r * nq;
        constexpr int U = NTH >= 512 ? 4 : 8;
        for (int base = tid; base < total; base += NTH * U) {
            f32x4 v[U];
            int rr[U], xx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rr[u] = r;
                xx[u] = x;
                const int w = wstart + 4 * x;
                const bool ok = (base + u * NTH < total) && r < nrows && w >= 0 && w + 3 < W;
                const int wc = ok ? w : 0;
                const int rc = ok ? r : 0;
                v[u] = ld4(src + (size_t)rc * cstride + wc);
                if (!ok) v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                x += dx;
                r += dr;
                if (x >= nq) {
                    x -= nq;
                    ++r;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * NTH < total) {
                    float *d = dst + rr[u] * dstride + 4 * xx[u];
                    d[0] = v[u][0];
                    d[1] = v[u][1];
                    d[2] = v[u][2];
                    d[3] = v[u][3];
                }
            }
        }
    } else {
        const int total = ROWS * ncols;
        const int dr = NTH / ncols, dx = NTH - dr * ncols;
        int r = tid / ncols, x = tid - r * ncols;
        constexpr int U = NTH >= 512 ? 8 : 16;
        for (int base = tid; base < total; base += NTH * U) {
            float v[U];
            int rr[U], xx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rr[u] = r;
                xx[u] = x;
                const int w = wstart + x;
                const bool ok = (base + u * NTH < total) && r < nrows && w >= 0 && w < W;
                v[u] = src[ok ? (size_t)r * cstride + w : 0];
                if (!ok) v[u] = 0.0f;
                x += dx;
                r += dr;
                if (x >= ncols) {
                    x -= ncols;
                    ++r;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (base + u * NTH < total) dst[rr[u] * dstride + xx[u]] = v[u];
        }
    }
}

// ---------------------------------------------------------------------------
// v2: both contractions on the matrix cores (exact-f32 v_mfma_f32_32x32x2_f32).
// WG = (row, 64 queries), 8 waves: (query tile = wave & 1) x (key group = wave >> 1).  With dq = (query tile origin) - (key tile origin), a multiple
// of 32, only tile pairs with 0 <= dq <= Dk touch the band; their entries have d in [-31, Dk+31].
//  phase 1  S^T[key][query] = K Q^T: A <- right-feature tile (rows = keys), B <- left tile
//           (cols = queries), accumulated over 32-channel chunks staged in LDS; each wave owns one
//           32-query tile x up to four 32-key tiles (64 accumulator registers).  Every accumulator
//           register is scattered, unconditionally and with immediate offsets, into the padded LDS
//           score sheet P[32 + d][query] (32 pad rows below d=0, 32+ above d=D-1).
//  phase 2  the pad rows are zeroed; masked softmax over d per query with the scores held in
//           registers (the score sheet never leaves the CU) + disparity expectation.
//  phase 3  O^T[c][query] = V^T P: A <- value tile (rows = channels), B <- P read back along the
//           band (address 65*q - 64*key: conflict-free, immediate offsets, no bounds test thanks to
//           the zero pad rows); waves split the key range and meet in LDS; stores are 128-B
//           W-contiguous row segments.
constexpr int A2_TQ = 64, A2_CK = 32, A2_PAD = 32;
constexpr int A2_NW = 8, A2_NT = A2_NW * 64, A2_NKG = A2_NW / 2, A2_TPW = 2;  // waves, threads, key groups, tiles/wave/pass

template <int NPT, bool VEC>  // NPT: scores per lane in the softmax phase, >= ceil(D / A2_NW)
__global__ __launch_bounds__(A2_NT) void cross_attention_mfma_kernel(const float *__restrict__ L,
                                                                   const float *__restrict__ R,
                                                                   const float *__restrict__ V,
                                                                   float *__restrict__ out, float *__restrict__ disp,
                                                                   int C, int Cv, int H, int W, int D, int Dk,
                                                                   int mask_mode, float scale)
{
    extern __shared__ f32x4 smem4[];
    const int KS = A2_TQ + Dk + 1;                 // odd row stride: row- and column-wise reads conflict-free
    const int NKT = (A2_TQ + Dk) / 32;             // key tiles covered by this WG
    const int PROWS = Dk + 2 * A2_PAD;
    float *P = reinterpret_cast<float *>(smem4);   // [PROWS][64], row = 32 + d
    float *Ks = P + (size_t)PROWS * A2_TQ;         // [CK][KS]  (K chunk, later V chunk)
    float *Qs = Ks + A2_CK * KS;                   // [CK][64]  (Q chunk; later reduction scratch)
    float *red = Qs;

    const int w0 = blockIdx.x * A2_TQ, h = blockIdx.y, b = blockIdx.z;
    const int kb = w0 - Dk;                        // global w of key column 0
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const size_t HW = (size_t)H * W;
    const int qt = wave & 1;                       // query tile of this wave
    const int qq = qt * 32 + l31;                  // query column of this lane
    const int NKH = (NKT + A2_NKG - 1) / A2_NKG;
    const int kg = wave >> 1;                      // key-tile group: tiles [kg*NKH, kt_end)
    const int kt_end = ((kg + 1) * NKH < NKT) ? (kg + 1) * NKH : NKT;
    // tile pair (qt, kt) touches the band iff 0 <= dq <= Dk, dq = Dk + (qt - kt)*32
    auto in_band = [&](int kt) { const int dq = Dk + (qt - kt) * 32; return dq >= 0 && dq <= Dk; };

    // ---------------- phase 1: band logits
    {
        f32x16 acc[A2_TPW];
        const int npass = (NKH + A2_TPW - 1) / A2_TPW;           // identical for every wave: the loop body has barriers
        for (int pass = 0; pass < npass; ++pass) {
            const int kt0 = kg * NKH + pass * A2_TPW;
#pragma unroll
            for (int t = 0; t < A2_TPW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            for (int c0 = 0; c0 < C; c0 += A2_CK) {
                const int ncc = (C - c0) < A2_CK ? (C - c0) : A2_CK;
                __syncthreads();
                stage_rows<A2_CK, VEC, A2_NT>(Ks, KS, R + ((size_t)b * C + c0) * HW + (size_t)h * W, HW, ncc, kb, A2_TQ + Dk, W, tid);
                stage_rows<A2_CK, VEC, A2_NT>(Qs, A2_TQ, L + ((size_t)b * C + c0) * HW + (size_t)h * W, HW, ncc, w0, A2_TQ, W, tid);
                __syncthreads();
                const float *qp = Qs + half * A2_TQ + qq;
                const float *kp = Ks + half * KS + kt0 * 32 + l31;
#pragma unroll 4
                for (int s = 0; s < A2_CK / 2; ++s) {
                    const float qv = qp[2 * s * A2_TQ];
#pragma unroll
                    for (int t = 0; t < A2_TPW; ++t) {
                        if (kt0 + t < kt_end && in_band(kt0 + t)) {   // wave-uniform
                            const float kv = kp[2 * s * KS + t * 32];
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qv, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            // scatter: row = 32 + d, d = Dk + qq - (kt*32 + ko), ko = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
            for (int t = 0; t < A2_TPW; ++t) {
                const int kt = kt0 + t;
                if (kt < kt_end && in_band(kt)) {
                    float *pp = P + (A2_PAD + Dk + qq - kt * 32 - 4 * half) * A2_TQ + qq;
#pragma unroll
                    for (int r = 0; r < 16; ++r) pp[-((r & 3) + 8 * (r >> 2)) * A2_TQ] = acc[t][r];
                }
            }
        }
    }
    __syncthreads();
    // ---------------- phase 2: zero the pad rows; masked softmax over d (A2_NW partial lanes per query)
    {
        for (int idx = tid; idx < A2_PAD * A2_TQ; idx += A2_NT) P[idx] = 0.0f;
        for (int idx = (A2_PAD + D) * A2_TQ + tid; idx < PROWS * A2_TQ; idx += A2_NT) P[idx] = 0.0f;
        const int q = tid & 63, part = tid >> 6;
        const int w = w0 + q;
        float *pc = P + (A2_PAD + part) * A2_TQ + q;
        float x[NPT];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int d = part + A2_NW * i;
            float v = -INFINITY;
            if (d < D) {
                v = pc[A2_NW * i * A2_TQ] * scale;
                if (w - d < 0) v = mask_mode == STEREO_MASK_ZERO ? 0.0f : -INFINITY;
            }
            x[i] = v;
            m = fmaxf(m, v);
        }
        red[part * 64 + q] = m;
        __syncthreads();
        m = red[q];
#pragma unroll
        for (int k = 1; k < A2_NW; ++k) m = fmaxf(m, red[k * 64 + q]);
        float s = 0.0f, e = 0.0f;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const float p = fast_exp(x[i] - m);   // x = -inf (d >= D or masked) -> 0
            x[i] = p;
            s += p;
            e = fmaf(p, (float)(part + A2_NW * i), e);
        }
        red[(A2_NW + part) * 64 + q] = s;
        red[(2 * A2_NW + part) * 64 + q] = e;
        __syncthreads();
        s = 0.0f, e = 0.0f;
#pragma unroll
        for (int k = 0; k < A2_NW; ++k) {
            s += red[(A2_NW + k) * 64 + q];
            e += red[(2 * A2_NW + k) * 64 + q];
        }
        const float inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < NPT; ++i)
            if (part + A2_NW * i < D) pc[A2_NW * i * A2_TQ] = x[i] * inv;
        if (part == 0 && disp && w < W) disp[((size_t)b * H + h) * W + w] = e * inv;
    }
    if (out == nullptr) return;
    // ---------------- phase 3: O^T = V^T P, 32 value channels per round
    for (int c0 = 0; c0 < Cv; c0 += 32) {
        const int ncc = (Cv - c0) < 32 ? (Cv - c0) : 32;
        __syncthreads();                           // P final / previous round's readers done
        stage_rows<32, VEC, A2_NT>(Ks, KS, V + ((size_t)b * Cv + c0) * HW + (size_t)h * W, HW, ncc, kb, A2_TQ + Dk, W, tid);
        __syncthreads();
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        for (int kt = kg * NKH; kt < kt_end; ++kt) {
            if (!in_band(kt)) continue;            // wave-uniform
            const float *ap = Ks + l31 * KS + kt * 32 + half;                                   // A[i = channel][k = key]
            const float *bp = P + (A2_PAD + Dk + qq - kt * 32 - half) * A2_TQ + qq;             // B[k = key][j = query]
#pragma unroll
            for (int s = 0; s < 16; ++s)
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[-2 * s * A2_TQ], o, 0, 0, 0);
        }
        // reduce the key groups through LDS (the value tile is no longer read: reuse Ks)
        __syncthreads();
        if (kg > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Ks[(((kg - 1) * 2 + qt) * 16 + r) * 64 + lane] = o[r];
        }
        __syncthreads();
        if (kg == 0) {
            const int w = w0 + qq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = o[r];
#pragma unroll
                for (int k = 1; k < A2_NKG; ++k) v += Ks[(((k - 1) * 2 + qt) * 16 + r) * 64 + lane];
                if (c < ncc && w < W) out[((size_t)b * Cv + c0 + c) * HW + (size_t)h * W + w] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------
// v4: scores never leave the registers.  One wave owns 32 queries of a row; a workgroup is A4_NW such waves over
// consecutive queries sharing ONE LDS tile of keys (32 channels x (32*A4_NW + Dk) keys, odd row stride).
//   phase 1  S^T tile t (32 keys x 32 queries) = K_t Q^T on v_mfma_f32_32x32x2_f32: A = K from LDS (lanes = keys),
//            B = Q straight from global (lanes = queries, 128-B coalesced per channel row); NT = Dk/32 + 1 key tiles
//            cover the band of a 32-query tile, all NT accumulators (16 VGPRs each) stay live.
//   phase 2  exact two-pass softmax over the lane's 16*NT scores of its query (column j = lane & 31); the two lane
//            halves hold the two halves of every key tile, so one xor-32 exchange per reduction.
//   phase 3  O^T (32 channels x 32 queries) = V^T P: the MFMA contracts over two keys per step and the order of the
//            contraction is free, so step s takes key (s&3)+8(s>>2) from lane half 0 and that key + 4 from half 1 --
//            which is exactly how the accumulator layout holds them.  The probabilities therefore feed the B
//            operand directly from the phase-1 accumulators: no LDS score sheet, no skew transform, no barriers.
// With V == K (value = right features, C <= 32) the workgroup stages one tile and runs barrier-free after it; LDS is
// ~41 KB (Dk = 192), so several workgroups share a CU and one's staging / softmax overlaps another's MFMA phases.
constexpr int A4_NW = 4;
constexpr int A4_TQ = 32 * A4_NW;

template <int NT, int MINW>
__global__ __launch_bounds__(A4_NW * 64, MINW) void cross_attention_reg_kernel(const float *__restrict__ L,
                                                                         const float *__restrict__ R,
                                                                         const float *__restrict__ V,
                                                                         float *__restrict__ out,
                                                                         float *__restrict__ disp, int C, int Cv, int H,
                                                                         int W, int D, int mask_mode, float scale,
                                                                         int same_kv)
{
    constexpr int Dk = 32 * (NT - 1);
    constexpr int NK = A4_TQ + Dk;               // keys in the tile
    constexpr int KS = NK + 1;                   // odd row stride: conflict-free for lanes = keys AND lanes = channels
    constexpr int NQD = 32 * (NK / 4);           // 16-B quads per staged 32-channel chunk
    constexpr int NTH = A4_NW * 64;
    constexpr int NSL = (NQD + NTH - 1) / NTH;
    __shared__ float Ks[32 * KS];

    const int w0 = blockIdx.x * A4_TQ, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const size_t HW = (size_t)H * W;
    const int kbase = w0 - Dk;                   // image column of LDS column 0 (multiple of 4)
    const int q = w0 + 32 * wave + l31;          // this lane's query
    const bool wave_active = w0 + 32 * wave < W; // wave-uniform

    auto stage = [&](const float *__restrict__ src, int nch, int c0) {  // channels c0..c0+31 of row h -> Ks
        const float *base = src + ((size_t)b * nch + c0) * HW + (size_t)h * W;
        f32x4 rk[NSL];
        unsigned okm = 0;
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int qi = tid + k * NTH;
            const int row = qi / (NK / 4), xq = qi - row * (NK / 4);
            const int w = kbase + 4 * xq;
            const bool ok = qi < NQD && c0 + row < nch && w >= 0 && w + 3 < W;
            rk[k] = ld4(base + (ok ? (size_t)row * HW + w : 0));
            okm |= ok ? (1u << k) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int qi = tid + k * NTH;
            if (qi < NQD) {
                const int row = qi / (NK / 4), xq = qi - row * (NK / 4);
                const bool ok = (okm >> k) & 1u;
                float *d = Ks + row * KS + 4 * xq;
                d[0] = ok ? rk[k][0] : 0.0f;
                d[1] = ok ? rk[k][1] : 0.0f;
                d[2] = ok ? rk[k][2] : 0.0f;
                d[3] = ok ? rk[k][3] : 0.0f;
            }
        }
    };

    // ---------------- phase 1: S^T = K Q^T, accumulated over 32-channel chunks
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    for (int c0 = 0; c0 < C; c0 += 32) {
        if (c0 > 0) __syncthreads();             // the previous chunk's readers are done
        stage(R, C, c0);
        float qv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int c = c0 + 2 * s + half;
            qv[s] = (c < C && q < W) ? L[((size_t)b * C + c) * HW + (size_t)h * W + q] : 0.0f;
        }
        __syncthreads();
        if (wave_active) {
            const float *kp = Ks + half * KS + 32 * wave + l31;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s * KS + 32 * t], qv[s], acc[t], 0, 0, 0);
            }
        }
    }

    // ---------------- phase 2: masked softmax over the band, in registers
    // accumulator element r of tile t: key row i = (r&3) + 8*(r>>2) + 4*half, disparity d = Dk - 32 t + l31 - i
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = Dk - 32 * t + l31 - ((r & 3) + 8 * (r >> 2) + 4 * half);
            float v = acc[t][r] * scale;
            if (d < 0 || d >= D) v = -INFINITY;
            else if (q - d < 0) v = mask_mode == STEREO_MASK_ZERO ? 0.0f : -INFINITY;
            acc[t][r] = v;
            m = fmaxf(m, v);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32));             // d = 0 is always in the band: m is finite
    float ssum = 0.0f, esum = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = Dk - 32 * t + l31 - ((r & 3) + 8 * (r >> 2) + 4 * half);
            const float pexp = fast_exp(acc[t][r] - m);
            acc[t][r] = pexp;
            ssum += pexp;
            esum = fmaf(pexp, (float)d, esum);
        }
    }
    ssum += __shfl_xor(ssum, 32);
    esum += __shfl_xor(esum, 32);
    const float inv = 1.0f / ssum;
    if (disp && half == 0 && q < W) disp[((size_t)b * H + h) * W + q] = esum * inv;
    if (out == nullptr) return;                  // workgroup-uniform
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= inv;

    // ---------------- phase 3: O^T = V^T P, 32 value channels per round
    for (int c0 = 0; c0 < Cv; c0 += 32) {
        if (!(same_kv && c0 == 0 && C <= 32)) {  // otherwise the K tile in LDS already is the value tile
            __syncthreads();
            stage(V, Cv, c0);
            __syncthreads();
        }
        if (wave_active) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.0f;
            const float *vp = Ks + l31 * KS + 32 * wave + 4 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * t + (s & 3) + 8 * (s >> 2)], acc[t][s], o, 0, 0, 0);
            }
            if (q < W) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (c < Cv) out[((size_t)b * Cv + c) * HW + (size_t)h * W + q] = o[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// v5: v4 with an ONLINE softmax and a persistent, prefetching workgroup, for C <= 32 and value == right features (one
// LDS tile is K and V).  The key tiles of a wave's band are visited one at a time, nearest first: 16 MFMAs give the
// 32x32 score tile, the running maximum is updated (one xor-32 exchange), the output accumulator and the running sums
// are rescaled, and the tile's probabilities go straight back into 16 MFMAs of P V.  Only one score tile is live
// (~110 VGPRs instead of 256).  Exponentials in base 2 with log2(e) and the logit scale folded into Q; the disparity
// expectation uses sum p*d = cb * sum p - sum p * i_r (i_r a per-register constant).
//
// Measured on MI355X (tools/mfma_coexec.hip, and by switching the phases of this kernel off one at a time): f32 MFMA
// and VALU work do NOT overlap on a SIMD, not even across waves, and the kernel's time is the plain sum of its phases
// -- operand fetch (K tile + Q, ~1.5 GB at HBM/L2 speed), QK MFMAs, softmax VALU, PV MFMAs.  So (1) every VALU
// instruction is paid in MFMA time: packed f32 math, no index divisions, one compare + select per masked score;
// (2) the fetch phase has to be hidden explicitly: a workgroup walks `tiles_per_wg` consecutive query tiles of the
// image (row-major, so the 192-key halo it re-reads was fetched by itself a moment ago and sits in L2) and the K
// quads and Q values of the NEXT tile are loaded into registers before the current tile's MFMAs start; they are
// committed to LDS between two barriers once every wave is done with the current tile.
template <int NT, int NWV, typename TIN>
__global__ __launch_bounds__(NWV * 64) void cross_attention_online_kernel(const TIN *__restrict__ L,
                                                                          const TIN *__restrict__ R,
                                                                          float *__restrict__ out,
                                                                          float *__restrict__ disp, int C, int H, int W,
                                                                          int D, int mask_mode, float scale,
                                                                          int tiles_per_wg, int ntiles)
{
    constexpr int Dk = 32 * (NT - 1);
    constexpr int TQW = 32 * NWV;                // queries per workgroup tile
    constexpr int NK = TQW + Dk;                 // keys in the tile (multiple of 32)
    constexpr int KS = NK + 4;                   // row stride: 16-B aligned rows, KS/4 odd (K reads conflict-free,
                                                 // V reads 2-way at worst -- 4 LDS cycles under a 64-cycle MFMA)
    constexpr int RMAX = (32 + NWV - 1) / NWV;   // staged rows per wave (row = wave, wave + NWV, ...)
    constexpr int NPASS = (NK / 4 + 63) / 64;    // 16-B quads per lane and row
    __shared__ f32x4 Ks4[32 * KS / 4];
    float *Ks = reinterpret_cast<float *>(Ks4);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;  // launcher: 32 * H * W < 2^31, the lane offsets below fit 32 bits
    const int nx = (W + TQW - 1) / TQW;
    const float sc2 = scale * 1.44269504088896340736f;  // logits in units of log2, folded into Q
    const int dl = l31 - 4 * half;                      // d = (Dk - 32 t + dl) - i_r
    const float *kp = Ks + half * KS + 32 * wave + l31;
    const float *vp = Ks + l31 * KS + 32 * wave + 4 * half;

    int tile = blockIdx.x * tiles_per_wg;
    const int tile_end = (tile + tiles_per_wg < ntiles) ? tile + tiles_per_wg : ntiles;
    if (tile >= tile_end) return;

    typename Quad<TIN>::raw rk[RMAX][NPASS];  // raw quads (16 B fp32 / 8 B fp16), widened at commit
    TIN qn[16];                               // raw as well: a conversion right behind each load would serialise the 16 loads
    unsigned okbits = 0;  // validity of the prefetched quads, applied at commit time (keeps the loads in flight)
    auto issue = [&](int tl) {  // K quads and Q values of tile tl -> registers
        const int x = tl % nx, bh = tl / nx;  // bh = b * H + h
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * TQW, kbase = w0 - Dk;
        const TIN *Rb = R + ((size_t)bb * C * HWu + (size_t)hh * W);  // workgroup-uniform bases
        const TIN *Lb = L + ((size_t)bb * C * HWu + (size_t)hh * W);
        okbits = 0;
#pragma unroll
        for (int pp = 0; pp < NPASS; ++pp) {
            const int w = kbase + 4 * (lane + 64 * pp);
            if (lane + 64 * pp < NK / 4 && w >= 0 && w + 3 < W) okbits |= 1u << pp;
        }
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * NWV;  // wave-uniform
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                const bool ok = ((okbits >> pp) & 1u) && row < C && row < 32;
                rk[rr][pp] = Quad<TIN>::ld(Rb + (ok ? (unsigned)row * HWu + (unsigned)(kbase + 4 * (lane + 64 * pp)) : 0u));
            }
        }
        const int q = w0 + 32 * wave + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int c = 2 * s + half;
            qn[s] = Lb[(c < C && q < W) ? (unsigned)c * HWu + (unsigned)q : 0u];
        }
    };
    auto commit = [&]() {  // registers -> the LDS tile; one ds_write_b128 per quad
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * NWV;
            if (row < 32) {
#pragma unroll
                for (int pp = 0; pp < NPASS; ++pp) {
                    if (lane + 64 * pp < NK / 4)
                        Ks4[(row * KS) / 4 + lane + 64 * pp] =
                            (((okbits >> pp) & 1u) && row < C) ? Quad<TIN>::widen(rk[rr][pp]) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };

    issue(tile);
    commit();
    float qv[16];
    for (;;) {
        const int x = tile % nx, bh = tile / nx;
        const int w0 = x * TQW, kbase = w0 - Dk;
        const int q = w0 + 32 * wave + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int c = 2 * s + half;
            qv[s] = (c < C && q < W) ? (float)qn[s] * sc2 : 0.0f;
        }
        __syncthreads();  // the tile is visible
        const bool more = tile + 1 < tile_end;
        if (more) issue(tile + 1);  // lands under this tile's MFMAs

        if (w0 + 32 * wave < W) {  // wave-uniform
            const bool neg_left = mask_mode == STEREO_MASK_NEGINF && kbase < 0;  // workgroup-uniform
            float m = -INFINITY, ssum = 0.0f, esc = 0.0f;
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll 1
            for (int t = NT - 1; t >= 0; --t) {  // nearest keys first: the d = 0 key makes the maximum finite at once
                // (a real loop: unrolled, the compiler hoists every tile's LDS operands and the occupancy is gone)
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = 0.0f;
                const float *kpt = kp + 32 * t;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(kpt[2 * s * KS], qv[s], a, 0, 0, 0);
                const int cb = Dk - 32 * t + dl;  // d of register r = cb - i_r,  i_r = (r&3) + 8*(r>>2)
                // tile t holds d in [Dk - 32 t - 31, Dk - 32 t + 31]: only the band's end tiles can leave [0, D)
                // (wave-uniform tests); i_r is a compile-time constant, so a check is one compare + one select.
                // Out-of-image keys need nothing in 'zero' mode: the tile is zero-padded, their logit is 0.
                if (t == NT - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (cb < (r & 3) + 8 * (r >> 2)) a[r] = -INFINITY;  // d < 0
                }
                if (Dk - 32 * t + 31 >= D) {
                    const int lim = cb - D;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (lim >= (r & 3) + 8 * (r >> 2)) a[r] = -INFINITY;  // d >= D
                }
                if (neg_left) {
                    const int lim = cb - q;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (lim > (r & 3) + 8 * (r >> 2)) a[r] = -INFINITY;  // key = q - d < 0
                }
                float mt = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; r += 2) mt = fmaxf(mt, fmaxf(a[r], a[r + 1]));  // v_max3_f32
                mt = fmaxf(mt, __shfl_xor(mt, 32));
                const float mn = fmaxf(m, mt);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);  // first tile: exp2(-inf) = 0 on zeros
                m = mn;
                const f32x2 nm = {-mn, -mn};
                f32x2 T2 = {0.0f, 0.0f}, U2 = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {  // register pairs: v_pk_add / v_pk_fma, two scores per issue
                    const f32x2 sh = (f32x2){a[2 * k], a[2 * k + 1]} + nm;
                    const f32x2 pe = {__builtin_amdgcn_exp2f(sh[0]), __builtin_amdgcn_exp2f(sh[1])};
                    a[2 * k] = pe[0];
                    a[2 * k + 1] = pe[1];
                    T2 += pe;
                    const float base = (float)(2 * (k & 1) + 8 * (k >> 1));  // i_r of the pair = base, base + 1
                    U2 = __builtin_elementwise_fma(pe, (f32x2){base, base}, U2);
                }
                const float ts = T2[0] + T2[1], ti = U2[0] + U2[1] + T2[1];
                ssum = fmaf(ssum, alpha, ts);
                esc = fmaf(esc, alpha, fmaf((float)cb, ts, -ti));  // sum p*d = cb * sum p - sum p*i_r
                if (out) {
                    const f32x2 al2 = {alpha, alpha};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const f32x2 ov = (f32x2){o[2 * k], o[2 * k + 1]} * al2;
                        o[2 * k] = ov[0];
                        o[2 * k + 1] = ov[1];
                    }
                    const float *vpt = vp + 32 * t;
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        o = __builtin_amdgcn_mfma_f32_32x32x2f32(vpt[(s & 3) + 8 * (s >> 2)], a[s], o, 0, 0, 0);
                }
            }
            float e = esc;  // dl is already folded into cb
            ssum += __shfl_xor(ssum, 32);
            e += __shfl_xor(e, 32);
            const float inv = 1.0f / ssum;
            if (q < W) {
                if (disp && half == 0) disp[(size_t)bh * W + q] = e * inv;
                if (out) {
                    const int hh = bh % H, bb = bh / H;
                    float *ob = out + ((size_t)bb * C * HWu + (size_t)hh * W);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (c < C) ob[(unsigned)c * HWu + (unsigned)q] = o[r] * inv;
                    }
                }
            }
        }
        if (!more) break;
        __syncthreads();  // every wave is done reading the tile
        commit();
        ++tile;
    }
}

// v5 launcher (also behind corr_soft_argmin: A1 + A4 without any volume is this kernel with mask 'zero', scale
// +-1/C and no output tensor -- SURVEY 8(c) identity I3).  Envelope: C <= 32, D <= 256, W % 4 == 0, 16-B aligned.
template <typename TIN>
int launch_online(const TIN *left, const TIN *right, float *out, float *disp, int B, int C, int H, int W, int D,
                  int mask_mode, float sc, hipStream_t st)
{
    if (C > 32 || D > 256 || D < 1 || W % 4 != 0 || !aligned_quad<TIN>(left) || !aligned_quad<TIN>(right) ||
        (long long)32 * H * W >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
    // waves (32 queries each) per workgroup: 8 (256 queries, the halo is re-fetched 1.75x) unless the row
    // is so short that half of them would idle.  (6 waves = 192 queries divides W = 960 exactly but
    // measured slower: its register budget admits a single 6-wave workgroup per CU.)
    int nwv = W > 128 ? 8 : 4;
    if (stereo_opt(OPT_ATT_WAVES) == 4 || stereo_opt(OPT_ATT_WAVES) == 8) nwv = stereo_opt(OPT_ATT_WAVES);  // test hook
    // persistent workgroups: two resident per CU walk contiguous runs of tiles (row-major)
    const int ntl = ((W + 32 * nwv - 1) / (32 * nwv)) * H * B;
    int tpw = (ntl + 511) / 512;
    if (stereo_opt(OPT_TILES_PER_WG) > 0) tpw = stereo_opt(OPT_TILES_PER_WG);  // test hook: short persistent runs
    const int nwg = (ntl + tpw - 1) / tpw;
#define ATT5N(NTV, NWVV)                                                                                        \
    hipLaunchKernelGGL((cross_attention_online_kernel<NTV, NWVV, TIN>), dim3(nwg), dim3(NWVV * 64), 0, st, left, right, out, \
                       disp, C, H, W, D, mask_mode, sc, tpw, ntl)
#define ATT5(NTV)                    \
    do {                             \
        if (nwv == 8) ATT5N(NTV, 8); \
        else ATT5N(NTV, 4);          \
    } while (0)
    if (nt <= 3) ATT5(3);
    else if (nt <= 5) ATT5(5);
    else if (nt <= 7) ATT5(7);
    else ATT5(9);
#undef ATT5N
#undef ATT5
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
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
    {   // v4: register-resident scores (aligned, W % 4 == 0, band <= 256 after rounding up to a multiple of 32)
        const bool vec = (W % 4 == 0) && aligned16(left) && aligned16(right) && (out == nullptr || aligned16(value));
        const int forced = stereo_opt(OPT_ATT_KERNEL);  // test hook: 2 = the LDS-score-sheet kernel below, 4 = v4
        if (vec && D <= 256 && forced != 2) {
            const int nt = (D + 31) / 32 + 1;
            const float sc = scale_mode == STEREO_SCALE_MEAN ? 1.0f / (float)C : 1.0f / sqrtf((float)C);
            const int same_kv = (out != nullptr && value == right && Cv == C) ? 1 : 0;
            dim3 grid4((W + A4_TQ - 1) / A4_TQ, H, B);
            if (C <= 32 && (out == nullptr || same_kv) && (forced == 9 || forced == 0)) {  // v9: split-fp16 matrix cores (default)
                const int rc9 = stereo_cross_attention_split16<float>(left, right, out, disp, B, C, H, W, D, mask_mode, sc,
                                                                      as_stream(stream));
                if (rc9 != STEREO_ERR_UNSUPPORTED) return rc9;
            }
            if (C <= 32 && (out == nullptr || same_kv) && forced == 7) {  // v7: split-bf16 (test hook)
                const int rc7 = stereo_cross_attention_split<float>(left, right, out, disp, B, C, H, W, D, mask_mode, sc,
                                                                    as_stream(stream));
                if (rc7 != STEREO_ERR_UNSUPPORTED) return rc7;
            }
            if (C <= 32 && (out == nullptr || same_kv) && forced != 4) {  // v5: fp32 online softmax (192 < D <= 256; test hook 5)
                const int rc5 = launch_online<float>(left, right, out, disp, B, C, H, W, D, mask_mode, sc, as_stream(stream));
                if (rc5 != STEREO_ERR_UNSUPPORTED) return rc5;
            }
            // Channel-chunked item pipeline (attention_chunked.hip).  Its split-bf16 form is the default for C > 32 at bands
            // of <= 128 (0.27 vs 0.38 ms at [8,256,136,240] D=48, 0.15 vs 0.19 ms at [4,256,96,312] D=64, 0.18 vs 0.29 ms
            // at D=128 against the two-pass kernel below); its fp32-MFMA form measured 5-10 % behind the two-pass kernel and
            // is reached through the test hook (6) only.  Hook 4 forces the two-pass kernel, 8 the split form.
            if (forced == 6 || forced == 8 || (forced == 0 && nt <= 5 && C > 32)) {
                const int rc6 = stereo_cross_attention_chunked<float>(left, right, value, out, disp, B, C, Cv, H, W, D,
                                                                      mask_mode, sc, forced != 6, as_stream(stream));
                if (rc6 != STEREO_ERR_UNSUPPORTED) return rc6;
            }
#define ATT4(NTV)                                                                                                      \
    hipLaunchKernelGGL((cross_attention_reg_kernel<NTV, (NTV <= 7 ? 2 : 1)>), grid4, dim3(A4_NW * 64), 0, as_stream(stream), left, right, value, \
                       out, disp, C, Cv, H, W, D, mask_mode, sc, same_kv)
            if (nt <= 3) ATT4(3);
            else if (nt <= 5) ATT4(5);
            else if (nt <= 7) ATT4(7);
            else ATT4(9);
#undef ATT4
            STEREO_LAUNCH_CHECK();
            return STEREO_OK;
        }
    }
    {   // v2: MFMA path whenever its LDS footprint fits
        const int Dk = (D + 31) / 32 * 32;
        size_t tiles = (size_t)A2_CK * (A2_TQ + Dk + 1) + (size_t)A2_CK * A2_TQ;       // K/V chunk + Q chunk
        const size_t redsz = (size_t)(A2_NKG - 1) * 2 * 16 * 64;                        // phase-3 partial tiles
        if (tiles < redsz) tiles = redsz;
        const size_t lds2 = ((size_t)(Dk + 2 * A2_PAD) * A2_TQ + tiles) * sizeof(float);
        const int npt = (D + A2_NW - 1) / A2_NW;
        if (lds2 <= 160 * 1024 && npt <= 64) {
            const bool vec = (W % 4 == 0) && aligned16(left) && aligned16(right) && (out == nullptr || aligned16(value));
#define ATT_K(N) (vec ? cross_attention_mfma_kernel<N, true> : cross_attention_mfma_kernel<N, false>)
            auto kern = npt <= 8 ? ATT_K(8) : npt <= 16 ? ATT_K(16) : npt <= 24 ? ATT_K(24) : npt <= 32 ? ATT_K(32) : ATT_K(64);
#undef ATT_K
            static std::atomic<unsigned long long> lds_done[10] = {};
            const int slot = (npt <= 8 ? 0 : npt <= 16 ? 1 : npt <= 24 ? 2 : npt <= 32 ? 3 : 4) * 2 + (vec ? 1 : 0);
            if (!lds_opt_in(kern, lds2, lds_done[slot])) return STEREO_ERR_UNSUPPORTED;
            const float scale2 = scale_mode == STEREO_SCALE_MEAN ? 1.0f / (float)C : 1.0f / sqrtf((float)C);
            dim3 grid2((W + A2_TQ - 1) / A2_TQ, H, B);
            hipLaunchKernelGGL(kern, grid2, dim3(A2_NT), lds2, as_stream(stream), left, right, value, out, disp, C, Cv, H,
                               W, D, Dk, mask_mode, scale2);
            STEREO_LAUNCH_CHECK();
            return STEREO_OK;
        }
    }
    return STEREO_ERR_UNSUPPORTED;  // band too long for the LDS score sheet (D > ~380)
}

// A1 + A4 with no volume at all, on the matrix cores (library-internal; STEREO_ERR_UNSUPPORTED outside the envelope).
template <typename TIN>
int stereo_corr_soft_argmin_mfma(const TIN *L, const TIN *R, float *disp, int B, int C, int H, int W, int D, float sgn,
                                 hipStream_t st)
{
    if (stereo_opt(OPT_ATT_KERNEL) == 0 || stereo_opt(OPT_ATT_KERNEL) == 9) {  // split-fp16 matrix cores where they apply
        const int rc = stereo_cross_attention_split16<TIN>(L, R, nullptr, disp, B, C, H, W, D, STEREO_MASK_ZERO, sgn / (float)C, st);
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    if (stereo_opt(OPT_ATT_KERNEL) == 7) {  // split-bf16 (test hook)
        const int rc = stereo_cross_attention_split<TIN>(L, R, nullptr, disp, B, C, H, W, D, STEREO_MASK_ZERO, sgn / (float)C, st);
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    return launch_online<TIN>(L, R, nullptr, disp, B, C, H, W, D, STEREO_MASK_ZERO, sgn / (float)C, st);
}
template int stereo_corr_soft_argmin_mfma<float>(const float *, const float *, float *, int, int, int, int, int, float,
                                                 hipStream_t);
template int stereo_corr_soft_argmin_mfma<f16>(const f16 *, const f16 *, float *, int, int, int, int, int, float, hipStream_t);

// A7 with fp16 features: left / right / value [B,C,H,W] fp16, out / disp fp32.  C <= 32 with value = right
// features -> the online-softmax kernel; anything else -> the channel-chunked kernel.  No slow path: outside both envelopes
// the call returns STEREO_ERR_UNSUPPORTED and the host layer widens the features itself.
extern "C" int stereo_cross_attention_f16(const void *left, const void *right, const void *value, float *out, float *disp,
                                          int B, int C, int Cv, int H, int W, int D, int mask_mode, int scale_mode,
                                          void *stream)
{
    if (B < 0 || C < 0 || Cv < 0 || H < 0 || W < 0 || D < 0) return STEREO_ERR_BAD_SHAPE;
    if (mask_mode != STEREO_MASK_ZERO && mask_mode != STEREO_MASK_NEGINF) return STEREO_ERR_BAD_SHAPE;
    if (scale_mode != STEREO_SCALE_SQRT_C && scale_mode != STEREO_SCALE_MEAN) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(left);
    STEREO_REQUIRE_PTR(right);
    if (out) STEREO_REQUIRE_PTR(value);
    if (D == 0 || C == 0) return STEREO_ERR_BAD_SHAPE;
    const f16 *L = static_cast<const f16 *>(left), *R = static_cast<const f16 *>(right), *V = static_cast<const f16 *>(value);
    const float sc = scale_mode == STEREO_SCALE_MEAN ? 1.0f / (float)C : 1.0f / sqrtf((float)C);
    const bool same_kv = out != nullptr && value == right && Cv == C;
    if (C <= 32 && (out == nullptr || same_kv) && (stereo_opt(OPT_ATT_KERNEL) == 9 || stereo_opt(OPT_ATT_KERNEL) == 0)) {
        const int rc = stereo_cross_attention_split16<f16>(L, R, out, disp, B, C, H, W, D, mask_mode, sc, as_stream(stream));
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    if (C <= 32 && (out == nullptr || same_kv) && stereo_opt(OPT_ATT_KERNEL) == 7) {
        const int rc = stereo_cross_attention_split<f16>(L, R, out, disp, B, C, H, W, D, mask_mode, sc, as_stream(stream));
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    if (C <= 32 && (out == nullptr || same_kv) && stereo_opt(OPT_ATT_KERNEL) != 6) {
        const int rc = launch_online<f16>(L, R, out, disp, B, C, H, W, D, mask_mode, sc, as_stream(stream));
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    return stereo_cross_attention_chunked<f16>(L, R, V, out, disp, B, C, Cv, H, W, D, mask_mode, sc,
                                               stereo_opt(OPT_ATT_KERNEL) != 6, as_stream(stream));  // split form where it applies
}
