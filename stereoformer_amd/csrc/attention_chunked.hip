// Epipolar cross-attention (A7) for ANY channel count / value tensor: the channel-chunked, software-pipelined form
// (attention.hip's v4 arithmetic -- scores stay in the MFMA accumulators, two-pass softmax in registers, probabilities feed
// the P V product straight from the accumulator registers -- inside v5's persistent, prefetching workgroup).
//
// Why: at C = 256 (RAFT-style features: [8,256,136,240] D=48, [4,256,96,312] D=64) the op moves 4 N C 4 bytes for
// 4 B H C band flops = ~11 flop/B, left of the f32 ridge: it is HBM-leaning, and v4's stage -> barrier -> MFMA -> barrier
// per 32-channel chunk exposes every chunk's load latency (21-23 % of the f32 MFMA peak, 32-39 % of HBM).  Here a
// workgroup walks a run of query tiles as ONE sequence of items -- the K chunks of a tile (32 channels x all keys), then
// its V chunks -- through two LDS buffers with ONE barrier per item: while item i is on the matrix cores, item i+1 sits
// in registers (raw 16-B / 8-B quads, masks applied at commit) and is committed to the other buffer after the MFMAs, then
// item i+2's loads are issued.  The pipeline runs across tile boundaries, so a tile's first K chunk lands under the
// previous tile's last V chunk.  The [B,H,W,D] score tensor never exists.
//
// Features may be fp32 or fp16 (Quad<T>, converted on the way into LDS; f32 accumulate): SURVEY 8(f) N2.
#include "common.h"

namespace {

template <int NT, int NWV, int MINW, typename TIN>
__global__ __launch_bounds__(NWV * 64, MINW) void cross_attention_chunked_kernel(
    const TIN *__restrict__ L, const TIN *__restrict__ R, const TIN *__restrict__ V, float *__restrict__ out,
    float *__restrict__ disp, int C, int Cv, int H, int W, int D, int mask_mode, float scale, int tiles_per_wg, int ntiles)
{
    constexpr int Dk = 32 * (NT - 1);
    constexpr int TQW = 32 * NWV;                // queries per workgroup tile
    constexpr int NK = TQW + Dk;                 // keys per staged chunk
    constexpr int KS = NK + 4;                   // row stride: 16-B aligned rows, KS / 4 odd
    constexpr int RMAX = (32 + NWV - 1) / NWV;   // staged rows per wave (row = wave, wave + NWV, ...)
    constexpr int NPASS = (NK / 4 + 63) / 64;    // quads per lane and row
    typedef typename Quad<TIN>::raw raw_t;
    extern __shared__ f32x4 smem4[];             // two buffers [32][KS]
    float *Kb = reinterpret_cast<float *>(smem4);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;  // launcher: 32 * H * W < 2^31
    const int nx = (W + TQW - 1) / TQW;
    const float sc2 = scale * 1.44269504088896340736f;  // logits in units of log2, folded into Q

    const int tile0 = blockIdx.x * tiles_per_wg;
    const int tile_end = (tile0 + tiles_per_wg < ntiles) ? tile0 + tiles_per_wg : ntiles;
    if (tile0 >= tile_end) return;
    const int nck = (C + 31) / 32, nvc = out ? (Cv + 31) / 32 : 0;
    const int ipt = nck + nvc;                   // items per tile
    const int total = (tile_end - tile0) * ipt;

    raw_t rk[RMAX][NPASS];
    TIN qn[16];                                  // raw: a conversion right behind each load would serialise the 16 loads
    unsigned okbits = 0;
    int oknrow = 0;                              // valid channel rows of the prefetched chunk
    // item -> (tile, chunk): kept as running counters (no division per item)
    int is_t = tile0, is_j = 0;                  // next item to ISSUE
    auto issue = [&]() {
        const int x = is_t % nx, bh = is_t / nx;
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * TQW, kbase = w0 - Dk;
        const bool isk = is_j < nck;
        const int c0 = 32 * (isk ? is_j : is_j - nck);
        const int nch = isk ? C : Cv;
        const TIN *base = (isk ? R : V) + (((size_t)bb * nch + c0) * HWu + (size_t)hh * W);  // workgroup-uniform
        oknrow = nch - c0 < 32 ? nch - c0 : 32;
        okbits = 0;
#pragma unroll
        for (int pp = 0; pp < NPASS; ++pp) {
            const int w = kbase + 4 * (lane + 64 * pp);
            if (lane + 64 * pp < NK / 4 && w >= 0 && w + 3 < W) okbits |= 1u << pp;
        }
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * NWV;     // wave-uniform
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                const bool ok = ((okbits >> pp) & 1u) && row < oknrow;
                rk[rr][pp] = Quad<TIN>::ld(base + (ok ? (unsigned)row * HWu + (unsigned)(kbase + 4 * (lane + 64 * pp)) : 0u));
            }
        }
        if (isk) {
            const TIN *Lb = L + (((size_t)bb * C + c0) * HWu + (size_t)hh * W);
            const int q = w0 + 32 * wave + l31;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int c = 2 * s + half;
                qn[s] = Lb[(c0 + c < C && q < W) ? (unsigned)c * HWu + (unsigned)q : 0u];
            }
        }
        if (++is_j == ipt) is_j = 0, ++is_t;
    };
    auto commit = [&](int buf) {                 // registers -> LDS buffer `buf`, one ds_write_b128 per quad
        f32x4 *dst = smem4 + buf * (32 * KS / 4);
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * NWV;
            if (row < 32) {
#pragma unroll
                for (int pp = 0; pp < NPASS; ++pp) {
                    if (lane + 64 * pp < NK / 4)
                        dst[(row * KS) / 4 + lane + 64 * pp] = (((okbits >> pp) & 1u) && row < oknrow)
                                                                    ? Quad<TIN>::widen(rk[rr][pp])
                                                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };

    f32x16 acc[NT];
    float qv[16];
    issue();
    commit(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) qv[s] = (float)qn[s] * sc2;

    int ct = tile0, cj = 0;                      // item being computed
    for (int it = 0; it < total; ++it) {
        const int x = ct % nx, bh = ct / nx;
        const int w0 = x * TQW;
        const int q = w0 + 32 * wave + l31;
        const bool wave_on = w0 + 32 * wave < W;  // wave-uniform
        const float *cur = Kb + (it & 1) * (32 * KS);
        __syncthreads();  // buffer it & 1 is complete; every wave is done with the other one (item it - 1)
        // item it + 1 -> registers, issued and committed INSIDE this iteration (hipcc waits vmcnt(0) on the loop back-edge, so
        // a prefetch carried across it would not overlap anything): it lands under this item's MFMAs
        if (it + 1 < total) issue();
        f32x16 o;
        if (cj < nck) {   // ---------------- K chunk: S^T += K_chunk Q_chunk^T
            if (cj == 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            }
            if (wave_on) {
                // one key tile at a time: 16 dependent MFMAs on acc[t] run back to back at the issue rate (64-cycle issue ==
                // 64-cycle accumulator latency), and the scheduling fence keeps the compiler from hoisting every tile's 16
                // LDS operands at once (16 NT live registers -> spills)
                const float *kp = cur + half * KS + 32 * wave + l31;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s * KS + 32 * t], qv[s], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (cj == nck - 1 && wave_on) {  // ---- masked softmax over the band, in registers (logits in log2 units)
                // element r of tile t: key row i_r + 4 half with i_r = (r&3) + 8 (r>>2) a compile-time constant, disparity
                // d = cb - i_r, cb = Dk - 32 t + l31 - 4 half.  Only the band's end tiles can leave [0, D) and only the row's
                // first tile has keys left of the image: wave-uniform tests, then one compare + select per score.
                const float left_fill = mask_mode == STEREO_MASK_ZERO ? 0.0f : -INFINITY;
                const bool left = w0 - Dk < 0;   // workgroup-uniform
                float m = -INFINITY;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int cb = Dk - 32 * t + l31 - 4 * half;
                    if (left) {
                        const int lim = cb - q;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (lim > (r & 3) + 8 * (r >> 2)) acc[t][r] = left_fill;  // key = q - d < 0
                    }
                    if (t == NT - 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (cb < (r & 3) + 8 * (r >> 2)) acc[t][r] = -INFINITY;  // d < 0
                    }
                    if (Dk - 32 * t + 31 >= D) {
                        const int lim = cb - D;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (lim >= (r & 3) + 8 * (r >> 2)) acc[t][r] = -INFINITY;  // d >= D
                    }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(acc[t][r], acc[t][r + 1]));
                }
                m = fmaxf(m, __shfl_xor(m, 32));  // d = 0 is always in the band: m is finite
                float ssum = 0.0f, esum = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float ts = 0.0f, ti = 0.0f;   // sum p, sum p * i_r of this tile: sum p d = cb sum p - sum p i_r
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pe = __builtin_amdgcn_exp2f(acc[t][r] - m);
                        acc[t][r] = pe;
                        ts += pe;
                        ti = fmaf(pe, (float)((r & 3) + 8 * (r >> 2)), ti);
                    }
                    ssum += ts;
                    esum += fmaf((float)(Dk - 32 * t + l31 - 4 * half), ts, -ti);
                }
                ssum += __shfl_xor(ssum, 32);
                esum += __shfl_xor(esum, 32);
                const float inv = 1.0f / ssum;
                if (disp && half == 0 && q < W) disp[(size_t)bh * W + q] = esum * inv;
                if (nvc > 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] *= inv;
                }
            }
        } else if (wave_on) {  // ---------------- V chunk: O^T = V_chunk^T P
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.0f;
            const float *vp = cur + l31 * KS + 32 * wave + 4 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * t + (s & 3) + 8 * (s >> 2)], acc[t][s], o, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // item it + 1 -> the other buffer (its last readers, item it - 1, passed this iteration's barrier); its Q values
        // become current
        if (it + 1 < total) {
            commit((it + 1) & 1);
            const bool nextk = (cj + 1 == ipt ? 0 : cj + 1) < nck;
            if (nextk) {
#pragma unroll
                for (int s = 0; s < 16; ++s) qv[s] = (float)qn[s] * sc2;
            }
        }
        // the V chunk's output rows leave AFTER the commit (whose vmcnt wait then covers loads only) and drain under the
        // next item's MFMAs
        if (cj >= nck && wave_on && q < W) {
            const int hh = bh % H, bb = bh / H;
            const int c0 = 32 * (cj - nck);
            float *ob = out + (((size_t)bb * Cv + c0) * HWu + (size_t)hh * W);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c0 + c < Cv) ob[(unsigned)c * HWu + (unsigned)q] = o[r];
            }
        }
        if (++cj == ipt) cj = 0, ++ct;
    }
}

// ---------------------------------------------------------------------------
// The same item pipeline on the BF16 matrix cores with SPLIT operands (attention_bf16.hip's formulation: every fp32 value is
// hi + mid + lo exactly, 6 piece products per fp32 product, P as hi + mid for the P V half; fp16 features: 2 pieces / 4
// products).  For the short bands of the C = 256 shapes (NT = 3: D <= 64) the executed matrix time per 32-channel chunk drops
// from 48 x 64 to 36 x 32 cycles per wave, which is what the fp32 kernels (v4, v6) sit on.  Chunk image in LDS:
// [buffer][piece][channel][key] bf16; K fragments through ds_read_b64_tr_b16, V^T fragments row-wise; the normalised
// probabilities are split ONCE per tile and parked, as bf16 pieces, in the registers of the score accumulators they came from.
struct PPieces {
    bf16x8 v[2][2];  // [k-step][hi, mid]
};
static_assert(sizeof(PPieces) == sizeof(f32x16), "P pieces reuse the accumulator's 16 registers");

template <int NT, int NWV, typename TIN>
__global__ __launch_bounds__(NWV * 64) void cross_attention_chunked_split_kernel(
    const TIN *__restrict__ L, const TIN *__restrict__ R, const TIN *__restrict__ V, float *__restrict__ out,
    float *__restrict__ disp, int C, int Cv, int H, int W, int D, int mask_mode, float scale, int tiles_per_wg, int ntiles)
{
    constexpr int NP = sizeof(TIN) == 4 ? 3 : 2;
    constexpr int Dk = 32 * (NT - 1);
    constexpr int TQW = 32 * NWV;
    constexpr int NK = TQW + Dk;
    constexpr int RSE = NK + 4;                  // image row stride (bf16 elements, 8-byte multiple)
    constexpr int PSE = 32 * RSE;                // piece stride
    constexpr int BSE = NP * PSE;                // buffer stride
    constexpr int RMAX = (32 + NWV - 1) / NWV;
    constexpr int NPASS = (NK / 4 + 63) / 64;
    typedef typename Quad<TIN>::raw raw_t;
    extern __shared__ __attribute__((aligned(16))) __bf16 ImgB[];   // [2][NP][32][RSE]

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;  // launcher: 32 * H * W < 2^30 (32-bit byte offsets)
    const int nx = (W + TQW - 1) / TQW;
    const float sc2 = scale * 1.44269504088896340736f;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    const int k_off = (8 * half + tq) * RSE + 32 * wave + 16 * tg1 + 4 * tp;   // + (16 s + 4 tt) RSE + 32 t
    const int v_off = l31 * RSE + 32 * wave + 4 * half;                        // + 32 t + 16 s (+ 8)

    const int tile0 = blockIdx.x * tiles_per_wg;
    const int tile_end = (tile0 + tiles_per_wg < ntiles) ? tile0 + tiles_per_wg : ntiles;
    if (tile0 >= tile_end) return;
    const int nck = (C + 31) / 32, nvc = out ? (Cv + 31) / 32 : 0;
    const int ipt = nck + nvc;
    const int total = (tile_end - tile0) * ipt;

    // Loads go through buffer resources: one per chunk, rows of W elements.  A key outside [0, W) is an out-of-range byte
    // offset and reads 0 (the zero padding of the band; also what ZERO-mask mode wants for the values); channel rows past
    // the tensor's last one get an empty resource.  No per-lane bound selects anywhere.
    raw_t rk[RMAX][NPASS];
    TIN qn[16];
    constexpr int ES = (int)sizeof(TIN);
    int is_t = tile0, is_j = 0;
    auto issue = [&]() {
        const int x = is_t % nx, bh = is_t / nx;
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * TQW, kbase = w0 - Dk;
        const bool isk = is_j < nck;
        const int c0 = 32 * (isk ? is_j : is_j - nck);
        const int nch = isk ? C : Cv;
        const int nrow = nch - c0 < 32 ? nch - c0 : 32;
        const TIN *base = (isk ? R : V) + (((size_t)bb * nch + c0) * HWu + (size_t)hh * W);
        int voff[NPASS];
#pragma unroll
        for (int pp = 0; pp < NPASS; ++pp) voff[pp] = (lane + 64 * pp < NK / 4) ? (kbase + 4 * (lane + 64 * pp)) * ES : -1;
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {       // one resource per row (the range check sees voffset + soffset on gfx950)
            const int row = wave + rr * NWV;
            const __amdgpu_buffer_rsrc_t rs = buffer_rsrc(base + (size_t)row * HWu, row < nrow ? (unsigned)W * ES : 0u);
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) rk[rr][pp] = BufLd<TIN>::quad(rs, voff[pp], 0);
        }
        if (isk) {
            const TIN *Lb = L + (((size_t)bb * C + c0) * HWu + (size_t)hh * W);
            const __amdgpu_buffer_rsrc_t rq = buffer_rsrc(Lb, ((unsigned)(nrow - 1) * HWu + (unsigned)W) * ES);
            int vq = (int)(((unsigned)(8 * half) * HWu + (unsigned)(w0 + 32 * wave + l31)) * ES);
#pragma unroll
            for (int i = 0; i < 16; ++i) {        // B-operand order: k-step i >> 3, element i & 7 -> channel 16 s + 8 half + j
                qn[i] = BufLd<TIN>::one(rq, vq, 0);
                vq += (int)((i == 7 ? 9u : 1u) * HWu * ES);   // running offset: one scalar step instead of 16 held ones
            }
        }
        if (++is_j == ipt) is_j = 0, ++is_t;
    };
    auto commit = [&](int buf) {
        __bf16 *img = ImgB + buf * BSE;
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * NWV;
            if (row < 32) {
#pragma unroll
                for (int pp = 0; pp < NPASS; ++pp) {
                    if (lane + 64 * pp < NK / 4) {
                        const f32x4 v = Quad<TIN>::widen(rk[rr][pp]);
                        bf16x2 a[NP], b[NP];
                        split_pair<NP>((f32x2){v[0], v[1]}, a);
                        split_pair<NP>((f32x2){v[2], v[3]}, b);
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            *reinterpret_cast<bf16x4 *>(&img[p * PSE + row * RSE + 4 * (lane + 64 * pp)]) = cat2(a[p], b[p]);
                    }
                }
            }
        }
    };
    bf16x8 qf[2][NP];
    auto take_q = [&]() {                        // the prefetched Q values (scaled, split) become the current B operand
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (float)qn[8 * s + j] * sc2;
            split_oct<NP>(x, qf[s]);
        }
    };

    f32x16 acc[NT];
    issue();
    commit(0);
    take_q();

    int ct = tile0, cj = 0;
    for (int it = 0; it < total; ++it) {
        const int x = ct % nx, bh = ct / nx;
        const int w0 = x * TQW;
        const int q = w0 + 32 * wave + l31;
        const bool wave_on = w0 + 32 * wave < W;  // wave-uniform
        const __bf16 *cur = ImgB + (it & 1) * BSE;
        __syncthreads();
        if (it + 1 < total) issue();
        f32x16 o;
        if (cj < nck) {   // ---------------- K chunk: S^T += K_chunk Q_chunk^T (6 / 4 piece products per k-step)
            if (cj == 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            }
            if (wave_on) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8 kf[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) {
                                const __bf16 *src = &cur[p * PSE + k_off + (16 * s + 4 * tt) * RSE + 32 * t];
                                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
                                const short tv[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                                for (int e = 0; e < 4; ++e) kf[p][4 * tt + e] = __builtin_bit_cast(__bf16, tv[e]);
                            }
                        }
                        f32x16 a = acc[t];
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[s][1], a, 0, 0, 0);
                        if (NP == 3) {
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[s][NP - 1], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[NP - 1], qf[s][0], a, 0, 0, 0);
                        }
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[s][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[s][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[s][0], a, 0, 0, 0);
                        acc[t] = a;
                    }
                }
            }
            if (cj == nck - 1 && wave_on) {  // ---- masked softmax over the band (as the fp32 chunked kernel), then P -> pieces
                const float left_fill = mask_mode == STEREO_MASK_ZERO ? 0.0f : -INFINITY;
                const bool left = w0 - Dk < 0;
                float m = -INFINITY;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int cb = Dk - 32 * t + l31 - 4 * half;
                    if (left) {
                        const int lim = cb - q;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (lim > (r & 3) + 8 * (r >> 2)) acc[t][r] = left_fill;
                    }
                    if (t == NT - 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (cb < (r & 3) + 8 * (r >> 2)) acc[t][r] = -INFINITY;
                    }
                    if (Dk - 32 * t + 31 >= D) {
                        const int lim = cb - D;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (lim >= (r & 3) + 8 * (r >> 2)) acc[t][r] = -INFINITY;
                    }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(acc[t][r], acc[t][r + 1]));
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                float ssum = 0.0f, esum = 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float ts = 0.0f, ti = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pe = __builtin_amdgcn_exp2f(acc[t][r] - m);
                        acc[t][r] = pe;
                        ts += pe;
                        ti = fmaf(pe, (float)((r & 3) + 8 * (r >> 2)), ti);
                    }
                    ssum += ts;
                    esum += fmaf((float)(Dk - 32 * t + l31 - 4 * half), ts, -ti);
                }
                ssum += __shfl_xor(ssum, 32);
                esum += __shfl_xor(esum, 32);
                const float inv = 1.0f / ssum;
                if (disp && half == 0 && q < W) disp[(size_t)bh * W + q] = esum * inv;
                if (nvc > 0) {   // normalised probabilities -> hi + mid bf16 pieces, parked in the accumulator's registers
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        PPieces pp;
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            float x[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] = acc[t][8 * s + e] * inv;
                            split_oct<2>(x, pp.v[s]);
                        }
                        acc[t] = __builtin_bit_cast(f32x16, pp);
                    }
                }
            }
        } else if (wave_on) {  // ---------------- V chunk: O^T = V_chunk^T P (5 / 4 piece products per k-step)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const PPieces pp = __builtin_bit_cast(PPieces, acc[t]);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 vf[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(&cur[p * PSE + v_off + 32 * t + 16 * s]);
                        const bf16x4 hi = *reinterpret_cast<const bf16x4 *>(&cur[p * PSE + v_off + 32 * t + 16 * s + 8]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) vf[p][e] = lo[e], vf[p][4 + e] = hi[e];
                    }
                    if (NP == 3) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[NP - 1], pp.v[s][0], o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pp.v[s][1], o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pp.v[s][1], o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pp.v[s][0], o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pp.v[s][0], o, 0, 0, 0);
                }
            }
        }
        if (it + 1 < total) {
            commit((it + 1) & 1);
            const bool nextk = (cj + 1 == ipt ? 0 : cj + 1) < nck;
            if (nextk) take_q();
        }
        if (cj >= nck && wave_on && q < W) {      // channels past Cv fall outside the resource and are dropped
            const int hh = bh % H, bb = bh / H;
            const int c0 = 32 * (cj - nck);
            const int nrow = Cv - c0 < 32 ? Cv - c0 : 32;
            float *ob = out + (((size_t)bb * Cv + c0) * HWu + (size_t)hh * W);
            const __amdgpu_buffer_rsrc_t ro = buffer_rsrc(ob, ((unsigned)(nrow - 1) * HWu + (unsigned)W) * 4u);
            int vo = (int)(((unsigned)(4 * half) * HWu + (unsigned)q) * 4u);
#pragma unroll
            for (int r = 0; r < 16; ++r) {        // accumulator row r -> channel (r & 3) + 8 (r >> 2) + 4 half
                const float ov = o[r];            // (a bit_cast straight off the vector element reads element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov), ro, vo, 0, 0);
                vo += (int)(((r & 3) == 3 ? 5u : 1u) * HWu * 4u);
            }
        }
        if (++cj == ipt) cj = 0, ++ct;
    }
}

template <int NT, int NWV, typename TIN>
int launch_chunked_split(const TIN *L, const TIN *R, const TIN *V, float *out, float *disp, int B, int C, int Cv, int H, int W,
                         int D, int mask_mode, float sc, hipStream_t st)
{
    constexpr int NP = sizeof(TIN) == 4 ? 3 : 2;
    constexpr int NK = 32 * NWV + 32 * (NT - 1);
    const size_t lds = (size_t)2 * NP * 32 * (NK + 4) * 2;
    auto kern = cross_attention_chunked_split_kernel<NT, NWV, TIN>;
    static std::atomic<unsigned long long> lds_done{0};
    if (lds > 160 * 1024 || !lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const long long ntl = (long long)((W + 32 * NWV - 1) / (32 * NWV)) * H * B;
    if (ntl >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    static std::atomic<int> per_cu{0};
    const int resident = 256 * workgroups_per_cu(kern, NWV * 64, lds, per_cu);
    int tpw = (int)((ntl + resident - 1) / resident);
    if (stereo_opt(OPT_TILES_PER_WG) > 0) tpw = stereo_opt(OPT_TILES_PER_WG);
    const int nwg = (int)((ntl + tpw - 1) / tpw);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NWV * 64), lds, st, L, R, V, out, disp, C, Cv, H, W, D, mask_mode, sc, tpw,
                       (int)ntl);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

template <int NT, int NWV, int MINW, typename TIN>
int launch_chunked(const TIN *L, const TIN *R, const TIN *V, float *out, float *disp, int B, int C, int Cv, int H, int W,
                   int D, int mask_mode, float sc, hipStream_t st)
{
    constexpr int NK = 32 * NWV + 32 * (NT - 1);
    const size_t lds = (size_t)2 * 32 * (NK + 4) * sizeof(float);
    auto kern = cross_attention_chunked_kernel<NT, NWV, MINW, TIN>;
    static std::atomic<unsigned long long> lds_done{0};
    if (lds > 160 * 1024 || !lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const long long ntl = (long long)((W + 32 * NWV - 1) / (32 * NWV)) * H * B;
    if (ntl >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    int per_cu = (int)((160 * 1024) / lds);      // resident workgroups per CU by LDS ...
    const int by_waves = 16 / NWV;               // ... and by two waves per SIMD x 2 (register budget of MINW)
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    const int resident = 256 * per_cu;
    int tpw = (int)((ntl + resident - 1) / resident);
    if (stereo_opt(OPT_TILES_PER_WG) > 0) tpw = stereo_opt(OPT_TILES_PER_WG);  // test hook: short persistent runs
    const int nwg = (int)((ntl + tpw - 1) / tpw);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NWV * 64), lds, st, L, R, V, out, disp, C, Cv, H, W, D, mask_mode, sc, tpw,
                       (int)ntl);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

// Library-internal (declared in common.h).  Envelope: W % 4 == 0, quad-aligned bases, D <= 192, 32 H W < 2^31;
// STEREO_ERR_UNSUPPORTED outside it (the caller then takes the older kernels).
template <typename TIN>
int stereo_cross_attention_chunked(const TIN *L, const TIN *R, const TIN *V, float *out, float *disp, int B, int C, int Cv,
                                   int H, int W, int D, int mask_mode, float sc, bool split_bf16_form, hipStream_t st)
{
    if (D < 1 || D > 192 || W % 4 != 0 || !aligned_quad<TIN>(L) || !aligned_quad<TIN>(R) || (out && !aligned_quad<TIN>(V)) ||
        (long long)32 * H * W >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
    // disparity only, bands of <= 128: the split-fp16 form with 16-channel items (attention_chunked16.hip); hook 8 keeps the
    // split-bf16 form below
    if (out == nullptr && disp != nullptr && split_bf16_form && stereo_opt(OPT_ATT_KERNEL) != 8) {
        const int rc16 = stereo_cross_attention_disp_chunked16<TIN>(L, R, disp, B, C, H, W, D, mask_mode, sc, st);
        if (rc16 != STEREO_ERR_UNSUPPORTED) return rc16;
    }
    // waves (32 queries each) per workgroup: the shape that wastes the fewest 32-query slots of a row -- 8 (256 queries),
    // 5 (160) or 4 (128); ties go to the wider tile (less halo re-read).  W = 240 -> 8 (one tile per row), 312 -> 5.
    const int qt = (W + 31) / 32;
    int nwv = 8, best = ((qt + 7) / 8) * 8;
    if (((qt + 4) / 5) * 5 < best) nwv = 5, best = ((qt + 4) / 5) * 5;
    if (((qt + 3) / 4) * 4 < best) nwv = 4, best = ((qt + 3) / 4) * 4;
    if (stereo_opt(OPT_ATT_WAVES) == 4 || stereo_opt(OPT_ATT_WAVES) == 5 || stereo_opt(OPT_ATT_WAVES) == 8)
        nwv = stereo_opt(OPT_ATT_WAVES);  // test hook
#define CH(NTV, NWVV, MW) return launch_chunked<NTV, NWVV, MW, TIN>(L, R, V, out, disp, B, C, Cv, H, W, D, mask_mode, sc, st)
#define CH_NT(NTV)                   \
    do {                             \
        if (nwv == 8) CH(NTV, 8, 2); \
        if (nwv == 5) CH(NTV, 5, 2); \
        CH(NTV, 4, 2);               \
    } while (0)
    // bands of <= 128: split operands on the bf16 matrix cores (32 H W < 2^30: its buffer resources address a 32-channel slab
    // of a row with 32-bit BYTE offsets)
    if (nt <= 5 && split_bf16_form && (long long)32 * H * W < (1LL << 30)) {
        // wide tiles: the band halo is loaded and split once per tile (W = 312: 8 waves 0.151 / 0.182 ms at D = 64 / 128, 5
        // exact-fit waves 0.147 / 0.215)
        if (stereo_opt(OPT_ATT_WAVES) == 0) nwv = qt <= 4 ? 4 : (qt == 5 ? 5 : 8);
        int rc;
#define SP(NTV)                                                                                                     \
    (nwv == 8   ? launch_chunked_split<NTV, 8, TIN>(L, R, V, out, disp, B, C, Cv, H, W, D, mask_mode, sc, st)       \
     : nwv == 5 ? launch_chunked_split<NTV, 5, TIN>(L, R, V, out, disp, B, C, Cv, H, W, D, mask_mode, sc, st)       \
                : launch_chunked_split<NTV, 4, TIN>(L, R, V, out, disp, B, C, Cv, H, W, D, mask_mode, sc, st))
        rc = nt <= 3 ? SP(3) : SP(5);
#undef SP
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    if (nt <= 3) CH_NT(3);
    if (nt <= 5) CH_NT(5);
    CH_NT(7);
#undef CH_NT
#undef CH
}

template int stereo_cross_attention_chunked<float>(const float *, const float *, const float *, float *, float *, int, int,
                                                   int, int, int, int, int, float, bool, hipStream_t);
template int stereo_cross_attention_chunked<f16>(const f16 *, const f16 *, const f16 *, float *, float *, int, int, int, int,
                                                 int, int, int, float, bool, hipStream_t);
