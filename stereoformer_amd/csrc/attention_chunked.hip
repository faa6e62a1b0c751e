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
                                   int H, int W, int D, int mask_mode, float sc, hipStream_t st)
{
    if (D < 1 || D > 192 || W % 4 != 0 || !aligned_quad<TIN>(L) || !aligned_quad<TIN>(R) || (out && !aligned_quad<TIN>(V)) ||
        (long long)32 * H * W >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
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
    if (nt <= 3) CH_NT(3);
    if (nt <= 5) CH_NT(5);
    CH_NT(7);
#undef CH_NT
#undef CH
}

template int stereo_cross_attention_chunked<float>(const float *, const float *, const float *, float *, float *, int, int,
                                                   int, int, int, int, int, float, hipStream_t);
template int stereo_cross_attention_chunked<f16>(const f16 *, const f16 *, const f16 *, float *, float *, int, int, int, int,
                                                 int, int, int, float, hipStream_t);
