// A1 (+ A4) for C > 32 on the f32 matrix cores: the correlation volume vol[b,d,h,w] = mean_c L[b,c,h,w] R[b,c,h,w-d]
// with the channels walked in 32-channel chunks.  corr_mfma.hip keeps one key tile of ALL channels in LDS (C <= 32); the
// reference's own use of CostVolume is 256 channels at 1/8 scale with 24 disparities (models/LocalCostVolume/baseline.py:63,
// 114: [B,256,H/8,W/8], D = 192 // 8), which until now fell to the VALU kernel.
//
// Structure = attention_chunked.hip's K phase + corr_mfma.hip's skew sheet: a persistent workgroup (4 waves = 128 queries of
// a row) walks a run of tiles as one sequence of items (32 channels x (128 + Dk) keys each) through two LDS buffers, one
// barrier per item, the next item's raw quads in registers under the current item's MFMAs; the NT band tiles accumulate in
// registers across the chunks (S^T[key][query], 1/C folded into Q).  After a tile's last chunk every accumulator register
// goes to row d of an LDS sheet [32 NT][128] with one ds_write_b32 (d = query - key runs along the tile diagonals; the
// (stride + 1) odd rows make it conflict-free), the optional soft-argmin runs on the registers, and the D complete rows
// leave as 512-B row-contiguous stores.  The w < d zeros come from the zero-padded key tile.
// HBM-bound on the feature reads (2 N C 4 bytes in, N D 4 out: 140 MB vs 0.7 GFLOP at the reference-native shape).
#include "common.h"

namespace {

constexpr int CC_NWV = 4;
constexpr int CC_TQW = 32 * CC_NWV;
constexpr int CC_RSTR = CC_TQW + 4;  // sheet row stride: 16-B aligned rows, (RSTR + 1) odd

template <int NT, bool ARGMIN, typename TIN>
__global__ __launch_bounds__(CC_NWV * 64) void corr_volume_chunked_kernel(const TIN *__restrict__ L, const TIN *__restrict__ R,
                                                                          float *__restrict__ vol, float *__restrict__ disp,
                                                                          int C, int H, int W, int D, float inv_c, float sgn,
                                                                          int tiles_per_wg, int ntiles)
{
    constexpr int Dk = 32 * (NT - 1);
    constexpr int NK = CC_TQW + Dk;
    constexpr int KS = NK + 4;
    constexpr int RMAX = 32 / CC_NWV;
    constexpr int NPASS = (NK / 4 + 63) / 64;
    typedef typename Quad<TIN>::raw raw_t;
    extern __shared__ f32x4 smem4[];                 // two key buffers [32][KS], then the sheet [32 NT][RSTR]
    float *Kb = reinterpret_cast<float *>(smem4);
    float *Sh = Kb + 2 * 32 * KS;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;  // launcher: 32 * H * W < 2^31
    const int nx = (W + CC_TQW - 1) / CC_TQW;

    const int tile0 = blockIdx.x * tiles_per_wg;
    const int tile_end = (tile0 + tiles_per_wg < ntiles) ? tile0 + tiles_per_wg : ntiles;
    if (tile0 >= tile_end) return;
    const int nck = (C + 31) / 32;
    const int total = (tile_end - tile0) * nck;

    raw_t rk[RMAX][NPASS];
    TIN qn[16];          // raw: a conversion right behind each load would serialise the 16 loads
    unsigned okbits = 0;
    int oknrow = 0;
    int is_t = tile0, is_j = 0;
    auto issue = [&]() {
        const int x = is_t % nx, bh = is_t / nx;
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * CC_TQW, kbase = w0 - Dk;
        const int c0 = 32 * is_j;
        const TIN *Rb = R + (((size_t)bb * C + c0) * HWu + (size_t)hh * W);
        const TIN *Lb = L + (((size_t)bb * C + c0) * HWu + (size_t)hh * W);
        oknrow = C - c0 < 32 ? C - c0 : 32;
        okbits = 0;
#pragma unroll
        for (int pp = 0; pp < NPASS; ++pp) {
            const int w = kbase + 4 * (lane + 64 * pp);
            if (lane + 64 * pp < NK / 4 && w >= 0 && w + 3 < W) okbits |= 1u << pp;
        }
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * CC_NWV;
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                const bool ok = ((okbits >> pp) & 1u) && row < oknrow;
                rk[rr][pp] = Quad<TIN>::ld(Rb + (ok ? (unsigned)row * HWu + (unsigned)(kbase + 4 * (lane + 64 * pp)) : 0u));
            }
        }
        const int q = w0 + 32 * wave + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int c = 2 * s + half;
            qn[s] = Lb[(c0 + c < C && q < W) ? (unsigned)c * HWu + (unsigned)q : 0u];
        }
        if (++is_j == nck) is_j = 0, ++is_t;
    };
    auto commit = [&](int buf) {
        f32x4 *dst = smem4 + buf * (32 * KS / 4);
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * CC_NWV;
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                if (lane + 64 * pp < NK / 4)
                    dst[(row * KS) / 4 + lane + 64 * pp] = (((okbits >> pp) & 1u) && row < oknrow)
                                                                ? Quad<TIN>::widen(rk[rr][pp])
                                                                : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    f32x16 acc[NT];
    float qv[16];
    issue();
    commit(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) qv[s] = (float)qn[s] * inv_c;

    int ct = tile0, cj = 0;
    for (int it = 0; it < total; ++it) {
        const int x = ct % nx, bh = ct / nx;
        const int w0 = x * CC_TQW;
        const int q = w0 + 32 * wave + l31;
        const bool wave_on = w0 + 32 * wave < W;  // wave-uniform
        const float *cur = Kb + (it & 1) * (32 * KS);
        __syncthreads();  // buffer it & 1 is complete; the other one (item it - 1) and the previous tile's sheet are free
        if (it + 1 < total) issue();  // lands under this item's MFMAs (issued and committed inside one iteration)
        if (cj == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        }
        if (wave_on) {
            const float *kp = cur + half * KS + 32 * wave + l31;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * s * KS + 32 * t], qv[s], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (it + 1 < total) {
            commit((it + 1) & 1);
#pragma unroll
            for (int s = 0; s < 16; ++s) qv[s] = (float)qn[s] * inv_c;
        }
        if (cj == nck - 1) {  // ---------------- the tile is complete: skew through the sheet, (soft-argmin,) flush
            if (wave_on) {
                // register r of tile t: d = cb - i_r with cb = Dk - 32 t + l31 - 4 half; rows outside [0, 32 NT) do not
                // exist (d >= -31 and d <= Dk + 31): the sheet has 32 NT rows and negative d are skipped
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int cb = Dk - 32 * t + l31 - 4 * half;
                    float *wp = Sh + cb * CC_RSTR + 32 * wave + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ir = (r & 3) + 8 * (r >> 2);
                        if (t < NT - 1 || cb >= ir) wp[-ir * CC_RSTR] = acc[t][r];  // only the nearest tile reaches d < 0
                    }
                }
                if (ARGMIN) {  // zero-filled logits take part (estimation.py:20): the key tile is zero-padded, nothing to mask
                    const float l2 = sgn * 1.44269504088896340736f;
                    float m = -INFINITY;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int cb = Dk - 32 * t + l31 - 4 * half;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ir = (r & 3) + 8 * (r >> 2);
                            float v = acc[t][r] * l2;
                            if ((t == NT - 1 && cb < ir) || cb - D >= ir) v = -INFINITY;  // d < 0 or d >= D
                            acc[t][r] = v;
                            m = fmaxf(m, v);
                        }
                    }
                    m = fmaxf(m, __shfl_xor(m, 32));
                    float ssum = 0.0f, esum = 0.0f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        float ts = 0.0f, ti = 0.0f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float pe = __builtin_amdgcn_exp2f(acc[t][r] - m);
                            ts += pe;
                            ti = fmaf(pe, (float)((r & 3) + 8 * (r >> 2)), ti);
                        }
                        ssum += ts;
                        esum += fmaf((float)(Dk - 32 * t + l31 - 4 * half), ts, -ti);
                    }
                    ssum += __shfl_xor(ssum, 32);
                    esum += __shfl_xor(esum, 32);
                    if (half == 0 && q < W) disp[(size_t)bh * W + q] = esum / ssum;
                }
            }
            __syncthreads();  // the sheet is complete
            const int hh = bh % H, bb = bh / H;
            float *vb = vol + ((size_t)bb * D * HWu + (size_t)hh * W);
            for (int qi = tid; qi < D * (CC_TQW / 4); qi += CC_NWV * 64) {  // D rows x 32 quads
                const int d = qi / (CC_TQW / 4), c4 = qi % (CC_TQW / 4);
                const int w = w0 + 4 * c4;
                if (w + 3 < W) st4(vb + (size_t)d * HWu + (unsigned)w, *reinterpret_cast<const f32x4 *>(Sh + d * CC_RSTR + 4 * c4));
            }
            // the next iteration's barrier separates these sheet reads from the next tile's sheet writes (>= 1 item later)
        }
        if (++cj == nck) cj = 0, ++ct;
    }
}

template <int NT, bool ARGMIN, typename TIN>
int launch_cc(const TIN *L, const TIN *R, float *vol, float *disp, int B, int C, int H, int W, int D, float sgn, hipStream_t st)
{
    constexpr int NK = CC_TQW + 32 * (NT - 1);
    const size_t lds = ((size_t)2 * 32 * (NK + 4) + (size_t)32 * NT * CC_RSTR) * sizeof(float);
    auto kern = corr_volume_chunked_kernel<NT, ARGMIN, TIN>;
    static std::atomic<unsigned long long> lds_done{0};
    if (lds > 160 * 1024 || !lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const long long ntl = (long long)((W + CC_TQW - 1) / CC_TQW) * H * B;
    if (ntl >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    const int resident = 256 * per_cu;
    // short launches (the reference-native 1/8-scale shape has ~2 tiles per resident workgroup) are latency-bound: rather
    // one tile per workgroup and every slot of the chip filled than a longer persistent run on half of them
    int tpw = (int)((ntl + 2 * resident - 1) / (2 * resident));
    if (stereo_opt(OPT_TILES_PER_WG) > 0) tpw = stereo_opt(OPT_TILES_PER_WG);  // test hook: short persistent runs
    const int nwg = (int)((ntl + tpw - 1) / tpw);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(CC_NWV * 64), lds, st, L, R, vol, disp, C, H, W, D, 1.0f / (float)C, sgn, tpw,
                       (int)ntl);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

// Library-internal (common.h).  Envelope: any C (meant for C > 32), D <= 128, W % 4 == 0, quad-aligned features, 16-B aligned
// volume, 32 H W < 2^31; STEREO_ERR_UNSUPPORTED outside it.  disp == nullptr: store-only (A1).
template <typename TIN>
int stereo_corr_volume_chunked(const TIN *L, const TIN *R, float *vol, float *disp, int B, int C, int H, int W, int D,
                               float sgn, hipStream_t st)
{
    if (D < 1 || D > 128 || W % 4 != 0 || !aligned_quad<TIN>(L) || !aligned_quad<TIN>(R) || !aligned16(vol) ||
        (long long)32 * H * W >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
#define CCL(NTV)                                                                             \
    return disp ? launch_cc<NTV, true, TIN>(L, R, vol, disp, B, C, H, W, D, sgn, st) \
                : launch_cc<NTV, false, TIN>(L, R, vol, disp, B, C, H, W, D, sgn, st)
    if (nt <= 2) CCL(2);
    if (nt <= 3) CCL(3);
    CCL(5);
#undef CCL
}

template int stereo_corr_volume_chunked<float>(const float *, const float *, float *, float *, int, int, int, int, int, float,
                                               hipStream_t);
template int stereo_corr_volume_chunked<f16>(const f16 *, const f16 *, float *, float *, int, int, int, int, int, float,
                                             hipStream_t);
