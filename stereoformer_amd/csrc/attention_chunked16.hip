// Epipolar cross-attention (A7), DISPARITY ONLY, any channel count, on the FP16 matrix cores with split operands: the
// channel-chunked item pipeline of attention_chunked.hip in attention_f16.hip's arithmetic.
//
// Why a third form: with no value tensor the op reads 2 N C 4 bytes for 2 N C band flops -- [4,256,96,312] D = 64 (cfg-5's
// coarse-to-fine start) is 245 MB = 31 us at 8 TB/s -- and the split-bf16 chunked kernel spent 79-85 us on it: three bf16
// pieces per value to split, six piece products per k-step, 124 KB of LDS per 8-wave workgroup (one per CU), and at W = 312
// its 256-query tiles leave 39 % of the query slots of a row empty.  The SIMDs, not the memory system, were the bound.  Here
//   * a value is TWO fp16 pieces of 16 x: 16 x = hi + lo, hi = fp16(16 x), lo = fp16(16 x - hi) -- UNSCALED low piece, so
//     the three leading products hi hi + hi lo + lo hi accumulate in ONE fp32 accumulator (scores = acc / 256): a third of
//     the splitting work (two v_fma_mix per value), half the matrix instructions, no second accumulator and no per-item
//     fold of one (attention_f16.hip's scaled low piece lo' = 2^11 lo needs one: 30 vector instructions and a dependent
//     matrix chain per item here, 23 of this launch's 62 us).  The price is in the small values: lo is a normal fp16 number
//     for |x| >= 2^-7 (22 significant bits, as before); below that it rounds on the subnormal grid, an ABSOLUTE error of
//     <= 2^-29 in x (the factor 16 buys four more binades of full precision) -- a score moves by <= C max|feature| 2^-28
//     scale, far below what the softmax resolves.  fp16 features are their own single piece, exactly;
//   * an item is 16 channels (one k-step of v_mfma_f32_32x32x16_f16) x all keys of the tile: two item buffers of a 5-wave
//     workgroup are 49 KB, so THREE workgroups share a CU (15 waves instead of 8) and 160-query tiles fit W = 312 exactly;
//   * item i + 1 is committed right behind the barrier that opens item i and item i + 2's loads are issued before item i's
//     matrix work: a load has a whole iteration to land.
// Softmax, masks and the disparity expectation are the chunked kernel's (two-pass over the band's score registers).
// Range: |feature| < 4 094 for fp32 features (16 x in fp16), < 65 504 for fp16 ones; beyond it inf / NaN, never silently wrong
// (test hook att_kernel = 8: the split-bf16 form, fp32's whole range).
// Reference: the empty CrossAttentionCostVolume module's contract as SURVEY.md section 8 rows A7 / A8 fix it.
#include "common.h"

namespace {

// DBG (dev ablations, not instantiated in the library): 1 = no matrix work, 3 = every load through an empty resource.
template <int NT, int NWV, typename TIN, int DBG = 0>
__global__ __launch_bounds__(NWV * 64, (NT <= 3 ? 4 : 3)) void cross_attention_disp_chunked16_kernel(
    const TIN *__restrict__ L, const TIN *__restrict__ R, float *__restrict__ disp, int C, int H, int W, int D, int mask_mode,
    float scale, int tiles_per_wg, int ntiles)
{
    constexpr int NP = sizeof(TIN) == 4 ? 2 : 1;  // pieces of a feature value
    constexpr int CH = 16;                        // channels per item
    constexpr int Dk = 32 * (NT - 1);
    constexpr int TQW = 32 * NWV;
    constexpr int NK = TQW + Dk;
    constexpr int RSD = (NK / 2 + 54 + 63) / 64 * 64;   // row slot in dwords: whole bank rows + room for the row's offset
    constexpr int RSE = 2 * RSD;
    constexpr int PSE = CH * RSE;                 // piece stride (f16 elements)
    constexpr int BSE = NP * PSE;                 // buffer stride
    constexpr int QPR = NK / 4;                   // 16-byte quads per channel row of an item
    constexpr int NQ = (CH * QPR + NWV * 64 - 1) / (NWV * 64);   // quads per lane: the item's quads dealt over ALL lanes (every
                                                  // wave runs the same code: no wave-dependent row count, no branch around a load)
    constexpr float PRE = NP == 2 ? 16.0f : 1.0f; // fp32 features are split as PRE x
    typedef typename Quad<TIN>::raw raw_t;
    extern __shared__ __attribute__((aligned(16))) _Float16 Im16c[];   // [2][NP][CH][RSE]

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;  // launcher: 32 * H * W < 2^30 (32-bit byte offsets)
    const int nx = (W + TQW - 1) / TQW;
    const float sc2 = scale * 1.44269504088896340736f;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // two fp32 values -> packed fp16 pieces of PRE x: hi = fp16(PRE x), lo = fp16(PRE x - hi), each ONE mixed-precision FMA
    // (fp32 arithmetic on the fp16 source hi, fp16 result into one half of the destination)
    float pre = PRE;
    asm volatile("" : "+v"(pre));
    auto split2 = [&](float x0, float x1, unsigned &h, unsigned &l) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(pre));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(pre));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(pre), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(pre), "v"(h));
    };
    // attention_f16.hip's image: channel row r starts 16 (r & 3) + 2 (r >> 2) dwords into its slot, so that the transposed
    // read of a 32-lane group (4 rows x 16 dwords) tiles the 64 banks
    auto row_off = [](int row) { return row * RSE + 32 * (row & 3) + 4 * (row >> 2); };
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    const int k_off = (8 * half + tq) * RSE + 32 * tq + 8 * half + 16 * tg1 + 4 * tp;   // + 4 tt RSE + 4 tt + key-tile column

    // workgroups go to the 8 XCDs round-robin; XCD k takes the k-th eighth of the tiles, so the tiles of an image row (which
    // share Dk of their keys) and of neighbouring rows meet in ONE L2
    // (any grid size: XCD k owns ceil((G - k) / 8) workgroups, as one contiguous run)
    const int gq = (int)(gridDim.x >> 3), gr = (int)(gridDim.x & 7), xcd = (int)(blockIdx.x & 7);
    const int wg = xcd * gq + (xcd < gr ? xcd : gr) + (int)(blockIdx.x >> 3);
    const int tile0 = wg * tiles_per_wg;
    const int tile_end = (tile0 + tiles_per_wg < ntiles) ? tile0 + tiles_per_wg : ntiles;
    if (tile0 >= tile_end) return;
    const int nck = (C + CH - 1) / CH;            // items per tile
    const int total = (tile_end - tile0) * nck;

    // Loads through ONE buffer resource per item (the item's channel rows; empty past the last item): a quad whose keys lie
    // outside [0, W) or whose row is past the tensor's last channel gets the out-of-range offset -1 and reads 0 -- the zero
    // padding of the band.
    raw_t rk[NQ];
    TIN qn[8];
    constexpr int ES = (int)sizeof(TIN);
    int g_off[NQ], g_kr[NQ];                      // lane-invariant: byte offset in the item; key | row << 10 | LDS element offset << 16
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int g = tid + NWV * 64 * k;
        const int row = g / QPR, cq = g - row * QPR;
        g_off[k] = (int)(((unsigned)row * HWu + 4u * (unsigned)cq) * ES);
        g_kr[k] = 4 * cq | ((row < 31 ? row : 31) << 10) | ((row_off(row < CH ? row : 0) + 4 * cq) << 16);   // row >= CH: no quad
    }
    // Running item cursors, all scalar: the tile index is taken apart (two divisions) only when a tile BEGINS -- done per item,
    // for the keys, the Q values and the tile being computed, that arithmetic was 230 scalar instructions per wave and item,
    // and the CU's one scalar unit serves all its waves (15 here).
    auto tile_row = [&](const TIN *T, int tl, int &x) -> const TIN * {
        const int bh = tl / nx;
        x = tl - bh * nx;
        const int bb = bh / H, hh = bh - bb * H;
        return T + ((size_t)bb * C * HWu + (size_t)hh * W);
    };
    int kx, qx;
    const TIN *kp = tile_row(R, tile0, kx), *qp = tile_row(L, tile0, qx);
    int kt = tile0, kj = 0, krows = C, qt = tile0, qj = 0, qrows = C;
    auto issue = [&](bool live) {
        const int kbase = kx * TQW - Dk;
        const int nrow = krows < CH ? krows : CH;
        const __amdgpu_buffer_rsrc_t rs = buffer_rsrc(kp, live ? ((unsigned)(nrow - 1) * HWu + (unsigned)W) * ES : 0u);
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            int kr = g_kr[k];
            asm volatile("" : "+v"(kr));          // decode here, every time: hoisted out of the loop the fields cost registers
            const bool ok = (unsigned)(kbase + (kr & 0x3ff)) < (unsigned)W && ((kr >> 10) & 31) < nrow;
            rk[k] = BufLd<TIN>::quad(rs, ok ? g_off[k] + kbase * ES : -1, 0);
        }
        kp += (size_t)CH * HWu, krows -= CH;
        if (++kj == nck) kj = 0, krows = C, kp = tile_row(R, ++kt, kx);
    };
    int vqj[8];                                   // lane-invariant Q offsets (channel 8 half + j, query 32 wave + l31); the tile's
#pragma unroll                                    // first query goes into the loads' SCALAR offset
    for (int j = 0; j < 8; ++j) vqj[j] = (int)(((unsigned)(8 * half + j) * HWu + (unsigned)(32 * wave + l31)) * ES);
    auto issue_q = [&](bool live) {               // the Q values of an item are issued one iteration later than its keys (below)
        const int nrow = qrows < CH ? qrows : CH;
        const __amdgpu_buffer_rsrc_t rq = buffer_rsrc(qp, live ? ((unsigned)(nrow - 1) * HWu + (unsigned)W) * ES : 0u);
        const int sq = qx * TQW * ES;
#pragma unroll
        for (int i = 0; i < 8; ++i) qn[i] = BufLd<TIN>::one(rq, vqj[i], sq);   // B operand: element i -> channel c0 + 8 half + i
        qp += (size_t)CH * HWu, qrows -= CH;
        if (++qj == nck) qj = 0, qrows = C, qp = tile_row(L, ++qt, qx);
    };
    auto commit = [&](int buf) {                  // registers -> piece images of buffer `buf`: one 8-byte store per quad, piece
        _Float16 *img = Im16c + buf * BSE;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            int kr = g_kr[k];
            asm volatile("" : "+v"(kr));
            if (((kr >> 10) & 31) < CH) {
                const int lo = (int)((unsigned)kr >> 16);
                if constexpr (NP == 2) {
                    const f32x4 v = Quad<TIN>::widen(rk[k]);
                    unsigned h0, l0, h1, l1;
                    split2(v[0], v[1], h0, l0);
                    split2(v[2], v[3], h1, l1);
                    *reinterpret_cast<u32x2 *>(&img[lo]) = (u32x2){h0, h1};
                    *reinterpret_cast<u32x2 *>(&img[PSE + lo]) = (u32x2){l0, l1};
                } else {
                    *reinterpret_cast<raw_t *>(&img[lo]) = rk[k];
                }
            }
        }
    };
    f16x8 qf[NP];
    auto take_q = [&]() {                         // the loaded Q values become the B operand (unscaled: the scale is on the scores)
        if constexpr (NP == 2) {
            unsigned hq[4], lq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split2((float)qn[2 * j], (float)qn[2 * j + 1], hq[j], lq[j]);
            qf[0] = __builtin_bit_cast(f16x8, (u32x4){hq[0], hq[1], hq[2], hq[3]});
            qf[NP - 1] = __builtin_bit_cast(f16x8, (u32x4){lq[0], lq[1], lq[2], lq[3]});
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[0][j] = (_Float16)qn[j];
        }
    };

    f32x16 ah[NT];                                // cleared here and at the end of every tile (not by a select per item)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ah[t][r] = 0.0f;
    // One loop, no prologue copy of it: iteration -1 commits item 0, issues item 1 and takes item 0's Q values; keys are
    // always issued BEFORE the Q values of the same item (the wait counts of the loop are merged over every way into it).
    issue(true);
    __builtin_amdgcn_sched_barrier(0);
    issue_q(true);
    __builtin_amdgcn_sched_barrier(0);

    int ct = tile0, cj = 0;
    int bh = tile0 / nx, w0 = (tile0 - bh * nx) * TQW;
    for (int it = -1; it < total; ++it) {
        const int q = w0 + 32 * wave + l31;
        const bool wave_on = w0 + 32 * wave < W;  // wave-uniform
        const _Float16 *cur = Im16c + (it & 1) * BSE;
        __syncthreads();  // buffer it & 1 is complete; every wave is done with the other one (item it - 1)
        // straight-line pipeline: past the last item the loads go through EMPTY resources (zeros, no memory access) and the
        // commit fills a buffer nobody reads -- with `if (it + 2 < total)` around them the compiler moved the loads of the Q
        // values in front of the conversion of the previous ones, into other registers, and copied them into the loop-carried
        // ones behind a wait for everything in flight
        commit((it + 1) & 1);
        issue(it + 2 < total && DBG != 3);
        if (wave_on && it >= 0 && DBG != 1) {   // ---------------- S^T += K_item Q_item^T: A = K through the transposed read, B = Q pieces
            f16x8 kf[NT][NP];                     // all fragments first, then the products tile-interleaved: three independent
#pragma unroll                                    // accumulation chains instead of three dependent instructions per tile
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    s16x4 kq[2];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const _Float16 *src = &cur[p * PSE + k_off + 4 * tt * RSE + 4 * tt + 32 * (wave + t)];
                        kq[tt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
                    }
                    kf[t][p] = __builtin_bit_cast(f16x8, __builtin_shufflevector(kq[0], kq[1], 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) ah[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t][0], qf[0], ah[t], 0, 0, 0);
            if constexpr (NP == 2) {
#pragma unroll
                for (int t = 0; t < NT; ++t) ah[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t][0], qf[1], ah[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) ah[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t][1], qf[0], ah[t], 0, 0, 0);
            }
        }
        if (cj == nck - 1 && wave_on && it >= 0) {  // ---- scores in exp2 units, masked softmax over the band, disparity expectation
            const float left_fill = mask_mode == STEREO_MASK_ZERO ? 0.0f : -INFINITY;
            const bool left = w0 - Dk < 0;   // workgroup-uniform
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ah[t][r] *= sc2 * (1.0f / (PRE * PRE));
                }
                // element r of tile t: key row i_r + 4 half with i_r = (r & 3) + 8 (r >> 2), disparity d = cb - i_r
                const int cb = Dk - 32 * t + l31 - 4 * half;
                if (left) {
                    const int lim = cb - q;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (lim > (r & 3) + 8 * (r >> 2)) ah[t][r] = left_fill;   // key = q - d < 0
                }
                if (t == NT - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (cb < (r & 3) + 8 * (r >> 2)) ah[t][r] = -INFINITY;    // d < 0
                }
                if (Dk - 32 * t + 31 >= D) {
                    const int lim = cb - D;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (lim >= (r & 3) + 8 * (r >> 2)) ah[t][r] = -INFINITY;  // d >= D
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(ah[t][r], ah[t][r + 1]));
            }
            m = fmaxf(m, __shfl_xor(m, 32));  // d = 0 is always in the band: m is finite
            float ssum = 0.0f, esum = 0.0f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float ts = 0.0f, ti = 0.0f;   // sum p, sum p i_r of this tile: sum p d = cb sum p - sum p i_r
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pe = __builtin_amdgcn_exp2f(ah[t][r] - m);
                    ts += pe;
                    ti = fmaf(pe, (float)((r & 3) + 8 * (r >> 2)), ti);
                }
                ssum += ts;
                esum += fmaf((float)(Dk - 32 * t + l31 - 4 * half), ts, -ti);
            }
            ssum += __shfl_xor(ssum, 32);
            esum += __shfl_xor(esum, 32);
            if (half == 0 && q < W) disp[(size_t)bh * W + q] = esum / ssum;
#pragma unroll
            for (int t = 0; t < NT; ++t)   // the next tile starts from zero (a wave that is off never accumulates)
#pragma unroll
                for (int r = 0; r < 16; ++r) ah[t][r] = 0.0f;
        }
        take_q();               // item it + 1's Q values (issued an iteration ago) -> B operand; item it + 2's take their registers
        __builtin_amdgcn_sched_barrier(0);
        issue_q(it + 2 < total && DBG != 3);
        if (it >= 0 && ++cj == nck) {
            cj = 0, ++ct;
            bh = ct / nx, w0 = (ct - bh * nx) * TQW;
        }
    }
}

template <int NT, int NWV, typename TIN, int DBG = 0>
int launch_disp_chunked16(const TIN *L, const TIN *R, float *disp, int B, int C, int H, int W, int D, int mask_mode, float sc,
                          hipStream_t st, long long *cost = nullptr)
{
    constexpr int NP = sizeof(TIN) == 4 ? 2 : 1;
    constexpr int NK = 32 * NWV + 32 * (NT - 1);
    constexpr int RSD = (NK / 2 + 54 + 63) / 64 * 64;
    const size_t lds = (size_t)2 * NP * 16 * (2 * RSD) * 2;
    auto kern = cross_attention_disp_chunked16_kernel<NT, NWV, TIN, DBG>;
    static std::atomic<unsigned long long> lds_done{0};
    if (lds > 160 * 1024 || !lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const long long ntl = (long long)((W + 32 * NWV - 1) / (32 * NWV)) * H * B;
    if (ntl >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    static std::atomic<int> per_cu{0};
    const int resident = 256 * workgroups_per_cu(kern, NWV * 64, lds, per_cu);
    int tpw = (int)((ntl + resident - 1) / resident);
    if (cost) {   // plan only: rounds of resident workgroups x workgroups sharing a CU x columns a tile stages (keys + queries)
        *cost = (long long)tpw * (resident / 256) * (NK + 32 * NWV);
        return STEREO_OK;
    }
    if (stereo_opt(OPT_TILES_PER_WG) > 0) tpw = stereo_opt(OPT_TILES_PER_WG);
    const int nwg = (int)((ntl + tpw - 1) / tpw);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NWV * 64), lds, st, L, R, disp, C, H, W, D, mask_mode, sc, tpw, (int)ntl);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

// Library-internal (declared in common.h).  Envelope: W % 4 == 0, quad-aligned bases, D <= 128, 32 H W < 2^30;
// STEREO_ERR_UNSUPPORTED outside it (the caller then takes the other chunked kernels).
template <typename TIN>
int stereo_cross_attention_disp_chunked16(const TIN *L, const TIN *R, float *disp, int B, int C, int H, int W, int D,
                                          int mask_mode, float sc, hipStream_t st)
{
    if (D < 1 || D > 128 || W % 4 != 0 || !aligned_quad<TIN>(L) || !aligned_quad<TIN>(R) || disp == nullptr ||
        (long long)32 * H * W >= (1LL << 30))
        return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
    // waves (32 queries each) per workgroup -- 8, 5 or 4: the shape with the least staged columns per CU over the launch,
    // rounds of resident workgroups x workgroups per CU x (keys + queries of a tile): that prices empty query slots of a
    // row's last tile, the halo, and a last round that leaves CUs idle.  [4,256,96,312] D = 64 -> 5 (768 tiles = one round of
    // 3 x 256 workgroups); [8,256,136,240] D = 48 -> 4.
    int nwv = 0;
    {
        long long best = -1;
        const int cand[3] = {8, 5, 4};
        for (int i = 0; i < 3; ++i) {
            long long c = -1;
#define PLAN(NTV, NWVV) launch_disp_chunked16<NTV, NWVV, TIN>(L, R, disp, B, C, H, W, D, mask_mode, sc, st, &c)
            const int rc = nt <= 3 ? (cand[i] == 8 ? PLAN(3, 8) : cand[i] == 5 ? PLAN(3, 5) : PLAN(3, 4))
                                   : (cand[i] == 8 ? PLAN(5, 8) : cand[i] == 5 ? PLAN(5, 5) : PLAN(5, 4));
#undef PLAN
            if (rc == STEREO_OK && (best < 0 || c < best)) best = c, nwv = cand[i];
        }
        if (nwv == 0) return STEREO_ERR_UNSUPPORTED;
    }
    if (stereo_opt(OPT_ATT_WAVES) == 4 || stereo_opt(OPT_ATT_WAVES) == 5 || stereo_opt(OPT_ATT_WAVES) == 8)
        nwv = stereo_opt(OPT_ATT_WAVES);  // test hook
#define DC(NTV)                                                                                                  \
    (nwv == 8   ? launch_disp_chunked16<NTV, 8, TIN>(L, R, disp, B, C, H, W, D, mask_mode, sc, st)               \
     : nwv == 5 ? launch_disp_chunked16<NTV, 5, TIN>(L, R, disp, B, C, H, W, D, mask_mode, sc, st)               \
                : launch_disp_chunked16<NTV, 4, TIN>(L, R, disp, B, C, H, W, D, mask_mode, sc, st))
    return nt <= 3 ? DC(3) : DC(5);
#undef DC
}

template int stereo_cross_attention_disp_chunked16<float>(const float *, const float *, float *, int, int, int, int, int, int,
                                                          float, hipStream_t);
template int stereo_cross_attention_disp_chunked16<f16>(const f16 *, const f16 *, float *, int, int, int, int, int, int, float,
                                                        hipStream_t);
