// Epipolar cross-attention (A7), C <= 32, value = right features, on the FP16 matrix cores with SPLIT operands (v9).
//
// attention_bf16.hip (v7) splits every fp32 operand into three bf16 pieces and runs six piece products per fp32 product: 22
// matrix instructions per 32-key tile (12 for Q K^T, 10 for P V).  Here every operand is TWO fp16 pieces and the three leading
// products are kept: 6 + 6 v_mfma_f32_32x32x16_f16 per tile, and the probabilities reach P V with 22 bits instead of v7's 16.
// Arithmetic (end of round 3; until then the low piece was scaled, x = hi + 2^-11 lo', and the cross terms had an accumulator
// of their own): a value is split as 16 x = hi + lo with an UNSCALED low piece -- hi = fp16(16 x), lo = fp16(16 x - hi), two
// v_fma_mix per value -- so hi hi + hi lo + lo hi accumulate in ONE fp32 accumulator (scores = acc / 256): no second
// accumulator, no zero accumulator, no combine per tile.  lo is a normal fp16 number for |x| >= 2^-7 (22-23 significant bits);
// below that it rounds on the subnormal grid, an absolute error of <= 2^-29 in x, invisible to the scores.  Probabilities are
// split the same way as 1024 p.  With the ring prefetching only a tile's own 256 keys the fp32-feature kernel of cfg-3 fits
// 125 registers: TWO 8-wave workgroups per CU (0.735 -> 0.65 ms).  fp16 FEATURES are their own single piece, exactly: the
// score scale sits on the accumulator and a tile costs 2 + 4.
// Range: |feature| < 4 094 (fp32 features: 16 x in fp16), < 65 504 (fp16 features); beyond it the result is inf / NaN, never
// silently wrong -- the fp32 kernels of attention.hip cover fp32's whole range (test hook att_kernel = 5 / 4).
//
// LDS image [piece][channel][key] f16 as in v7 (one image serves V^T row-wise and K through ds_read_b64_tr_b16), but channel
// row r starts 16 (r & 3) + 2 (r >> 2) dwords into a row slot that is a whole number of 256-B bank rows: the transposed K read
// of a 32-lane group touches 4 channel rows x 16 dwords -- with that offset the four tile the 64 banks exactly -- and the
// row-wise V^T read (32 channel rows at one key offset, 2 dwords each) lands on 32 distinct even banks.  (v7's packed rows,
// 226 dwords apart, made every transposed read 2-way conflicted: SQ_LDS_BANK_CONFLICT = 21 % of the LDS cycles.)
// Softmax, masks, disparity expectation, persistence and register prefetch are v5's / v7's.
// Key ring (256-query tiles with fp32 features, D > 128): a persistent workgroup takes consecutive tiles of an image row, and
// consecutive tiles share NK - 256 keys -- the image is a ring of 512 key columns and a tile that continues its
// predecessor's row stages only its 256 new keys (cfg-3: 0.78 -> 0.735 ms, disparity only 0.50 -> 0.45 ms).
#include "common.h"

namespace {

template <int NT, int NWV, typename TIN>
__global__ __launch_bounds__(NWV * 64, ((NWV == 8 && NT >= 6) ? 4 : 1)) void cross_attention_split16_kernel(const TIN *__restrict__ L, const TIN *__restrict__ R,
                                                                           float *__restrict__ out, float *__restrict__ disp,
                                                                           int C, int H, int W, int D, int mask_mode,
                                                                           float scale, int tiles_per_wg, int ntiles)
{
    constexpr int NP = sizeof(TIN) == 4 ? 2 : 1;  // pieces of a feature value
    constexpr int Dk = 32 * (NT - 1);
    constexpr int TQW = 32 * NWV;
    constexpr int NK = TQW + Dk;
    // RING (256-query tiles, long bands): consecutive tiles of an image row share NK - 256 keys, so the key image is a ring of
    // 512 columns -- a tile that continues its predecessor's row stages only its 256 NEW keys (loads, splits, LDS stores:
    // -43 % of the staging of such a tile; staging was 0.17-0.2 ms of the 0.8-ms cfg-3 launch) over the oldest 256 columns,
    // which nobody needs any more; the first tile of a row or of a workgroup stages all NK.  Window start = ring column
    // `rbase` (0 or 256), key tile k of the window sits at column (rbase + 32 k) mod 512.
    // (fp32 features only: the fp16-feature instantiation sits at 127 registers = two workgroups per CU, and its staging has
    // no splits to save)
    constexpr bool RING = NWV == 8 && NT >= 6 && NP == 2;
    constexpr int RCOLS = 512;
    constexpr int RSD = ((RING ? RCOLS : NK) / 2 + 62 + 63) / 64 * 64;   // row slot in dwords: whole bank rows + room for the offset
    constexpr int RSE = 2 * RSD;                        // ... in f16 elements
    constexpr int PSE = 32 * RSE;                       // piece stride
    constexpr int RMAX = (32 + NWV - 1) / NWV;
    constexpr int NPASS = (NK / 4 + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) _Float16 Im16[];   // [NP][32][RSE]

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;  // launcher: 32 * H * W < 2^30 (32-bit byte offsets)
    const int nx = (W + TQW - 1) / TQW;
    const float sc2 = scale * 1.44269504088896340736f;
    // One-accumulator arithmetic (end of round 3, from attention_chunked16.hip): a feature value is two fp16 pieces of PRE x with
    // an UNSCALED low piece -- hi = fp16(PRE x), lo = fp16(PRE x - hi), two v_fma_mix per value -- so that hi hi + hi lo + lo hi
    // accumulate in ONE fp32 accumulator (no second accumulator, no zero accumulator, no combine per tile); lo is a normal fp16
    // number for |x| >= 2^-7 and rounds on the subnormal grid below (absolute error <= 2^-29 in x).  The probabilities are
    // split the same way as PP p (PP = 2^10: normal low piece down to p = 2^-17).
    constexpr float PRE = NP == 2 ? 16.0f : 1.0f, PP = 1024.0f;
    float pre = PRE;
    asm volatile("" : "+v"(pre));
    auto split2 = [&](float x0, float x1, unsigned &h, unsigned &l) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(pre));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(pre));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(pre), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(pre), "v"(h));
    };
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
    const int dl = l31 - 4 * half;
    auto row_off = [](int row) { return row * RSE + 32 * (row & 3) + 4 * (row >> 2); };   // elements
    // transposed K reads: lane 4 q + p of its 16-lane group supplies row (channel) R0 + q, keys K0 + 4 p .. ; R0 = 8 half + 16 s
    // + 4 tt is a multiple of 4, so the row's offset splits into a lane part and a compile-time (s, tt) part
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    const int k_off = (8 * half + tq) * RSE + 32 * tq + 8 * half + 16 * tg1 + 4 * tp;  // + (16 s + 4 tt) RSE + 4 (4 s + tt) + column of the key tile
    const int v_off = row_off(l31) + 4 * half;                                        // + column of the key tile + 16 s (+ 8 for elements 4..7)

    int tile = blockIdx.x * tiles_per_wg;
    const int tile_end = (tile + tiles_per_wg < ntiles) ? tile + tiles_per_wg : ntiles;
    if (tile >= tile_end) return;

    // Loads go through buffer resources (common.h): a key outside [0, W) is an out-of-range byte offset and reads 0 -- the zero
    // padding of the band and what ZERO-mask mode wants for the values; channel rows >= C get an empty resource, Q channels
    // >= C fall past the end of theirs.  No per-lane bound selects.
    typename Quad<TIN>::raw rk[RMAX][RING ? 1 : NPASS];   // RING: a tile only ever prefetches its own 256 keys (below)
    TIN qn[16];
    constexpr int ES = (int)sizeof(TIN);
    bool incr = false;   // the prefetched tile continues the current one's row: only its 256 new keys are in `rk`
    int rbase = 0;       // ring column of the current image's first key
    // RING staging: every tile loads exactly its own 256 keys [w0, w0 + 256) -- one quad per lane and row.  A tile that
    // continues its predecessor's row (incr) appends them to the ring; any other tile restarts the ring at column 0 with its
    // Dk halo keys in columns 0 .. Dk - 1 and its own keys behind them.  The halo of a row's FIRST tile lies left of the image
    // (zeros, written at commit); only the first tile of a workgroup's run can start mid-row, and its halo is loaded and
    // committed synchronously by `stage_halo` before the loop.  (Until the end of round 3 a restart prefetched all 448 keys:
    // 16 more registers held for the whole kernel, the difference between one and two workgroups per CU.)
    bool halo_staged = false;
    auto issue = [&](int tl, bool first) {
        const int x = tl % nx, bh = tl / nx;
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * TQW, kbase = w0 - Dk;
        const TIN *Rb = R + ((size_t)bb * C * HWu + (size_t)hh * W);
        const TIN *Lb = L + ((size_t)bb * C * HWu + (size_t)hh * W);
        incr = RING && !first && x > 0;
        int voff[NPASS];
        if constexpr (RING) {
            voff[0] = (w0 + 4 * lane) * ES;      // keys w0 .. w0 + 255
        } else {
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) voff[pp] = (lane + 64 * pp < NK / 4) ? (kbase + 4 * (lane + 64 * pp)) * ES : -1;
        }
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {       // one resource per row: the range check sees voffset + soffset on gfx950
            const int row = wave + rr * NWV;
            const __amdgpu_buffer_rsrc_t rs = buffer_rsrc(Rb + (size_t)row * HWu, row < C ? (unsigned)W * ES : 0u);
            rk[rr][0] = BufLd<TIN>::quad(rs, voff[0], 0);
            if constexpr (!RING) {
#pragma unroll
                for (int pp = 1; pp < NPASS; ++pp) rk[rr][pp] = BufLd<TIN>::quad(rs, voff[pp], 0);
            }
        }
        const __amdgpu_buffer_rsrc_t rq = buffer_rsrc(Lb, ((unsigned)(C - 1) * HWu + (unsigned)W) * ES);
        int vq = (int)(((unsigned)(8 * half) * HWu + (unsigned)(w0 + 32 * wave + l31)) * ES);
#pragma unroll
        for (int i = 0; i < 16; ++i) {           // B operand order: k-step s = i >> 3, element j = i & 7 -> channel 16 s + 8 half + j
            qn[i] = BufLd<TIN>::one(rq, vq, 0);
            vq += (int)((i == 7 ? 9u : 1u) * HWu * ES);
        }
    };
    auto put_quad = [&](const typename Quad<TIN>::raw &q, int off) {   // one quad -> the piece image(s) at element offset `off`
        if constexpr (NP == 2) {
            const f32x4 v = Quad<TIN>::widen(q);
            unsigned h0, l0, h1, l1;
            split2(v[0], v[1], h0, l0);
            split2(v[2], v[3], h1, l1);
            *reinterpret_cast<u32x2v *>(&Im16[off]) = (u32x2v){h0, h1};
            *reinterpret_cast<u32x2v *>(&Im16[PSE + off]) = (u32x2v){l0, l1};
        } else {
            *reinterpret_cast<typename Quad<TIN>::raw *>(&Im16[off]) = q;
        }
    };
    auto commit = [&]() {  // registers -> the f16 piece images: one 8-byte store per quad and piece
        if constexpr (RING) {
            // a continuing tile's 256 keys follow the current window (columns rbase + NK ..., modulo 512) and the window moves on
            // by 256; a restart puts them behind the Dk halo columns and the ring at column 0
            const int c0 = incr ? (rbase + NK + 4 * lane) & (RCOLS - 1) : Dk + 4 * lane;
            const bool zero_halo = !incr && !halo_staged && lane < Dk / 4;
            rbase = incr ? (rbase + 256) & (RCOLS - 1) : 0;
            halo_staged = false;
#pragma unroll
            for (int rr = 0; rr < RMAX; ++rr) {
                const int row = wave + rr * NWV;
                if (row < 32) {
                    const int ro = row_off(row);
                    put_quad(rk[rr][0], ro + c0);
                    if (zero_halo) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2v *>(&Im16[p * PSE + ro + 4 * lane]) = (u32x2v){0u, 0u};
                    }
                }
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < RMAX; ++rr) {
                const int row = wave + rr * NWV;
                if (row < 32) {
                    const int ro = row_off(row);
#pragma unroll
                    for (int pp = 0; pp < NPASS; ++pp) {
                        if (lane + 64 * pp < NK / 4) put_quad(rk[rr][pp], ro + 4 * (lane + 64 * pp));
                    }
                }
            }
        }
    };
    auto stage_halo = [&](int tl) {   // RING, first tile of the run, mid-row: keys [w0 - Dk, w0) -> columns 0 .. Dk - 1, synchronously
        const int x = tl % nx, bh = tl / nx;
        const int hh = bh % H, bb = bh / H;
        if (x == 0) return;           // (left of the image: commit writes zeros)
        const TIN *Rb = R + ((size_t)bb * C * HWu + (size_t)hh * W);
        const int voff = lane < Dk / 4 ? (x * TQW - Dk + 4 * lane) * ES : -1;
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * NWV;
            const __amdgpu_buffer_rsrc_t rs = buffer_rsrc(Rb + (size_t)row * HWu, row < C ? (unsigned)W * ES : 0u);
            rk[rr][0] = BufLd<TIN>::quad(rs, voff, 0);
        }
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * NWV;
            if (row < 32 && lane < Dk / 4) put_quad(rk[rr][0], row_off(row) + 4 * lane);
        }
        halo_staged = true;
    };

    if constexpr (RING) stage_halo(tile);
    issue(tile, true);
    commit();
    f16x8 qf[2][NP];
    for (;;) {
        const int x = tile % nx, bh = tile / nx;
        const int w0 = x * TQW, kbase = w0 - Dk;
        const int q = w0 + 32 * wave + l31;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (NP == 2) {
                unsigned hq[4], lq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split2((float)qn[8 * s + 2 * j], (float)qn[8 * s + 2 * j + 1], hq[j], lq[j]);
                qf[s][0] = __builtin_bit_cast(f16x8, (u32x4w){hq[0], hq[1], hq[2], hq[3]});
                qf[s][1] = __builtin_bit_cast(f16x8, (u32x4w){lq[0], lq[1], lq[2], lq[3]});
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[s][0][j] = (_Float16)qn[8 * s + j];
            }
        }
        __syncthreads();  // the image is visible
        const bool more = tile + 1 < tile_end;
        if (more) issue(tile + 1, false);

        if (w0 + 32 * wave < W) {  // wave-uniform
            const bool neg_left = mask_mode == STEREO_MASK_NEGINF && kbase < 0;
            float m = -INFINITY, ssum = 0.0f, esc = 0.0f;
            f32x16 oh;
#pragma unroll
            for (int r = 0; r < 16; ++r) oh[r] = 0.0f;
#pragma unroll 1
            for (int t = NT - 1; t >= 0; --t) {
                f32x16 ah;
                int ct = 32 * (wave + t);    // column of key tile wave + t of the window (wave-uniform)
                if constexpr (RING) ct = (ct + rbase) & (RCOLS - 1);
                // ---- S^T tile = K_t Q^T: A = K through the transposed read, B = Q pieces
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x8 kf[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const _Float16 *src = &Im16[p * PSE + k_off + (16 * s + 4 * tt) * RSE + 4 * (4 * s + tt) + ct];
                            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
                            const short tv[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) kf[p][4 * tt + e] = __builtin_bit_cast(_Float16, tv[e]);
                        }
                    }
                    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[s][0], s == 0 ? z16 : ah, 0, 0, 0);   // (constant 0: an inline operand)
                    if constexpr (NP == 2) {
                        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[s][1], ah, 0, 0, 0);
                        ah = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[s][0], ah, 0, 0, 0);
                    }
                }
                // scores in exp2 units: acc x scale x log2 e / PRE^2
                f32x16 a;
                {   // (packed f32: one v_pk_mul_f32 per register pair)
                    const float su = sc2 * (1.0f / (PRE * PRE));
                    const f32x2 sh2 = {su, su};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const f32x2 v = (f32x2){ah[2 * k], ah[2 * k + 1]} * sh2;
                        a[2 * k] = v[0];
                        a[2 * k + 1] = v[1];
                    }
                }
                // ---- masks + online softmax (v5)
                const int cb = Dk - 32 * t + dl;
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
                for (int r = 0; r < 16; r += 2) mt = fmaxf(mt, fmaxf(a[r], a[r + 1]));
                mt = fmaxf(mt, __shfl_xor(mt, 32));
                const float mn = fmaxf(m, mt);
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                m = mn;
                // packed-f32 VALU on register pairs (2 x the scalar rate): register 2k sits at key offset 2 (k & 1) + 8 (k >> 1)
                // of its tile, register 2k + 1 one further
                const f32x2 nm = {-mn, -mn};
                f32x2 T2 = {0.0f, 0.0f}, U2 = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x2 sh = (f32x2){a[2 * k], a[2 * k + 1]} + nm;
                    const f32x2 pe = {__builtin_amdgcn_exp2f(sh[0]), __builtin_amdgcn_exp2f(sh[1])};
                    a[2 * k] = pe[0];
                    a[2 * k + 1] = pe[1];
                    T2 += pe;
                    const float base = (float)(2 * (k & 1) + 8 * (k >> 1));
                    U2 = __builtin_elementwise_fma(pe, (f32x2){base, base}, U2);
                }
                const float ts = T2[0] + T2[1], ti = U2[0] + U2[1] + T2[1];
                ssum = fmaf(ssum, alpha, ts);
                esc = fmaf(esc, alpha, fmaf((float)cb, ts, -ti));
                if (out) {
                    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {  // wave-uniform: the running maximum rarely moves after the nearest tiles
#pragma unroll
                        for (int r = 0; r < 16; ++r) oh[r] *= alpha;
                    }
                    // ---- O^T += V_t^T P: A = V^T rows from the image (accumulator key order), B = P as hi + 2^-11 lo'
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        // P = hi + 2^-11 lo': hi by one packed conversion per pair, lo' = 2^11 p - 2^11 hi by ONE mixed-precision
                        // FMA per element (fp32 arithmetic on the fp16 source 2^11 hi, fp16 result into one half of the
                        // destination) -- the compiler's form of split_oct_f16 is two conversions back to fp32, a packed
                        // subtraction, a packed multiply and a packed conversion per pair: 56 vector instructions per key tile
                        // instead of 32, in a loop that is bound by its vector instructions
                        f16x8 pq[2];
                        {
                            typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                            u32x4v hv, lv;
                            float cpp = PP;
                            asm volatile("" : "+v"(cpp));
#pragma unroll
                            for (int e = 0; e < 4; ++e) {   // hi = fp16(PP p), lo = fp16(PP p - hi): four mixed-precision FMAs per pair
                                unsigned hu, lu;
                                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hu) : "v"(a[8 * s + 2 * e]), "v"(cpp));
                                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hu) : "v"(a[8 * s + 2 * e + 1]), "v"(cpp));
                                asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lu) : "v"(a[8 * s + 2 * e]), "v"(cpp), "v"(hu));
                                asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                                    : "+v"(lu)
                                    : "v"(a[8 * s + 2 * e + 1]), "v"(cpp), "v"(hu));
                                hv[e] = hu;
                                lv[e] = lu;
                            }
                            pq[0] = __builtin_bit_cast(f16x8, hv), pq[1] = __builtin_bit_cast(f16x8, lv);
                        }
                        f16x8 vf[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const f16x4 lo = *reinterpret_cast<const f16x4 *>(&Im16[p * PSE + v_off + ct + 16 * s]);
                            const f16x4 hi = *reinterpret_cast<const f16x4 *>(&Im16[p * PSE + v_off + ct + 16 * s + 8]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) vf[p][e] = lo[e], vf[p][4 + e] = hi[e];
                        }
                        oh = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], pq[0], oh, 0, 0, 0);
                        oh = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], pq[1], oh, 0, 0, 0);
                        if constexpr (NP == 2) oh = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1], pq[0], oh, 0, 0, 0);
                    }
                }
            }
            float e = esc;
            ssum += __shfl_xor(ssum, 32);
            e += __shfl_xor(e, 32);
            const float inv = 1.0f / ssum;
            if (q < W) {
                if (disp && half == 0) disp[(size_t)bh * W + q] = e * inv;
                if (out) {
                    const int hh = bh % H, bb = bh / H;
                    float *ob = out + ((size_t)bb * C * HWu + (size_t)hh * W);
                    const __amdgpu_buffer_rsrc_t ro = buffer_rsrc(ob, ((unsigned)(C - 1) * HWu + (unsigned)W) * 4u);
                    int vo = (int)(((unsigned)(4 * half) * HWu + (unsigned)q) * 4u);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {   // accumulator row r -> channel (r & 3) + 8 (r >> 2) + 4 half; >= C: dropped
                        const float ov = oh[r] * (inv * (1.0f / (PRE * PP)));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov), ro, vo, 0, 0);
                        vo += (int)(((r & 3) == 3 ? 5u : 1u) * HWu * 4u);
                    }
                }
            }
        }
        if (!more) break;
        __syncthreads();  // every wave is done reading the image
        commit();
        ++tile;
    }
}

template <int NT, int NWV, typename TIN>
int launch_split16(const TIN *left, const TIN *right, float *out, float *disp, int B, int C, int H, int W, int D, int mask_mode,
                   float sc, hipStream_t st)
{
    constexpr int NK = 32 * NWV + 32 * (NT - 1);
    constexpr int RSD = (((NWV == 8 && NT >= 6 && sizeof(TIN) == 4) ? 512 : NK) / 2 + 62 + 63) / 64 * 64;   // (the kernel's RING / RCOLS)
    const size_t lds = (size_t)(sizeof(TIN) == 4 ? 2 : 1) * 32 * RSD * 4;
    auto kern = cross_attention_split16_kernel<NT, NWV, TIN>;
    static std::atomic<unsigned long long> lds_done{0};
    static std::atomic<int> per_cu{0};
    if (lds > 160 * 1024 || !lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const long long ntl = (long long)((W + 32 * NWV - 1) / (32 * NWV)) * H * B;
    if (ntl >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    // persistent: every resident workgroup takes an equal run of tiles
    const int resident = 256 * workgroups_per_cu(kern, NWV * 64, lds, per_cu);
    int tpw = (int)((ntl + resident - 1) / resident);
    if (stereo_opt(OPT_TILES_PER_WG) > 0) tpw = stereo_opt(OPT_TILES_PER_WG);  // test hook: short persistent runs
    const int nwg = (int)((ntl + tpw - 1) / tpw);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NWV * 64), lds, st, left, right, out, disp, C, H, W, D, mask_mode, sc, tpw,
                       (int)ntl);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

// Library-internal (common.h).  Envelope: C <= 32, D <= 192, W % 4 == 0, quad-aligned features, 32 H W < 2^30.
template <typename TIN>
int stereo_cross_attention_split16(const TIN *left, const TIN *right, float *out, float *disp, int B, int C, int H, int W, int D,
                                   int mask_mode, float sc, hipStream_t st)
{
    // (32 H W < 2^30: the buffer resources address a 32-channel slab of a row with 32-bit BYTE offsets)
    if (C > 32 || D > 192 || D < 1 || W % 4 != 0 || !aligned_quad<TIN>(left) || !aligned_quad<TIN>(right) ||
        (long long)32 * H * W >= (1LL << 30))
        return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
    // waves (32 queries each) per workgroup: 8 (256-query tiles: the band halo is loaded and split once per tile, so wide tiles
    // win even where they leave slots idle); rows of <= 192 / <= 128 queries take one 6- / 4-wave tile
    const int qt = (W + 31) / 32;
    int nwv = qt <= 4 ? 4 : (qt <= 6 ? 6 : 8);
    const int forced = stereo_opt(OPT_ATT_WAVES);  // test hook
    if (forced == 4 || forced == 6 || forced == 8) nwv = forced;
#define SPL16(NTV)                                                                                            \
    do {                                                                                                      \
        if (nwv == 8) return launch_split16<NTV, 8, TIN>(left, right, out, disp, B, C, H, W, D, mask_mode, sc, st); \
        if (nwv == 6) return launch_split16<NTV, 6, TIN>(left, right, out, disp, B, C, H, W, D, mask_mode, sc, st); \
        return launch_split16<NTV, 4, TIN>(left, right, out, disp, B, C, H, W, D, mask_mode, sc, st);               \
    } while (0)
    if (nt <= 3) SPL16(3);
    if (nt <= 5) SPL16(5);
    SPL16(7);
#undef SPL16
}

template int stereo_cross_attention_split16<float>(const float *, const float *, float *, float *, int, int, int, int, int, int,
                                                   float, hipStream_t);
template int stereo_cross_attention_split16<f16>(const f16 *, const f16 *, float *, float *, int, int, int, int, int, int, float,
                                                 hipStream_t);
