// A1 (+ A4) on the f32 matrix cores: correlation volume vol[b,d,h,w] = mean_c L[b,c,h,w] * R[b,c,h,w-d] (0 for w < d),
// optionally folded into the soft-argmin in the same pass (vol stored AND regressed, never read back).
//
// Why a second formulation next to volume.hip's VALU kernel: that kernel is FMA-issue bound (v_fma_f32 holds a SIMD
// for 4 cycles) and its stores do not overlap its FMAs (one 12-wave workgroup per CU; vmcnt is one queue for loads and
// stores, so a wave's next operand loads wait for its own stores): 1.42 ms = 37 % of the HBM roofline at cfg-2.
// v_mfma_f32_32x32x2_f32 does the same exact-fp32 products at twice the rate, but an MFMA tile is (key x query) while
// the volume is (disparity x query): d = query - key runs along the tile's diagonals.  The skew goes through LDS:
//
//   * a workgroup = 8 waves = 256 consecutive queries of one image row, one LDS tile of keys (32 channels x
//     (256 + Dk) columns, zero-padded outside the image: the w < d zeros of the volume come for free);
//   * step s = 0 .. NT-1 handles, per wave, the 32-key tile at distance 32 s: 16 MFMAs (A = K from LDS, lanes = keys;
//     B = Q from registers with 1/C folded in) give S^T[key][query]; accumulator register r of lane (j, half) is the
//     volume entry (d = 32 s + j - i_r - 4 half, query j): it is written to the LDS "sheet" row d mod 96, column =
//     query -- one ds_write_b32 with an IMMEDIATE offset per register (the row of register r is a compile-time
//     distance from the lane's base row; only every third step, where the 96-row ring wraps, needs a select);
//   * after the barrier of step s the 32 sheet rows d = 32 (s-1) .. 32 s - 1 are complete (a row takes entries from two
//     neighbouring key tiles) and leave as 32 x 1 KiB row-contiguous global stores while step s+1 runs: the ring has
//     three 32-row blocks -- being read, complete, being written;
//   * ARGMIN: the online softmax of attention.hip's v5 kernel runs on the same accumulators (packed-f32 VALU, one
//     xor-32 exchange per tile) and the disparity leaves at the end of the tile.
// Persistent workgroups walk runs of tiles; the next tile's K quads / Q values are prefetched into registers under the
// current tile's MFMAs.  C <= 32 (one MFMA K-tile), W % 4 == 0, D <= 192 (LDS: 58 KB keys + 100 KB sheet; up to 256 for
// A1 / A1+A4 with six waves per workgroup).
#include "common.h"

namespace {

// Workgroup shape: CM_NWV waves = 32 * CM_NWV queries, a ring of RINGB = 3 32-row sheet blocks (being read / complete /
// being written), one barrier per step, 1-KiB row stores; 158 KB of LDS at D = 192 -> one workgroup per CU.  (A
// 128-query shape with a two-block ring fits two workgroups per CU but measured slower, 1.20 vs 1.13 ms for A1.)
// CPG2 > 0 adds the GROUP PASS in front (A3 + group mean + A4 in ONE kernel, cfg-2's whole volume stage): for each of
// the G groups of 2 * CPG2 channels the same step loop runs with CPG2 MFMAs per key tile (the group's channel pairs) and
// flushes into volg[b, g]; then the all-channel pass produces the mean volume and the disparity as before.  The key
// tile and the Q registers are fetched once for both; the 25.5 GB group-wise volume leaves through the same
// row-contiguous flush path (which sustains > 6 TB/s on its own) and is never read back.
// F16P (round 3; CPG2 == 0: A1 / A1 + A4 with the volume stored): the matrix phase on v_mfma_f32_32x32x16_f16 with split operands
// (common.h: x = hi + 2^-11 lo', two fp32 accumulators; fp16 features are their own single piece) -- 6 (2) MFMAs of 32 cycles per
// key tile instead of 16 of 64.  The key image keeps its footprint ([piece][channel][key] f16: 2 x 2 B = the fp32 image's 4 B per
// element); K fragments come through ds_read_b64_tr_b16, Q pieces sit in registers, 1/C moves to the accumulator.
template <int NT, bool ARGMIN, int CM_NWV, int RINGB, int CPG2 = 0, bool MEANPASS = true, typename TIN = float, bool F16P = false>
__global__ __launch_bounds__(CM_NWV * 64) void corr_volume_mfma_kernel(const TIN *__restrict__ L,
                                                                       const TIN *__restrict__ R,
                                                                       float *__restrict__ vol, float *__restrict__ disp,
                                                                       int C, int H, int W, int D, float inv_c, float sgn,
                                                                       int tiles_per_wg, int ntiles,
                                                                       float *__restrict__ volg, int G, float inv_cpg)
{
    constexpr int CM_TQW = 32 * CM_NWV;  // queries per workgroup tile
    constexpr int CM_RSTR = CM_TQW + 4;  // sheet row stride (floats): 16-B aligned rows, (RSTR + 1) odd -> conflict-free
                                         // diagonal writes
    constexpr int CM_RING = 32 * RINGB;  // sheet rows
    constexpr int Dk = 32 * (NT - 1);
    constexpr int NK = CM_TQW + Dk;
    constexpr int KS = NK + 4;
    constexpr int RMAX = (32 + CM_NWV - 1) / CM_NWV;  // staged rows per wave (row = wave, wave + NWV, ... < 32)
    constexpr int NPASS = (NK / 4 + 63) / 64;  // 16-B quads per lane and row
    static_assert(!F16P || CPG2 == 0, "the split-fp16 matrix phase serves the all-channel pass only");
    constexpr int NPK = sizeof(TIN) == 4 ? 2 : 1;   // F16P: pieces per feature value
    extern __shared__ f32x4 smem4[];
    f32x4 *Ks4 = smem4;
    float *Ks = reinterpret_cast<float *>(smem4);
    _Float16 *Kh = reinterpret_cast<_Float16 *>(smem4);   // F16P: [NPK][32][KS] f16 in the same bytes
    float *Sh = Ks + 32 * KS;  // [CM_RING][CM_RSTR]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;  // launcher: 32 * H * W < 2^31
    const int nx = (W + CM_TQW - 1) / CM_TQW;
    const int dl = l31 - 4 * half;                   // d of register r at step s = 32 s + dl - i_r
    const float *kp = Ks + half * KS + 32 * wave + l31;
    // sheet address of this lane's register r at step s: ((32 (s % RINGB) + dl - i_r) mod RING) * RSTR + 32 wave + l31
    //   = lane_base + 32 (s % RINGB) * RSTR + (27 - i_r) * RSTR    with lane_base = (dl - 27) * RSTR + column  (may be < 0)
    const int lane_base = (dl - 27) * CM_RSTR + 32 * wave + l31;

    int tile = blockIdx.x * tiles_per_wg;
    const int tile_end = (tile + tiles_per_wg < ntiles) ? tile + tiles_per_wg : ntiles;
    if (tile >= tile_end) return;

    typename Quad<TIN>::raw rk[RMAX][NPASS];  // raw quads (16 B fp32 / 8 B fp16), widened at commit
    TIN qn[16];                               // raw as well: a conversion right behind each load would serialise the 16 loads
    unsigned okbits = 0;
    auto issue = [&](int tl) {  // K quads and Q values of tile tl -> registers
        const int x = tl % nx, bh = tl / nx;
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * CM_TQW, kbase = w0 - Dk;
        const TIN *Rb = R + ((size_t)bb * C * HWu + (size_t)hh * W);
        const TIN *Lb = L + ((size_t)bb * C * HWu + (size_t)hh * W);
        okbits = 0;
#pragma unroll
        for (int pp = 0; pp < NPASS; ++pp) {
            const int w = kbase + 4 * (lane + 64 * pp);
            if (lane + 64 * pp < NK / 4 && w >= 0 && w + 3 < W) okbits |= 1u << pp;
        }
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * CM_NWV;
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                const bool ok = ((okbits >> pp) & 1u) && row < C && row < 32;
                rk[rr][pp] = Quad<TIN>::ld(Rb + (ok ? (unsigned)row * HWu + (unsigned)(kbase + 4 * (lane + 64 * pp)) : 0u));
            }
        }
        const int q = w0 + 32 * wave + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // f32 MFMA: k-step s contracts channels 2 s + half; f16 MFMA: k-step s >> 3, element s & 7 -> channel 16 (s >> 3) + 8 half + (s & 7)
            const int c = F16P ? 16 * (s >> 3) + 8 * half + (s & 7) : 2 * s + half;
            qn[s] = Lb[(c < C && q < W) ? (unsigned)c * HWu + (unsigned)q : 0u];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * CM_NWV;
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                if (row < 32 && lane + 64 * pp < NK / 4) {
                    const bool ok = ((okbits >> pp) & 1u) && row < C;
                    if constexpr (!F16P) {
                        Ks4[(row * KS) / 4 + lane + 64 * pp] = ok ? Quad<TIN>::widen(rk[rr][pp]) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    } else if constexpr (NPK == 2) {
                        const f32x4 v = ok ? Quad<TIN>::widen(rk[rr][pp]) : (f32x4){0.f, 0.f, 0.f, 0.f};
                        f16x2 a[2], b[2];
                        split_pair_f16((f32x2){v[0], v[1]}, a);
                        split_pair_f16((f32x2){v[2], v[3]}, b);
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            *reinterpret_cast<f16x4 *>(&Kh[(p * 32 + row) * KS + 4 * (lane + 64 * pp)]) =
                                __builtin_shufflevector(a[p], b[p], 0, 1, 2, 3);
                    } else {
                        const f16x4 z = {0, 0, 0, 0};
                        *reinterpret_cast<f16x4 *>(&Kh[row * KS + 4 * (lane + 64 * pp)]) =
                            ok ? __builtin_bit_cast(f16x4, rk[rr][pp]) : z;
                    }
                }
            }
        }
    };
    // 32 complete sheet rows (block blk, disparities dbase ..) -> global, 1 KiB per wave store
    auto flush_block = [&](float *__restrict__ dst, int blk, int dbase, size_t row_off, int w0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int qi = tid + k * (CM_NWV * 64);
            const int row = qi / (CM_TQW / 4), c4 = qi % (CM_TQW / 4);
            const int d = dbase + row, w = w0 + 4 * c4;
            if (d < D && w + 3 < W) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(Sh + (32 * blk + row) * CM_RSTR + 4 * c4);
                st4(dst + row_off + (size_t)d * HWu + (unsigned)w, v);
            }
        }
    };

    issue(tile);
    commit();
    float qv[16];
    f16x8 qf[2][NPK];   // F16P: Q pieces per k-step
    // F16P: transposed K reads (attention_bf16.hip): lane 4 q + p of its 16-lane group supplies channel row R0 + q, keys K0 + 4 p ..
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
    const int k_off = (8 * half + tq) * KS + 32 * wave + 16 * tg1 + 4 * tp;   // + (16 s + 4 tt) KS + 32 (key tile)
    const float l2 = sgn * 1.44269504088896340736f;  // mean -> logit in units of log2
    for (;;) {
        const int x = tile % nx, bh = tile / nx;
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * CM_TQW;
        const int q = w0 + 32 * wave + l31;
        const size_t row_off = (size_t)bb * D * HWu + (size_t)hh * W;  // vol[bb, 0, hh, 0]
        if constexpr (!F16P) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int c = 2 * s + half;
                qv[s] = (c < C && q < W) ? (float)qn[s] * inv_c : 0.0f;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float xq[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xq[j] = (16 * s + 8 * half + j < C && q < W) ? (float)qn[8 * s + j] : 0.0f;
                if constexpr (NPK == 2) {
                    split_oct_f16(xq, qf[s]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) qf[s][0][j] = (_Float16)xq[j];
                }
            }
        }
        __syncthreads();  // the key tile is visible; the previous tile's last sheet reads are done
        const bool more = tile + 1 < tile_end;
        if (more) issue(tile + 1);

        const bool wave_on = w0 + 32 * wave < W;  // wave-uniform
        // accumulator tile of key tile s -> sheet rows (32 s + dl - i_r) mod RING (rows outside [0, D) land in ring
        // slots nobody reads)
        auto sheet_write = [&](const f32x16 &a, int sb) {
            float *wp = Sh + lane_base + 32 * sb * CM_RSTR;
            if (sb != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) wp[(27 - ((r & 3) + 8 * (r >> 2))) * CM_RSTR] = a[r];
            } else {  // the ring wraps: rows with dl < i_r are RING rows up
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ir = (r & 3) + 8 * (r >> 2);
                    float *p = wp + (27 - ir) * CM_RSTR;
                    if (dl < ir) p += CM_RING * CM_RSTR;
                    *p = a[r];
                }
            }
        };
        if (CPG2 > 0) {  // ---------------- group pass: volg[bb, g, :, hh, w0 ..]
            float qg[16];  // Q scaled for a group mean (1 / channels per group)
#pragma unroll
            for (int t = 0; t < 16; ++t) qg[t] = qv[t] * (inv_cpg / inv_c);
#pragma unroll
            for (int g = 0; g < 16 / (CPG2 > 0 ? CPG2 : 1); ++g) {
                if (g < G) {  // workgroup-uniform
                    const size_t goff = ((size_t)bb * G + g) * D * HWu + (size_t)hh * W;
#pragma unroll 1
                    for (int s = 0; s < NT; ++s) {
                        if (wave_on) {
                            f32x16 a;
#pragma unroll
                            for (int r = 0; r < 16; ++r) a[r] = 0.0f;
                            const float *kpt = kp + 32 * (NT - 1 - s) + 2 * g * CPG2 * KS;
#pragma unroll
                            for (int t = 0; t < CPG2; ++t)
                                a = __builtin_amdgcn_mfma_f32_32x32x2f32(kpt[2 * t * KS], qg[g * CPG2 + t], a, 0, 0, 0);
                            sheet_write(a, s % RINGB);
                        }
                        __syncthreads();
                        if (s >= 1) flush_block(volg, (s + RINGB - 1) % RINGB, 32 * (s - 1), goff, w0);
                    }
                    __syncthreads();  // the last flush's sheet reads vs the next pass's first writes
                }
            }
        }
        float m = -INFINITY, ssum = 0.0f, esc = 0.0f;
#pragma unroll 1
        for (int s = 0; MEANPASS && s < NT; ++s) {  // (MEANPASS = false: A3 alone, the group pass above was all)
            const int sb = s % RINGB;  // ring block of the rows 32 s .. 32 s + 31
            if (wave_on) {
                f32x16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = 0.0f;
                if constexpr (!F16P) {
                    const float *kpt = kp + 32 * (NT - 1 - s);
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        a = __builtin_amdgcn_mfma_f32_32x32x2f32(kpt[2 * t * KS], qv[t], a, 0, 0, 0);
                } else {
                    f32x16 al = a;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        f16x8 kf[NPK];
#pragma unroll
                        for (int p = 0; p < NPK; ++p) {
#pragma unroll
                            for (int tt = 0; tt < 2; ++tt) {
                                const _Float16 *src = &Kh[p * 32 * KS + k_off + (16 * ks + 4 * tt) * KS + 32 * (NT - 1 - s)];
                                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)src);
                                const short tv[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                                for (int e = 0; e < 4; ++e) kf[p][4 * tt + e] = __builtin_bit_cast(_Float16, tv[e]);
                            }
                        }
                        if constexpr (NPK == 2) {
                            al = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[ks][1], al, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[ks][0], a, 0, 0, 0);
                            al = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[ks][0], al, 0, 0, 0);
                        } else {
                            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[ks][0], a, 0, 0, 0);
                        }
                    }
                    const float il = inv_c * SPLIT16_LO_UNSCALE;
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] = NPK == 2 ? fmaf(al[r], il, a[r] * inv_c) : a[r] * inv_c;
                }
                sheet_write(a, sb);
                if (ARGMIN) {
                    const int cb = 32 * s + dl;  // d of register r = cb - i_r
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] *= l2;
                    if (s == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (cb < (r & 3) + 8 * (r >> 2)) a[r] = -INFINITY;  // d < 0
                    }
                    if (32 * s + 31 >= D) {
                        const int lim = cb - D;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (lim >= (r & 3) + 8 * (r >> 2)) a[r] = -INFINITY;  // d >= D
                    }
                    float mt = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) mt = fmaxf(mt, fmaxf(a[r], a[r + 1]));
                    mt = fmaxf(mt, __shfl_xor(mt, 32));
                    const float mn = fmaxf(m, mt);
                    const float alpha = __builtin_amdgcn_exp2f(m - mn);
                    m = mn;
                    const f32x2 nm = {-mn, -mn};
                    f32x2 T2 = {0.0f, 0.0f}, U2 = {0.0f, 0.0f};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const f32x2 sh = (f32x2){a[2 * k], a[2 * k + 1]} + nm;
                        const f32x2 pe = {__builtin_amdgcn_exp2f(sh[0]), __builtin_amdgcn_exp2f(sh[1])};
                        T2 += pe;
                        const float base = (float)(2 * (k & 1) + 8 * (k >> 1));
                        U2 = __builtin_elementwise_fma(pe, (f32x2){base, base}, U2);
                    }
                    const float ts = T2[0] + T2[1], ti = U2[0] + U2[1] + T2[1];
                    ssum = fmaf(ssum, alpha, ts);
                    esc = fmaf(esc, alpha, fmaf((float)cb, ts, -ti));
                }
            }
            __syncthreads();  // step s is on the sheet: block (s-1) % RINGB is complete
            if (s >= 1) flush_block(vol, (s + RINGB - 1) % RINGB, 32 * (s - 1), row_off, w0);
        }
        // the last complete block: rows 32 (NT-2) .. Dk - 1 needed no further step?  No: they take entries from tile
        // NT-1 as well, which the loop has just written -> complete after its barrier.  (Rows >= Dk never exist.)
        if (ARGMIN && wave_on) {
            float e = esc;
            ssum += __shfl_xor(ssum, 32);
            e += __shfl_xor(e, 32);
            if (half == 0 && q < W) disp[(size_t)bh * W + q] = e / ssum;
        }
        if (!more) break;
        __syncthreads();  // every wave is done with the key tile
        commit();
        ++tile;
    }
}

template <int NT, bool ARGMIN, int NWV, int RINGB, int CPG2 = 0, bool MEANPASS = true, typename TIN = float, bool F16P = false>
int launch_cm(const TIN *L, const TIN *R, float *vol, float *disp, int B, int C, int H, int W, int D, float sgn,
              hipStream_t st, float *volg = nullptr, int G = 0)
{
    constexpr int Dk = 32 * (NT - 1), TQW = 32 * NWV;
    const size_t lds = ((size_t)32 * (TQW + Dk + 4) + (size_t)32 * RINGB * (TQW + 4)) * sizeof(float);
    auto kern = corr_volume_mfma_kernel<NT, ARGMIN, NWV, RINGB, CPG2, MEANPASS, TIN, F16P>;
    if (lds > 160 * 1024) return STEREO_ERR_UNSUPPORTED;
    static std::atomic<unsigned long long> lds_done{0};  // per instantiation: a bit per device
    if (!lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const long long ntl = (long long)((W + TQW - 1) / TQW) * H * B;
    if (ntl >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    const int resident = 256 * (int)((160 * 1024) / lds);  // persistent workgroups: as many as the LDS admits per CU
    int tpw = (int)((ntl + resident - 1) / resident);
    if (stereo_opt(OPT_TILES_PER_WG) > 0) tpw = stereo_opt(OPT_TILES_PER_WG);  // test hook: short persistent runs
    const int nwg = (int)((ntl + tpw - 1) / tpw);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NWV * 64), lds, st, L, R, vol, disp, C, H, W, D, 1.0f / (float)C, sgn, tpw,
                       (int)ntl, volg, G, CPG2 > 0 ? 1.0f / (float)(2 * CPG2) : 1.0f);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

// Returns STEREO_ERR_UNSUPPORTED when the shape is outside the kernel's envelope (the caller then takes the VALU path).
// disp == nullptr: store-only (A1).
template <typename TIN>
int stereo_corr_volume_mfma(const TIN *L, const TIN *R, float *vol, float *disp, int B, int C, int H, int W, int D,
                            float sgn, hipStream_t st)
{
    if (C > 32) return stereo_corr_volume_chunked<TIN>(L, R, vol, disp, B, C, H, W, D, sgn, st);  // 32-channel chunks
    if (D > 256 || D < 1 || W % 4 != 0 || !aligned_quad<TIN>(L) || !aligned_quad<TIN>(R) || !aligned16(vol))
        return STEREO_ERR_UNSUPPORTED;
    if ((long long)32 * H * W >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
    if (nt > 7)  // 192 < D <= 256: six waves (192 queries) keep key tile + sheet inside the 160 KB of LDS
        return disp ? launch_cm<9, true, 6, 3, 0, true, TIN>(L, R, vol, disp, B, C, H, W, D, sgn, st)
                    : launch_cm<9, false, 6, 3, 0, true, TIN>(L, R, vol, disp, B, C, H, W, D, sgn, st);
    const bool f16p = stereo_opt(OPT_CORR_F16P) != 0;   // test hook: 0 = the f32-MFMA matrix phase
#define CM_CASE(NTV)                                                                                              \
    do {                                                                                                          \
        if (f16p)                                                                                                 \
            return disp ? launch_cm<NTV, true, 8, 3, 0, true, TIN, true>(L, R, vol, disp, B, C, H, W, D, sgn, st)  \
                        : launch_cm<NTV, false, 8, 3, 0, true, TIN, true>(L, R, vol, disp, B, C, H, W, D, sgn, st); \
        return disp ? launch_cm<NTV, true, 8, 3, 0, true, TIN>(L, R, vol, disp, B, C, H, W, D, sgn, st)            \
                    : launch_cm<NTV, false, 8, 3, 0, true, TIN>(L, R, vol, disp, B, C, H, W, D, sgn, st);          \
    } while (0)
    if (nt <= 3) CM_CASE(3);
    if (nt <= 5) CM_CASE(5);
    CM_CASE(7);
#undef CM_CASE
}

// A3 + group mean + A4 fused (see the kernel's group pass).  volg [B,G,D,H,W], mean [B,D,H,W], disp [B,H,W].
// mean == nullptr && disp == nullptr: A3 alone (group pass only).
template <typename TIN>
int stereo_gwc_mean_argmin_mfma(const TIN *L, const TIN *R, float *volg, float *mean, float *disp, int B, int C, int H,
                                int W, int D, int G, float sgn, hipStream_t st)
{
    const bool only_groups = mean == nullptr && disp == nullptr;
    if (C > 32 || D > 192 || D < 1 || W % 4 != 0 || G < 1 || C % G != 0 || !aligned_quad<TIN>(L) || !aligned_quad<TIN>(R) ||
        !aligned16(volg) || (!only_groups && (!aligned16(mean) || mean == nullptr || disp == nullptr)))
        return STEREO_ERR_UNSUPPORTED;
    if ((long long)32 * H * W >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    const int cpg = C / G;
    if (cpg != 2 && cpg != 4 && cpg != 8 && cpg != 16) return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;  // (D <= 192 here: the fused launch is instantiated for the 8-wave shape only)
#define CMG(NTV, P2)                                                                                          \
    return only_groups ? launch_cm<NTV, false, 8, 3, P2, false, TIN>(L, R, mean, disp, B, C, H, W, D, sgn, st, volg, G) \
                       : launch_cm<NTV, true, 8, 3, P2, true, TIN>(L, R, mean, disp, B, C, H, W, D, sgn, st, volg, G)
#define CMG_NT(NTV)                \
    do {                           \
        if (cpg == 2) CMG(NTV, 1); \
        if (cpg == 4) CMG(NTV, 2); \
        if (cpg == 8) CMG(NTV, 4); \
        CMG(NTV, 8);               \
    } while (0)
    if (nt <= 3) CMG_NT(3);
    if (nt <= 5) CMG_NT(5);
    CMG_NT(7);
#undef CMG_NT
#undef CMG
}

template int stereo_corr_volume_mfma<float>(const float *, const float *, float *, float *, int, int, int, int, int, float,
                                            hipStream_t);
template int stereo_corr_volume_mfma<f16>(const f16 *, const f16 *, float *, float *, int, int, int, int, int, float,
                                          hipStream_t);
template int stereo_gwc_mean_argmin_mfma<float>(const float *, const float *, float *, float *, float *, int, int, int, int,
                                                int, int, float, hipStream_t);
template int stereo_gwc_mean_argmin_mfma<f16>(const f16 *, const f16 *, float *, float *, float *, int, int, int, int, int,
                                              int, float, hipStream_t);
