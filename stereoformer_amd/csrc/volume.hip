// Cost-volume builders (A1 correlation, A2 difference/concat, A3 group-wise).
//
// Correlation / group-wise volume:   vol[b,g,d,h,w] = mean_{c in g} L[b,c,h,w] * R[b,c,h,w-d]
// (0.0 for w < d).  Common scheme of the three kernels below: a workgroup owns an image-row segment of TW=256 pixels;
// the left tile [CC][TW] and the right tile with its disparity halo [CC][TW+Dpad] are staged in LDS with coalesced
// 16-byte loads along W; each wave owns a strip of DC disparities and keeps a DC x 4 accumulator block in registers
// (lane = 4 consecutive pixels), reading the right tile as a sliding window of ds_read_b128.  Every store is a full
// 1-KiB W-contiguous wave store (16 B/lane); the w<d zeros are written by the same kernel (the reference's new_zeros
// pass does not exist here).
//   gwc_volume_kernel       generic (ragged W, D > 256): one (batch, group) per workgroup, a barrier per chunk
//   gwc_volume_pipe_kernel  the fast path: software pipeline over rows x groups x chunks (default)
// Epilogues: store the volume, fold it into a soft-argmin (A1+A4, nothing stored), or both.
#include "common.h"

namespace {

constexpr int TW = 256;  // pixels per tile: 64 lanes x 4
constexpr int NW = 8;    // waves per workgroup
constexpr int NT = NW * 64;
constexpr int CCMAX = 8;  // channels staged per LDS round

// EPI_STORE: write the volume.  EPI_ARGMIN: fold every disparity plane into an
// online softmax-expectation instead (A1+A4 fused; `vol` is then disp [B,H,W]
// and G must be 1): each wave keeps (max, sum, d-weighted sum) for its own
// disparity chunks, the 8 partial states meet in LDS at the end.
enum { EPI_STORE = 0, EPI_ARGMIN = 1, EPI_BOTH = 2 };  // BOTH (pipelined kernel only): volume -> vol, disparity -> aux

template <int DC, bool VEC, int EPI>
__global__ __launch_bounds__(NT) void gwc_volume_kernel(const float *__restrict__ L, const float *__restrict__ R,
                                                        float *__restrict__ vol, int C, int H, int W, int D, int G,
                                                        int CPG, int Dpad, int CC, float inv_cpg, int exact_div,
                                                        float sgn)
{
    extern __shared__ f32x4 smem4[];
    float *Rs = reinterpret_cast<float *>(smem4);  // [CC][RS]
    const int RS = TW + Dpad;
    float *Ls = Rs + CC * RS;  // [CC][TW]

    const int w0 = blockIdx.x * TW;
    const int h = blockIdx.y;
    const int b = blockIdx.z / G, g = blockIdx.z % G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wl = lane * 4;

    const size_t HW = (size_t)H * W;
    const float *Lrow = L + ((size_t)b * C + (size_t)g * CPG) * HW + (size_t)h * W;
    const float *Rrow = R + ((size_t)b * C + (size_t)g * CPG) * HW + (size_t)h * W;
    float *vbase = vol + (((size_t)b * G + g) * D) * HW + (size_t)h * W;

    float sm_m[4], sm_s[4], sm_e[4];  // EPI_ARGMIN state
#pragma unroll
    for (int i = 0; i < 4; ++i) sm_m[i] = -INFINITY, sm_s[i] = 0.0f, sm_e[i] = 0.0f;

    const int npass = (D + NW * DC - 1) / (NW * DC);
    for (int pass = 0; pass < npass; ++pass) {
        const int d0 = (pass * NW + wave) * DC;
        const bool active = d0 < D;  // wave-uniform

        f32x4 acc[DC];
#pragma unroll
        for (int i = 0; i < DC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int c0 = 0; c0 < CPG; c0 += CC) {
            const int ncc = (CPG - c0) < CC ? (CPG - c0) : CC;
            __syncthreads();  // previous round's readers are done
            // ---- stage right tile (with halo) and left tile
            const int rq = RS / 4;
            for (int idx = tid; idx < ncc * rq; idx += NT) {
                const int c = idx / rq, x = (idx - c * rq) * 4;
                const int w = w0 - Dpad + x;
                const float *src = Rrow + (size_t)(c0 + c) * HW;
                f32x4 v;
                if (VEC && w >= 0 && w + 3 < W) {
                    v = ld4(src + w);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (w + i >= 0 && w + i < W) ? src[w + i] : 0.0f;
                }
                *reinterpret_cast<f32x4 *>(Rs + c * RS + x) = v;
            }
            const int lq = TW / 4;
            for (int idx = tid; idx < ncc * lq; idx += NT) {
                const int c = idx / lq, x = (idx - c * lq) * 4;
                const int w = w0 + x;
                const float *src = Lrow + (size_t)(c0 + c) * HW;
                f32x4 v;
                if (VEC && w + 3 < W) {
                    v = ld4(src + w);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (w + i < W) ? src[w + i] : 0.0f;
                }
                *reinterpret_cast<f32x4 *>(Ls + c * TW + x) = v;
            }
            __syncthreads();
            if (active) {
                // window start in the right tile: x = Dpad + wl - d0 - DC  (>= 0, multiple of 4)
                const float *rwin = Rs + (Dpad + wl - d0 - DC);
                for (int c = 0; c < ncc; ++c) {
                    const f32x4 lv = *reinterpret_cast<const f32x4 *>(Ls + c * TW + wl);
                    float win[DC + 4];
#pragma unroll
                    for (int q = 0; q < DC / 4 + 1; ++q) {
                        const f32x4 t = *reinterpret_cast<const f32x4 *>(rwin + c * RS + 4 * q);
                        win[4 * q + 0] = t[0];
                        win[4 * q + 1] = t[1];
                        win[4 * q + 2] = t[2];
                        win[4 * q + 3] = t[3];
                    }
#pragma unroll
                    for (int dd = 0; dd < DC; ++dd) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[dd][i] = fmaf(lv[i], win[DC + i - dd], acc[dd][i]);
                    }
                }
            }
        }
        if (active && EPI == EPI_ARGMIN) {
            const int w = w0 + wl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x[DC];
                float cm = sm_m[i];
#pragma unroll
                for (int dd = 0; dd < DC; ++dd) {
                    const float m = exact_div ? acc[dd][i] / (float)CPG : acc[dd][i] * inv_cpg;
                    x[dd] = (w + i >= d0 + dd) ? sgn * m : sgn * 0.0f;  // zero-filled logits take part (estimation.py:20)
                    if (d0 + dd < D) cm = fmaxf(cm, x[dd]);
                }
                const float sc = (sm_m[i] == cm) ? 1.0f : fast_exp(sm_m[i] - cm);
                float ss = sm_s[i] * sc, ee = sm_e[i] * sc;
#pragma unroll
                for (int dd = 0; dd < DC; ++dd)
                    if (d0 + dd < D) {
                        const float p = fast_exp(x[dd] - cm);
                        ss += p;
                        ee = fmaf(p, (float)(d0 + dd), ee);
                    }
                sm_m[i] = cm, sm_s[i] = ss, sm_e[i] = ee;
            }
        }
        if (active && EPI == EPI_STORE) {
            const int w = w0 + wl;
#pragma unroll
            for (int dd = 0; dd < DC; ++dd) {
                const int d = d0 + dd;
                if (d < D) {
                    f32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float m = exact_div ? acc[dd][i] / (float)CPG : acc[dd][i] * inv_cpg;
                        o[i] = (w + i >= d) ? m : 0.0f;
                    }
                    float *dst = vbase + (size_t)d * HW + w;
                    if (VEC && w + 3 < W) {
                        st4(dst, o);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (w + i < W) dst[i] = o[i];
                    }
                }
            }
        }
    }
    if (EPI == EPI_ARGMIN) {
        __syncthreads();
        float *stt = reinterpret_cast<float *>(smem4);  // [3][NW][TW]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            stt[(0 * NW + wave) * TW + wl + i] = sm_m[i];
            stt[(1 * NW + wave) * TW + wl + i] = sm_s[i];
            stt[(2 * NW + wave) * TW + wl + i] = sm_e[i];
        }
        __syncthreads();
        if (tid < TW && w0 + tid < W) {
            float M = -INFINITY;
#pragma unroll
            for (int k = 0; k < NW; ++k) M = fmaxf(M, stt[(0 * NW + k) * TW + tid]);
            float S = 0.0f, E = 0.0f;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float mk = stt[(0 * NW + k) * TW + tid];
                const float sc = (mk == -INFINITY) ? 0.0f : fast_exp(mk - M);
                S = fmaf(stt[(1 * NW + k) * TW + tid], sc, S);
                E = fmaf(stt[(2 * NW + k) * TW + tid], sc, E);
            }
            vol[((size_t)b * H + h) * W + w0 + tid] = E / S;
        }
    }
}

// ---------------------------------------------------------------------------
// Software-pipelined variant (the fast path: W % 4 == 0, Dpad <= 256).
// One workgroup = (row, 256-pixel tile, batch, disparity part) and walks ALL groups and channel
// chunks as a sequence of items (g, c0): while the waves run the FMA/LDS loop of item i, the
// 16-byte global loads of item i+1 are already in flight into registers; the store phase of a
// group overlaps the next group's loads the same way.  NWV waves per workgroup, each owning DC
// disparities; the store variant may split the disparity range over several small workgroups so
// that more independent workgroups share a CU (barriers and phases of different workgroups overlap).
template <int DC, int NWV, int EPI>
__global__ __launch_bounds__(NWV * 64) void gwc_volume_pipe_kernel(const float *__restrict__ L,
                                                                   const float *__restrict__ R,
                                                                   float *__restrict__ vol, float *__restrict__ aux,
                                                                   int C, int H, int W, int D, int G, int CPG, int Dpad,
                                                                   int CC, float inv_cpg, int exact_div, float sgn,
                                                                   int nparts, int RB)
{
    constexpr bool DO_STORE = EPI == EPI_STORE || EPI == EPI_BOTH;
    constexpr bool DO_ARGMIN = EPI == EPI_ARGMIN || EPI == EPI_BOTH;
    constexpr int TWG = TW;  // pixels per workgroup
    extern __shared__ f32x4 smem4[];
    float *Rs = reinterpret_cast<float *>(smem4);  // [CC][RS]
    const int RS = TWG + Dpad;
    float *Ls = Rs + CC * RS;  // [CC][TWG]

    constexpr int NTV = NWV * 64;
    constexpr int NSR = (CCMAX * (TWG + 256) / 4 + NTV - 1) / NTV;  // staging slots per lane: right tile (Dpad <= 256)
    constexpr int NSL = (CCMAX * TWG / 4 + NTV - 1) / NTV;          //                         left tile
    static_assert(NSR <= 8 && NSL <= 8, "okmask holds 8 + 8 slot bits");
    const int w0 = blockIdx.x * TWG, h0 = blockIdx.y * RB;  // RB consecutive rows per workgroup
    const int nrows = (H - h0) < RB ? (H - h0) : RB;
    const int b = blockIdx.z / nparts, dpart = blockIdx.z - b * nparts;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wl = lane * 4;
    const size_t HW = (size_t)H * W;
    const float *Lb = L + (size_t)b * C * HW;
    const float *Rb = R + (size_t)b * C * HW;

    const int d0 = (dpart * NWV + wave) * DC;
    const bool active = d0 < D;  // wave-uniform

    const int rq = RS / 4, lq = TWG / 4;
    int rc[NSR], rx[NSR], lc[NSL], lx[NSL];
#pragma unroll
    for (int k = 0; k < NSR; ++k) {
        const int q = tid + k * NTV;
        rc[k] = q / rq;
        rx[k] = (q - rc[k] * rq) * 4;
    }
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        const int q = tid + k * NTV;
        lc[k] = q / lq;
        lx[k] = (q - lc[k] * lq) * 4;
    }

    // Per-slot 32-bit element offsets inside one (batch, row-0, chunk-0) frame, computed ONCE: every item's
    // address is then a wave-uniform base (row, chunk) + this per-lane offset, i.e. the SGPR-base form of
    // global_load with no per-item 64-bit VALU address arithmetic.
    unsigned roff[NSR], loff[NSL], wmask = 0;
#pragma unroll
    for (int k = 0; k < NSR; ++k) {
        const int w = w0 - Dpad + rx[k];
        const bool okw = rc[k] < CC && w >= 0 && w + 3 < W;
        roff[k] = okw ? (unsigned)rc[k] * (unsigned)HW + (unsigned)w : 0u;
        wmask |= okw ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        const int w = w0 + lx[k];
        const bool okw = lc[k] < CC && w + 3 < W;
        loff[k] = okw ? (unsigned)lc[k] * (unsigned)HW + (unsigned)w : 0u;
        wmask |= okw ? (1u << (8 + k)) : 0u;
    }
    f32x4 pr[NSR], pl[NSL];
    unsigned okmask = 0;  // validity of the prefetched quads; applied at COMMIT time so that the loads stay
                          // in flight (a select on the loaded value right here would force an immediate vmcnt wait)
    auto issue_loads = [&](int h, int cbase, int ncc) {  // channels cbase .. cbase+ncc-1 of row h
        const float *Rrow = Rb + ((size_t)h * W + (size_t)cbase * HW);  // wave-uniform
        const float *Lrow = Lb + ((size_t)h * W + (size_t)cbase * HW);
        okmask = wmask;
#pragma unroll
        for (int k = 0; k < NSR; ++k) {
            const bool okc = rc[k] < ncc;  // only the tail chunk of a group can fail this
            pr[k] = ld4(Rrow + (okc ? roff[k] : 0u));
            if (!okc) okmask &= ~(1u << k);
        }
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const bool okc = lc[k] < ncc;
            pl[k] = ld4(Lrow + (okc ? loff[k] : 0u));
            if (!okc) okmask &= ~(1u << (8 + k));
        }
    };

    const int nchunk = (CPG + CC - 1) / CC;
    const int per_row = G * nchunk;
    const int nitems = nrows * per_row;  // the pipeline runs across groups, chunks AND rows: the stores of one
                                         // row drain while the next row is being computed
    const int BUF = CC * (RS + TWG);  // floats per LDS buffer; two buffers, ONE barrier per item
    auto item_loads = [&](int item) {
        const int r2 = item / per_row, i2 = item - r2 * per_row;
        const int g2 = i2 / nchunk, ck2 = i2 - g2 * nchunk;
        const int c2 = ck2 * CC;
        issue_loads(h0 + r2, g2 * CPG + c2, (CPG - c2) < CC ? (CPG - c2) : CC);
    };
    auto commit = [&](int buf) {  // prefetched registers -> LDS buffer `buf`
        float *rs = Rs + buf * BUF, *ls = Ls + buf * BUF;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NSR; ++k)
            if (rc[k] < CC) *reinterpret_cast<f32x4 *>(rs + rc[k] * RS + rx[k]) = (okmask >> k) & 1u ? pr[k] : zero;
#pragma unroll
        for (int k = 0; k < NSL; ++k)
            if (lc[k] < CC)
                *reinterpret_cast<f32x4 *>(ls + lc[k] * TWG + lx[k]) = (okmask >> (8 + k)) & 1u ? pl[k] : zero;
    };
    item_loads(0);
    commit(0);
    if (nitems > 1) item_loads(1);
    __syncthreads();

    f32x4 acc[DC];
    for (int it = 0; it < nitems; ++it) {
        const int r = it / per_row, ir = it - r * per_row;
        const int g = ir / nchunk, ck = ir - g * nchunk;
        const int h = h0 + r;
        const int c0 = ck * CC;
        const int ncc = (CPG - c0) < CC ? (CPG - c0) : CC;
        if (ck == 0) {
#pragma unroll
            for (int i = 0; i < DC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const float *Rcur = Rs + (it & 1) * BUF, *Lcur = Ls + (it & 1) * BUF;
        if (active) {
            const float *rwin = Rcur + (Dpad + wl - d0 - DC);
            for (int c = 0; c < ncc; ++c) {
                const f32x4 lv = *reinterpret_cast<const f32x4 *>(Lcur + c * TWG + wl);
                float win[DC + 4];
#pragma unroll
                for (int q = 0; q < DC / 4 + 1; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(rwin + c * RS + 4 * q);
                    win[4 * q + 0] = t[0];
                    win[4 * q + 1] = t[1];
                    win[4 * q + 2] = t[2];
                    win[4 * q + 3] = t[3];
                }
#pragma unroll
                for (int dd = 0; dd < DC; ++dd) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[dd][i] = fmaf(lv[i], win[DC + i - dd], acc[dd][i]);
                }
            }
        }
        // Item it+1 (in registers since the previous iteration) -> the other LDS buffer (its last readers, item
        // it-1, passed the barrier that ended the previous iteration), then item it+2's loads are issued.
        // Placed AFTER this item's math and BEFORE its stores: the commit's vmcnt wait then only has to cover
        // loads issued a whole item ago and the PREVIOUS group's stores, which had this item's math to drain
        // (vmcnt counts loads and stores in one in-order queue).
        if (it + 1 < nitems) commit((it + 1) & 1);
        // the group's sums become means: x 2^-k when C/G is a power of two (exact), else ONE uniform branch to a true
        // division pass (bit-faithful to the reference's mean) -- not a branch per element in the epilogues below
        if (ck == nchunk - 1 && active && exact_div) {
#pragma unroll
            for (int dd = 0; dd < DC; ++dd)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[dd][i] = acc[dd][i] / (float)CPG;
        }
        const float mscale = exact_div ? 1.0f : inv_cpg;
        if (DO_ARGMIN && ck == nchunk - 1 && active) {
            // soft-argmin epilogue, placed before the next loads are issued (the prefetch registers are dead here).  The
            // store+argmin variant (EPI_BOTH) still spills 13 VGPRs at its 168-register budget; measured cost ~nil.
            const int w = w0 + wl;
            // G == 1 here (launcher), so the end of the group is the end of the row: this wave's softmax state over
            // its DC planes goes straight to the LDS exchange (no state carried across items -- with the store
            // epilogue in the same kernel those registers, and a DC-float logit array, spilled to scratch).  The
            // logit of (dd, i) is recomputed from the accumulator in both passes.
            float *stt = reinterpret_cast<float *>(smem4) + 2 * BUF;  // [3][NWV][TW], its own LDS region
            auto logit = [&](int dd, int i) {
                const float m = acc[dd][i] * mscale;
                return (w + i >= d0 + dd) ? sgn * m : sgn * 0.0f;
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float cm = -INFINITY;
#pragma unroll
                for (int dd = 0; dd < DC; ++dd)
                    if (d0 + dd < D) cm = fmaxf(cm, logit(dd, i));
                float ss = 0.0f, ee = 0.0f;
#pragma unroll
                for (int dd = 0; dd < DC; ++dd)
                    if (d0 + dd < D) {
                        const float p = fast_exp(logit(dd, i) - cm);
                        ss += p;
                        ee = fmaf(p, (float)(d0 + dd), ee);
                    }
                stt[(0 * NWV + wave) * TW + wl + i] = cm;
                stt[(1 * NWV + wave) * TW + wl + i] = ss;
                stt[(2 * NWV + wave) * TW + wl + i] = ee;
            }
        }
        if (it + 2 < nitems) item_loads(it + 2);
        if (DO_STORE && ck == nchunk - 1 && active) {  // group finished: store epilogue
            const int w = w0 + wl;
            if (DO_STORE) {
                float *vbase = vol + (((size_t)b * G + g) * D) * HW + (size_t)h * W;
                const bool interior = (w0 >= d0 + DC - 1) && (w0 + TW <= W) && (d0 + DC <= D);  // wave-uniform
                if (interior) {  // no w<d zeros, no ragged edge: straight scaled stores
#pragma unroll
                    for (int dd = 0; dd < DC; ++dd) {
                        f32x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = acc[dd][i] * mscale;
                        st4(vbase + (size_t)(d0 + dd) * HW + w, o);
                    }
                } else {
#pragma unroll
                    for (int dd = 0; dd < DC; ++dd) {
                        const int d = d0 + dd;
                        if (d < D) {
                            f32x4 o;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float m = acc[dd][i] * mscale;
                                o[i] = (w + i >= d) ? m : 0.0f;
                            }
                            if (w + 3 < W) st4(vbase + (size_t)d * HW + w, o);
                        }
                    }
                }
            }
        }
        if (DO_ARGMIN && ir == per_row - 1) {  // row finished: combine the waves' softmax states
            // (single disparity part: the launcher guarantees nparts == 1).  Uses its own LDS region.
            float *stt = reinterpret_cast<float *>(smem4) + 2 * BUF;  // [3][NWV][TW]
            if (!active) {  // a wave past the last disparity contributes the neutral state
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    stt[(0 * NWV + wave) * TW + wl + i] = -INFINITY;
                    stt[(1 * NWV + wave) * TW + wl + i] = 0.0f;
                    stt[(2 * NWV + wave) * TW + wl + i] = 0.0f;
                }
            }
            __syncthreads();
            if (tid < TW && w0 + tid < W) {
                float M = -INFINITY;
#pragma unroll
                for (int k = 0; k < NWV; ++k) M = fmaxf(M, stt[(0 * NWV + k) * TW + tid]);
                float S = 0.0f, E = 0.0f;
#pragma unroll
                for (int k = 0; k < NWV; ++k) {
                    const float mk = stt[(0 * NWV + k) * TW + tid];
                    const float sc = (mk == -INFINITY) ? 0.0f : fast_exp(mk - M);
                    S = fmaf(stt[(1 * NWV + k) * TW + tid], sc, S);
                    E = fmaf(stt[(2 * NWV + k) * TW + tid], sc, E);
                }
                (EPI == EPI_BOTH ? aux : vol)[((size_t)b * H + h) * W + w0 + tid] = E / S;
            }
        }
        __syncthreads();  // all waves: done reading buffer it&1, buffer (it+1)&1 fully written
    }
}

template <int DC, int NWV, int EPI>
int launch_pipe(const float *L, const float *R, float *vol, int B, int C, int H, int W, int D, int G, float sgn,
                hipStream_t st, float *aux = nullptr)
{
    const int CPG = C / G;
    const int Dpad = ((D + DC - 1) / DC) * DC;
    const int CC = CPG < CCMAX ? CPG : CCMAX;
    size_t lds = (size_t)2 * CC * (size_t)(TW + Dpad + TW) * sizeof(float);  // two staging buffers
    if (EPI != EPI_STORE) lds += (size_t)3 * NWV * TW * sizeof(float);         // + softmax-state exchange
    int RB = 4;                                                                // rows per workgroup
    while (RB > 1 && (long long)((H + RB - 1) / RB) * ((W + TW - 1) / TW) * B * ((D + NWV * DC - 1) / (NWV * DC)) < 2048)
        RB >>= 1;                                                              // keep >= 8 workgroups per CU
    const int nparts = (D + NWV * DC - 1) / (NWV * DC);
    if (Dpad > 256 || lds > 160 * 1024 || !fits_grid_yz((long long)B * nparts) || (EPI != EPI_STORE && nparts != 1))
        return STEREO_ERR_UNSUPPORTED;
    if ((long long)CCMAX * H * W + W >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;  // 32-bit per-slot offsets
    static std::atomic<unsigned long long> lds_done{0};
    if (!lds_opt_in(gwc_volume_pipe_kernel<DC, NWV, EPI>, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const bool pow2 = (CPG & (CPG - 1)) == 0;
    dim3 pgrid((W + TW - 1) / TW, (H + RB - 1) / RB, (unsigned)(B * nparts));
    hipLaunchKernelGGL((gwc_volume_pipe_kernel<DC, NWV, EPI>), pgrid, dim3(NWV * 64), lds, st, L, R, vol, aux, C, H, W, D,
                       G, CPG, Dpad, CC, 1.0f / (float)CPG, pow2 ? 0 : 1, sgn, nparts, RB);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

// Pipelined-path configuration (DC disparities per wave, NWV waves per workgroup), measured on MI355X at cfg-2:
// (16, 12) is the fastest for both epilogues; (8, 8) for short disparity ranges.
template <int EPI>
int pipe_dispatch(const float *L, const float *R, float *vol, int B, int C, int H, int W, int D, int G, float sgn,
                  hipStream_t st)
{
    if (D <= 64) return launch_pipe<8, 8, EPI>(L, R, vol, B, C, H, W, D, G, sgn, st);
    return launch_pipe<16, 12, EPI>(L, R, vol, B, C, H, W, D, G, sgn, st);
}

template <int DC, int EPI>
int launch_gwc(const float *L, const float *R, float *vol, int B, int C, int H, int W, int D, int G, float sgn,
               hipStream_t st)
{
    const int CPG = C / G;
    const int Dpad = ((D + DC - 1) / DC) * DC;
    const int CC = CPG < CCMAX ? CPG : CCMAX;
    size_t lds = (size_t)CC * (size_t)(TW + Dpad + TW) * sizeof(float);
    if (EPI == EPI_ARGMIN && lds < (size_t)3 * NW * TW * sizeof(float)) lds = (size_t)3 * NW * TW * sizeof(float);
    if (lds > 160 * 1024) return STEREO_ERR_UNSUPPORTED;
    const bool pow2 = (CPG & (CPG - 1)) == 0;
    const bool vec = (W % 4 == 0) && aligned16(L) && aligned16(R) && aligned16(vol);
    dim3 grid((W + TW - 1) / TW, H, (unsigned)(B * G));
    auto kern = vec ? gwc_volume_kernel<DC, true, EPI> : gwc_volume_kernel<DC, false, EPI>;
    static std::atomic<unsigned long long> lds_done[2] = {{0}, {0}};
    if (!lds_opt_in(kern, lds, lds_done[vec ? 1 : 0])) return STEREO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, L, R, vol, C, H, W, D, G, CPG, Dpad, CC, 1.0f / (float)CPG,
                       pow2 ? 0 : 1, sgn);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

template <int EPI>
int gwc_dispatch(const float *L, const float *R, float *vol, int B, int C, int H, int W, int D, int G, float sgn,
                 void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || D < 0 || G <= 0) return STEREO_ERR_BAD_SHAPE;
    if (C % G != 0) return STEREO_ERR_BAD_SHAPE;  // GWC_Cost_Volume.py:7
    if (B == 0 || C == 0 || H == 0 || W == 0 || D == 0) {
        // C == 0 with D > 0: mean over nothing; the reference would give NaN.  Not a use case.
        return C == 0 && B && H && W && D ? STEREO_ERR_BAD_SHAPE : STEREO_OK;
    }
    STEREO_REQUIRE_PTR(L);
    STEREO_REQUIRE_PTR(R);
    STEREO_REQUIRE_PTR(vol);
    if (!fits_grid_yz(H) || !fits_grid_yz((long long)B * G)) return STEREO_ERR_UNSUPPORTED;
    if (D > 4096) return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if ((W % 4 == 0) && aligned16(L) && aligned16(R) && aligned16(vol) && (EPI == EPI_STORE || G == 1)) {
        const int rc = pipe_dispatch<EPI>(L, R, vol, B, C, H, W, D, G, sgn, st);
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;  // otherwise: generic kernel below
    }
    if (D <= NW * 8) return launch_gwc<8, EPI>(L, R, vol, B, C, H, W, D, G, sgn, st);
    if (D <= NW * 16) return launch_gwc<16, EPI>(L, R, vol, B, C, H, W, D, G, sgn, st);
    return launch_gwc<24, EPI>(L, R, vol, B, C, H, W, D, G, sgn, st);
}

// ---------------------------------------------------------------- A2 (copy-bound volumes)
// out[b, c, d, h, w]: one thread per 4 consecutive w of one (b,c,d,h) row.
template <bool CONCAT>
__global__ __launch_bounds__(256) void shift_volume_kernel(const float *__restrict__ L, const float *__restrict__ R,
                                                           float *__restrict__ vol, int C, int H, int W, int D,
                                                           long long nquads, int vec_ok)
{
    const int Wq = (W + 3) / 4;
    const int Cout = CONCAT ? 2 * C : C;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
         q += (long long)gridDim.x * blockDim.x) {
        const int wq = (int)(q % Wq);
        long long r = q / Wq;
        const int h = (int)(r % H);
        r /= H;
        const int d = (int)(r % D);
        r /= D;
        const int c = (int)(r % Cout);
        const int b = (int)(r / Cout);
        const int w = wq * 4;
        float *dst = vol + ((((size_t)b * Cout + c) * D + d) * H + h) * (size_t)W + w;
        const int cs = (CONCAT && c >= C) ? c - C : c;
        const float *lrow = L + (((size_t)b * C + cs) * H + h) * (size_t)W;
        const float *rrow = R + (((size_t)b * C + cs) * H + h) * (size_t)W;
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = 0.0f;
            if (w + i < W && w + i >= d) {
                if (CONCAT)
                    v = (c < C) ? lrow[w + i] : rrow[w + i - d];
                else
                    v = lrow[w + i] - rrow[w + i - d];
            }
            o[i] = v;
        }
        if (vec_ok && w + 3 < W) {
            st4(dst, o);  // 16-B W-contiguous store
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (w + i < W) dst[i] = o[i];
        }
    }
}

// Fast path (W % 4 == 0, 16-B aligned): a workgroup writes whole rows of DS consecutive disparity planes of one
// (b, c, h).  Each lane keeps its left quad and a (DS+4)-float right window in registers (five aligned 16-B loads,
// L2 hits: the features are ~1000x smaller than the volume), so the kernel is a pure row-contiguous store stream.
template <bool CONCAT, int DS>
__global__ __launch_bounds__(256) void shift_volume_rows_kernel(const float *__restrict__ L,
                                                                const float *__restrict__ R, float *__restrict__ vol,
                                                                int C, int H, int W, int D)
{
    const int Cout = CONCAT ? 2 * C : C;
    const int h = blockIdx.y, bc = blockIdx.z;
    const int b = bc / Cout, c = bc - b * Cout;
    const int cs = (CONCAT && c >= C) ? c - C : c;
    const bool left_copy = CONCAT && c < C;  // workgroup-uniform
    const size_t HW = (size_t)H * W;
    const float *lrow = L + (((size_t)b * C + cs) * H + h) * (size_t)W;
    const float *rrow = R + (((size_t)b * C + cs) * H + h) * (size_t)W;
    const int d0 = blockIdx.x * DS;
    float *dst0 = vol + ((size_t)bc * D + d0) * HW + (size_t)h * W;
    for (int w = threadIdx.x * 4; w < W; w += 1024) {
        f32x4 lv = {0.f, 0.f, 0.f, 0.f};
        if (!CONCAT || left_copy) lv = ld4(lrow + w);
        float win[DS + 4];  // win[k] = R[w - d0 - DS + k]; the quad of plane d0+dd is win[DS-dd .. DS-dd+3]
        if (!left_copy) {
            const int base = w - d0 - DS;  // multiple of 4: every quad is entirely in or out of the row
#pragma unroll
            for (int q = 0; q < DS / 4 + 1; ++q) {
                const f32x4 t = (base + 4 * q >= 0) ? ld4(rrow + base + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
                win[4 * q + 0] = t[0], win[4 * q + 1] = t[1], win[4 * q + 2] = t[2], win[4 * q + 3] = t[3];
            }
        }
#pragma unroll
        for (int dd = 0; dd < DS; ++dd) {
            const int d = d0 + dd;
            if (d < D) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = left_copy ? lv[i] : (CONCAT ? win[DS - dd + i] : lv[i] - win[DS - dd + i]);
                    o[i] = (w + i >= d) ? v : 0.0f;
                }
                st4(dst0 + (size_t)dd * HW + w, o);
            }
        }
    }
}

template <bool CONCAT>
int shift_dispatch(const float *L, const float *R, float *vol, int B, int C, int H, int W, int D, void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || D < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || C == 0 || H == 0 || W == 0 || D == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(L);
    STEREO_REQUIRE_PTR(R);
    STEREO_REQUIRE_PTR(vol);
    const long long BC = (long long)B * (CONCAT ? 2 * C : C);
    if (W % 4 == 0 && aligned16(L) && aligned16(R) && aligned16(vol) && fits_grid_yz(H) && fits_grid_yz(BC)) {
        constexpr int DS = 16;
        hipLaunchKernelGGL((shift_volume_rows_kernel<CONCAT, DS>), dim3((unsigned)((D + DS - 1) / DS), H, (unsigned)BC),
                           dim3(256), 0, as_stream(stream), L, R, vol, C, H, W, D);
        STEREO_LAUNCH_CHECK();
        return STEREO_OK;
    }
    const long long nquads = (long long)B * (CONCAT ? 2 * C : C) * D * H * ((W + 3) / 4);
    long long blocks = (nquads + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(shift_volume_kernel<CONCAT>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), L, R, vol,
                       C, H, W, D, nquads, (W % 4 == 0 && aligned16(vol)) ? 1 : 0);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

static bool corr_mfma_enabled() { return stereo_opt(OPT_CORR_MFMA) != 0; }

extern "C" int stereo_corr_volume_f32(const float *left, const float *right, float *vol, int B, int C, int H, int W,
                                      int D, void *stream)
{
    if (B > 0 && C > 0 && H > 0 && W > 0 && D > 0 && left && right && vol && corr_mfma_enabled()) {
        const int rc = stereo_corr_volume_mfma(left, right, vol, nullptr, B, C, H, W, D, 1.0f, as_stream(stream));
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    return gwc_dispatch<EPI_STORE>(left, right, vol, B, C, H, W, D, 1, 1.0f, stream);
}

extern "C" int stereo_gwc_volume_f32(const float *left, const float *right, float *vol, int B, int C, int H, int W,
                                     int D, int G, void *stream)
{
    if (B > 0 && C > 0 && H > 0 && W > 0 && D > 0 && G > 0 && C % G == 0 && left && right && vol && corr_mfma_enabled()) {
        const int rc = G == 1 ? stereo_corr_volume_mfma(left, right, vol, nullptr, B, C, H, W, D, 1.0f, as_stream(stream))
                              : stereo_gwc_mean_argmin_mfma(left, right, vol, nullptr, nullptr, B, C, H, W, D, G, 1.0f,
                                                            as_stream(stream));
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    return gwc_dispatch<EPI_STORE>(left, right, vol, B, C, H, W, D, G, 1.0f, stream);
}

extern "C" int stereo_diff_volume_f32(const float *left, const float *right, float *vol, int B, int C, int H, int W,
                                      int D, void *stream)
{
    return shift_dispatch<false>(left, right, vol, B, C, H, W, D, stream);
}

extern "C" int stereo_concat_volume_f32(const float *left, const float *right, float *vol, int B, int C, int H, int W,
                                        int D, void *stream)
{
    return shift_dispatch<true>(left, right, vol, B, C, H, W, D, stream);
}

// A1 + A4 in one pass: the correlation volume is stored AND folded into the soft-argmin while it is produced.
extern "C" int stereo_corr_volume_soft_argmin_f32(const float *left, const float *right, float *vol, float *disp,
                                                  int B, int C, int H, int W, int D, int match_similarity,
                                                  void *stream)
{
    if (D == 0 && B > 0 && H > 0 && W > 0) return STEREO_ERR_BAD_SHAPE;  // softmax over an empty axis
    if (B < 0 || C < 0 || H < 0 || W < 0 || D < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    if (C == 0) return STEREO_ERR_BAD_SHAPE;
    STEREO_REQUIRE_PTR(left);
    STEREO_REQUIRE_PTR(right);
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    const float sgn = match_similarity ? 1.0f : -1.0f;
    hipStream_t st = as_stream(stream);
    if (corr_mfma_enabled()) {
        const int rc = stereo_corr_volume_mfma(left, right, vol, disp, B, C, H, W, D, sgn, st);
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    if ((W % 4 == 0) && aligned16(left) && aligned16(right) && aligned16(vol) && fits_grid_yz(H)) {
        int rc = STEREO_ERR_UNSUPPORTED;
        if (D <= 64) rc = launch_pipe<8, 8, EPI_BOTH>(left, right, vol, B, C, H, W, D, 1, sgn, st, disp);
        else if (D <= 192) rc = launch_pipe<16, 12, EPI_BOTH>(left, right, vol, B, C, H, W, D, 1, sgn, st, disp);
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    const int rc = gwc_dispatch<EPI_STORE>(left, right, vol, B, C, H, W, D, 1, 1.0f, stream);
    if (rc != STEREO_OK) return rc;
    return gwc_dispatch<EPI_ARGMIN>(left, right, disp, B, C, H, W, D, 1, sgn, stream);
}

// A3 + group-mean + A4 in one call.  The mean over groups of the group-wise volume IS the correlation volume of
// all C channels (equal group sizes; SURVEY 8(c) identity I1), so the disparity and the mean volume are produced from
// the LDS-staged feature tiles (1 GB of input) instead of re-reading the 25.5 GB group-wise volume.
extern "C" int stereo_gwc_volume_soft_argmin_f32(const float *left, const float *right, float *vol, float *mean,
                                                 float *disp, int B, int C, int H, int W, int D, int G,
                                                 int match_similarity, void *stream)
{
    if (D == 0 && B > 0 && H > 0 && W > 0) return STEREO_ERR_BAD_SHAPE;
    if (B > 0 && C > 0 && H > 0 && W > 0 && G > 0 && left && right && vol && mean && disp && corr_mfma_enabled() &&
        stereo_opt(OPT_GWC_FUSED) != 0) {  // one kernel for the whole volume stage where its envelope allows
        const int rcf = stereo_gwc_mean_argmin_mfma(left, right, vol, mean, disp, B, C, H, W, D, G,
                                                    match_similarity ? 1.0f : -1.0f, as_stream(stream));
        if (rcf != STEREO_ERR_UNSUPPORTED) return rcf;
    }
    const int rc = gwc_dispatch<EPI_STORE>(left, right, vol, B, C, H, W, D, G, 1.0f, stream);
    if (rc != STEREO_OK) return rc;
    if (B == 0 || C == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(disp);
    if (mean == nullptr)
        return gwc_dispatch<EPI_ARGMIN>(left, right, disp, B, C, H, W, D, 1, match_similarity ? 1.0f : -1.0f, stream);
    return stereo_corr_volume_soft_argmin_f32(left, right, mean, disp, B, C, H, W, D, match_similarity, stream);
}

extern "C" int stereo_corr_soft_argmin_f32(const float *left, const float *right, float *disp, int B, int C, int H,
                                           int W, int D, int match_similarity, void *stream)
{
    if (D == 0 && B > 0 && H > 0 && W > 0) return STEREO_ERR_BAD_SHAPE;  // softmax over an empty axis
    if (B > 0 && C > 0 && H > 0 && W > 0 && left && right && disp && corr_mfma_enabled()) {
        const int rc = stereo_corr_soft_argmin_mfma(left, right, disp, B, C, H, W, D, match_similarity ? 1.0f : -1.0f,
                                                    as_stream(stream));
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    return gwc_dispatch<EPI_ARGMIN>(left, right, disp, B, C, H, W, D, 1, match_similarity ? 1.0f : -1.0f, stream);
}

// ---------------------------------------------------------------- fp16 feature operands
// left / right [B,C,H,W] fp16 (what an fp16 extractor hands over), every output fp32: the matrix-core kernels widen the
// features on the way into LDS, so the arithmetic is the fp32 path's on the fp16-rounded values and the read side of the
// launch is half the bytes.  These entry points have NO slow path: outside the matrix-core envelope (C <= 32, W % 4 == 0,
// D <= 192 / 256, 8-B aligned bases, channels per group in {2,4,8,16}) they return STEREO_ERR_UNSUPPORTED and the host
// layer widens the features itself before calling the fp32 entry point.
namespace {
int f16_args(const void *left, const void *right, int B, int C, int H, int W, int D, bool softmax)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || D < 0) return STEREO_ERR_BAD_SHAPE;
    if (softmax && D == 0 && B > 0 && H > 0 && W > 0) return STEREO_ERR_BAD_SHAPE;  // softmax over an empty axis
    if (B == 0 || H == 0 || W == 0 || D == 0) return 1;                           // nothing to do
    if (C == 0) return STEREO_ERR_BAD_SHAPE;
    if (left == nullptr || right == nullptr) return STEREO_ERR_NULL_POINTER;
    return STEREO_OK;
}
}  // namespace

extern "C" int stereo_corr_volume_f16(const void *left, const void *right, float *vol, int B, int C, int H, int W, int D,
                                      void *stream)
{
    const int chk = f16_args(left, right, B, C, H, W, D, false);
    if (chk != STEREO_OK) return chk > 0 ? STEREO_OK : chk;
    STEREO_REQUIRE_PTR(vol);
    return stereo_corr_volume_mfma<f16>(static_cast<const f16 *>(left), static_cast<const f16 *>(right), vol, nullptr, B, C, H,
                                        W, D, 1.0f, as_stream(stream));
}

extern "C" int stereo_gwc_volume_f16(const void *left, const void *right, float *vol, int B, int C, int H, int W, int D,
                                     int G, void *stream)
{
    if (G <= 0 || (C >= 0 && C % G != 0)) return STEREO_ERR_BAD_SHAPE;
    const int chk = f16_args(left, right, B, C, H, W, D, false);
    if (chk != STEREO_OK) return chk > 0 ? STEREO_OK : chk;
    STEREO_REQUIRE_PTR(vol);
    const f16 *L = static_cast<const f16 *>(left), *R = static_cast<const f16 *>(right);
    if (G == 1) return stereo_corr_volume_mfma<f16>(L, R, vol, nullptr, B, C, H, W, D, 1.0f, as_stream(stream));
    return stereo_gwc_mean_argmin_mfma<f16>(L, R, vol, nullptr, nullptr, B, C, H, W, D, G, 1.0f, as_stream(stream));
}

extern "C" int stereo_corr_volume_soft_argmin_f16(const void *left, const void *right, float *vol, float *disp, int B,
                                                  int C, int H, int W, int D, int match_similarity, void *stream)
{
    const int chk = f16_args(left, right, B, C, H, W, D, true);
    if (chk != STEREO_OK) return chk > 0 ? STEREO_OK : chk;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(disp);
    return stereo_corr_volume_mfma<f16>(static_cast<const f16 *>(left), static_cast<const f16 *>(right), vol, disp, B, C, H, W,
                                        D, match_similarity ? 1.0f : -1.0f, as_stream(stream));
}

extern "C" int stereo_gwc_volume_soft_argmin_f16(const void *left, const void *right, float *vol, float *mean, float *disp,
                                                 int B, int C, int H, int W, int D, int G, int match_similarity,
                                                 void *stream)
{
    if (G <= 0 || (C >= 0 && C % G != 0)) return STEREO_ERR_BAD_SHAPE;
    const int chk = f16_args(left, right, B, C, H, W, D, true);
    if (chk != STEREO_OK) return chk > 0 ? STEREO_OK : chk;
    STEREO_REQUIRE_PTR(vol);
    STEREO_REQUIRE_PTR(mean);
    STEREO_REQUIRE_PTR(disp);
    return stereo_gwc_mean_argmin_mfma<f16>(static_cast<const f16 *>(left), static_cast<const f16 *>(right), vol, mean, disp,
                                            B, C, H, W, D, G, match_similarity ? 1.0f : -1.0f, as_stream(stream));
}

extern "C" int stereo_corr_soft_argmin_f16(const void *left, const void *right, float *disp, int B, int C, int H, int W,
                                           int D, int match_similarity, void *stream)
{
    const int chk = f16_args(left, right, B, C, H, W, D, true);
    if (chk != STEREO_OK) return chk > 0 ? STEREO_OK : chk;
    STEREO_REQUIRE_PTR(disp);
    return stereo_corr_soft_argmin_mfma<f16>(static_cast<const f16 *>(left), static_cast<const f16 *>(right), disp, B, C, H,
                                             W, D, match_similarity ? 1.0f : -1.0f, as_stream(stream));
}
