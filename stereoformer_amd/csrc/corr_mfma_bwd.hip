// Backward of the correlation volume (A1, and the mean path of the fused volume stage) on the f32 matrix cores:
//     dL[c, w] = 1/C * sum_d g[d, w]     * R[c, w - d]
//     dR[c, v] = 1/C * sum_d g[d, v + d] * L[c, v + d]
// The forward kernel (corr_mfma.hip) turns MFMA tiles (key x query) into volume rows (disparity x query) by WRITING the
// accumulators along the diagonals of an LDS sheet; the backward runs the same geometry in reverse: 32 rows of the
// gradient volume are loaded as 1-KiB contiguous pieces into the sheet (ring of three 32-row blocks: in use / next /
// being filled), and the MFMA's B operand G[key][query] = g[query - key][query] is READ along the diagonals -- one
// ds_read_b32 at an immediate offset per MFMA.  A = the other feature tensor from a zero-padded LDS tile (lanes =
// channels), the accumulator is the 32-channel x 32-query gradient tile of a wave, summed over the band's key tiles.
//
// dR is the same computation in mirrored coordinates u = W-1-v with the gradient sheared by d:
//     dR~[c, u] = 1/C * sum_d g~[d, u] * L~[c, u - d],   g~[d, u] = g[d, (W-1+d) - u],   L~[c, x] = L[c, W-1-x]
// so FLIP = true only changes how the sheet and the feature tile are FILLED (row d of g is read backwards from column
// W-1+d: contiguous, 4-byte aligned) and where the result is stored.  C <= 32, D <= 192, W % 4 == 0.

#include "common.h"

namespace {

constexpr int CB_NWV = 8;
constexpr int CB_TQW = 32 * CB_NWV;
constexpr int CB_RSTR = CB_TQW + 4;
constexpr int CB_RING = 96;
constexpr int CB_PD = 3;  // gradient blocks in flight (registers)

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned quad (sheared rows)

// four consecutive mirrored columns x .. x+3 of a row whose mirrored column x is src[off - x]  (off = W-1 (+ d))
__device__ __forceinline__ f32x4 load_rev4(const float *__restrict__ row, int off, int x, int W)
{
    const int hi = off - x;  // source column of mirrored column x; the quad covers hi-3 .. hi
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (hi >= 3 && hi < W) {
        const f32x4u t = *reinterpret_cast<const f32x4u *>(row + (hi - 3));
        v = (f32x4){t[3], t[2], t[1], t[0]};
    } else if (hi - 3 >= W || hi < 0) {
        // entirely outside the row (the sheared gradient's v + d >= W corner, or past the left edge): zeros, no loads
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sc = hi - i;
            if (sc >= 0 && sc < W) v[i] = row[sc];
        }
    }
    return v;
}

template <int NT, bool FLIP>
__global__ __launch_bounds__(CB_NWV * 64) void corr_volume_bwd_mfma_kernel(const float *__restrict__ G,
                                                                           const float *__restrict__ X,
                                                                           float *__restrict__ dOut, int C, int H, int W,
                                                                           int D, float inv_c, int tiles_per_wg, int ntiles)
{
    constexpr int Dk = 32 * (NT - 1);
    constexpr int NK = CB_TQW + Dk;
    constexpr int KS = NK + 4;
    constexpr int RMAX = 32 / CB_NWV;
    constexpr int NPASS = (NK / 4 + 63) / 64;
    extern __shared__ f32x4 smem4[];
    f32x4 *Ks4 = smem4;
    float *Ks = reinterpret_cast<float *>(smem4);
    float *Sh = Ks + 32 * KS;  // [CB_RING][CB_RSTR]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned HWu = (unsigned)H * (unsigned)W;
    const int nx = (W + CB_TQW - 1) / CB_TQW;
    const int dl = l31 - 4 * half;
    const float *vp = Ks + l31 * KS + 32 * wave + 4 * half;                   // A: X[channel = l31][key]
    const int lane_base = (dl - 27) * CB_RSTR + 32 * wave + l31;             // B: sheet diagonal of this lane

    for (int tile = blockIdx.x * tiles_per_wg; tile < ntiles && tile < (blockIdx.x + 1) * tiles_per_wg; ++tile) {
        const int x = tile % nx, bh = tile / nx;
        const int hh = bh % H, bb = bh / H;
        const int w0 = x * CB_TQW, kbase = w0 - Dk;  // (mirrored coordinates when FLIP)
        const float *Xb = X + ((size_t)bb * C * HWu + (size_t)hh * W);
        const float *Gb = G + ((size_t)bb * D * HWu + (size_t)hh * W);
        __syncthreads();  // the previous tile is done with both LDS regions
        // ---- feature tile: channels 0..31 (zero rows above C), columns kbase .. kbase + NK - 1, zero outside the image
#pragma unroll
        for (int rr = 0; rr < RMAX; ++rr) {
            const int row = wave + rr * CB_NWV;
#pragma unroll
            for (int pp = 0; pp < NPASS; ++pp) {
                const int xq = lane + 64 * pp;
                if (xq < NK / 4) {
                    const int w = kbase + 4 * xq;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (row < C) {
                        if (FLIP) v = load_rev4(Xb + (size_t)row * HWu, W - 1, w, W);
                        else if (w >= 0 && w + 3 < W) v = ld4(Xb + (size_t)row * HWu + w);
                    }
                    Ks4[(row * KS) / 4 + xq] = v;
                }
            }
        }
        // ---- sheet block filler: rows d = 32 blk .. +31 of the gradient volume, this tile's 256 columns
        auto load_block = [&](int blk, f32x4 (&q)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int qi = tid + k * (CB_NWV * 64);
                const int row = qi >> 6, c4 = qi & 63;
                const int d = 32 * blk + row, w = w0 + 4 * c4;
                q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (d < D && w + 3 < W) {
                    if (FLIP) q[k] = load_rev4(Gb + (size_t)d * HWu, W - 1 + d, w, W);
                    else q[k] = ld4(Gb + (size_t)d * HWu + w);
                }
            }
        };
        auto store_block = [&](int slot, const f32x4 (&q)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int qi = tid + k * (CB_NWV * 64);
                const int row = qi >> 6, c4 = qi & 63;
                *reinterpret_cast<f32x4 *>(Sh + (32 * slot + row) * CB_RSTR + 4 * c4) = q[k];
            }
        };
        // gradient blocks travel PD steps ahead of their use in registers: one step's arithmetic (~0.5 us) is far shorter
        // than a loaded memory system's latency, and the LDS footprint allows one workgroup per CU
        f32x4 qn[CB_PD][4];
        {
            load_block(0, qn[0]);
#pragma unroll
            for (int j = 1; j < CB_PD; ++j)
                if (j < NT) load_block(j, qn[j]);
            store_block(0, qn[0]);
            const f32x4 z[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            store_block(2, z);  // rows -31 .. -1 of step 0 (d < 0): zeros
        }
        __syncthreads();

        const bool wave_on = w0 + 32 * wave < W;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            if (s + CB_PD < NT) load_block(s + CB_PD, qn[s % CB_PD]);  // its previous content (block s) is in the sheet
            if (wave_on) {
                const int sb = s % 3;
                const float *rp = Sh + lane_base + 32 * sb * CB_RSTR;
                const float *vpt = vp + 32 * (NT - 1 - s);
                if (sb != 0) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int at = (t & 3) + 8 * (t >> 2);
                        o = __builtin_amdgcn_mfma_f32_32x32x2f32(vpt[at], rp[(27 - at) * CB_RSTR], o, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int at = (t & 3) + 8 * (t >> 2);
                        const float gv = rp[(27 - at) * CB_RSTR + (dl < at ? CB_RING * CB_RSTR : 0)];
                        o = __builtin_amdgcn_mfma_f32_32x32x2f32(vpt[at], gv, o, 0, 0, 0);
                    }
                }
            }
            if (s + 1 < NT) store_block((s + 1) % 3, qn[(s + 1) % CB_PD]);  // slot of block s-2: its last readers passed the previous barrier
            __syncthreads();
        }
        if (wave_on) {
            const int q = w0 + 32 * wave + l31;  // (mirrored) output column
            const int qs = FLIP ? W - 1 - q : q;
            if (q < W) {
                float *ob = dOut + ((size_t)bb * C * HWu + (size_t)hh * W);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (c < C) ob[(unsigned)c * HWu + (unsigned)qs] = o[r] * inv_c;
                }
            }
        }
    }
}

template <int NT, bool FLIP>
int launch_cb(const float *G, const float *X, float *dOut, int B, int C, int H, int W, int D, hipStream_t st)
{
    constexpr int Dk = 32 * (NT - 1);
    const size_t lds = ((size_t)32 * (CB_TQW + Dk + 4) + (size_t)CB_RING * CB_RSTR) * sizeof(float);
    auto kern = corr_volume_bwd_mfma_kernel<NT, FLIP>;
    if (lds > 160 * 1024) return STEREO_ERR_UNSUPPORTED;
    static std::atomic<unsigned long long> lds_done{0};
    if (!lds_opt_in(kern, lds, lds_done)) return STEREO_ERR_UNSUPPORTED;
    const long long ntl = (long long)((W + CB_TQW - 1) / CB_TQW) * H * B;
    if (ntl >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    const int tpw = (int)((ntl + 1023) / 1024);  // ~4 workgroups per CU over the launch (one resident at a time)
    const int nwg = (int)((ntl + tpw - 1) / tpw);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(CB_NWV * 64), lds, st, G, X, dOut, C, H, W, D, 1.0f / (float)C, tpw, (int)ntl);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

int stereo_corr_volume_bwd_fused(const float *left, const float *right, const float *grad_vol, float *grad_left,
                                 float *grad_right, int B, int C, int H, int W, int D, hipStream_t st);  // corr_bwd_fused.hip

// grad_vol [B,D,H,W] -> grad_left / grad_right [B,C,H,W] (either may be null).  STEREO_ERR_UNSUPPORTED outside the
// envelope (the caller then runs the VALU backward).
int stereo_corr_volume_bwd_mfma(const float *left, const float *right, const float *grad_vol, float *grad_left,
                                float *grad_right, int B, int C, int H, int W, int D, hipStream_t st)
{
    if (grad_left && grad_right && stereo_opt(OPT_CORR_BWD_FUSED)) {
        const int rc = stereo_corr_volume_bwd_fused(left, right, grad_vol, grad_left, grad_right, B, C, H, W, D, st);
        if (rc != STEREO_ERR_UNSUPPORTED) return rc;
    }
    if (C > 32 || D > 192 || D < 1 || W % 4 != 0 || !aligned16(left) || !aligned16(right) || !aligned16(grad_vol))
        return STEREO_ERR_UNSUPPORTED;
    if ((long long)32 * H * W >= (1LL << 31) || (long long)D * H * W >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    const int nt = (D + 31) / 32 + 1;
#define CB_CASE(NTV)                                                                                   \
    do {                                                                                               \
        if (grad_left) {                                                                               \
            const int rc = launch_cb<NTV, false>(grad_vol, right, grad_left, B, C, H, W, D, st);       \
            if (rc != STEREO_OK) return rc;                                                            \
        }                                                                                              \
        if (grad_right) {                                                                              \
            const int rc = launch_cb<NTV, true>(grad_vol, left, grad_right, B, C, H, W, D, st);        \
            if (rc != STEREO_OK) return rc;                                                            \
        }                                                                                              \
        return STEREO_OK;                                                                              \
    } while (0)
    if (nt <= 3) CB_CASE(3);
    if (nt <= 5) CB_CASE(5);
    CB_CASE(7);
#undef CB_CASE
}
