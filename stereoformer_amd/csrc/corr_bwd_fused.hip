// Backward of the correlation volume (A1; the mean path of the fused volume stage; Q K^T / P V of A7 under grad) with BOTH
// feature gradients from ONE read of the gradient volume:
//     dL[c, w] = 1/C * sum_d g[d, w]     * R[c, w - d]
//     dR[c, v] = 1/C * sum_d g[d, v + d] * L[c, v + d]
// corr_mfma_bwd.hip runs two launches (dL, then dR in mirrored coordinates), each streaming the 4-GB volume through the f32
// matrix cores; the volume is 6 x the size of everything else the backward touches, so this kernel reads it once.
//
// Geometry.  A workgroup walks whole image rows left to right in RECTANGLES of 128 columns x all D disparities; a rectangle is
// streamed as 32-row BLOCKS (step s = rows 32 s .. 32 s + 31) through a three-block LDS sheet, exactly the ring of the
// two-launch kernel.  Waves 0-3 own the four 32-column dL tiles of the rectangle: at step s wave i multiplies the key tile
// k = w0 + 32 i - 32 s .. + 31 of R (keys x channels) by the sheet read along its diagonals, B[k][w] = g[w - k][w].  Waves 4-7
// compute dR from the same sheet: the KEYS of dR are the rectangle's own columns, B[w][v] = g[w - v][w] (sheet columns read
// downwards), and the outputs v = w - d lie up to D columns to the LEFT of the rectangle.  Wave i takes key tile
// (i + s) mod 4 at step s, which makes its output tile v0 = 128 (j - r) + 32 i with r = (i + s) div 4 in {0, 1, 2}: tile i of
// this rectangle, of the previous one or of the one before.  Because the workgroup walks the row, those three output tiles
// are simply three accumulators the wave carries from rectangle to rectangle: after rectangle j the tile of rectangle j - 2
// is complete and is stored -- no atomics, no partial buffers, a fixed summation order.  R's keys reach 192 columns to the left
// of the rectangle: they are still in LDS from the rectangles before (ring of 448 keys); L is needed for the rectangle only.
//
// Arithmetic.  Split-fp16 on v_mfma_f32_32x32x16_f16 (6 per wave and step): x = hi + lo, products hi hi + hi lo + lo hi in ONE
// fp32 accumulator.  Features are split once per rectangle when they enter LDS, as 16 x = hi + lo with an unscaled low piece
// (a normal fp16 number for |x| >= 2^-7, an absolute error of <= 2^-29 in x below that; |feature| < 4 094 -- until the end
// of round 3 the low piece carried 2^11 and its products a second accumulator: 1.49 -> 1.38 ms).  GRADIENTS have no natural
// scale (a mean-reduced loss hands over 1e-9), so each wave scales its 32 x 32 gradient tile per OUTPUT COLUMN by the power of
// two that puts the column's largest element into [2^14, 2^15) before splitting -- a per-column scale commutes with the
// product -- runs the six MFMAs from a zero accumulator and adds acc / (16 sigma) to its fp32 totals.  The
// gradient's low piece stays UNSCALED: against a column maximum >= 2^14 its absolute error (fp16's subnormal spacing) is 2^-39
// of the largest term of the dot product.  Result: as close to fp64 as the f32 matrix cores for gradients from 3e-9 to 2e7
// (tests/test_gpu_backward.py::test_corr_volume_backward_single_read_kernel).
// Measured alternatives (cfg-2 size, 8 pairs; against this kernel's 1.49 ms before the one-accumulator form): all sixteen v_mfma_f32_32x32x2_f32 per step on the raw values
// (no operand preparation at all) 1.81 ms, f32 MFMAs for the dR waves only 1.66 ms -- the f32 matrix pipe at 64 cycles per
// instruction is the slower resource.  This form runs the vector pipe at ~40 % and the matrix pipe at 9 %: its step time
// (1.5 us for 16 KB of volume per CU) is the dependent chain LDS read -> column maximum -> split -> MFMA -> totals of two waves
// per SIMD behind one barrier per step (in-kernel cycle stamps: a dR wave needs ~3 000 cycles per step, a dL wave ~2 150
// plus its mover burst; the dR waves' sheet reads -- columns, 2-way bank-conflicted -- come back after ~440 cycles against
// ~90 for the dL waves' diagonals, and a mover's store burst still waits ~750 cycles on average for loads issued two steps
// earlier: a third block in flight would need a second register set per mover wave, and the dR waves set the step anyway).
// Ablations (test option corr_bwd_fused = 2 / 3): without arithmetic 1.06 ms (5.0 TB/s), without the volume loads 1.16 ms.
//
// Data movement.  One workgroup per CU (143 KB of LDS).  The dL waves, whose arithmetic is the shorter one, move the data:
// their two wave pairs take the blocks in turn, block t + 2 is issued at the top of step t into the pair's ONE statically named
// register set and goes to the sheet at the end of step t + 1 (two steps = ~3 us in flight; the flat stream of blocks
// continues across rectangles and rows), so every mover wave does one burst per step, loads or stores, and reaches the
// barrier about when the dR waves do.  Tried and measured slower: all eight waves taking turns with four blocks in flight
// (1.53 ms: the pair whose turn it was arrived 1 000-1 300 cycles late, every step); every wave moving four rows of every
// block through four rotating register sets pinned outside the compiler's allocation, loads and waits in inline assembly
// (1.78 ms: the per-step scalar bookkeeping then sits on all eight waves' critical path).  The next rectangle's features
// travel with the pairs' own blocks (loaded right before a block's loads, stored with it).
//
// C <= 32, D <= 192 (the R ring holds 192 + 128 keys in use + 128 being filled), W % 4 == 0, both gradients requested;
// anything else: STEREO_ERR_UNSUPPORTED (the caller runs the two-launch kernel).

#include "common.h"

namespace {

constexpr int FB_TW = 128;            // rectangle columns
constexpr int FB_NQ = FB_TW / 32;     // 32-column tiles per rectangle = waves per role
constexpr int FB_NTH = 2 * FB_NQ * 64;
constexpr int FB_RS = FB_TW + 2;      // sheet row stride (dwords): even (8-byte stores), (RS + 1) odd (dL diagonal reads conflict-free)
constexpr int FB_RING = 96;           // sheet rows: three 32-row blocks
constexpr int FB_RK = 448;            // R ring (keys): 320 in use + 128 being filled; a multiple of 32
constexpr int FB_RB = 912;            // bytes per channel row of an R plane: 16 x odd (conflict-free ds_read_b128), >= 2 * FB_RK
constexpr int FB_LB = 272;            // bytes per channel row of an L plane: 16 x 17
constexpr int FB_SHEET_BYTES = FB_RING * FB_RS * 4;
constexpr int FB_RPLANE = 32 * FB_RB, FB_LPLANE = 32 * FB_LB;
constexpr int FB_LDS = FB_SHEET_BYTES + 2 * FB_RPLANE + 4 * FB_LPLANE;

struct FbCursor {
    int row, j, s;  // image row (b * H + h), rectangle, block
};

// gradient tile of one wave and step: 16 values per lane (two MFMAs x 8 k-slots) -> scaled split-fp16 operands
template <bool DR>
__device__ __forceinline__ float fb_gradient_tile(const float *__restrict__ Sh, int cs, int w0, int cb, f16x8 (&bh)[2],
                                                  f16x8 (&bl)[2])
{
    constexpr int SG = DR ? 1 : -1;
    float raw[16];
    if (cs != 0) {
        const float *p = Sh + (32 * cs + w0) * FB_RS + cb;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int kk = 16 * (q >> 3) + (q & 7);
            raw[q] = p[SG * kk * FB_RS + (DR ? kk : 0)];
        }
    } else {  // block s-1 sits in the last slot of the ring
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int kk = 16 * (q >> 3) + (q & 7);
            const int ro = w0 + SG * kk;
            raw[q] = Sh[(ro + (ro < 0 ? FB_RING : 0)) * FB_RS + cb + (DR ? kk : 0)];
        }
    }
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) m = fmaxf(m, fabsf(raw[q]));
    {   // the other 16 keys of this output column live in lane ^ 32
        const unsigned mu = __float_as_uint(m);
        const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
        m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    int ex = (int)(__float_as_uint(m) >> 23);
    ex = ex < 16 ? 16 : (ex > 254 ? 254 : ex);
    const float sig = __uint_as_float((unsigned)(268 - ex) << 23);  // column max -> [2^14, 2^15)
    // hi = fp16(x sigma), lo = fp16(x sigma - hi) UNSCALED: lo's absolute error is <= 2^-25 (fp16's subnormal spacing) against
    // a column maximum of >= 2^14, i.e. 2^-39 of the largest term of the dot product
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    u32x4v hv[2], lv[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x2 t = (f32x2){raw[2 * q], raw[2 * q + 1]} * sig;
        const f16x2 h = __builtin_convertvector(t, f16x2);
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        // x sigma - hi in ONE mixed-precision FMA per element (fp32 arithmetic on an fp16 source, fp16 result into one half
        // of the destination) -- the compiler's own choice is two conversions back to fp32 and a packed FMA
        unsigned lu;
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lu) : "v"(raw[2 * q]), "v"(sig), "v"(hu));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lu) : "v"(raw[2 * q + 1]), "v"(sig), "v"(hu));
        hv[q >> 2][q & 3] = hu;
        lv[q >> 2][q & 3] = lu;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) bh[j] = __builtin_bit_cast(f16x8, hv[j]), bl[j] = __builtin_bit_cast(f16x8, lv[j]);
    return __uint_as_float((unsigned)(ex - 14) << 23);  // 1 / sigma
}

constexpr float FB_PRE = 16.0f;   // features are split as 16 x (|feature| < 4 094)
// eight fp32 values -> hi / unscaled-lo fp16 pieces of FB_PRE x: two mixed-precision FMAs per value
__device__ __forceinline__ void fb_split_oct(const float *x, f16x8 *p)
{
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    float pre = FB_PRE;
    asm volatile("" : "+v"(pre));
    u32x4v h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hu, lu;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hu) : "v"(x[2 * j]), "v"(pre));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hu) : "v"(x[2 * j + 1]), "v"(pre));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lu) : "v"(x[2 * j]), "v"(pre), "v"(hu));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lu) : "v"(x[2 * j + 1]), "v"(pre), "v"(hu));
        h[j] = hu, l[j] = lu;
    }
    p[0] = __builtin_bit_cast(f16x8, h), p[1] = __builtin_bit_cast(f16x8, l);
}

__device__ __forceinline__ void fb_products(const unsigned char *__restrict__ ahi, const unsigned char *__restrict__ alo,
                                            const f16x8 (&bh)[2], const f16x8 (&bl)[2], float inv, f32x16 &tot)
{
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const f16x8 ah0 = *reinterpret_cast<const f16x8 *>(ahi), ah1 = *reinterpret_cast<const f16x8 *>(ahi + 32);
    const f16x8 al0 = *reinterpret_cast<const f16x8 *>(alo), al1 = *reinterpret_cast<const f16x8 *>(alo + 32);
    // ONE accumulator: the feature pieces are 16 x = hi + lo with an UNSCALED low piece (store_features), the gradient's low piece
    // is unscaled as well, so hi hi + hi lo + lo hi share it (until the end of round 3 the features' low piece carried 2^11 and
    // the lo hi products an accumulator of their own: 16 more registers and 16 more FMAs per call)
    f32x16 a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[0], z, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[1], a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[0], a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[1], a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[0], a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[1], a0, 0, 0, 0);
    const float invu = inv * (1.0f / FB_PRE);
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[r] = fmaf(a0[r], invu, tot[r]);
}

__device__ __forceinline__ void fb_store_tile(float *__restrict__ out, int C, unsigned HWu, int col, int W, int half,
                                              const f32x16 &tot, float inv_c)
{
    if (col < W) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (c < C) out[(unsigned)c * HWu + (unsigned)col] = tot[r] * inv_c;
        }
    }
}

__global__ __launch_bounds__(FB_NTH) void corr_volume_bwd_fused_kernel(const float *__restrict__ G, const float *__restrict__ Lf,
                                                                       const float *__restrict__ Rf, float *__restrict__ dL,
                                                                       float *__restrict__ dR, int C, int H, int W, int D,
                                                                       float inv_c, int nrows, int rows_per_wg, int dbg)
{
    // dbg = the test option corr_bwd_fused: 1 product; 2 / 3 ablations for tools/a1bwd_ab.py (2: no arithmetic, 3: no
    // volume loads -- results are then meaningless, only the time is read)
    extern __shared__ f32x4 smem4[];
    float *Sh = reinterpret_cast<float *>(smem4);
    unsigned char *Rhi = reinterpret_cast<unsigned char *>(smem4) + FB_SHEET_BYTES;
    unsigned char *Rlo = Rhi + FB_RPLANE;
    unsigned char *Lhi = Rlo + FB_RPLANE;  // [2 buffers][hi, lo]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: roles, phases and cursors stay on the SALU
    const int l31 = lane & 31, half = lane >> 5;
    const bool role_dr = wave >= FB_NQ;
    const int wi = wave & (FB_NQ - 1);
    const unsigned HWu = (unsigned)H * (unsigned)W;
    const int nrect = (W + FB_TW - 1) / FB_TW;
    const int NT = max((D + 31) / 32 + 1, 3);  // blocks per rectangle; blocks at d >= D are zeros
    const int row_begin = blockIdx.x * rows_per_wg;
    const int row_end = min(nrows, row_begin + rows_per_wg);
    if (row_begin >= row_end) return;
    const int nitems = (row_end - row_begin) * nrect * NT;

    // ---- gradient blocks: item (row, j, s) = rows 32 s .. + 31 of the volume slice of `row`, columns 128 j .. + 127.
    //      The dL waves move the data (their arithmetic is ~800 cycles per step shorter than the dR waves': one accumulator,
    //      conflict-free sheet reads): item t belongs to the wave PAIR t mod 2 of them (even wave: rows 0-15, odd wave: rows
    //      16-31; a lane two columns, so one 512-byte row segment per wave instruction).  A pair issues item t + 2 at the top of
    //      step t and stores it at the end of step t + 1: every mover wave has ONE statically named register set, in flight for
    //      two steps (~3 us), and does one burst per step -- loads or stores.  (Measured with in-kernel cycle stamps: with
    //      all eight waves taking turns the pair whose turn it was reached the step's barrier 1 000-1 300 cycles after the
    //      others every step; a rotating register set shared by all waves makes the compiler wait for everything in flight;
    //      so does overwriting a load's destination with zeros first -- hence clamped addresses and the zeroing at store time.)
    const bool mover = wave < FB_NQ;
    const int my_phase = (wave >> 1) & 1, my_rows = 16 * (wave & 1);
    f32x2 q[16];
    int q_d0 = D;        // first volume row of the registers' content (>= D: a zero block, nothing was loaded)
    bool q_colok = false;
    bool q_full = false;  // every row and column of the registers' content is inside the volume
    auto load_item = [&](const FbCursor &c) {
        const int b = c.row / H, h = c.row - b * H;
        const int col = FB_TW * c.j + 2 * lane;
        q_d0 = dbg == 3 ? D : 32 * c.s + my_rows;
        q_colok = col < W;
        q_full = q_d0 + 16 <= D && FB_TW * (c.j + 1) <= W;
        if (q_d0 < D) {
            const float *gb = G + ((size_t)b * D * HWu + (size_t)h * W) + (size_t)q_d0 * HWu;  // wave-uniform
            const int co = min(col, W - 2);
            if (q_d0 + 16 <= D) {
#pragma unroll
                for (int k = 0; k < 16; ++k) q[k] = *reinterpret_cast<const f32x2 *>(gb + (size_t)k * HWu + co);
            } else {
                const int kmax = D - 1 - q_d0;
#pragma unroll
                for (int k = 0; k < 16; ++k) q[k] = *reinterpret_cast<const f32x2 *>(gb + (size_t)min(k, kmax) * HWu + co);
            }
        }
    };
    auto store_item = [&](int slot) {
        float *sp = Sh + (32 * slot + my_rows) * FB_RS + 2 * lane;
        if (q_full) {
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x2 *>(sp + k * FB_RS) = q[k];
        } else if (q_d0 < D) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool ok = q_colok && q_d0 + k < D;
                *reinterpret_cast<f32x2 *>(sp + k * FB_RS) = ok ? q[k] : (f32x2){0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x2 *>(sp + k * FB_RS) = (f32x2){0.f, 0.f};
        }
    };
    auto advance = [&](FbCursor &c) {
        if (++c.s == NT) {
            c.s = 0;
            if (++c.j == nrect) c.j = 0, ++c.row;
        }
    };
    // ---- features of a rectangle, 8 channels per mover wave: lane = (channel 8 w + lane / 16 (+ 4), eight columns).  A pair
    //      fetches its half of the NEXT rectangle right before the first gradient item it issues inside the current one and
    //      stores it together with that item one step later: one wait covers both.
    const int fc = 8 * wave + (lane >> 4), fx = (lane & 15) * 8;
    f32x4 fl[2][2], fr[2][2];   // [channel fc, fc + 4][columns fx .. fx + 3, fx + 4 .. fx + 7]
    int f_col = 0;
    bool f_pending = false;
    auto load_features = [&](int row, int j) {
        const int b = row / H, h = row - b * H;
        f_col = FB_TW * j + fx;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t off = (size_t)b * C * HWu + (size_t)min(fc + 4 * u, C - 1) * HWu + (size_t)h * W;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int col = min(f_col + 4 * k, W - 4);
                fl[u][k] = ld4(Lf + off + col);
                fr[u][k] = ld4(Rf + off + col);
            }
        }
    };
    auto store_features = [&](int lb, int rb) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = fc + 4 * u;
            float x[8];
            f16x8 p[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = (ch < C && f_col + 4 * (k >> 2) < W) ? fl[u][k >> 2][k & 3] : 0.0f;
            fb_split_oct(x, p);
            unsigned char *lp = Lhi + lb * 2 * FB_LPLANE + ch * FB_LB + 2 * fx;
            *reinterpret_cast<f16x8 *>(lp) = p[0];
            *reinterpret_cast<f16x8 *>(lp + FB_LPLANE) = p[1];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = (ch < C && f_col + 4 * (k >> 2) < W) ? fr[u][k >> 2][k & 3] : 0.0f;
            fb_split_oct(x, p);
            int pos = rb + fx;
            pos = pos >= FB_RK ? pos - FB_RK : pos;
            *reinterpret_cast<f16x8 *>(Rhi + ch * FB_RB + 2 * pos) = p[0];
            *reinterpret_cast<f16x8 *>(Rlo + ch * FB_RB + 2 * pos) = p[1];
        }
    };

    // ---- prologue
    FbCursor lc = {row_begin, 0, 0};  // this wave pair's next item
    int li = my_phase;                // ... and its index in the stream
    if (my_phase == 1) advance(lc);
    if (mover) {
        load_features(row_begin, 0);
        if (li < nitems) load_item(lc);
        store_features(0, 0);
        if (my_phase == 0) {
            store_item(0);
            li += 2;
            advance(lc), advance(lc);
            if (li < nitems) load_item(lc);
        } else {
            // "block -1" of the first step (d < 0): zeros
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x2 *>(Sh + (64 + my_rows + k) * FB_RS + 2 * lane) = (f32x2){0.f, 0.f};
        }
    }
    __syncthreads();

    f32x16 t0, t1, t2;  // dL waves: t0; dR waves: output tiles of rectangles j, j-1, j-2
#pragma unroll
    for (int r = 0; r < 16; ++r) t0[r] = t1[r] = t2[r] = 0.0f;

    FbCursor cc = {row_begin, 0, 0};
    int cs = 0;        // ring slot of the current block; block s-1 is in slot cs-1 (mod 3)
    int rbase = 0;     // ring position of the current rectangle's first key
    int lbuf = 0;
    const int w0 = role_dr ? 8 * half - l31 : l31 - 8 * half;
#pragma unroll 1
    for (int t = 0; t < nitems; ++t) {
        // next rectangle of this workgroup
        int nrow = cc.row, nj = cc.j + 1;
        if (nj == nrect) nj = 0, ++nrow;
        const bool has_next = nrow < row_end;
        const bool last_step = cc.s == NT - 1;
        if (mover && (t & 1) == my_phase) {
            if (cc.s < 2 && has_next) load_features(nrow, nj), f_pending = true;
            // item t went to the sheet at the end of step t - 1: fetch item t + 2 into the same registers
            if (t > 0) {
                li += 2;
                advance(lc), advance(lc);
                if (li < nitems) load_item(lc);
            }
        }

        if (dbg == 2) {
        } else if (!role_dr) {
            const int kb = FB_TW * cc.j + 32 * wi - 32 * cc.s;  // first key of the step's key tile
            if (kb >= 0 && FB_TW * cc.j + 32 * wi < W) {
                f16x8 bh[2], bl[2];
                const float inv = fb_gradient_tile<false>(Sh, cs, w0, 32 * wi + l31, bh, bl);
                int pos = rbase + 32 * wi - 32 * cc.s;
                pos = pos < 0 ? pos + FB_RK : (pos >= FB_RK ? pos - FB_RK : pos);
                const int ao = l31 * FB_RB + 2 * (pos + 8 * half);
                fb_products(Rhi + ao, Rlo + ao, bh, bl, inv, t0);
            }
        } else {
            const int kt = (wi + cc.s) & (FB_NQ - 1), r = (wi + cc.s) / FB_NQ;
            if (cc.j - r >= 0 && FB_TW * cc.j + 32 * kt < W) {
                f16x8 bh[2], bl[2];
                const float inv = fb_gradient_tile<true>(Sh, cs, w0, 32 * kt + 8 * half, bh, bl);
                const unsigned char *ap = Lhi + lbuf * 2 * FB_LPLANE + l31 * FB_LB + 2 * (32 * kt + 8 * half);
                if (r == 0) fb_products(ap, ap + FB_LPLANE, bh, bl, inv, t0);
                else if (r == 1) fb_products(ap, ap + FB_LPLANE, bh, bl, inv, t1);
                else fb_products(ap, ap + FB_LPLANE, bh, bl, inv, t2);
            }
        }

        // -- item t + 1 into the sheet slot of item t - 2 (its last readers passed the previous barrier)
        const int ns = cs == 2 ? 0 : cs + 1;
        const bool my_store = mover && ((t + 1) & 1) == my_phase;
        if (my_store && t + 1 < nitems) store_item(ns);
        const int nb = rbase + FB_TW >= FB_RK ? rbase + FB_TW - FB_RK : rbase + FB_TW;
        if (f_pending && (my_store || last_step)) store_features(lbuf ^ 1, nb), f_pending = false;

        // -- results of the rectangle (after the LDS stores: the next wait of this wave is a step away)
        if (last_step) {
            const int b = cc.row / H, h = cc.row - b * H;
            const int oc = FB_TW * cc.j + 32 * wi + l31;
            if (!role_dr) {
                fb_store_tile(dL + ((size_t)b * C * HWu + (size_t)h * W), C, HWu, oc, W, half, t0, inv_c);
#pragma unroll
                for (int r = 0; r < 16; ++r) t0[r] = 0.0f;
            } else {
                float *ob = dR + ((size_t)b * C * HWu + (size_t)h * W);
                if (cc.j >= 2) fb_store_tile(ob, C, HWu, oc - 2 * FB_TW, W, half, t2, inv_c);
                if (cc.j == nrect - 1) {  // end of the row: nothing to the right contributes any more
                    if (cc.j >= 1) fb_store_tile(ob, C, HWu, oc - FB_TW, W, half, t1, inv_c);
                    fb_store_tile(ob, C, HWu, oc, W, half, t0, inv_c);
#pragma unroll
                    for (int r2 = 0; r2 < 16; ++r2) t0[r2] = t1[r2] = t2[r2] = 0.0f;
                } else {
#pragma unroll
                    for (int r2 = 0; r2 < 16; ++r2) t2[r2] = t1[r2], t1[r2] = t0[r2], t0[r2] = 0.0f;
                }
            }
            rbase = nb, lbuf ^= 1;
        }
        advance(cc);
        cs = ns;
        __syncthreads();
    }
}

}  // namespace

// grad_vol [B,D,H,W] -> grad_left AND grad_right [B,C,H,W] from one pass over grad_vol.  STEREO_ERR_UNSUPPORTED outside the
// envelope (the caller then runs the two-launch kernel of corr_mfma_bwd.hip).
int stereo_corr_volume_bwd_fused(const float *left, const float *right, const float *grad_vol, float *grad_left,
                                 float *grad_right, int B, int C, int H, int W, int D, hipStream_t st)
{
    if (!grad_left || !grad_right || C > 32 || C < 1 || D > 192 || D < 1 || W % 4 != 0 || B < 1 || H < 1 || W < 4)
        return STEREO_ERR_UNSUPPORTED;
    if (!aligned16(left) || !aligned16(right) || !aligned16(grad_vol)) return STEREO_ERR_UNSUPPORTED;
    if ((long long)32 * H * W >= (1LL << 31) || (long long)D * H * W >= (1LL << 31) || (long long)B * H >= (1LL << 30))
        return STEREO_ERR_UNSUPPORTED;
    auto kern = corr_volume_bwd_fused_kernel;
    static std::atomic<unsigned long long> lds_done{0};
    if (!lds_opt_in(kern, FB_LDS, lds_done)) return STEREO_ERR_UNSUPPORTED;
    static std::atomic<int> ncu_cache{0};
    int ncu = ncu_cache.load(std::memory_order_relaxed);
    if (ncu <= 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1)
            ncu = 256;
        ncu_cache.store(ncu, std::memory_order_relaxed);
    }
    const int nrows = B * H;
    const int rpw = (nrows + ncu - 1) / ncu;  // one resident workgroup per CU, contiguous rows each
    const int nwg = (nrows + rpw - 1) / rpw;
    if ((long long)rpw * ((W + FB_TW - 1) / FB_TW) * ((D + 31) / 32 + 3) >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;  // item counter
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(FB_NTH), FB_LDS, st, grad_vol, left, right, grad_left, grad_right, C, H, W, D,
                       1.0f / (float)C, nrows, rpw, stereo_opt(OPT_CORR_BWD_FUSED));
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
