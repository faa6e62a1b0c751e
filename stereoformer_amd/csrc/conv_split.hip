// A14's convolutions on the bf16 matrix cores with SPLIT operands: stride-1 "same" 3x3 (and 1x1) convolutions of the RAFT-style
// update block (update/update.py:6-33,66-91,130-178) over channels-last activations, fp32 in / fp32 out, fp32-faithful.
//
// Why not the library: MIOpen's fp32 implicit GEMM already runs at ~83 % of the f32-MFMA peak on these shapes (130 of 157
// TFLOP/s at 384 -> 128 channels, 136 x 240, batch 8), and an update-block iteration is 1.2 TFLOP of it -- 78 % of cfg-4's time,
// with another 17 % in the torch.cat / bias / ReLU passes around the library calls.  v_mfma_f32_32x32x16_bf16 does a 32x32x16
// block in 32 cycles against 8 x 64 for the f32 instruction; with every fp32 operand split exactly into three bf16 pieces
// (hi + mid + lo) and the six leading piece products accumulated in fp32 (the dropped ones are <= 2^-24 relative: fp32 rounding
// level), the same product costs 6 x 32 cycles: 2.7 x less matrix time at fp32 accuracy.  And a kernel of our own takes its
// K dimension from SEVERAL source tensors (the concatenation [h | motion | interp] is never materialised), adds the bias, applies
// the ReLU and scatters its output channels to several destination tensors (convz | convr in one launch).
//
//   GEMM view      M = pixels (A operand, from LDS), N = output channels (B operand, pre-packed weights), K = (tap, channel)
//   workgroup      4 waves, tile = 4 rows x 32 columns of pixels x 128 output channels; wave wn owns all four M blocks (2 rows x
//                  16 columns each) x channels 32 wn .. + 31 (one N block): 4 accumulator blocks = 64 registers, four waves per
//                  SIMD.  (fp16 activations: 8 rows, wave (wm, wn) = rows 4 wm .. x 64 channels, so an A fragment feeds two MFMAs.)
//   K loop         sources in concatenation order, 16 input channels at a time: the (4 + 2) x (32 + 2) halo tile of the chunk is
//                  fetched into registers (16-B quads through one buffer resource per source image: pixels outside the image read
//                  0 = the zero padding), split into its bf16 pieces and written to LDS as [piece][pixel][16 ch + pad]
//                  (48-B pixel stride: the A fragments are conflict-free ds_read_b128 at compile-time offsets per tap); the next
//                  chunk's quads are in flight while the 9 taps x 24 MFMAs of the current one run
//   weights        packed once per module by conv_pack_kernel into the B-fragment order [k-step][n block][piece][lane][8]: a
//                  fragment is one 16-B load per lane, 1 KB contiguous per wave, served by L2 (2.6 MB for 384 -> 128 channels)
//   epilogue       + bias, optional ReLU, every accumulator register is a 128-B row of 32 consecutive output channels of one pixel
//   ceiling        power: the bf16 matrix pipe sustains 1.47-1.66 PFLOP/s on operands with random bits on this part (2.4 on
//                  constant ones; tools/mfma_rate_bf16.hip), and the kernel issues 87-98 % of that MFMA rate
#include "common.h"
#include <type_traits>

namespace {

constexpr int CV_TW = 32;               // output pixel tile: TH (8 or 4) rows x 32 columns
constexpr int CV_PXB = 48;             // LDS bytes per pixel and piece: 16 channels x 2 B + 16 B pad (conflict-free b128 reads)
constexpr int CV_MAXSRC = 4, CV_MAXNT = 5, CV_MAXNB = 4 * CV_MAXNT;   // <= 640 output channels per launch

struct ConvSrcs {
    const void *p[CV_MAXSRC];
    int c[CV_MAXSRC];
    int n;
};
struct ConvDsts {          // one entry per N BLOCK of 32 output channels (a wave's MFMA column block)
    void *p[CV_MAXNB];
    const void *add[CV_MAXNB];   // per-element addend (a ConvGRU context tensor) or null; geometry of the destination
    const void *aux0[CV_MAXNB];  // epilogue operands (hidden state / update gate), geometry of the destination
    const void *aux1[CV_MAXNB];
    int stride[CV_MAXNB];  // channels of the destination tensor (pixel stride)
    int coff[CV_MAXNB];    // first channel of this block inside it
    int nvalid[CV_MAXNB];  // output channels of this block that exist (<= 32; 0: none)
    int epi[CV_MAXNB];     // STEREO_EPI_*
};

// Arithmetic forms (STEREO_CONV_* of include/stereo_hip.h).  A form says how an activation / weight value becomes NP matrix-core
// operands ("pieces"), which piece products are executed (PA[q] x PB[q]) and into which of NACC fp32 accumulators (PC[q]).
//   F32_F16X3 (default for fp32 activations): x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16(2^11 (x - hi)) -- 22-23 significant
//       bits; the products hi hi -> accumulator 0, hi lo' and lo' hi -> accumulator 1, result = acc0 + 2^-11 acc1 (the dropped
//       lo lo term is <= 2^-22 of the product).  Every piece product is exact in fp32 (11 x 11 bits), so the error is that of
//       the fp32 accumulation plus <= 2^-22 relative per operand: fp32-faithful for |x| < 65 504 (the fp16 range -- the range
//       the reference's own mixed_precision flag assumes for this block; beyond it the result is inf / NaN, never silently
//       wrong; below fp16's normal range, |x| < 2^-14, the error becomes absolute, <= 2^-36 per operand).  Three MFMAs per fp32
//       product block instead of six.
//   F32_BF16X6: x = hi + mid + lo exactly (three bf16 pieces, fp32's exponent range), the six leading piece products.
//   F32_BF16X3 ("high"): two bf16 pieces, three products, ~2^-16 relative.   F16: fp16 activations, one product.
template <int AR>
struct ConvAr;
template <>
struct ConvAr<STEREO_CONV_F32_BF16X6> {
    typedef float tin;
    typedef bf16x8 frag;
    typedef __bf16 welem;
    static constexpr int NP = 3, NPB = 3, NPROD = 6, NACC = 1;
    // smallest first: (a1 b1) (a0 b2) (a2 b0) (a0 b1) (a1 b0) (a0 b0)
    static constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0}, PC[6] = {0, 0, 0, 0, 0, 0};
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct ConvAr<STEREO_CONV_F32_BF16X3> {
    typedef float tin;
    typedef bf16x8 frag;
    typedef __bf16 welem;
    static constexpr int NP = 2, NPB = 2, NPROD = 3, NACC = 1;
    static constexpr int PA[6] = {0, 1, 0, 0, 0, 0}, PB[6] = {1, 0, 0, 0, 0, 0}, PC[6] = {0, 0, 0, 0, 0, 0};
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct ConvAr<STEREO_CONV_F16> {
    typedef f16 tin;
    typedef f16x8 frag;
    typedef _Float16 welem;
    static constexpr int NP = 1, NPB = 1, NPROD = 1, NACC = 1;
    static constexpr int PA[6] = {0, 0, 0, 0, 0, 0}, PB[6] = {0, 0, 0, 0, 0, 0}, PC[6] = {0, 0, 0, 0, 0, 0};
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <>
struct ConvAr<STEREO_CONV_F32_F16X3> {
    typedef float tin;
    typedef f16x8 frag;
    typedef _Float16 welem;
    static constexpr int NP = 2, NPB = 2, NPROD = 3, NACC = 2;
    // the two products of accumulator 1 are separated by the one of accumulator 0 (no back-to-back dependent MFMAs)
    static constexpr int PA[6] = {0, 0, 1, 0, 0, 0}, PB[6] = {1, 0, 0, 0, 0, 0}, PC[6] = {1, 0, 1, 0, 0, 0};
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
// compile-time loop: the body sees its index as a constant expression (a `#pragma unroll` loop this long is only partially
// unrolled and then indexes the fragment buffers dynamically)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// F32_F16X3S: the same three products into ONE accumulator -- which frees the registers for a wave tile of 4 M blocks x 2 N
// blocks (8-row pixel tile, 64 channels per wave): half the LDS fragment reads per MFMA of the 4 x 1 tile above.  What makes
// one accumulator possible is that the WEIGHTS are ours to scale: output channel n's weights are multiplied by 2^k[n] (exact)
// so that their maximum lies in [2^13, 2^14), and packed as THREE fp16 pieces b0 = fp16(w'), b1 = fp16(w' - b0) (unscaled: a
// normal fp16 wherever |w'| >= 2^-3, i.e. down to 2^-17 of the channel's largest weight) and b2 = 2^-11 b0 (exact above the
// same threshold).  With the activation pieces a0 = fp16(x), a1 = fp16(2^11 (x - a0)) the products a0 b0, a0 b1 and a1 b2 =
// (x - a0) b0 all carry the scale 2^k[n]; the epilogue multiplies by 2^-k[n] (exact).  Same error budget as F32_F16X3.
template <>
struct ConvAr<STEREO_CONV_F32_F16X3S> {
    typedef float tin;
    typedef f16x8 frag;
    typedef _Float16 welem;
    static constexpr int NP = 2, NPB = 3, NPROD = 3, NACC = 1;
    // (per block the two N blocks' chains are interleaved by `products`, so dependent MFMAs are two apart)
    static constexpr int PA[6] = {0, 1, 0, 0, 0, 0}, PB[6] = {1, 2, 0, 0, 0, 0}, PC[6] = {0, 0, 0, 0, 0, 0};
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
constexpr float CV_LO_UNSCALE = SPLIT16_LO_UNSCALE;

// eight fp32 values -> the form's NP operands
template <int AR>
__device__ __forceinline__ void conv_split_oct(const float *x, typename ConvAr<AR>::frag *p)
{
    if constexpr (AR == STEREO_CONV_F32_BF16X6 || AR == STEREO_CONV_F32_BF16X3) {
        split_oct<ConvAr<AR>::NP>(x, p);
    } else if constexpr (AR == STEREO_CONV_F16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p[0][j] = (_Float16)x[j];
    } else if constexpr (AR == STEREO_CONV_F32_F16X3S) {   // weights (already scaled by 2^k[n]): b0, b1 = unscaled low piece, b2 = 2^-11 b0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 b0 = (_Float16)x[j];
            p[0][j] = b0;
            p[1][j] = (_Float16)(x[j] - (float)b0);
            p[2][j] = (_Float16)((float)b0 * SPLIT16_LO_UNSCALE);
        }
    } else {
        split_oct_f16(x, p);
    }
}

// F32_F16X3S: per output channel the power of two that brings its largest |weight| into [2^13, 2^14) -- wscale[n] = 2^k[n] for
// the packing kernel, wscale[Cpad + n] = 2^-k[n] for the convolution's epilogue.  One workgroup per channel.
__global__ __launch_bounds__(256) void conv_wscale_kernel(const float *__restrict__ w, float *__restrict__ wscale, int Cout, int Cpad,
                                                          int per_channel)
{
    __shared__ float red[256];
    const int n = blockIdx.x;
    float m = 0.0f;
    if (n < Cout)
        for (int i = threadIdx.x; i < per_channel; i += 256) m = fmaxf(m, fabsf(w[(size_t)n * per_channel + i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int k = 0;
        const float mx = red[0];
        if (mx > 0.0f && mx < INFINITY) {
            int e;
            frexpf(mx, &e);              // mx = f 2^e, f in [0.5, 1)  ->  mx 2^(14 - e) in [2^13, 2^14)
            k = 14 - e;
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
        }
        wscale[n] = ldexpf(1.0f, k);
        wscale[Cpad + n] = ldexpf(1.0f, -k);
    }
}

// weights [Cout][Cin][KS][KS] fp32 -> packed[(((chunk * TAPS + tap) * NB + nb) * NP + piece) * 64 + lane][8], where lane l of
// n block nb holds output channel 32 nb + (l & 31), input channels 16 chunk + 8 (l >> 5) + 0..7 (the MFMA's B fragment);
// the pieces of the form (F16: the weight rounded to fp16, what autocast hands the library)
template <int AR>
__global__ void conv_pack_kernel(const float *__restrict__ w, typename ConvAr<AR>::welem *__restrict__ packed, int Cout, int Cin,
                                 int taps, int NB, const float *__restrict__ wscale)
{
    typedef ConvAr<AR> Ar;
    constexpr int NP = Ar::NPB;
    const long long total = (long long)(Cin / 16) * taps * NB * 64;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i % 64);
    const int nb = (int)((i / 64) % NB);
    const int tap = (int)((i / 64 / NB) % taps);
    const int chunk = (int)(i / 64 / NB / taps);
    const int n = 32 * nb + (lane & 31);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 16 * chunk + 8 * (lane >> 5) + j;
        x[j] = n < Cout ? w[((size_t)n * Cin + c) * taps + tap] : 0.0f;
        if constexpr (AR == STEREO_CONV_F32_F16X3S) x[j] *= wscale[n];     // (exact: a power of two)
    }
    const size_t base = ((((size_t)chunk * taps + tap) * NB + nb) * NP) * 64 + lane;
    typename Ar::frag pc[NP];
    conv_split_oct<AR>(x, pc);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<typename Ar::frag *>(packed + (base + (size_t)p * 64) * 8) = pc[p];
}

template <int AR>
constexpr int conv_min_wg_per_cu()
{
    return AR == STEREO_CONV_F16 ? 2 : ((AR == STEREO_CONV_F32_F16X3 || AR == STEREO_CONV_F32_F16X3S) ? 2 : 4);
}

template <int KS, int TH, int AR>
__global__ __launch_bounds__(256, TH == 2 ? 4 : conv_min_wg_per_cu<AR>()) void conv_split_kernel(ConvSrcs src,
                                                                                    const typename ConvAr<AR>::welem *__restrict__ packed,
                                                                                    const float *__restrict__ bias, ConvDsts dst, int relu,
                                                                                    int H, int W, int tiles_x, int tiles_y, int NB, int Cout,
                                                                                    int ntn, int nbatch, int tiles_per_xcd,
                                                                                    const float *__restrict__ oscale)
{
    typedef ConvAr<AR> Ar;
    typedef typename Ar::tin TIN;
    typedef typename Ar::frag frag;
    typedef typename Ar::welem welem;
    constexpr int NP = Ar::NP, NPB = Ar::NPB, NPROD = Ar::NPROD, NACC = Ar::NACC;   // A pieces (LDS image), B pieces (weights)
    constexpr int ES = (int)sizeof(TIN), QCH = 16 / ES, QPP = 16 / QCH;   // channels per 16-B quad, quads per pixel and chunk
    constexpr int HALO = KS / 2, TAPS = KS * KS;
    constexpr int WN = TH == 8 ? 2 : 4;          // waves along the output channels (the other 4 / WN along the rows)
    constexpr int MBW = TH == 2 ? 2 : 4;         // M blocks (2 rows x 16 columns) per wave: 2-row tiles (small images) have two
    constexpr int NBW = 4 / WN;                  // 32-channel N blocks per wave
    constexpr int HCOLS = CV_TW + 2 * HALO, HROWS = TH + 2 * HALO, NPX = HROWS * HCOLS;
    // LDS row stride: a multiple of the 256-B bank row.  ds_read_b128 is served in four 16-lane groups that mix the two pixel
    // rows of an M block ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} ...; MI355X_MICROARCH.md, LDS): with 48-B pixels the 16-B
    // slots of a group's row-0 lanes are {0,3,6,9,4,7,10,13} (+ c) and those of its row-1 lanes the complement exactly when
    // the rows are a whole number of bank rows apart -- the packed 34-pixel row (1632 B) made every group 2-way conflicted
    // (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE: the LDS array, not the MFMA pipe, set the pace).
    constexpr int ROWB = (HCOLS * CV_PXB + 255) / 256 * 256;
    constexpr int PS = HROWS * ROWB;             // piece stride (bytes)
    constexpr int NQ = NPX * QPP, NPASS = (NQ + 255) / 256;   // 16-B quads of the halo tile's chunk
    // PIPE (the split-fp16 form: three MFMAs per fragment pair, two waves per SIMD -- too little of either to hide an LDS
    // round trip behind): two chunk images in LDS, ONE barrier per chunk, A fragments fetched a block ahead and B fragments a
    // tap ahead of the MFMAs that use them
    constexpr bool PIPE = AR == STEREO_CONV_F32_F16X3 || AR == STEREO_CONV_F32_F16X3S;
    constexpr int IMG = NP * PS;                 // one chunk image
    __shared__ __attribute__((aligned(16))) char img[(PIPE ? 2 : 1) * IMG];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroup -> (pixel tile, N tile), XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own
    // L2), so XCD x = id mod 8 takes the x-th CONTIGUOUS eighth of the pixel-tile sequence, N tiles fastest: the workgroups
    // that share a pixel tile's activations (convz | convr) and the tiles that share halo rows / columns meet in one L2.
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int nt = jx % ntn;
    int t = xcd * tiles_per_xcd + jx / ntn;
    if (t >= tiles_x * tiles_y * nbatch) return;   // (whole workgroup, before any barrier)
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = tx * CV_TW, y0 = ty * TH;

    // halo-tile quads of this thread: pixel index inside the image (or -1) and LDS byte offset
    int pix[NPASS], lw[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int q = tid + 256 * i;
        const int px = q / QPP, cq = q % QPP;
        const int hy = px / HCOLS, hx = px - hy * HCOLS;
        const int y = y0 - HALO + hy, x = x0 - HALO + hx;
        pix[i] = (q < NQ && y >= 0 && y < H && x >= 0 && x < W) ? y * W + x : -1;
        lw[i] = q < NQ ? hy * ROWB + hx * CV_PXB + cq * QCH * 2 : -1;
    }
    u32x4_t rk[NPASS];   // raw 16-B quads: 4 floats / 8 halves
    int is_s = 0, is_c = 0;  // source / first channel of the chunk to issue next
    auto issue = [&]() {
        const int Cs = src.c[is_s];
        const __amdgpu_buffer_rsrc_t rs = buffer_rsrc(static_cast<const TIN *>(src.p[is_s]) + (size_t)b * H * W * Cs,
                                                      (unsigned)H * (unsigned)W * (unsigned)Cs * (unsigned)ES);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            // 0x80000000: beyond any image of < 2 GB (launcher), whatever the chunk offset added to it
            const int voff = pix[i] >= 0 ? (pix[i] * Cs + is_c + QCH * (tid % QPP)) * ES : (int)0x80000000;
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        }
        is_c += 16;
        if (is_c >= Cs) is_c = 0, ++is_s;
    };
    auto commit = [&](int buf) {
        char *const im = img + buf * IMG;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if (lw[i] >= 0) {
                if constexpr (AR == STEREO_CONV_F32_BF16X6 || AR == STEREO_CONV_F32_BF16X3) {
                    const f32x4 v = __builtin_bit_cast(f32x4, rk[i]);
                    bf16x2 a[NP], c[NP];
                    split_pair<NP>((f32x2){v[0], v[1]}, a);
                    split_pair<NP>((f32x2){v[2], v[3]}, c);
#pragma unroll
                    for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4 *>(im + p * PS + lw[i]) = cat2(a[p], c[p]);
                } else if constexpr (AR == STEREO_CONV_F32_F16X3 || AR == STEREO_CONV_F32_F16X3S) {
                    const f32x4 v = __builtin_bit_cast(f32x4, rk[i]);
                    f16x2 a[2], c[2];
                    split_pair_f16((f32x2){v[0], v[1]}, a);
                    split_pair_f16((f32x2){v[2], v[3]}, c);
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        *reinterpret_cast<f16x4 *>(im + p * PS + lw[i]) = __builtin_shufflevector(a[p], c[p], 0, 1, 2, 3);
                } else {
                    *reinterpret_cast<u32x4_t *>(im + lw[i]) = rk[i];
                }
            }
        }
    };

    int nchunks = 0;
#pragma unroll
    for (int s = 0; s < CV_MAXSRC; ++s)
        if (s < src.n) nchunks += src.c[s] / 16;

    f32x16 acc[MBW][NBW][NACC];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][a][r] = 0.0f;

    // M block mb = (row pair rp = mb >> 1, column half ch = mb & 1) is 2 rows x 16 columns of pixels: MFMA row i of the block is
    // pixel (4 wm + 2 rp + (i >> 4), 16 ch + (i & 15)) of the tile, so an image whose width is a multiple of 16 (240, 120 ...)
    // wastes no matrix work in its last tile -- blocks entirely outside the image are skipped (wave-uniform).
    // A fragment of (mb, tap (ky, kx), piece p): that pixel shifted by the tap in the halo tile, channels 8 half ..
    const char *abase = img + (4 * wm + (l31 >> 4)) * ROWB + (l31 & 15) * CV_PXB + half * 16;
    // B fragments of this wave: n blocks 4 nt + NBW wn + {0 .. NBW - 1}
    const welem *wbase = packed + ((size_t)(4 * nt + NBW * wn) * NPB * 64 + lane) * 8;
    const size_t kstep = (size_t)NB * NPB * 64 * 8;   // elements per k-step (one tap of one chunk)

    // which of this wave's four M blocks lie inside the image (wave-uniform): all of them (the common case), the left column
    // half only (last tile of a row whose width is 16 mod 32), or some other subset (bottom edge).  The K loop -- and with it
    // the barrier sequence -- is chosen for the WORKGROUP: the waves of an 8-row tile (wm = 0, 1) can see different subsets at
    // the bottom edge, and then all of them take the generic loop with their own predicates.
    bool mbv[4] = {false, false, false, false};
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) mbv[mb] = y0 + 4 * wm + 2 * (mb >> 1) < H && x0 + 16 * (mb & 1) < W;
    int mode = -1;
#pragma unroll
    for (int w = 0; w < 4 / WN; ++w) {
        bool v[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) v[mb] = mb >= MBW || (y0 + 4 * w + 2 * (mb >> 1) < H && x0 + 16 * (mb & 1) < W);
        const bool right1 = MBW == 2 ? !v[1] : (!v[1] && !v[3]);
        const int m = (v[0] && v[1] && v[2] && v[3]) ? 0 : ((v[0] && v[2] && right1) ? 1 : 2);
        mode = (w == 0 || m == mode) ? m : 2;
    }

    // one chunk's 9 taps; MODE selects the M blocks.
    auto load_b = [&](frag (&bf)[NBW][NPB], const welem *wt) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int p = 0; p < NPB; ++p) bf[nb][p] = *reinterpret_cast<const frag *>(wt + ((size_t)nb * NPB + p) * 64 * 8);
    };
    auto load_a = [&](frag (&af)[NP], int buf, int mb, int ky, int kx) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            af[p] = *reinterpret_cast<const frag *>(abase + buf * IMG + p * PS + (2 * (mb >> 1) + ky) * ROWB +
                                                    (16 * (mb & 1) + kx) * CV_PXB);
    };
    auto products = [&](int m0, const frag (&a0)[NP], const frag (&bf)[NBW][NPB]) {
#pragma unroll
        for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
                acc[m0][nb][Ar::PC[q]] = Ar::mma(a0[Ar::PA[q]], bf[nb][Ar::PB[q]], acc[m0][nb][Ar::PC[q]]);
    };
    auto single = [&](int m0, int ky, int kx, const frag (&bf)[NBW][NPB]) {
        frag a0[NP];
        load_a(a0, 0, m0, ky, kx);
        products(m0, a0, bf);
    };
    auto chunk = [&](auto mode_tag, const welem *wk) {
        constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            frag bc[NBW][NPB];
            load_b(bc, wk + (size_t)tap * kstep);
            if (MODE == 0) {
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb) {
                    single(mb, ky, kx, bc);
                    __builtin_amdgcn_sched_barrier(0);   // (hoisting every block's fragments over the unrolled taps spills)
                }
            } else if (MODE == 1) {
#pragma unroll
                for (int mb = 0; mb < MBW; mb += 2) {
                    single(mb, ky, kx, bc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb)
                    if (mbv[mb]) single(mb, ky, kx, bc);
            }
        }
    };

    // the mode is fixed for the tile: one clean K loop per mode (a switch inside the loop makes the accumulators flow through
    // three alternative paths per iteration, which costs registers)
    auto run = [&](auto mode_tag) {
        issue();
        for (int g = 0; g < nchunks; ++g) {
            commit(0);
            __syncthreads();   // the chunk's pieces are visible
            if (g + 1 < nchunks) issue();
            chunk(mode_tag, wbase + (size_t)g * TAPS * kstep);
            __syncthreads();   // every wave is done with the chunk before the next commit overwrites it
        }
    };
    // PIPE: chunk g is computed out of image g & 1 while chunk g + 1 is committed (half-way through, when its loads have long
    // landed) into the other one and chunk g + 2's loads are issued: one barrier per chunk.  The (tap, block) steps of a chunk
    // are one unrolled sequence: step s issues the A fragments of step s + 1 and, on the first block of a tap, the B fragments
    // of the NEXT tap (of the next chunk after the last tap), then its own MFMAs.
    auto run_pipe = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;   // 0: all four M blocks; 1: blocks 0 and 2
        constexpr int NBLK = MODE == 0 ? MBW : MBW / 2, BSTEP = MODE == 0 ? 1 : 2, NSTEP = TAPS * NBLK;
        frag bq[2][NBW][NPB], aq[2][NP];
        issue();
        load_b(bq[0], wbase);
        commit(0);
        if (nchunks > 1) issue();
        __syncthreads();
        for (int g = 0; g < nchunks; ++g) {
            const int cur = g & 1;
            const welem *wk = wbase + (size_t)g * TAPS * kstep;
            load_a(aq[0], cur, 0, 0, 0);
            static_for<0, NSTEP>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int tap = s / NBLK, mb = (s % NBLK) * BSTEP;
                if constexpr (s % NBLK == 0) {
                    if (tap + 1 < TAPS || g + 1 < nchunks) load_b(bq[(tap + 1) & 1], wk + (size_t)(tap + 1) * kstep);
                }
                if constexpr (s + 1 < NSTEP) {
                    constexpr int ntap = (s + 1) / NBLK, nmb = ((s + 1) % NBLK) * BSTEP;
                    load_a(aq[(s + 1) & 1], cur, nmb, ntap / KS, ntap % KS);
                }
                if constexpr (s == NSTEP / 2) {
                    if (g + 1 < nchunks) {
                        commit(cur ^ 1);
                        if (g + 2 < nchunks) issue();
                    }
                }
                // (pin the issue order: the machine scheduler otherwise sinks every load to just in front of its first use)
                __builtin_amdgcn_sched_barrier(0);
                products(mb, aq[s & 1], bq[tap & 1]);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr ((TAPS & 1) != 0) {   // the next chunk's first tap landed in bq[1]
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int p = 0; p < NPB; ++p) bq[0][nb][p] = bq[1][nb][p];
            }
            __syncthreads();   // every wave is done with image `cur`; the commits into the other one are visible
        }
    };
    if constexpr (PIPE) {
        // (an edge tile with some other subset of blocks inside the image computes all four: pixels outside the image read as
        // zeros and are never stored)
        if (mode == 1) run_pipe(std::integral_constant<int, 1>{});
        else run_pipe(std::integral_constant<int, 0>{});
    } else {
        if (mode == 0) run(std::integral_constant<int, 0>{});
        else if (mode == 1) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
    }

    // ---- epilogue: register r of lane (l31, half) holds MFMA row i = (r & 3) + 8 (r >> 2) + 4 half of its M block, i.e. pixel
    // (row + (r >> 3), 16 ch + (r & 3) + 8 ((r >> 2) & 1) + 4 half), output channel 32 (NBW wn + nb) + l31: a 128-B row per
    // access.  STEREO_EPI_*: bias (+ per-element addend), then ReLU / the ConvGRU gate math (update.py:28-32) in place of the
    // separate pointwise launches: sigmoid -> z; sigmoid x h -> r h (convq's first source); tanh and the (1 - z) h + z q blend.
    // Destinations are per N block: several modules can share one 128-channel tile (convz | convb of a 64-channel GRU).
    auto epilogue = [&](auto epi_tag, auto nb_tag) {
        constexpr int EPI = decltype(epi_tag)::value, nb = decltype(nb_tag)::value;
        const int gb = 4 * nt + NBW * wn + nb;             // this wave's N block among the launch's
        const int ostride = dst.stride[gb], coff = dst.coff[gb], nvalid = dst.nvalid[gb];
        const size_t obase = (size_t)b * H * W * ostride;  // this batch image inside the destination-shaped tensors
        TIN *ob = static_cast<TIN *>(dst.p[gb]) + obase;
        const TIN *addp = dst.add[gb] ? static_cast<const TIN *>(dst.add[gb]) + obase : nullptr;
        const TIN *a0p = dst.aux0[gb] ? static_cast<const TIN *>(dst.aux0[gb]) + obase : nullptr;
        const TIN *a1p = dst.aux1[gb] ? static_cast<const TIN *>(dst.aux1[gb]) + obase : nullptr;
        const int cg = 32 * gb + l31;
        const float bv = (bias != nullptr && cg < Cout) ? bias[cg] : 0.0f;
        float osc = 1.0f;       // F32_F16X3S: 2^-k of this output channel
        if constexpr (AR == STEREO_CONV_F32_F16X3S) osc = oscale[cg];
        if (l31 < nvalid) {
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) {
#pragma unroll
                for (int rh = 0; rh < 16; rh += 8) {   // (one row of the block: 8 registers)
                    // the operand rows of 8 outputs are requested together, then the math, then the stores
                    int o[8];                          // element offsets inside this batch image (< 2^31: launcher)
                    bool in[8];
                    float ad[8], u0[8], u1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = rh + j;
                        const int y = y0 + 4 * wm + 2 * (mb >> 1) + (r >> 3);
                        const int x = x0 + 16 * (mb & 1) + (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
                        in[j] = y < H && x < W;
                        o[j] = (y * W + x) * ostride + coff + l31;
                        ad[j] = (in[j] && addp != nullptr) ? (float)addp[o[j]] : 0.0f;
                        if constexpr (EPI >= STEREO_EPI_SIGMOID_MUL) u0[j] = in[j] ? (float)a0p[o[j]] : 0.0f;
                        if constexpr (EPI >= STEREO_EPI_GRU_BLEND) u1[j] = in[j] ? (float)a1p[o[j]] : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = rh + j;
                        float v = acc[mb][nb][0][r];
                        if constexpr (NACC == 2) v = fmaf(acc[mb][nb][1][r], CV_LO_UNSCALE, v);
                        if constexpr (AR == STEREO_CONV_F32_F16X3S) v *= osc;
                        v = (v + bv) + ad[j];
                        if constexpr (EPI == STEREO_EPI_LINEAR) {
                            if (relu) v = fmaxf(v, 0.0f);
                        } else if constexpr (EPI == STEREO_EPI_SIGMOID) {
                            v = gate_sigmoid(v);
                        } else if constexpr (EPI == STEREO_EPI_SIGMOID_MUL) {
                            v = gate_sigmoid(v) * u0[j];
                        } else {
                            const float q = EPI == STEREO_EPI_GRU_BLEND_SOFTSIGN ? gate_softsign(v) : gate_tanh(v);
                            v = gate_blend(u1[j], u0[j], q);
                        }
                        if (in[j]) ob[o[j]] = (TIN)v;
                    }
                }
            }
        }
    };
    static_for<0, NBW>([&](auto nbc) {
        const int epi = dst.epi[4 * nt + NBW * wn + decltype(nbc)::value];
        if (epi == STEREO_EPI_LINEAR) epilogue(std::integral_constant<int, STEREO_EPI_LINEAR>{}, nbc);
        else if (epi == STEREO_EPI_SIGMOID) epilogue(std::integral_constant<int, STEREO_EPI_SIGMOID>{}, nbc);
        else if (epi == STEREO_EPI_SIGMOID_MUL) epilogue(std::integral_constant<int, STEREO_EPI_SIGMOID_MUL>{}, nbc);
        else if (epi == STEREO_EPI_GRU_BLEND) epilogue(std::integral_constant<int, STEREO_EPI_GRU_BLEND>{}, nbc);
        else epilogue(std::integral_constant<int, STEREO_EPI_GRU_BLEND_SOFTSIGN>{}, nbc);
    });
}

// ---- 3x3 convolutions with a handful of output channels (flow_head.conv2: 256 -> 2, update.py:11-14): a 32-channel MFMA N
// block would be 1/16 full, and the library's implicit GEMM takes 0.46 ms for these 2.4 GFLOP.  Plain fp32 FMAs instead: a
// thread owns one pixel of an 8 x 32 tile, the halo tile goes through LDS 32 input channels at a time ([pixel][32 + 4 pad]:
// conflict-free 16-B reads), the weights are wave-uniform (scalar loads).  Channels-last in, NCHW out (the consumer adds the
// result to the NCHW coordinate grid).
template <int COUT, typename TIN>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const TIN *__restrict__ x, const float *__restrict__ wt,
                                                            const float *__restrict__ bias, float *__restrict__ out, int Cin,
                                                            int H, int W, int tiles_x, int tiles_y, int relu)
{
    constexpr int HC = 34, HR = 10, NPX = HC * HR, PXS = 36;   // halo tile, floats per pixel in LDS
    constexpr int ES = (int)sizeof(TIN), QCH = 16 / ES, QPP = 32 / QCH;   // channels per 16-B quad, quads per pixel and chunk
    constexpr int NQ = NPX * QPP, NPASS = (NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[NPX * PXS];
    // the chunk's weights [tap][32 channels][COUT] go through LDS as well: read back as wave-wide broadcasts.  (As scalar loads
    // -- they are wave-uniform -- they shared the LDS reads' counter, and scalar loads return out of order: every wait in the tap
    // loop drained both, 288 times per chunk: 0.12 ms for a 0.27-GB read.)
    constexpr int WCH = 9 * 32 * COUT;
    __shared__ __attribute__((aligned(16))) float wl[WCH];
    const int tid = threadIdx.x, row = tid >> 5, col = tid & 31;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = 32 * tx, y0 = 8 * ty;
    const __amdgpu_buffer_rsrc_t rs = buffer_rsrc(x + (size_t)b * H * W * Cin, (unsigned)H * (unsigned)W * (unsigned)Cin * (unsigned)ES);
    int voff[NPASS], lw[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int q = tid + 256 * i;
        const int px = q / QPP, cq = q % QPP;
        const int hy = px / HC, hx = px - hy * HC;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        voff[i] = (q < NQ && yy >= 0 && yy < H && xx >= 0 && xx < W) ? ((yy * W + xx) * Cin + QCH * cq) * ES : (int)0x80000000;
        lw[i] = q < NQ ? px * PXS + QCH * cq : -1;
    }
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.0f;
    const float *tp = tile + (row * HC + col) * PXS;
    u32x4_t rk[NPASS];
    constexpr int WPT = (WCH + 255) / 256;   // weight floats per thread and chunk
    float wr[WPT];
    auto issue = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i)
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], c0 * ES, 0);   // (soffset is range-checked with voffset)
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + 256 * i;             // element (tap, channel, output) of the chunk
            const int tap = e / (32 * COUT), r = e - tap * (32 * COUT);
            wr[i] = e < WCH ? wt[((size_t)tap * Cin + c0) * COUT + r] : 0.0f;
        }
    };
    issue(0);
    for (int c0 = 0; c0 < Cin; c0 += 32) {
        __syncthreads();   // the previous chunk has been consumed
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if (lw[i] >= 0) {
                if constexpr (ES == 4) {
                    *reinterpret_cast<u32x4_t *>(tile + lw[i]) = rk[i];
                } else {   // fp16 activations: widened on the way into LDS
                    const f16x8 h = __builtin_bit_cast(f16x8, rk[i]);
                    f32x4 lo, hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) lo[e] = (float)h[e], hi[e] = (float)h[4 + e];
                    *reinterpret_cast<f32x4 *>(tile + lw[i]) = lo;
                    *reinterpret_cast<f32x4 *>(tile + lw[i] + 4) = hi;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (tid + 256 * i < WCH) wl[tid + 256 * i] = wr[i];
        __syncthreads();
        if (c0 + 32 < Cin) issue(c0 + 32);   // the next chunk's loads fly under this chunk's arithmetic
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float *pp = tp + ((tap / 3) * HC + tap % 3) * PXS;
            const float *wk = wl + tap * 32 * COUT;   // [tap][channel][COUT]
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(pp + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (COUT == 2) {   // (the flow head: both outputs in one v_pk_fma_f32)
                        const f32x2 w2 = *reinterpret_cast<const f32x2 *>(wk + (4 * q + i) * 2);
                        const f32x2 a2 = __builtin_elementwise_fma((f32x2){v[i], v[i]}, w2, (f32x2){acc[0], acc[1]});
                        acc[0] = a2[0], acc[1] = a2[1];
                    } else {
#pragma unroll
                        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v[i], wk[(4 * q + i) * COUT + o], acc[o]);
                    }
                }
            }
        }
    }
    const int y = y0 + row, xo = x0 + col;
    if (y < H && xo < W) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) out[(((size_t)b * COUT + o) * H + y) * W + xo] = relu ? fmaxf(acc[o], 0.0f) : acc[o];
    }
}

// ---- KxK convolutions with a handful of INPUT channels (convf1: 7x7 over the 2-channel flow, 64 output channels,
// update.py:79,87): 98 multiply-adds per output, nothing for a matrix core to do.  A thread owns one pixel of an 8 x 32 tile and
// 16 output channels at a time; the (8 + K - 1) x (32 + K - 1) x Cin input tile sits in LDS, the weights are wave-uniform
// (scalar loads).  NCHW fp32 in, channels-last out with the ReLU applied.
template <int K>
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                             const float *__restrict__ bias, float *__restrict__ out, int Cin,
                                                             int Cout, int H, int W, int tiles_x, int tiles_y, int ostride, int relu)
{
    constexpr int HC = 32 + K - 1, HR = 8 + K - 1, HALO = K / 2, MAXC = 4;
    __shared__ float tile[MAXC * HR * HC];
    // the weights of ONE group of 16 output channels, [tap][channel][16], staged through LDS per group and read back as
    // wave-wide broadcasts.  As scalar loads (they are wave-uniform) every tap's 64 bytes were fetched right before use and
    // waited for -- ~400 dependent scalar-load round trips per wave: 0.14-0.16 ms for a 25-us arithmetic job.
    __shared__ __attribute__((aligned(16))) float wsh[K * K * MAXC * 16];
    const int tid = threadIdx.x, row = tid >> 5, col = tid & 31;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = 32 * tx, y0 = 8 * ty;
    for (int i = tid; i < Cin * HR * HC; i += 256) {
        const int c = i / (HR * HC), r = i - c * (HR * HC);
        const int hy = r / HC, hx = r - hy * HC;
        const int yy = y0 - HALO + hy, xx = x0 - HALO + hx;
        tile[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(((size_t)b * Cin + c) * H + yy) * W + xx] : 0.0f;
    }
    __syncthreads();
    const int y = y0 + row, xo = x0 + col;
    float *op = out + (((size_t)b * H + y) * W + xo) * ostride;
    for (int o0 = 0; o0 < Cout; o0 += 16) {
        __syncthreads();   // the previous group's weights have been consumed (first round: the tile is complete)
        for (int i = tid; i < K * K * Cin * 4; i += 256) {   // 16-byte quads of [tap][channel][16]
            const int tc = i >> 2, q4 = (i & 3) * 4;
            *reinterpret_cast<f32x4 *>(wsh + tc * 16 + q4) = *reinterpret_cast<const f32x4 *>(wt + (size_t)tc * Cout + o0 + q4);
        }
        __syncthreads();
        f32x2 acc[8];    // channel pairs: v_pk_fma_f32 does two multiply-adds in the issue slot of one
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = bias ? (f32x2){bias[o0 + 2 * o], bias[o0 + 2 * o + 1]} : (f32x2){0.0f, 0.0f};
        for (int c = 0; c < Cin; ++c) {
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float v = tile[(c * HR + row + ky) * HC + col + kx];
                    const f32x2 vv = {v, v};
                    const f32x2 *wk = reinterpret_cast<const f32x2 *>(wsh + ((ky * K + kx) * Cin + c) * 16);   // wave-uniform
#pragma unroll
                    for (int o = 0; o < 8; ++o) acc[o] = __builtin_elementwise_fma(vv, wk[o], acc[o]);
                }
            }
        }
        if (y < H && xo < W) {
#pragma unroll
            for (int o = 0; o < 8; o += 2) {
                f32x4 v = {acc[o][0], acc[o][1], acc[o + 1][0], acc[o + 1][1]};
                if (relu) v = __builtin_elementwise_max(v, (f32x4){0.0f, 0.0f, 0.0f, 0.0f});
                st4(op + o0 + 2 * o, v);
            }
        }
    }
}

// convf1 proper -- 7x7 over TWO input channels (the flow), 32 or 64 output channels -- as an implicit GEMM on the fp16 matrix
// cores (end of round 3): K = 49 taps x 2 channels = 98 (padded to 7 k-steps of 16), M = pixels, N = output channels, operands
// split as 16 x = hi + lo with an unscaled low piece, three products per k-step into ONE accumulator (attention_f16.hip's
// arithmetic: fp32-faithful for |x| < 4 094).  A workgroup takes 32 x 8 pixels: the fp32 input tile (with halo) sits in LDS as in
// the vector kernel above, a lane gathers the 8 taps of its k-group for its pixel (tap offsets are compile-time constants
// selected by the lane's half), splits them in registers; the weights are split ONCE per workgroup into an [N][K] fp16 image
// whose rows a lane reads as one 16-byte B fragment.  98 x 64 multiply-adds per pixel are 42 matrix instructions per 32 pixels
// instead of ~2 000 LDS reads and 3 100 packed FMAs per pixel on the vector kernel: 86 -> ~15 us at [8,2,136,240] -> 64.
template <int NT>
__global__ __launch_bounds__(256) void conv7_cin2_mfma_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                              const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                                                              int tiles_x, int tiles_y, int ostride, int relu, int ntiles)
{
    constexpr int K = 7, CIN = 2, HC = 32 + K - 1, HR = 8 + K - 1, HALO = K / 2, KK = 112, KROW = KK + 8, COUT = 32 * NT;
    __shared__ float tile[CIN * HR * HC + 4];
    __shared__ __attribute__((aligned(16))) _Float16 wB[2][COUT][KROW];   // [piece][out channel][k = tap * 2 + channel]
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    float pre = 16.0f;
    asm volatile("" : "+v"(pre));
    auto split2 = [&](float a0, float a1, unsigned &h, unsigned &l) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(pre));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(pre));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(pre), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(pre), "v"(h));
    };
    // weights [tap][channel][COUT] fp32 -> pieces [COUT][k], once per workgroup; k >= 98: zeros (the A operand reads tap 0
    // there: finite)
    {   // (all of a thread's weight loads in flight together: COUT * 56 / 256 = 7 NT pairs per thread)
        constexpr int NW = COUT * (KK / 2) / 256;
        static_assert(COUT * (KK / 2) % 256 == 0, "weight pairs per thread");
        float w0[NW], w1[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + 256 * u, n = i % COUT, kp = i / COUT;   // k = 2 kp, 2 kp + 1 (one tap, both channels)
            w0[u] = 2 * kp < K * K * CIN ? wt[(size_t)(2 * kp) * COUT + n] : 0.0f;
            w1[u] = 2 * kp + 1 < K * K * CIN ? wt[(size_t)(2 * kp + 1) * COUT + n] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + 256 * u, n = i % COUT, kp = i / COUT;
            unsigned h, l;
            split2(w0[u], w1[u], h, l);
            *reinterpret_cast<unsigned *>(&wB[0][n][2 * kp]) = h;
            *reinterpret_cast<unsigned *>(&wB[1][n][2 * kp]) = l;
        }
    }
  for (int tt = blockIdx.x; tt < ntiles; tt += gridDim.x) {
    int t = tt;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = 32 * tx, y0 = 8 * ty;
    __syncthreads();   // the previous tile's gathers are done
    for (int i = tid; i < CIN * HR * HC; i += 256) {
        const int c = i / (HR * HC), r = i - c * (HR * HC);
        const int hy = r / HC, hx = r - hy * HC;
        const int yy = y0 - HALO + hy, xx = x0 - HALO + hx;
        tile[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(((size_t)b * CIN + c) * H + yy) * W + xx] : 0.0f;
    }
    __syncthreads();
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    const int pbase = (2 * wave) * HC + l31;                         // this lane's pixel of the wave's first row
#pragma unroll
    for (int s = 0; s < KK / 16; ++s) {
        f16x8 ap[2][2];                                              // [row of the wave][piece]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // k = 16 s + 8 half + j -> tap k / 2 = (ky, kx), channel k & 1: two compile-time offsets, the lane's half picks
                constexpr int dummy = 0;
                (void)dummy;
                const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;
                const int t0 = k0 < K * K * CIN ? k0 / 2 : 0, t1 = k1 < K * K * CIN ? k1 / 2 : 0;
                const int o0 = ((k0 & 1) * HR + t0 / K) * HC + t0 % K, o1 = ((k1 & 1) * HR + t1 / K) * HC + t1 % K;
                a[j] = tile[pbase + mt * HC + (half ? o1 : o0)];
            }
            unsigned h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split2(a[2 * j], a[2 * j + 1], h[j], l[j]);
            ap[mt][0] = __builtin_bit_cast(f16x8, (u32x4v){h[0], h[1], h[2], h[3]});
            ap[mt][1] = __builtin_bit_cast(f16x8, (u32x4v){l[0], l[1], l[2], l[3]});
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f16x8 bh = *reinterpret_cast<const f16x8 *>(&wB[0][32 * nt + l31][16 * s + 8 * half]);
            const f16x8 bl = *reinterpret_cast<const f16x8 *>(&wB[1][32 * nt + l31][16 * s + 8 * half]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[mt][0], bh, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[mt][0], bl, acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[mt][1], bh, acc[mt][nt], 0, 0, 0);
            }
        }
    }
    // accumulator register r of lane (l31, half): pixel column (r & 3) + 8 (r >> 2) + 4 half of the row, output channel l31
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int y = y0 + 2 * wave + mt;
        if (y < H) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float bv = bias ? bias[32 * nt + l31] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xo = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = fmaf(acc[mt][nt][r], 1.0f / 256.0f, bv);
                    if (relu) v = fmaxf(v, 0.0f);
                    if (xo < W) out[(((size_t)b * H + y) * W + xo) * ostride + 32 * nt + l31] = v;
                }
            }
        }
    }
  }
}

}  // namespace

extern "C" int stereo_conv_small_cin_nhwc_f32(const float *x, const float *weight_tco, const float *bias, float *out, int B, int Cin,
                                              int Cout, int H, int W, int ksize, int out_stride, int relu, void *stream)
{
    if (B < 0 || Cin < 1 || Cout < 1 || H < 0 || W < 0 || out_stride < Cout) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(x);
    STEREO_REQUIRE_PTR(weight_tco);
    STEREO_REQUIRE_PTR(out);
    if (Cin > 4 || Cout % 16 != 0 || Cout > 64 || out_stride % 4 != 0 || !aligned16(out) || !aligned16(weight_tco) ||
        (ksize != 3 && ksize != 5 && ksize != 7))
        return STEREO_ERR_UNSUPPORTED;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
    const long long nt = (long long)tiles_x * tiles_y * B;
    if (nt >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    if (ksize == 7 && Cin == 2 && (Cout == 64 || Cout == 32) && stereo_opt(OPT_CORR_F16P)) {   // convf1: matrix cores (option 0: vector kernel)
        const unsigned nwg = (unsigned)nt;   // one tile per workgroup: the launch is bound by a workgroup's own latency (persistent runs of
                                             // two tiles with the weights split once measured slower: 66 vs 57 us at [8,2,136,240])
        if (Cout == 64)
            hipLaunchKernelGGL((conv7_cin2_mfma_kernel<2>), dim3(nwg), dim3(256), 0, as_stream(stream), x, weight_tco, bias, out, H, W,
                               tiles_x, tiles_y, out_stride, relu, (int)nt);
        else
            hipLaunchKernelGGL((conv7_cin2_mfma_kernel<1>), dim3(nwg), dim3(256), 0, as_stream(stream), x, weight_tco, bias, out, H, W,
                               tiles_x, tiles_y, out_stride, relu, (int)nt);
        STEREO_LAUNCH_CHECK();
        return STEREO_OK;
    }
#define SMALLCIN(KV)                                                                                                          \
    hipLaunchKernelGGL((conv_small_cin_kernel<KV>), dim3((unsigned)nt), dim3(256), 0, as_stream(stream), x, weight_tco, bias, out, \
                       Cin, Cout, H, W, tiles_x, tiles_y, out_stride, relu)
    if (ksize == 7) SMALLCIN(7);
    else if (ksize == 5) SMALLCIN(5);
    else SMALLCIN(3);
#undef SMALLCIN
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

namespace {
template <typename TIN>
int conv3x3_small(const TIN *x, const float *weight_tco, const float *bias, float *out_nchw, int B, int Cin, int Cout, int H, int W,
                  int relu, void *stream)
{
    if (B < 0 || Cin < 0 || Cout < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || H == 0 || W == 0 || Cout == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(x);
    STEREO_REQUIRE_PTR(weight_tco);
    STEREO_REQUIRE_PTR(out_nchw);
    if (Cin < 32 || Cin % 32 != 0 || Cout > 4 || !aligned16(x) || (long long)H * W * Cin * (long long)sizeof(TIN) >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
    const long long nt = (long long)tiles_x * tiles_y * B;
    if (nt >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
#define SMALL(CO)                                                                                                              \
    hipLaunchKernelGGL((conv3x3_small_kernel<CO, TIN>), dim3((unsigned)nt), dim3(256), 0, as_stream(stream), x, weight_tco, bias, \
                       out_nchw, Cin, H, W, tiles_x, tiles_y, relu)
    if (Cout == 1) SMALL(1);
    else if (Cout == 2) SMALL(2);
    else if (Cout == 3) SMALL(3);
    else SMALL(4);
#undef SMALL
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
}  // namespace

extern "C" int stereo_conv3x3_small_nhwc_f32(const float *x, const float *weight_tco, const float *bias, float *out_nchw, int B,
                                             int Cin, int Cout, int H, int W, void *stream)
{
    return conv3x3_small<float>(x, weight_tco, bias, out_nchw, B, Cin, Cout, H, W, 0, stream);
}
extern "C" int stereo_conv3x3_small_nhwc_ex(const void *x, int x_is_f16, const float *weight_tco, const float *bias, float *out_nchw,
                                            int B, int Cin, int Cout, int H, int W, int relu, void *stream)
{
    if (x_is_f16) return conv3x3_small<f16>(static_cast<const f16 *>(x), weight_tco, bias, out_nchw, B, Cin, Cout, H, W, relu, stream);
    return conv3x3_small<float>(static_cast<const float *>(x), weight_tco, bias, out_nchw, B, Cin, Cout, H, W, relu, stream);
}
extern "C" int stereo_conv3x3_small_nhwc_f16(const void *x, const float *weight_tco, const float *bias, float *out_nchw, int B,
                                             int Cin, int Cout, int H, int W, void *stream)
{
    return conv3x3_small<f16>(static_cast<const f16 *>(x), weight_tco, bias, out_nchw, B, Cin, Cout, H, W, 0, stream);
}

namespace {

template <int AR>
long long weight_bytes(int Cin, int ksize, long long NB)
{
    return (long long)(Cin / 16) * ksize * ksize * NB * ConvAr<AR>::NPB * 64 * 8 * 2;
}

template <int AR>
long long packed_bytes(int Cout, int Cin, int ksize)
{
    if (Cout < 1 || Cin < 16 || Cin % 16 != 0 || (ksize != 1 && ksize != 3)) return 0;
    const long long NB = 4LL * ((Cout + 127) / 128);
    return weight_bytes<AR>(Cin, ksize, NB) + (AR == STEREO_CONV_F32_F16X3S ? 2LL * NB * 32 * 4 : 0);   // (+ the channel scales)
}

template <int AR>
int pack_weights(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream)
{
    if (packed_bytes<AR>(Cout, Cin, ksize) == 0) return STEREO_ERR_BAD_SHAPE;
    STEREO_REQUIRE_PTR(weight);
    STEREO_REQUIRE_PTR(packed);
    if (!aligned16(packed)) return STEREO_ERR_UNSUPPORTED;
    const int NB = 4 * ((Cout + 127) / 128), taps = ksize * ksize;
    const long long total = (long long)(Cin / 16) * taps * NB * 64;
    float *wscale = nullptr;
    if (AR == STEREO_CONV_F32_F16X3S) {   // channel scales behind the weights: [2^k | 2^-k] x (NB * 32)
        wscale = reinterpret_cast<float *>(static_cast<char *>(packed) + weight_bytes<AR>(Cin, ksize, NB));
        hipLaunchKernelGGL(conv_wscale_kernel, dim3((unsigned)(NB * 32)), dim3(256), 0, as_stream(stream), weight, wscale, Cout,
                           NB * 32, Cin * taps);
    }
    hipLaunchKernelGGL((conv_pack_kernel<AR>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), weight,
                       static_cast<typename ConvAr<AR>::welem *>(packed), Cout, Cin, taps, NB, wscale);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

template <int AR>
int conv_nhwc(const void *const *srcs, const int *src_channels, int nsrc, const void *packed, const float *bias,
              void *const *outs, const int *out_channels, const int *out_strides, int nout, const int *epilogue,
              const void *const *addend, const void *const *aux0, const void *const *aux1, int ksize, int relu, int B, int H,
              int W, void *stream)
{
    typedef typename ConvAr<AR>::tin TIN;
    if (nsrc < 1 || nsrc > CV_MAXSRC || nout < 1 || nout > CV_MAXNB || (ksize != 1 && ksize != 3) || B < 0 || H < 0 || W < 0)
        return STEREO_ERR_BAD_SHAPE;
    if (srcs == nullptr || src_channels == nullptr || outs == nullptr || out_channels == nullptr) return STEREO_ERR_NULL_POINTER;
    STEREO_REQUIRE_PTR(packed);
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    ConvSrcs s{};
    ConvDsts d{};
    s.n = nsrc;
    int csum = 0;   // input channels of the launch (= the packed weights' Cin)
    for (int i = 0; i < nsrc; ++i) {
        csum += src_channels[i];
        STEREO_REQUIRE_PTR(srcs[i]);
        if (src_channels[i] < 16 || src_channels[i] % 16 != 0) return STEREO_ERR_UNSUPPORTED;
        if (!aligned16(srcs[i]) || (long long)H * W * src_channels[i] * (long long)sizeof(TIN) >= (1LL << 31))
            return STEREO_ERR_UNSUPPORTED;
        s.p[i] = srcs[i];
        s.c[i] = src_channels[i];
    }
    // destinations: every one but the last must fill whole 32-channel N blocks (a wave's MFMA column block); a 128-channel N
    // tile (= a workgroup) may serve several of them
    int Cout = 0, nblk = 0;
    for (int i = 0; i < nout; ++i) {
        STEREO_REQUIRE_PTR(outs[i]);
        const int c = out_channels[i];
        const int cs = out_strides ? out_strides[i] : c;   // channels of the destination tensor (>= c: the rest is the caller's)
        if (c < 1 || cs < c || (i + 1 < nout && c % 32 != 0)) return STEREO_ERR_UNSUPPORTED;
        if ((long long)H * W * cs >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;     // 32-bit element offsets in the epilogue
        const int epi = epilogue ? epilogue[i] : STEREO_EPI_LINEAR;
        if (epi < STEREO_EPI_LINEAR || epi > STEREO_EPI_GRU_BLEND_SOFTSIGN) return STEREO_ERR_BAD_SHAPE;
        if (epi != STEREO_EPI_LINEAR && relu) return STEREO_ERR_BAD_SHAPE;
        const void *ad = addend ? addend[i] : nullptr, *x0 = aux0 ? aux0[i] : nullptr, *x1 = aux1 ? aux1[i] : nullptr;
        if (epi >= STEREO_EPI_SIGMOID_MUL && x0 == nullptr) return STEREO_ERR_NULL_POINTER;
        if (epi >= STEREO_EPI_GRU_BLEND && x1 == nullptr) return STEREO_ERR_NULL_POINTER;
        for (int o = 0; o < c; o += 32) {
            if (nblk >= CV_MAXNB) return STEREO_ERR_UNSUPPORTED;
            d.p[nblk] = outs[i];
            d.add[nblk] = ad;
            d.aux0[nblk] = x0;
            d.aux1[nblk] = x1;
            d.stride[nblk] = cs;
            d.coff[nblk] = o;
            d.nvalid[nblk] = c - o < 32 ? c - o : 32;
            d.epi[nblk] = epi;
            ++nblk;
        }
        Cout += c;
    }
    const int ntn = (nblk + 3) / 4;
    for (int k = nblk; k < 4 * ntn; ++k) d.p[k] = outs[0], d.stride[k] = 1, d.nvalid[k] = 0, d.epi[k] = STEREO_EPI_LINEAR;   // (unused blocks of the last tile)
    const int NB = 4 * ((Cout + 127) / 128);
    const int tiles_x = (W + CV_TW - 1) / CV_TW;
    // pixel tile: 4 rows x 32 columns, a wave = 4 rows x 32 output channels (an 8-row tile with 64 channels per wave measured equal
    // at the large shapes and 7-18 % behind at 1/8 and 1/16 scale: finer workgroups spread a launch more evenly over the chip)
    // (fp16 activations: one MFMA per fragment instead of six makes the 4-row shape LDS-bound -- every A fragment feeds a single
    // MFMA; the 8-row shape, a wave = 4 rows x 64 channels, halves the fragment reads per MFMA)
    constexpr int th = (sizeof(TIN) == 4 && AR != STEREO_CONV_F32_F16X3S) ? 4 : 8;
    typedef typename ConvAr<AR>::welem welem;
    const float *oscale = nullptr;
    if (AR == STEREO_CONV_F32_F16X3S)
        oscale = reinterpret_cast<const float *>(static_cast<const char *>(packed) + weight_bytes<AR>(csum, ksize, NB)) + NB * 32;
    // Launches with a SHORT K loop (<= 128 input channels at small sizes, <= 64 at any size) or very few tiles take 2-row pixel
    // tiles: twice as many workgroups of half the work, four resident per CU instead of two.  With few chunks per workgroup the
    // prologue / epilogue and the tile quantisation dominate (576 four-row tiles on 512 slots run as two rounds, the second one
    // eighth full): the LowCNN updater's convolutions at [8,.,72,120] run 10-20 % faster (tools/conv_rows_ab.py); with 384 input
    // channels the 4-row tile's weight reuse wins at every size.
    if constexpr (AR == STEREO_CONV_F32_F16X3) {
        const long long nt4 = (long long)tiles_x * ((H + 3) / 4) * B * ntn;
        const int forced = stereo_opt(OPT_CONV_ROWS);   // test hook
        const bool small = csum <= 64 || (csum <= 128 && nt4 <= 1024) || nt4 <= 320;
        if (forced == 2 || (forced != 4 && small)) {
            const int tiles_y2 = (H + 1) / 2;
            const long long nt2 = (long long)tiles_x * tiles_y2 * B;
            const long long per2 = (nt2 + 7) / 8;
            dim3 grid2((unsigned)(per2 * 8 * ntn));
            if (ksize == 3)
                hipLaunchKernelGGL((conv_split_kernel<3, 2, AR>), grid2, dim3(256), 0, as_stream(stream), s,
                                   static_cast<const welem *>(packed), bias, d, relu, H, W, tiles_x, tiles_y2, NB, Cout, ntn, B,
                                   (int)per2, oscale);
            else
                hipLaunchKernelGGL((conv_split_kernel<1, 2, AR>), grid2, dim3(256), 0, as_stream(stream), s,
                                   static_cast<const welem *>(packed), bias, d, relu, H, W, tiles_x, tiles_y2, NB, Cout, ntn, B,
                                   (int)per2, oscale);
            STEREO_LAUNCH_CHECK();
            return STEREO_OK;
        }
    }
    const int tiles_y = (H + th - 1) / th;
    const long long nt = (long long)tiles_x * tiles_y * B;
    const long long per_xcd = (nt + 7) / 8;
    if (per_xcd * 8 * ntn >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(per_xcd * 8 * ntn));
#define CONV_LAUNCH(KSV)                                                                                                        \
    hipLaunchKernelGGL((conv_split_kernel<KSV, th, AR>), grid, dim3(256), 0, as_stream(stream), s, static_cast<const welem *>(packed), \
                       bias, d, relu, H, W, tiles_x, tiles_y, NB, Cout, ntn, B, (int)per_xcd, oscale)
    if (ksize == 3) CONV_LAUNCH(3);
    else CONV_LAUNCH(1);
#undef CONV_LAUNCH
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

#define CONV_FORMS(CALL)                                       \
    switch (arith) {                                           \
    case STEREO_CONV_F32_BF16X6: return CALL(STEREO_CONV_F32_BF16X6); \
    case STEREO_CONV_F32_BF16X3: return CALL(STEREO_CONV_F32_BF16X3); \
    case STEREO_CONV_F16: return CALL(STEREO_CONV_F16);       \
    case STEREO_CONV_F32_F16X3: return CALL(STEREO_CONV_F32_F16X3); \
    case STEREO_CONV_F32_F16X3S: return CALL(STEREO_CONV_F32_F16X3S); \
    default: break;                                            \
    }

extern "C" long long stereo_conv_packed_bytes_ex(int arith, int Cout, int Cin, int ksize)
{
#define CALL(A) packed_bytes<A>(Cout, Cin, ksize)
    CONV_FORMS(CALL)
#undef CALL
    return 0;
}
extern "C" int stereo_conv_pack_weights_ex(int arith, const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream)
{
#define CALL(A) pack_weights<A>(weight, packed, Cout, Cin, ksize, stream)
    CONV_FORMS(CALL)
#undef CALL
    return STEREO_ERR_BAD_SHAPE;
}
extern "C" int stereo_conv_nhwc_ex(int arith, const void *const *srcs, const int *src_channels, int nsrc, const void *packed,
                                   const float *bias, void *const *outs, const int *out_channels, const int *out_strides, int nout,
                                   const int *epilogue, const void *const *addend, const void *const *aux0, const void *const *aux1,
                                   int ksize, int relu, int B, int H, int W, void *stream)
{
#define CALL(A) \
    conv_nhwc<A>(srcs, src_channels, nsrc, packed, bias, outs, out_channels, out_strides, nout, epilogue, addend, aux0, aux1, ksize, relu, B, H, W, stream)
    CONV_FORMS(CALL)
#undef CALL
    return STEREO_ERR_BAD_SHAPE;
}

// the round-2 entry points: fixed arithmetic form, linear epilogue
extern "C" long long stereo_conv_packed_bytes(int Cout, int Cin, int ksize) { return packed_bytes<STEREO_CONV_F32_BF16X6>(Cout, Cin, ksize); }
extern "C" long long stereo_conv_packed_bytes_bf16x3(int Cout, int Cin, int ksize) { return packed_bytes<STEREO_CONV_F32_BF16X3>(Cout, Cin, ksize); }
extern "C" long long stereo_conv_packed_bytes_f16(int Cout, int Cin, int ksize) { return packed_bytes<STEREO_CONV_F16>(Cout, Cin, ksize); }
extern "C" int stereo_conv_pack_weights(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream)
{
    return pack_weights<STEREO_CONV_F32_BF16X6>(weight, packed, Cout, Cin, ksize, stream);
}
extern "C" int stereo_conv_pack_weights_bf16x3(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream)
{
    return pack_weights<STEREO_CONV_F32_BF16X3>(weight, packed, Cout, Cin, ksize, stream);
}
extern "C" int stereo_conv_pack_weights_f16(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream)
{
    return pack_weights<STEREO_CONV_F16>(weight, packed, Cout, Cin, ksize, stream);
}
extern "C" int stereo_conv_nhwc_f32(const float *const *srcs, const int *src_channels, int nsrc, const void *packed,
                                    const float *bias, float *const *outs, const int *out_channels, const int *out_strides,
                                    int nout, int ksize, int relu, int B, int H, int W, void *stream)
{
    return conv_nhwc<STEREO_CONV_F32_BF16X6>(reinterpret_cast<const void *const *>(srcs), src_channels, nsrc, packed, bias,
                                             reinterpret_cast<void *const *>(outs), out_channels, out_strides, nout, nullptr, nullptr,
                                             nullptr, nullptr, ksize, relu, B, H, W, stream);
}
extern "C" int stereo_conv_nhwc_f32_bf16x3(const float *const *srcs, const int *src_channels, int nsrc, const void *packed,
                                           const float *bias, float *const *outs, const int *out_channels, const int *out_strides,
                                           int nout, int ksize, int relu, int B, int H, int W, void *stream)
{
    return conv_nhwc<STEREO_CONV_F32_BF16X3>(reinterpret_cast<const void *const *>(srcs), src_channels, nsrc, packed, bias,
                                             reinterpret_cast<void *const *>(outs), out_channels, out_strides, nout, nullptr, nullptr,
                                             nullptr, nullptr, ksize, relu, B, H, W, stream);
}
extern "C" int stereo_conv_nhwc_f16(const void *const *srcs, const int *src_channels, int nsrc, const void *packed,
                                    const float *bias, void *const *outs, const int *out_channels, const int *out_strides,
                                    int nout, int ksize, int relu, int B, int H, int W, void *stream)
{
    return conv_nhwc<STEREO_CONV_F16>(srcs, src_channels, nsrc, packed, bias, outs, out_channels, out_strides, nout, nullptr, nullptr,
                                      nullptr, nullptr, ksize, relu, B, H, W, stream);
}
