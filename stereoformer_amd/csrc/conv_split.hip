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
//   workgroup      4 waves, tile = 8 rows x 32 columns of pixels x 128 output channels; wave (wm, wn) owns rows 4 wm .. 4 wm + 3
//                  (four 32-pixel M blocks) x channels 64 wn .. + 63 (two N blocks): 8 accumulator blocks = 128 registers
//   K loop         sources in concatenation order, 16 input channels at a time: the (8 + 2) x (32 + 2) halo tile of the chunk is
//                  fetched into registers (16-B quads through one buffer resource per source image: pixels outside the image read
//                  0 = the zero padding), split into its three bf16 pieces and written to LDS as [piece][pixel][16 ch + pad]
//                  (48-B pixel stride: the A fragments are conflict-free ds_read_b128 at compile-time offsets per tap); the next
//                  chunk's quads are in flight while the 9 taps x 48 MFMAs of the current one run
//   weights        packed once per module by conv_pack_kernel into the B-fragment order [k-step][n block][piece][lane][8]: a
//                  fragment is one 16-B load per lane, 1 KB contiguous per wave, served by L2 (2.6 MB for 384 -> 128 channels)
//   epilogue       + bias, optional ReLU, every accumulator register is a 128-B row of 32 consecutive output channels of one pixel
#include "common.h"

namespace {

constexpr int CV_TH = 8, CV_TW = 32;   // output pixel tile
constexpr int CV_PXB = 48;             // LDS bytes per pixel and piece: 16 channels x bf16 + 16 B pad (conflict-free b128 reads)
constexpr int CV_MAXSRC = 4, CV_MAXNT = 4;

struct ConvSrcs {
    const float *p[CV_MAXSRC];
    int c[CV_MAXSRC];
    int n;
};
struct ConvDsts {          // one entry per N tile of 128 output channels
    float *p[CV_MAXNT];
    int stride[CV_MAXNT];  // channels of the destination tensor (pixel stride)
    int coff[CV_MAXNT];    // first channel of this tile inside it
    int nvalid[CV_MAXNT];  // output channels of this tile that exist (<= 128)
};

// weights [Cout][Cin][KS][KS] fp32 -> packed[(((chunk * TAPS + tap) * NB + nb) * 3 + piece) * 64 + lane][8] bf16, where lane l of
// n block nb holds output channel 32 nb + (l & 31), input channels 16 chunk + 8 (l >> 5) + 0..7 (the MFMA's B fragment)
__global__ void conv_pack_kernel(const float *__restrict__ w, __bf16 *__restrict__ packed, int Cout, int Cin, int taps, int NB)
{
    const long long total = (long long)(Cin / 16) * taps * NB * 64;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i % 64);
    const int nb = (int)((i / 64) % NB);
    const int tap = (int)((i / 64 / NB) % taps);
    const int chunk = (int)(i / 64 / NB / taps);
    const int n = 32 * nb + (lane & 31);
    bf16x8 pc[3];
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = 16 * chunk + 8 * (lane >> 5) + j;
        x[j] = n < Cout ? w[((size_t)n * Cin + c) * taps + tap] : 0.0f;
    }
    split_oct<3>(x, pc);
    const size_t base = ((((size_t)chunk * taps + tap) * NB + nb) * 3) * 64 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8 *>(packed + (base + (size_t)p * 64) * 8) = pc[p];
}

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(ConvSrcs src, const __bf16 *__restrict__ packed,
                                                         const float *__restrict__ bias, ConvDsts dst, int relu, int H, int W,
                                                         int tiles_x, int tiles_y, int NB, int Cout)
{
    constexpr int HALO = KS / 2, TAPS = KS * KS;
    constexpr int HCOLS = CV_TW + 2 * HALO, HROWS = CV_TH + 2 * HALO, NPX = HROWS * HCOLS;
    constexpr int PS = NPX * CV_PXB;             // piece stride (bytes)
    constexpr int NQ = NPX * 4, NPASS = (NQ + 255) / 256;
    __shared__ __attribute__((aligned(16))) char img[3 * PS];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int nt = blockIdx.y;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int x0 = tx * CV_TW, y0 = ty * CV_TH;

    // halo-tile quads of this thread: pixel index inside the image (or -1) and LDS byte offset
    int pix[NPASS], lw[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int q = tid + 256 * i;
        const int px = q >> 2, cq = q & 3;
        const int hy = px / HCOLS, hx = px - hy * HCOLS;
        const int y = y0 - HALO + hy, x = x0 - HALO + hx;
        pix[i] = (q < NQ && y >= 0 && y < H && x >= 0 && x < W) ? y * W + x : -1;
        lw[i] = q < NQ ? px * CV_PXB + cq * 8 : -1;
    }
    f32x4 rk[NPASS];
    int is_s = 0, is_c = 0;  // source / first channel of the chunk to issue next
    auto issue = [&]() {
        const int Cs = src.c[is_s];
        const __amdgpu_buffer_rsrc_t rs = buffer_rsrc(src.p[is_s] + (size_t)b * H * W * Cs, (unsigned)H * (unsigned)W * (unsigned)Cs * 4u);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            // 0x80000000: beyond any image of < 2 GB (launcher), whatever the chunk offset added to it
            const int voff = pix[i] >= 0 ? (pix[i] * Cs + is_c + 4 * (tid & 3)) * 4 : (int)0x80000000;
            rk[i] = BufLd<float>::quad(rs, voff, 0);
        }
        is_c += 16;
        if (is_c >= Cs) is_c = 0, ++is_s;
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if (lw[i] >= 0) {
                bf16x2 a[3], c[3];
                split_pair<3>((f32x2){rk[i][0], rk[i][1]}, a);
                split_pair<3>((f32x2){rk[i][2], rk[i][3]}, c);
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x4 *>(img + p * PS + lw[i]) = cat2(a[p], c[p]);
            }
        }
    };

    int nchunks = 0;
#pragma unroll
    for (int s = 0; s < CV_MAXSRC; ++s)
        if (s < src.n) nchunks += src.c[s] / 16;

    f32x16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

    // A fragment of (M block mb, tap (ky, kx), piece p): pixel (4 wm + mb + ky, l31 + kx) of the halo tile, channels 8 half ..
    const char *abase = img + ((4 * wm) * HCOLS + l31) * CV_PXB + half * 16;
    // B fragments of this wave: n blocks 4 nt + 2 wn + {0, 1}
    const __bf16 *wbase = packed + ((size_t)(4 * nt + 2 * wn) * 3 * 64 + lane) * 8;
    const size_t kstep = (size_t)NB * 3 * 64 * 8;   // elements per k-step (one tap of one chunk)

    issue();
    for (int g = 0; g < nchunks; ++g) {
        commit();
        __syncthreads();   // the chunk's pieces are visible
        if (g + 1 < nchunks) issue();
        const __bf16 *wk = wbase + (size_t)g * TAPS * kstep;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            bf16x8 bf[2][3];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[nb][p] = *reinterpret_cast<const bf16x8 *>(wk + (size_t)tap * kstep + ((size_t)nb * 3 + p) * 64 * 8);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                if (y0 + 4 * wm + mb < H) {   // wave-uniform: rows below the image do no work
                    bf16x8 af[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        af[p] = *reinterpret_cast<const bf16x8 *>(abase + p * PS + ((mb + ky) * HCOLS + kx) * CV_PXB);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        f32x16 a = acc[mb][nb];   // smallest terms first
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[nb][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[nb][2], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[nb][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[nb][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[nb][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[nb][0], a, 0, 0, 0);
                        acc[mb][nb] = a;
                    }
                }
            }
        }
        __syncthreads();   // every wave is done with the chunk before the next commit overwrites it
    }

    // ---- epilogue: register r of lane (l31, half) = pixel x0 + (r & 3) + 8 (r >> 2) + 4 half, channel 64 wn + 32 nb + l31
    float *ob = dst.p[nt];
    const int ostride = dst.stride[nt], coff = dst.coff[nt], nvalid = dst.nvalid[nt];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int cl = 64 * wn + 32 * nb + l31;
        const int cg = 128 * nt + cl;
        const float bv = (bias != nullptr && cg < Cout) ? bias[cg] : 0.0f;
        if (cl < nvalid) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int y = y0 + 4 * wm + mb;
                if (y < H) {
                    float *orow = ob + ((size_t)b * H + y) * W * ostride + coff + cl;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = acc[mb][nb][r] + bv;
                        if (relu) v = fmaxf(v, 0.0f);
                        if (x < W) orow[(size_t)x * ostride] = v;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" long long stereo_conv_packed_bytes(int Cout, int Cin, int ksize)
{
    if (Cout < 1 || Cin < 16 || Cin % 16 != 0 || (ksize != 1 && ksize != 3)) return 0;
    const long long NB = 4LL * ((Cout + 127) / 128);
    return (long long)(Cin / 16) * ksize * ksize * NB * 3 * 64 * 8 * 2;
}

extern "C" int stereo_conv_pack_weights(const float *weight, void *packed, int Cout, int Cin, int ksize, void *stream)
{
    if (stereo_conv_packed_bytes(Cout, Cin, ksize) == 0) return STEREO_ERR_BAD_SHAPE;
    STEREO_REQUIRE_PTR(weight);
    STEREO_REQUIRE_PTR(packed);
    if (!aligned16(packed)) return STEREO_ERR_UNSUPPORTED;
    const int NB = 4 * ((Cout + 127) / 128), taps = ksize * ksize;
    const long long total = (long long)(Cin / 16) * taps * NB * 64;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), weight,
                       static_cast<__bf16 *>(packed), Cout, Cin, taps, NB);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_conv_nhwc_f32(const float *const *srcs, const int *src_channels, int nsrc, const void *packed,
                                    const float *bias, float *const *outs, const int *out_channels, const int *out_strides,
                                    int nout, int ksize, int relu, int B, int H, int W, void *stream)
{
    if (nsrc < 1 || nsrc > CV_MAXSRC || nout < 1 || nout > CV_MAXNT || (ksize != 1 && ksize != 3) || B < 0 || H < 0 || W < 0)
        return STEREO_ERR_BAD_SHAPE;
    if (srcs == nullptr || src_channels == nullptr || outs == nullptr || out_channels == nullptr) return STEREO_ERR_NULL_POINTER;
    STEREO_REQUIRE_PTR(packed);
    if (B == 0 || H == 0 || W == 0) return STEREO_OK;
    ConvSrcs s{};
    ConvDsts d{};
    s.n = nsrc;
    int Cin = 0;
    for (int i = 0; i < nsrc; ++i) {
        STEREO_REQUIRE_PTR(srcs[i]);
        if (src_channels[i] < 16 || src_channels[i] % 16 != 0) return STEREO_ERR_UNSUPPORTED;
        if (!aligned16(srcs[i]) || (long long)H * W * src_channels[i] * 4 >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
        s.p[i] = srcs[i];
        s.c[i] = src_channels[i];
        Cin += src_channels[i];
    }
    // destinations: every one but the last must fill whole 128-channel N tiles
    int Cout = 0, ntn = 0;
    for (int i = 0; i < nout; ++i) {
        STEREO_REQUIRE_PTR(outs[i]);
        const int c = out_channels[i];
        const int cs = out_strides ? out_strides[i] : c;   // channels of the destination tensor (>= c: the rest is the caller's)
        if (c < 1 || cs < c || (i + 1 < nout && c % 128 != 0)) return STEREO_ERR_UNSUPPORTED;
        for (int o = 0; o < c; o += 128) {
            if (ntn >= CV_MAXNT) return STEREO_ERR_UNSUPPORTED;
            d.p[ntn] = outs[i];
            d.stride[ntn] = cs;
            d.coff[ntn] = o;
            d.nvalid[ntn] = c - o < 128 ? c - o : 128;
            ++ntn;
        }
        Cout += c;
    }
    const int tiles_x = (W + CV_TW - 1) / CV_TW, tiles_y = (H + CV_TH - 1) / CV_TH;
    const long long nt = (long long)tiles_x * tiles_y * B;
    if (nt >= (1LL << 31)) return STEREO_ERR_UNSUPPORTED;
    const int NB = 4 * ((Cout + 127) / 128);
    (void)Cin;
    dim3 grid((unsigned)nt, (unsigned)ntn);
    if (ksize == 3)
        hipLaunchKernelGGL(conv_split_kernel<3>, grid, dim3(256), 0, as_stream(stream), s, static_cast<const __bf16 *>(packed), bias,
                           d, relu, H, W, tiles_x, tiles_y, NB, Cout);
    else
        hipLaunchKernelGGL(conv_split_kernel<1>, grid, dim3(256), 0, as_stream(stream), s, static_cast<const __bf16 *>(packed), bias,
                           d, relu, H, W, tiles_x, tiles_y, NB, Cout);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
