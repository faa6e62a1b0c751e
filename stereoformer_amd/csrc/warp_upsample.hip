// disp_warp (A12), convex upsampling (A13), bilinear resize and the general
// bilinear_sampler.  All are gather / scatter-free streaming kernels: lanes run
// along W so every load and store of a wave is a contiguous segment.
#include "common.h"

namespace {

inline unsigned grid_for(long long nthreads)
{
    long long blocks = (nthreads + 255) / 256;
    const long long cap = 256LL * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// ------------------------------------------------------------------ A12
// utils/disparity_warper.py:83-106: grid (x-disp, y) normalised with (w-1),(h-1)
// (:55-56) then grid_sample with the default align_corners=False (:100).
__global__ __launch_bounds__(256) void disp_warp_kernel(const float *__restrict__ img, const float *__restrict__ disp,
                                                        float *__restrict__ warped, float *__restrict__ mask, int Ci,
                                                        int H, int W, long long npix, int padding_border)
{
    const long long HW = (long long)H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npix;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float ux, uy;
        disp_warp_coords(x, y, disp[t], W, H, ux, uy);
        if (mask) {  // ones sampled with zeros padding, thresholded (:102-105)
            const float fx = floorf(ux), fy = floorf(uy);
            const float ax = ux - fx, ay = uy - fy;
            float msum = 0.0f;
#pragma unroll
            for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                for (int jx = 0; jx < 2; ++jx) {
                    const float yy = fy + (float)jy, xx = fx + (float)jx;
                    if (yy >= 0.0f && yy <= (float)(H - 1) && xx >= 0.0f && xx <= (float)(W - 1))
                        msum += (jx ? ax : 1.0f - ax) * (jy ? ay : 1.0f - ay);
                }
            float mv = msum;
            if (mv < 0.9999f) mv = 0.0f;
            if (mv > 0.0f) mv = 1.0f;
            for (int c = 0; c < Ci; ++c) mask[((size_t)b * Ci + c) * HW + p] = mv;
        }
        float sx = ux, sy = uy;
        if (padding_border) {
            sx = clampf(sx, 0.0f, (float)(W - 1));
            sy = clampf(sy, 0.0f, (float)(H - 1));
        }
        const float fx = floorf(sx), fy = floorf(sy);
        const float ax = sx - fx, ay = sy - fy;
        const int ix = (int)fx, iy = (int)fy;
        const bool x0 = ix >= 0 && ix <= W - 1, x1 = ix + 1 >= 0 && ix + 1 <= W - 1;
        const bool y0 = iy >= 0 && iy <= H - 1, y1 = iy + 1 >= 0 && iy + 1 <= H - 1;
        const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
        for (int c = 0; c < Ci; ++c) {
            const float *im = img + ((size_t)b * Ci + c) * HW;
            float s = 0.0f;
            if (y0 && x0) s += im[(size_t)iy * W + ix] * w00;
            if (y0 && x1) s += im[(size_t)iy * W + ix + 1] * w01;
            if (y1 && x0) s += im[(size_t)(iy + 1) * W + ix] * w10;
            if (y1 && x1) s += im[(size_t)(iy + 1) * W + ix + 1] * w11;
            warped[((size_t)b * Ci + c) * HW + p] = s;
        }
    }
}

// ------------------------------------------------------------------ A13
// upsample_convex8 (models/utils/upsample.py:22-38) / RAFT upsample_flow
// (raft_stereo.py:110-122).  One workgroup = one coarse row segment of XT
// coarse pixels: mask planes are read with lanes along x (contiguous), the
// f x (XT*f) output tile is assembled in LDS and written as whole W-contiguous
// rows (16 B/lane).
constexpr int XT = 64;

template <int F, int CDT>
__global__ __launch_bounds__(256) void upsample_convex_kernel(const float *__restrict__ disp,
                                                              const float *__restrict__ mask, float *__restrict__ out,
                                                              int Cd, int H, int W)
{
    constexpr int F2 = F * F;
    __shared__ float nb[CDT][3][XT + 2];        // f * disp neighbourhood rows y-1..y+1, x0-1..x0+XT
    __shared__ float tile[CDT][F][XT * F + 4];  // output rows

    const int x0 = blockIdx.x * XT, y = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x;
    const size_t HW = (size_t)H * W;
    const int xl = tid % XT, sub0 = tid / XT;  // 4 sub-position lanes per pixel
    const int x = x0 + xl;

    for (int c0 = 0; c0 < Cd; c0 += CDT) {
        const int nc = (Cd - c0) < CDT ? (Cd - c0) : CDT;
        __syncthreads();
        for (int idx = tid; idx < nc * 3 * (XT + 2); idx += 256) {
            const int c = idx / (3 * (XT + 2)), rem = idx - c * 3 * (XT + 2);
            const int r = rem / (XT + 2), xx = x0 - 1 + (rem - r * (XT + 2)), yy = y - 1 + r;
            const float *dp = disp + ((size_t)n * Cd + c0 + c) * HW;
            nb[c][r][rem - r * (XT + 2)] =
                (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)F * dp[(size_t)yy * W + xx] : 0.0f;
        }
        __syncthreads();
        // ---- softmax over the 9 mask logits for every (i,j,x): lanes along x (contiguous loads)
        for (int ij = sub0; ij < F2; ij += 256 / XT) {
            const int i = ij / F, j = ij - i * F;
            float mk[9];
            if (x < W) {
                float mx = -INFINITY, s = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    mk[k] = mask[((size_t)n * 9 * F2 + (size_t)(k * F2 + ij)) * HW + (size_t)y * W + x];
                    mx = fmaxf(mx, mk[k]);
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    mk[k] = fast_exp(mk[k] - mx);
                    s += mk[k];
                }
                const float inv = 1.0f / s;
#pragma unroll
                for (int k = 0; k < 9; ++k) mk[k] *= inv;
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) mk[k] = 0.0f;
            }
#pragma unroll
            for (int c = 0; c < CDT; ++c) {
                if (c < nc) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc = fmaf(mk[k], nb[c][k / 3][xl + k % 3], acc);
                    tile[c][i][xl * F + j] = acc;
                }
            }
        }
        __syncthreads();
        // ---- write F rows of XT*F floats per channel
        const int rowlen = ((W - x0) < XT ? (W - x0) : XT) * F;
        for (int c = 0; c < nc; ++c) {
            float *op = out + (((size_t)n * Cd + c0 + c) * (size_t)(F * H) + (size_t)F * y) * (size_t)(F * W) +
                        (size_t)F * x0;
            for (int idx = tid; idx < F * (XT * F); idx += 256) {
                const int i = idx / (XT * F), col = idx - i * (XT * F);
                if (col < rowlen) op[(size_t)i * (F * W) + col] = tile[c][i][col];
            }
        }
    }
}

// The same with a CHANNELS-LAST mask, mask [N][H*W][9 F F] (what the update blocks' 1x1 mask projection writes on the
// convolution kernel): a pixel's 9 F F logits are contiguous, so the lanes of a wave run along the F F sub-positions of
// 64 / (F F) pixels and every mask load is one contiguous 256-byte piece; the softmax is per lane, the output rows leave
// through the same LDS tile.  Saves the NHWC -> NCHW copy of the mask (159 MB at [8,576,72,120]: 0.15 ms of a 5.2-ms loop).
template <int F, int CDT>
__global__ __launch_bounds__(256) void upsample_convex_nhwc_kernel(const float *__restrict__ disp, const float *__restrict__ mask,
                                                                   float *__restrict__ out, int Cd, int H, int W)
{
    constexpr int F2 = F * F, PPW = 64 / F2;     // pixels per wave and pass
    __shared__ float nb[CDT][3][XT + 2];
    __shared__ float tile[CDT][F][XT * F + 4];
    const int x0 = blockIdx.x * XT, y = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t HW = (size_t)H * W;
    const int ij = lane % F2, psub = lane / F2, i = ij / F, j = ij - i * F;
    for (int c0 = 0; c0 < Cd; c0 += CDT) {
        const int nc = (Cd - c0) < CDT ? (Cd - c0) : CDT;
        __syncthreads();
        for (int idx = tid; idx < nc * 3 * (XT + 2); idx += 256) {
            const int c = idx / (3 * (XT + 2)), rem = idx - c * 3 * (XT + 2);
            const int r = rem / (XT + 2), xx = x0 - 1 + (rem - r * (XT + 2)), yy = y - 1 + r;
            const float *dp = disp + ((size_t)n * Cd + c0 + c) * HW;
            nb[c][r][rem - r * (XT + 2)] =
                (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)F * dp[(size_t)yy * W + xx] : 0.0f;
        }
        __syncthreads();
        for (int xl = PPW * wave + psub; xl < XT; xl += 4 * PPW) {   // wave-uniform trip count
            const int x = x0 + xl;
            float mk[9];
            if (x < W) {
                const float *mp = mask + ((size_t)n * HW + (size_t)y * W + x) * (9 * F2) + ij;
                float mx = -INFINITY, s = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    mk[k] = mp[k * F2];
                    mx = fmaxf(mx, mk[k]);
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    mk[k] = fast_exp(mk[k] - mx);
                    s += mk[k];
                }
                const float inv = 1.0f / s;
#pragma unroll
                for (int k = 0; k < 9; ++k) mk[k] *= inv;
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) mk[k] = 0.0f;
            }
#pragma unroll
            for (int c = 0; c < CDT; ++c) {
                if (c < nc) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc = fmaf(mk[k], nb[c][k / 3][xl + k % 3], acc);
                    tile[c][i][xl * F + j] = acc;
                }
            }
        }
        __syncthreads();
        const int rowlen = ((W - x0) < XT ? (W - x0) : XT) * F;
        for (int c = 0; c < nc; ++c) {
            float *op = out + (((size_t)n * Cd + c0 + c) * (size_t)(F * H) + (size_t)F * y) * (size_t)(F * W) +
                        (size_t)F * x0;
            for (int idx = tid; idx < F * (XT * F); idx += 256) {
                const int ii = idx / (XT * F), col = idx - ii * (XT * F);
                if (col < rowlen) op[(size_t)ii * (F * W) + col] = tile[c][ii][col];
            }
        }
    }
}

// generic factor fallback (any f): one thread per output pixel
__global__ __launch_bounds__(256) void upsample_convex_generic_kernel(const float *__restrict__ disp,
                                                                      const float *__restrict__ mask,
                                                                      float *__restrict__ out, int Cd, int H, int W,
                                                                      int f, long long nout)
{
    const size_t HW = (size_t)H * W;
    const int f2 = f * f;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nout;
         t += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(t % ((long long)f * W));
        long long r = t / ((long long)f * W);
        const int Y = (int)(r % ((long long)f * H));
        r /= (long long)f * H;
        const int c = (int)(r % Cd), n = (int)(r / Cd);
        const int x = X / f, j = X - x * f, y = Y / f, i = Y - y * f;
        float mk[9], mx = -INFINITY, s = 0.0f;
        for (int k = 0; k < 9; ++k) {
            mk[k] = mask[((size_t)n * 9 * f2 + (size_t)(k * f2 + i * f + j)) * HW + (size_t)y * W + x];
            mx = fmaxf(mx, mk[k]);
        }
        for (int k = 0; k < 9; ++k) {
            mk[k] = fast_exp(mk[k] - mx);
            s += mk[k];
        }
        float acc = 0.0f;
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W)
                                ? (float)f * disp[((size_t)n * Cd + c) * HW + (size_t)yy * W + xx]
                                : 0.0f;
            acc = fmaf(mk[k] / s, v, acc);
        }
        out[t] = acc;
    }
}

// ------------------------------------------------------------------ resize
// F.interpolate(mode='bilinear', size=(Ho,Wo), align_corners=ac); out *= scale
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                              int Hi, int Wi, int Ho, int Wo, int align_corners,
                                                              float scale, float rh, float rw, long long nout)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nout;
         t += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(t % Wo);
        long long r = t / Wo;
        const int y = (int)(r % Ho);
        const long long nc = r / Ho;
        float sy, sx;
        if (align_corners) {
            sy = rh * (float)y;
            sx = rw * (float)x;
        } else {
            sy = __fsub_rn(__fmul_rn(rh, (float)y + 0.5f), 0.5f);
            sx = __fsub_rn(__fmul_rn(rw, (float)x + 0.5f), 0.5f);
            sy = sy < 0.0f ? 0.0f : sy;
            sx = sx < 0.0f ? 0.0f : sx;
        }
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 > Hi - 1 ? Hi - 1 : y0;
        x0 = x0 > Wi - 1 ? Wi - 1 : x0;
        const int y1 = y0 < Hi - 1 ? y0 + 1 : y0, x1 = x0 < Wi - 1 ? x0 + 1 : x0;
        const float ay = sy - (float)y0, ax = sx - (float)x0;
        const float *im = in + (size_t)nc * Hi * Wi;
        const float top = im[(size_t)y0 * Wi + x0] * (1.0f - ax) + im[(size_t)y0 * Wi + x1] * ax;
        const float bot = im[(size_t)y1 * Wi + x0] * (1.0f - ax) + im[(size_t)y1 * Wi + x1] * ax;
        out[t] = scale * (top * (1.0f - ay) + bot * ay);
    }
}

// ------------------------------------------------------------------ bilinear_sampler
// models/IterativeLocalCostVolume/utils/utils.py:59-76: pixel coords -> [-1,1]
// (y untouched when H == 1) -> grid_sample(align_corners=True, zeros padding)
__global__ __launch_bounds__(256) void bilinear_sampler_kernel(const float *__restrict__ img,
                                                               const float *__restrict__ coords,
                                                               float *__restrict__ out, float *__restrict__ mask,
                                                               int C, int H, int W, int Ho, int Wo, long long npts)
{
    const long long HoWo = (long long)Ho * Wo;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npts;
         t += (long long)gridDim.x * blockDim.x) {
        const long long n = t / HoWo, q = t - n * HoWo;
        const float cx = coords[t * 2 + 0], cy = coords[t * 2 + 1];
        const float xg = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, cx), (float)(W - 1)), 1.0f);
        const float yg = H > 1 ? __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, cy), (float)(H - 1)), 1.0f) : cy;
        const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(xg, 1.0f), 2.0f), (float)(W - 1));
        const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(yg, 1.0f), 2.0f), (float)(H - 1));
        const float fx = floorf(ix), fy = floorf(iy);
        const float ax = ix - fx, ay = iy - fy;
        const int x0 = (int)fx, y0 = (int)fy;
        const bool vx0 = x0 >= 0 && x0 <= W - 1, vx1 = x0 + 1 >= 0 && x0 + 1 <= W - 1;
        const bool vy0 = y0 >= 0 && y0 <= H - 1, vy1 = y0 + 1 >= 0 && y0 + 1 <= H - 1;
        const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
        for (int c = 0; c < C; ++c) {
            const float *im = img + ((size_t)n * C + c) * (size_t)H * W;
            float s = 0.0f;
            if (vy0 && vx0) s += im[(size_t)y0 * W + x0] * w00;
            if (vy0 && vx1) s += im[(size_t)y0 * W + x0 + 1] * w01;
            if (vy1 && vx0) s += im[(size_t)(y0 + 1) * W + x0] * w10;
            if (vy1 && vx1) s += im[(size_t)(y0 + 1) * W + x0 + 1] * w11;
            out[((size_t)n * C + c) * HoWo + q] = s;
        }
        if (mask) mask[t] = (xg > -1.0f && yg > -1.0f && xg < 1.0f && yg < 1.0f) ? 1.0f : 0.0f;
    }
}

}  // namespace

extern "C" int stereo_disp_warp_f32(const float *img, const float *disp, float *warped, float *mask, int B, int Ci,
                                    int H, int W, int padding_border, void *stream)
{
    if (B < 0 || Ci < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || Ci == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(img);
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(warped);
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(disp_warp_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), img, disp, warped, mask,
                       Ci, H, W, n, padding_border);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_upsample_convex_f32(const float *disp, const float *mask, float *out, int N, int Cd, int H, int W,
                                          int f, void *stream)
{
    if (N < 0 || Cd < 0 || H < 0 || W < 0 || f <= 0) return STEREO_ERR_BAD_SHAPE;
    if (N == 0 || Cd == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(mask);
    STEREO_REQUIRE_PTR(out);
    hipStream_t st = as_stream(stream);
    const bool tiled = fits_grid_yz(H) && fits_grid_yz(N);
    dim3 grid((W + XT - 1) / XT, H, N);
    if (tiled && f == 8 && Cd == 1) {
        hipLaunchKernelGGL((upsample_convex_kernel<8, 1>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    } else if (tiled && f == 8) {
        hipLaunchKernelGGL((upsample_convex_kernel<8, 2>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    } else if (tiled && f == 4 && Cd == 1) {
        hipLaunchKernelGGL((upsample_convex_kernel<4, 1>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    } else if (tiled && f == 4) {
        hipLaunchKernelGGL((upsample_convex_kernel<4, 2>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    } else {
        const long long nout = (long long)N * Cd * f * H * f * W;
        hipLaunchKernelGGL(upsample_convex_generic_kernel, dim3(grid_for(nout)), dim3(256), 0, st, disp, mask, out, Cd,
                           H, W, f, nout);
    }
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_upsample_convex_nhwc_mask_f32(const float *disp, const float *mask, float *out, int N, int Cd, int H, int W,
                                                    int f, void *stream)
{
    if (N < 0 || Cd < 0 || H < 0 || W < 0 || f <= 0) return STEREO_ERR_BAD_SHAPE;
    if (N == 0 || Cd == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(disp);
    STEREO_REQUIRE_PTR(mask);
    STEREO_REQUIRE_PTR(out);
    if (!fits_grid_yz(H) || !fits_grid_yz(N) || (f != 8 && f != 4)) return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    dim3 grid((W + XT - 1) / XT, H, N);
    if (f == 8 && Cd == 1) hipLaunchKernelGGL((upsample_convex_nhwc_kernel<8, 1>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    else if (f == 8) hipLaunchKernelGGL((upsample_convex_nhwc_kernel<8, 2>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    else if (Cd == 1) hipLaunchKernelGGL((upsample_convex_nhwc_kernel<4, 1>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    else hipLaunchKernelGGL((upsample_convex_nhwc_kernel<4, 2>), grid, dim3(256), 0, st, disp, mask, out, Cd, H, W);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_resize_bilinear_f32(const float *in, float *out, int NC, int Hi, int Wi, int Ho, int Wo,
                                          int align_corners, float scale, void *stream)
{
    if (NC < 0 || Hi < 0 || Wi < 0 || Ho < 0 || Wo < 0) return STEREO_ERR_BAD_SHAPE;
    if (NC == 0 || Ho == 0 || Wo == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(in);
    STEREO_REQUIRE_PTR(out);
    if (Hi == 0 || Wi == 0) return STEREO_ERR_BAD_SHAPE;
    float rh, rw;
    if (align_corners) {
        rh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f;
        rw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f;
    } else {
        rh = (float)Hi / (float)Ho;
        rw = (float)Wi / (float)Wo;
    }
    const long long nout = (long long)NC * Ho * Wo;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(nout)), dim3(256), 0, as_stream(stream), in, out, Hi, Wi,
                       Ho, Wo, align_corners, scale, rh, rw, nout);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

// The same on channels-last tensors, in [N][Hi][Wi][C] -> out [N][Ho][Wo][C] (C % 4 == 0): a thread owns four channels of one
// output pixel (16-B loads of the four taps); the same formula per element.
__global__ __launch_bounds__(256) void resize_bilinear_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int C4,
                                                                   int Hi, int Wi, int Ho, int Wo, int align_corners,
                                                                   float scale, float rh, float rw, long long nout)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nout;
         t += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(t % C4);
        long long r = t / C4;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const long long n = r / Ho;
        float sy, sx;
        if (align_corners) {
            sy = rh * (float)y;
            sx = rw * (float)x;
        } else {
            sy = __fsub_rn(__fmul_rn(rh, (float)y + 0.5f), 0.5f);
            sx = __fsub_rn(__fmul_rn(rw, (float)x + 0.5f), 0.5f);
            sy = sy < 0.0f ? 0.0f : sy;
            sx = sx < 0.0f ? 0.0f : sx;
        }
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 > Hi - 1 ? Hi - 1 : y0;
        x0 = x0 > Wi - 1 ? Wi - 1 : x0;
        const int y1 = y0 < Hi - 1 ? y0 + 1 : y0, x1 = x0 < Wi - 1 ? x0 + 1 : x0;
        const float ay = sy - (float)y0, ax = sx - (float)x0;
        const float *im = in + (size_t)n * Hi * Wi * 4 * C4 + 4 * cq;
        const f32x4 a = ld4(im + ((size_t)y0 * Wi + x0) * 4 * C4), b = ld4(im + ((size_t)y0 * Wi + x1) * 4 * C4);
        const f32x4 c = ld4(im + ((size_t)y1 * Wi + x0) * 4 * C4), d = ld4(im + ((size_t)y1 * Wi + x1) * 4 * C4);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float top = a[i] * (1.0f - ax) + b[i] * ax;
            const float bot = c[i] * (1.0f - ax) + d[i] * ax;
            o[i] = scale * (top * (1.0f - ay) + bot * ay);
        }
        st4(out + t * 4, o);
    }
}

extern "C" int stereo_resize_bilinear_nhwc_f32(const float *in, float *out, int N, int C, int Hi, int Wi, int Ho, int Wo,
                                               int align_corners, float scale, void *stream)
{
    if (N < 0 || C < 0 || Hi < 0 || Wi < 0 || Ho < 0 || Wo < 0) return STEREO_ERR_BAD_SHAPE;
    if (N == 0 || C == 0 || Ho == 0 || Wo == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(in);
    STEREO_REQUIRE_PTR(out);
    if (Hi == 0 || Wi == 0) return STEREO_ERR_BAD_SHAPE;
    if (C % 4 != 0 || !aligned16(in) || !aligned16(out)) return STEREO_ERR_UNSUPPORTED;
    float rh, rw;
    if (align_corners) {
        rh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f;
        rw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f;
    } else {
        rh = (float)Hi / (float)Ho;
        rw = (float)Wi / (float)Wo;
    }
    const long long nout = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(resize_bilinear_nhwc_kernel, dim3(grid_for(nout)), dim3(256), 0, as_stream(stream), in, out, C / 4, Hi, Wi,
                       Ho, Wo, align_corners, scale, rh, rw, nout);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

// pool2x / pool4x of the update block (update/update.py:119-123: F.avg_pool2d(x, 3, stride=2, padding=1) / (x, 5, stride=4,
// padding=1): zero padding, count_include_pad -- the divisor is always k*k) on channels-last tensors: a thread owns four
// channels of one output pixel (16-B loads of its k x k taps, one 16-B store).  The framework's NHWC pooling kernel takes
// 0.14 ms for [8,128,136,240] -> [8,128,68,120]; this one runs at the read rate of the input.
__global__ __launch_bounds__(256) void avg_pool_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int C4, int Hi,
                                                            int Wi, int Ho, int Wo, int k, int stride, int pad, float inv,
                                                            long long nout)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nout; t += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(t % C4);
        long long r = t / C4;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const long long n = r / Ho;
        const float *im = in + (size_t)n * Hi * Wi * 4 * C4 + 4 * cq;
        f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int ky = 0; ky < k; ++ky) {
            const int yy = y * stride - pad + ky;
            if (yy < 0 || yy >= Hi) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = x * stride - pad + kx;
                if (xx >= 0 && xx < Wi) s += ld4(im + ((size_t)yy * Wi + xx) * 4 * C4);
            }
        }
        st4(out + t * 4, s * inv);
    }
}

extern "C" int stereo_avg_pool_nhwc_f32(const float *in, float *out, int N, int C, int Hi, int Wi, int ksize, int stride, int pad,
                                        void *stream)
{
    if (N < 0 || C < 0 || Hi < 0 || Wi < 0 || ksize < 1 || stride < 1 || pad < 0 || 2 * pad > ksize) return STEREO_ERR_BAD_SHAPE;
    const int Ho = (Hi + 2 * pad - ksize) / stride + 1, Wo = (Wi + 2 * pad - ksize) / stride + 1;   // floor mode
    if (Hi + 2 * pad < ksize || Wi + 2 * pad < ksize) return STEREO_ERR_BAD_SHAPE;
    if (N == 0 || C == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(in);
    STEREO_REQUIRE_PTR(out);
    if (C % 4 != 0 || !aligned16(in) || !aligned16(out)) return STEREO_ERR_UNSUPPORTED;
    const long long nout = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(avg_pool_nhwc_kernel, dim3(grid_for(nout)), dim3(256), 0, as_stream(stream), in, out, C / 4, Hi, Wi, Ho, Wo,
                       ksize, stride, pad, 1.0f / (float)(ksize * ksize), nout);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_bilinear_sampler_f32(const float *img, const float *coords, float *out, float *mask, int N, int C,
                                           int H, int W, int Ho, int Wo, void *stream)
{
    if (N < 0 || C < 0 || H < 0 || W < 0 || Ho < 0 || Wo < 0) return STEREO_ERR_BAD_SHAPE;
    if (N == 0 || Ho == 0 || Wo == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(img);
    STEREO_REQUIRE_PTR(coords);
    STEREO_REQUIRE_PTR(out);
    if (H == 0 || W == 0) return STEREO_ERR_BAD_SHAPE;
    const long long npts = (long long)N * Ho * Wo;
    hipLaunchKernelGGL(bilinear_sampler_kernel, dim3(grid_for(npts)), dim3(256), 0, as_stream(stream), img, coords, out,
                       mask, C, H, W, Ho, Wo, npts);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}
