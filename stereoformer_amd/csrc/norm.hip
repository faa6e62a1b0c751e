// Extractor epilogue for the feature hand-over: InstanceNorm2d (affine optional, eps inside the sqrt,
// biased variance over H x W -- what nn.InstanceNorm2d computes in eval and in training, extractor.py:135,160-162) fused
// with the ReLU that follows it (extractor.py:187-188), written in the layout / precision the cost-volume loaders take:
// fp32 or fp16, NCHW.  HBM-bound: the plane statistics need one read of x, the normalisation a second one (a 540 x 960
// plane is 2 MB, far beyond LDS): 8 B read + 2 or 4 B written per element.
//
//   pass 1  stats_kernel: grid (chunks, B*C); a workgroup sums its chunk of one plane with 16-B loads (fp32 partials of
//           <= 32 quads per lane, combined in fp64) and adds (sum, sum of squares) to the plane's two fp64 accumulators
//           with global_atomic_add_f64 -- the workspace is caller-owned and zeroed on the stream by the launcher.
//   pass 2  apply_kernel: y = max((x - mean) * rstd * gamma + beta, 0) with mean / rstd derived in fp64 per workgroup.
#include "common.h"

namespace {

constexpr int IN_THREADS = 256;
constexpr int IN_QUADS = 32;  // 16-B quads per lane and workgroup chunk (32 KiB of x per wave)

__global__ __launch_bounds__(IN_THREADS) void instance_norm_stats_kernel(const float *__restrict__ x,
                                                                          double *__restrict__ stats, long long HW)
{
    const long long plane = blockIdx.y;
    const float *xp = x + plane * HW;
    const long long q0 = (long long)blockIdx.x * IN_THREADS * IN_QUADS;  // first quad of this chunk
    const long long nq = HW / 4;
    float s = 0.0f, ss = 0.0f;
#pragma unroll 8
    for (int k = 0; k < IN_QUADS; ++k) {
        const long long q = q0 + (long long)k * IN_THREADS + threadIdx.x;
        if (q < nq) {
            const f32x4 v = ld4(xp + 4 * q);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
    if (blockIdx.x == gridDim.x - 1) {  // ragged tail of the plane (HW % 4 elements)
        const long long t = 4 * nq + threadIdx.x;
        if (t < HW) {
            const float v = xp[t];
            s += v;
            ss += v * v;
        }
    }
    double ds = (double)s, dss = (double)ss;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ds += __shfl_xor(ds, off);
        dss += __shfl_xor(dss, off);
    }
    __shared__ double red[2][IN_THREADS / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[0][wave] = ds, red[1][wave] = dss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < IN_THREADS / 64; ++w) a += red[0][w], b += red[1][w];
        atomicAdd(stats + 2 * plane, a);
        atomicAdd(stats + 2 * plane + 1, b);
    }
}

template <typename TOUT>
__device__ __forceinline__ void store_quad(TOUT *p, f32x4 v);
template <>
__device__ __forceinline__ void store_quad<float>(float *p, f32x4 v) { st4(p, v); }
template <>
__device__ __forceinline__ void store_quad<f16>(f16 *p, f32x4 v)
{
    *reinterpret_cast<f16x4 *>(p) = __builtin_convertvector(v, f16x4);  // round to nearest even
}

template <typename TOUT>
__global__ __launch_bounds__(IN_THREADS) void instance_norm_apply_kernel(const float *__restrict__ x,
                                                                          const double *__restrict__ stats,
                                                                          const float *__restrict__ gamma,
                                                                          const float *__restrict__ beta,
                                                                          TOUT *__restrict__ out, int C, long long HW,
                                                                          float eps, int relu)
{
    const long long plane = blockIdx.y;
    const int c = (int)(plane % C);
    const double mean = stats[2 * plane] / (double)HW;
    double var = stats[2 * plane + 1] / (double)HW - mean * mean;  // biased, like F.instance_norm
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] * rstd : rstd;
    const float sh = (beta ? beta[c] : 0.0f) - (float)mean * g;  // y = x * g + sh
    const float *xp = x + plane * HW;
    TOUT *op = out + plane * HW;
    const long long q0 = (long long)blockIdx.x * IN_THREADS * IN_QUADS;
    const long long nq = HW / 4;
    const float lo = relu ? 0.0f : -INFINITY;
#pragma unroll 8
    for (int k = 0; k < IN_QUADS; ++k) {
        const long long q = q0 + (long long)k * IN_THREADS + threadIdx.x;
        if (q < nq) {
            const f32x4 v = ld4(xp + 4 * q);
            f32x4 y;
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = fmaxf(fmaf(v[i], g, sh), lo);
            store_quad<TOUT>(op + 4 * q, y);
        }
    }
    if (blockIdx.x == gridDim.x - 1) {
        const long long t = 4 * nq + threadIdx.x;
        if (t < HW) op[t] = (TOUT)fmaxf(fmaf(xp[t], g, sh), lo);
    }
}

// ---- channels-last input (what MIOpen's convolutions hand over when the extractor runs in torch.channels_last): x is
// [B][H*W][C]; the output stays NCHW, so the transpose the cost-volume loaders would otherwise need as a pass of its own rides
// on the normalisation's second read.  C % 4 == 0 and 256 % (C / 4) == 0 (C = 4 .. 256 in powers of two, or 12 .. ): a thread
// keeps ONE channel quad for its whole chunk.
constexpr int INH_PASSES = 64;  // pixels per thread and chunk in the statistics pass

template <typename TIN>
__global__ __launch_bounds__(IN_THREADS) void instance_norm_stats_nhwc_kernel(const TIN *__restrict__ x,
                                                                               double *__restrict__ stats, int C, long long HW)
{
    extern __shared__ double acc[];  // [2][C]
    const int b = blockIdx.y, CQ = C / 4, PP = IN_THREADS / CQ;
    const int cq = threadIdx.x % CQ, pi = threadIdx.x / CQ;
    for (int i = threadIdx.x; i < 2 * C; i += IN_THREADS) acc[i] = 0.0;
    __syncthreads();
    const TIN *xb = x + (size_t)b * HW * C + 4 * cq;
    const long long p0 = (long long)blockIdx.x * PP * INH_PASSES + pi;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < INH_PASSES; ++k) {
        const long long p = p0 + (long long)k * PP;
        if (p < HW) {
            const f32x4 v = Quad<TIN>::widen(Quad<TIN>::ld(xb + p * C));
            s += v;
            ss += v * v;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        atomicAdd(&acc[4 * cq + i], (double)s[i]);
        atomicAdd(&acc[C + 4 * cq + i], (double)ss[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += IN_THREADS) {
        const int c = i % C, which = i / C;
        atomicAdd(stats + 2 * ((size_t)b * C + c) + which, acc[i]);
    }
}

template <typename TIN, typename TOUT, int TP>
__global__ __launch_bounds__(IN_THREADS) void instance_norm_apply_nhwc_kernel(const TIN *__restrict__ x,
                                                                               const double *__restrict__ stats,
                                                                               const float *__restrict__ gamma,
                                                                               const float *__restrict__ beta,
                                                                               TOUT *__restrict__ out, int C, long long HW,
                                                                               float eps, int relu)
{
    extern __shared__ float tile[];  // [C][TP + 1] + g[C] + sh[C]
    float *gs = tile + (size_t)C * (TP + 1), *shs = gs + C;
    const int b = blockIdx.y, CQ = C / 4;
    for (int c = threadIdx.x; c < C; c += IN_THREADS) {
        if (stats == nullptr) {  // plain relayout (stereo_relayout_nhwc_to_nchw): y = x
            gs[c] = 1.0f;
            shs[c] = 0.0f;
            continue;
        }
        const size_t plane = (size_t)b * C + c;
        const double mean = stats[2 * plane] / (double)HW;
        double var = stats[2 * plane + 1] / (double)HW - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] * rstd : rstd;
        gs[c] = g;
        shs[c] = (beta ? beta[c] : 0.0f) - (float)mean * g;
    }
    __syncthreads();
    const long long p0 = (long long)blockIdx.x * TP;
    const float lo = relu ? 0.0f : -INFINITY;
    const TIN *xb = x + ((size_t)b * HW + p0) * C;
    for (int idx = threadIdx.x; idx < TP * CQ; idx += IN_THREADS) {   // coalesced along the channels of consecutive pixels
        const int p = idx / CQ, cq = idx % CQ;
        if (p0 + p < HW) {
            const f32x4 v = Quad<TIN>::widen(Quad<TIN>::ld(xb + (size_t)p * C + 4 * cq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                tile[(4 * cq + i) * (TP + 1) + p] = fmaxf(fmaf(v[i], gs[4 * cq + i], shs[4 * cq + i]), lo);
        }
    }
    __syncthreads();
    TOUT *ob = out + (size_t)b * C * HW + p0;
    for (int idx = threadIdx.x; idx < C * (TP / 4); idx += IN_THREADS) {  // rows of TP pixels per channel
        const int c = idx / (TP / 4), pq = idx % (TP / 4);
        if (p0 + 4 * pq + 3 < HW) {
            const float *t = tile + c * (TP + 1) + 4 * pq;
            store_quad<TOUT>(ob + (size_t)c * HW + 4 * pq, (f32x4){t[0], t[1], t[2], t[3]});
        }
    }
}

template <typename TIN, typename TOUT>
int launch_nhwc(const TIN *x, const double *stats, const float *gamma, const float *beta, TOUT *out, int B, int C, long long HW,
                float eps, int relu, hipStream_t st)
{
#define APPLY(TPV)                                                                                                         \
    hipLaunchKernelGGL((instance_norm_apply_nhwc_kernel<TIN, TOUT, TPV>), dim3((unsigned)((HW + TPV - 1) / TPV), (unsigned)B), \
                       dim3(IN_THREADS), ((size_t)C * (TPV + 1) + 2 * (size_t)C) * sizeof(float), st, x, stats, gamma, beta, out, \
                       C, HW, eps, relu)
    if (C <= 64) APPLY(128);
    else if (C <= 128) APPLY(64);
    else APPLY(32);
#undef APPLY
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace

extern "C" long long stereo_instance_norm_workspace_bytes(int B, int C)
{
    return B > 0 && C > 0 ? (long long)B * C * 2 * (long long)sizeof(double) : 0;
}

extern "C" int stereo_instance_norm_relu_f32(const float *x, const float *gamma, const float *beta, void *out,
                                             void *workspace, int out_is_f16, int B, int C, int H, int W, float eps,
                                             int relu, void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || !(eps >= 0.0f)) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || C == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(x);
    STEREO_REQUIRE_PTR(out);
    STEREO_REQUIRE_PTR(workspace);
    const long long HW = (long long)H * W, planes = (long long)B * C;
    // 16-B loads per plane need HW % 4 == 0 for every plane base to stay aligned; 8-B fp16 stores likewise
    if (!aligned16(x) || HW % 4 != 0 || !fits_grid_yz(planes) || (reinterpret_cast<uintptr_t>(workspace) & 7u) != 0 ||
        (out_is_f16 ? (reinterpret_cast<uintptr_t>(out) & 7u) != 0 : !aligned16(out)))
        return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    double *stats = static_cast<double *>(workspace);
    if (hipMemsetAsync(stats, 0, (size_t)planes * 2 * sizeof(double), st) != hipSuccess) return STEREO_ERR_LAUNCH;
    const long long per_wg = (long long)IN_THREADS * IN_QUADS * 4;
    const unsigned chunks = (unsigned)((HW + per_wg - 1) / per_wg);
    dim3 grid(chunks, (unsigned)planes);
    hipLaunchKernelGGL(instance_norm_stats_kernel, grid, dim3(IN_THREADS), 0, st, x, stats, HW);
    STEREO_LAUNCH_CHECK();
    if (out_is_f16)
        hipLaunchKernelGGL(instance_norm_apply_kernel<f16>, grid, dim3(IN_THREADS), 0, st, x, stats, gamma, beta,
                           static_cast<f16 *>(out), C, HW, eps, relu);
    else
        hipLaunchKernelGGL(instance_norm_apply_kernel<float>, grid, dim3(IN_THREADS), 0, st, x, stats, gamma, beta,
                           static_cast<float *>(out), C, HW, eps, relu);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

extern "C" int stereo_instance_norm_relu_nhwc(const void *x, int x_is_f16, const float *gamma, const float *beta, void *out,
                                              void *workspace, int out_is_f16, int B, int C, int H, int W, float eps, int relu,
                                              void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || !(eps >= 0.0f)) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || C == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(x);
    STEREO_REQUIRE_PTR(out);
    STEREO_REQUIRE_PTR(workspace);
    const long long HW = (long long)H * W;
    const uintptr_t xa = x_is_f16 ? 7u : 15u;
    if (C % 4 != 0 || C > 256 || IN_THREADS % (C / 4) != 0 || HW % 4 != 0 || !fits_grid_yz(B) ||
        (reinterpret_cast<uintptr_t>(x) & xa) != 0 || (reinterpret_cast<uintptr_t>(workspace) & 7u) != 0 ||
        (out_is_f16 ? (reinterpret_cast<uintptr_t>(out) & 7u) != 0 : !aligned16(out)) || HW / 32 >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    double *stats = static_cast<double *>(workspace);
    if (hipMemsetAsync(stats, 0, (size_t)B * C * 2 * sizeof(double), st) != hipSuccess) return STEREO_ERR_LAUNCH;
    const long long per_wg = (long long)(IN_THREADS / (C / 4)) * INH_PASSES;  // pixels per statistics workgroup
    dim3 sgrid((unsigned)((HW + per_wg - 1) / per_wg), (unsigned)B);
    const size_t slds = (size_t)2 * C * sizeof(double);
    if (x_is_f16)
        hipLaunchKernelGGL(instance_norm_stats_nhwc_kernel<f16>, sgrid, dim3(IN_THREADS), slds, st, static_cast<const f16 *>(x),
                           stats, C, HW);
    else
        hipLaunchKernelGGL(instance_norm_stats_nhwc_kernel<float>, sgrid, dim3(IN_THREADS), slds, st,
                           static_cast<const float *>(x), stats, C, HW);
    STEREO_LAUNCH_CHECK();
    if (x_is_f16)
        return out_is_f16 ? launch_nhwc<f16, f16>(static_cast<const f16 *>(x), stats, gamma, beta, static_cast<f16 *>(out), B, C,
                                                  HW, eps, relu, st)
                          : launch_nhwc<f16, float>(static_cast<const f16 *>(x), stats, gamma, beta, static_cast<float *>(out),
                                                    B, C, HW, eps, relu, st);
    return out_is_f16 ? launch_nhwc<float, f16>(static_cast<const float *>(x), stats, gamma, beta, static_cast<f16 *>(out), B, C,
                                                HW, eps, relu, st)
                      : launch_nhwc<float, float>(static_cast<const float *>(x), stats, gamma, beta, static_cast<float *>(out),
                                                  B, C, HW, eps, relu, st);
}

// NCHW -> channels-last relayout (the update loops take their hidden states and context tensors channels-last; the
// reference's extractors hand them over NCHW).  A workgroup transposes 32 channels x 64 pixels through LDS: 16-byte loads
// along the pixels of a channel plane, 16-byte stores along the channels of a pixel.
namespace {
__global__ __launch_bounds__(256) void relayout_nchw_to_nhwc_kernel(const float *__restrict__ x, float *__restrict__ out, int C,
                                                                    long long HW)
{
    __shared__ float tile[32][64 + 1];
    const int tid = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32, b = blockIdx.z;
    const float *xb = x + ((size_t)b * C + c0) * HW + p0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, c = q >> 4, pq = (q & 15) * 4;   // channel row, pixel quad
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p0 + pq < HW) v = ld4(xb + (size_t)c * HW + pq);          // HW % 4 == 0: a quad is entirely in or out
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[c][pq + e] = v[e];
    }
    __syncthreads();
    float *ob = out + ((size_t)b * HW + p0) * C + c0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, px = q >> 3, cq = (q & 7) * 4;   // pixel, channel quad
        if (p0 + px < HW) {
            const f32x4 v = {tile[cq][px], tile[cq + 1][px], tile[cq + 2][px], tile[cq + 3][px]};
            st4(ob + (size_t)px * C + cq, v);
        }
    }
}
}  // namespace

extern "C" int stereo_relayout_nchw_to_nhwc_f32(const float *x, float *out, int B, int C, int H, int W, void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || C == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(x);
    STEREO_REQUIRE_PTR(out);
    const long long HW = (long long)H * W;
    if (C % 32 != 0 || HW % 4 != 0 || !fits_grid_yz(B) || !fits_grid_yz(C / 32) || !aligned16(x) || !aligned16(out) ||
        (HW + 63) / 64 >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(relayout_nchw_to_nhwc_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)(C / 32), (unsigned)B), dim3(256), 0,
                       as_stream(stream), x, out, C, HW);
    STEREO_LAUNCH_CHECK();
    return STEREO_OK;
}

/* NHWC -> NCHW relayout of a feature tensor (the apply pass above with the identity map): what a channels-last extractor
 * output needs in front of the cost-volume loaders when no normalisation closes the stage (raft_stereo's fnet ends in a
 * plain 1x1 convolution). */
extern "C" int stereo_relayout_nhwc_to_nchw(const void *x, int x_is_f16, void *out, int out_is_f16, int B, int C, int H, int W,
                                            void *stream)
{
    if (B < 0 || C < 0 || H < 0 || W < 0) return STEREO_ERR_BAD_SHAPE;
    if (B == 0 || C == 0 || H == 0 || W == 0) return STEREO_OK;
    STEREO_REQUIRE_PTR(x);
    STEREO_REQUIRE_PTR(out);
    const long long HW = (long long)H * W;
    if (C % 4 != 0 || C > 256 || HW % 4 != 0 || !fits_grid_yz(B) || (reinterpret_cast<uintptr_t>(x) & (x_is_f16 ? 7u : 15u)) != 0 ||
        (out_is_f16 ? (reinterpret_cast<uintptr_t>(out) & 7u) != 0 : !aligned16(out)) || HW / 32 >= (1LL << 31))
        return STEREO_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (x_is_f16)
        return out_is_f16 ? launch_nhwc<f16, f16>(static_cast<const f16 *>(x), nullptr, nullptr, nullptr, static_cast<f16 *>(out),
                                                  B, C, HW, 0.0f, 0, st)
                          : launch_nhwc<f16, float>(static_cast<const f16 *>(x), nullptr, nullptr, nullptr,
                                                    static_cast<float *>(out), B, C, HW, 0.0f, 0, st);
    return out_is_f16 ? launch_nhwc<float, f16>(static_cast<const float *>(x), nullptr, nullptr, nullptr, static_cast<f16 *>(out),
                                                B, C, HW, 0.0f, 0, st)
                      : launch_nhwc<float, float>(static_cast<const float *>(x), nullptr, nullptr, nullptr,
                                                  static_cast<float *>(out), B, C, HW, 0.0f, 0, st);
}
