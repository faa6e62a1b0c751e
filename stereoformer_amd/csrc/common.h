// Shared helpers for the gfx950 kernels behind include/stereo_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stereo_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Test hooks (stereo_test_option in include/stereo_hip.h): process-wide relaxed atomics, read once per launch.  They
// only steer WHICH kernel serves a call (forcing a fallback onto a shape the fast path would take, or a short
// persistent run so that small test inputs exercise the multi-tile loop); no environment variable is read anywhere.
enum StereoOpt {
    OPT_CORR_MFMA = 0,   // 1 (default): matrix-core volume kernels where their envelope allows; 0: VALU kernels
    OPT_GWC_FUSED,       // 1 (default): A3 + group mean + A4 in one launch; 0: separate launches
    OPT_ATT_KERNEL,      // 0 (default): by shape; 2: LDS-score-sheet; 4: two-pass register; 5: fp32 online; 6: chunked; 7: split-bf16; 9: split-fp16
    OPT_ATT_WAVES,       // 0 (default): by row length; 4 / 5 / 8: waves per workgroup of the persistent attention kernels
    OPT_TILES_PER_WG,    // 0 (default): sized to the resident workgroups; n > 0: tiles per persistent workgroup
    OPT_PROB_SINGLE_PASS,  // 1 (default): softmax column in registers (D <= 192); 0: two-pass kernel
    OPT_CONV_ROWS,       // 0 (default): pixel-tile rows of the split-fp16 convolution by launch size; 2 / 4: forced
    OPT_CORR_F16P,       // 1 (default): split-fp16 matrix phase of the A1 volume kernel; 0: f32 MFMA
    OPT_CORR_BWD_FUSED,  // 1 (default): A1 backward reads the gradient volume once (corr_bwd_fused.hip); 0: two launches;
                         // 2 / 3: timing ablations of the one-read kernel (no arithmetic / no volume loads)
    OPT_COUNT
};
int stereo_opt(int which);

// > 64 KB of dynamic LDS needs a one-time opt-in per kernel and device; idempotent, so a benign race between host
// threads (DataParallel replicas) at worst sets it twice.  `done` is a per-instantiation bit mask over devices.
#include <atomic>
template <typename K>
static inline bool lds_opt_in(K kern, size_t bytes, std::atomic<unsigned long long> &done)
{
    if (bytes <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const bool track = dev >= 0 && dev < 64;
    if (track && (done.load(std::memory_order_acquire) >> dev & 1ull)) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
        return false;
    if (track) done.fetch_or(1ull << dev, std::memory_order_release);
    return true;
}

// Workgroups of a kernel instantiation that fit one CU (registers, waves, LDS), asked of the runtime once per instantiation:
// what a persistent launch divides its tiles by.
template <typename K>
static inline int workgroups_per_cu(K kern, int block, size_t lds, std::atomic<int> &cache)
{
    int n = cache.load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(kern), block, lds) != hipSuccess || n < 1)
        n = 1;
    cache.store(n, std::memory_order_relaxed);
    return n;
}

#define STEREO_REQUIRE_PTR(p) \
    do {                      \
        if ((p) == nullptr) return STEREO_ERR_NULL_POINTER; \
    } while (0)

#define STEREO_LAUNCH_CHECK()                                        \
    do {                                                             \
        if (hipGetLastError() != hipSuccess) return STEREO_ERR_LAUNCH; \
    } while (0)

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// grid.y / grid.z are limited to 65535
static inline bool fits_grid_yz(long long v) { return v >= 1 && v <= 65535; }

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void st4_nt(float *p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); }

// ---- feature operand types: the matrix-core kernels take their FEATURE inputs as fp32 or fp16 and
// convert on the way into LDS; all arithmetic and every output stays fp32.  `Quad<T>`: 4 consecutive elements as they sit
// in memory (16 B / 8 B), kept raw in the prefetch registers and widened at commit time.
typedef _Float16 f16;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <typename T>
struct Quad;
template <>
struct Quad<float> {
    typedef f32x4 raw;
    static __device__ __forceinline__ raw ld(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
    static __device__ __forceinline__ f32x4 widen(raw r) { return r; }
};
template <>
struct Quad<f16> {
    typedef f16x4 raw;
    static __device__ __forceinline__ raw ld(const f16 *p) { return *reinterpret_cast<const f16x4 *>(p); }
    static __device__ __forceinline__ f32x4 widen(raw r) { return __builtin_convertvector(r, f32x4); }
};
// The same loads through a buffer resource (raw, stride 0): the byte offset `voff` is range-checked against the resource's
// num_records and an out-of-range lane reads 0 without touching memory (negative offsets wrap to huge unsigned ones).  On
// gfx950 the check sees voff + soff (tools/buf_probe.hip), so a row index must go into the resource's base, not into soff.
// Row- and channel-bound selects of a kernel's loaders disappear into it.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00020000);
}
template <typename T>
struct BufLd;
template <>
struct BufLd<float> {
    static __device__ __forceinline__ f32x4 quad(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    }
    static __device__ __forceinline__ float one(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
};
template <>
struct BufLd<f16> {
    static __device__ __forceinline__ f16x4 quad(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
    static __device__ __forceinline__ f16 one(__amdgpu_buffer_rsrc_t r, int voff, int soff)
    {
        return __builtin_bit_cast(f16, __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0));
    }
};
// Split-bf16 operands (attention_bf16.hip, attention_chunked.hip): x = p[0] + p[1] (+ p[2]) exactly -- three bf16 pieces carry
// an fp32 value's 24 mantissa bits, two an fp16 value's 11.  Two values at a time: one v_cvt_pk_bf16_f32, the two widenings
// (shift / mask) and one v_pk_add_f32 per piece level.
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int NP>
__device__ __forceinline__ void split_pair(f32x2 x, bf16x2 *p)
{
    p[0] = __builtin_convertvector(x, bf16x2);
    const f32x2 r = x - __builtin_convertvector(p[0], f32x2);
    p[1] = __builtin_convertvector(r, bf16x2);
    if (NP == 3) p[2] = __builtin_convertvector(r - __builtin_convertvector(p[1], f32x2), bf16x2);
}
__device__ __forceinline__ bf16x4 cat2(bf16x2 a, bf16x2 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// eight fp32 values -> NP bf16x8 MFMA operands
template <int NP>
__device__ __forceinline__ void split_oct(const float *x, bf16x8 *p)
{
    bf16x2 t[4][NP];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair<NP>((f32x2){x[2 * j], x[2 * j + 1]}, t[j]);
#pragma unroll
    for (int q = 0; q < NP; ++q) p[q] = cat4(cat2(t[0][q], t[1][q]), cat2(t[2][q], t[3][q]));
}
// Split-fp16 operands (conv_split.hip, attention_f16.hip): x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16(2^11 (x - hi)): 22-23
// significant bits for 2^-14 <= |x| < 65 504 (absolute error <= 2^-36 below, inf / NaN above -- never silently wrong).  The
// piece products hi hi | hi lo', lo' hi go to two fp32 accumulators (each product is exact in fp32: 11 x 11 bits); result =
// acc0 + 2^-11 acc1.  Two values at a time: v_cvt_pk_f16_f32, two widenings, v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
constexpr float SPLIT16_LO_SCALE = 2048.0f, SPLIT16_LO_UNSCALE = 1.0f / 2048.0f;
__device__ __forceinline__ void split_pair_f16(f32x2 x, f16x2 *p)
{
    p[0] = __builtin_convertvector(x, f16x2);
    const f32x2 r = (x - __builtin_convertvector(p[0], f32x2)) * SPLIT16_LO_SCALE;
    p[1] = __builtin_convertvector(r, f16x2);
}
// eight fp32 values -> hi / lo' MFMA operands
__device__ __forceinline__ void split_oct_f16(const float *x, f16x8 *p)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f16x2 t[2];
        split_pair_f16((f32x2){x[2 * j], x[2 * j + 1]}, t);
        p[0][2 * j] = t[0][0], p[0][2 * j + 1] = t[0][1];
        p[1][2 * j] = t[1][0], p[1][2 * j + 1] = t[1][1];
    }
}
// base pointers of quad loads: 16 B for fp32, 8 B for fp16
template <typename T>
static inline bool aligned_quad(const void *p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(T) - 1)) == 0; }

// v_exp_f32 based exp: |rel err| ~1e-6 on the softmax range; far inside the 1e-3 disparity bar
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// ConvGRU gate functions (update.py:28-32): shared by the pointwise kernels and the convolution epilogues so that the fused
// and the separate forms agree bit for bit.  Hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): |error| <= 3e-7
// absolute on both -- fp32 rounding level for values in (0, 1) / (-1, 1) -- at 4-5 instructions instead of the ~30-40 of
// expf + IEEE division / tanhf, which matters inside a convolution epilogue (64 gate values per lane).
__device__ __forceinline__ float gate_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float gate_tanh(float x)
{
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// (1 - z) h + z q with the contraction spelled out (the compiler's own choice differs from kernel to kernel)
__device__ __forceinline__ float gate_blend(float z, float h, float q) { return fmaf(z, q, (1.0f - z) * h); }
__device__ __forceinline__ float gate_softsign(float x) { return x / (1.0f + fabsf(x)); }

// ste_ceil / ste_floor forward values (fixed_local_cost_volume.py:5-9):
// ceil(x) - x + x evaluated left to right in fp32 (no contraction: these are
// plain adds, and the library is built without fast-math).
__device__ __forceinline__ float ste_ceil_f(float x) { return (ceilf(x) - x) + x; }
__device__ __forceinline__ float ste_floor_f(float x) { return (floorf(x) - x) + x; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Sampling position of disp_warp (utils/disparity_warper.py:83-106: grid (x - disp, y) normalised with (w-1), (h-1) -- :55-56 --,
// then grid_sample's un-normalisation with align_corners=False -- :100), in the reference's own order of fp32 operations.
__device__ __forceinline__ void disp_warp_coords(int x, int y, float d, int W, int H, float &ux, float &uy)
{
    const float gx = (float)x + (-d);
    const float gy = (float)y + 0.0f;
    const float nx = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(gx, (float)(W - 1))), 1.0f);
    const float ny = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(gy, (float)(H - 1))), 1.0f);
    ux = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(nx, 1.0f), (float)W), 1.0f), 2.0f);
    uy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(ny, 1.0f), (float)H), 1.0f), 2.0f);
}

// One interpolated tap of build_cost_volume_from_volume
// (fixed_local_cost_volume.py:61-80).  v -> vol[b,0,h,w], stride = H*W.
__device__ __forceinline__ float sample_tap(const float *__restrict__ v, size_t stride, int D, float c)
{
    const float cc = clampf(ste_ceil_f(c), 0.0f, (float)(D - 1));
    const float cf = clampf(ste_floor_f(c), 0.0f, (float)(D - 1));
    const int ic = (int)cc, ifl = (int)cf;  // .long() truncates toward zero
    const float floor_rate = (float)ic - c;
    const float ceil_rate = 1.0f - floor_rate;
    return v[(size_t)ifl * stride] * floor_rate + v[(size_t)ic * stride] * ceil_rate;
}

// corr_mfma.hip: A1 (+ A4 when disp != nullptr) on the f32 matrix cores.  Library-internal (not part of the C ABI);
// returns STEREO_ERR_UNSUPPORTED outside its envelope (C <= 32, D <= 256, W % 4 == 0, quad-aligned).  TIN = float | f16:
// the feature operands' storage type (widened on the way into LDS; arithmetic and outputs are fp32 either way).
template <typename TIN>
int stereo_corr_volume_mfma(const TIN *L, const TIN *R, float *vol, float *disp, int B, int C, int H, int W, int D,
                            float sgn, hipStream_t st);
// A3 + group mean + A4 in one kernel (even channels per group <= 16); same envelope, same return convention.
template <typename TIN>
int stereo_gwc_mean_argmin_mfma(const TIN *L, const TIN *R, float *volg, float *mean, float *disp, int B, int C, int H,
                                int W, int D, int G, float sgn, hipStream_t st);
// attention.hip: A1 + A4 without any volume (= cross-attention, mask 'zero', scale +-1/C, no output tensor).
template <typename TIN>
int stereo_corr_soft_argmin_mfma(const TIN *L, const TIN *R, float *disp, int B, int C, int H, int W, int D, float sgn,
                                 hipStream_t st);
// corr_mfma_bwd.hip: backward of A1 (G = 1) on the f32 matrix cores; either output may be null.
int stereo_corr_volume_bwd_mfma(const float *left, const float *right, const float *grad_vol, float *grad_left,
                                float *grad_right, int B, int C, int H, int W, int D, hipStream_t st);
// attention_chunked.hip: A7 for any C / Cv / value tensor, channel-chunked and software-pipelined; features fp32 or fp16.
template <typename TIN>
int stereo_cross_attention_chunked(const TIN *L, const TIN *R, const TIN *V, float *out, float *disp, int B, int C, int Cv,
                                   int H, int W, int D, int mask_mode, float sc, bool split_bf16_form, hipStream_t st);
// corr_chunked.hip: A1 (+ A4) for C > 32, channels in 32-channel chunks (D <= 128); same conventions as above.
template <typename TIN>
int stereo_corr_volume_chunked(const TIN *L, const TIN *R, float *vol, float *disp, int B, int C, int H, int W, int D,
                               float sgn, hipStream_t st);
// attention_f16.hip: A7 for C <= 32, value = right features, on the fp16 matrix cores with two-piece operands (hi + 2^-11 lo'),
// three products, two accumulators: the default there.
// attention_chunked16.hip: disparity-only A7 for any channel count on the fp16 matrix cores with split operands (16-channel
// items); STEREO_ERR_UNSUPPORTED outside its envelope (D <= 128, W % 4 == 0, quad-aligned, 32 H W < 2^30).
template <typename TIN>
int stereo_cross_attention_disp_chunked16(const TIN *L, const TIN *R, float *disp, int B, int C, int H, int W, int D,
                                          int mask_mode, float sc, hipStream_t st);
template <typename TIN>
int stereo_cross_attention_split16(const TIN *left, const TIN *right, float *out, float *disp, int B, int C, int H, int W, int D,
                                   int mask_mode, float sc, hipStream_t st);
// attention_bf16.hip: A7 for C <= 32, value = right features, on the bf16 matrix cores with split (hi + mid + lo) operands.
template <typename TIN>
int stereo_cross_attention_split(const TIN *left, const TIN *right, float *out, float *disp, int B, int C, int H, int W, int D,
                                 int mask_mode, float sc, hipStream_t st);
