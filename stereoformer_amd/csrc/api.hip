// ABI version + error strings for include/stereo_hip.h.
#include "common.h"

extern "C" int stereo_abi_version(void) { return 1; }

extern "C" const char *stereo_strerror(int code)
{
    switch (code) {
        case STEREO_OK: return "ok";
        case STEREO_ERR_NULL_POINTER: return "null pointer argument";
        case STEREO_ERR_BAD_SHAPE: return "bad shape (negative extent, C % G != 0, empty softmax axis ...)";
        case STEREO_ERR_UNSUPPORTED: return "size outside the supported range of the gfx950 kernels";
        case STEREO_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown stereo_hip error";
    }
}

// ---- test hooks -------------------------------------------------------------------------------------------------
#include <string.h>

namespace {
const int kOptDefault[OPT_COUNT] = {1, 1, 0, 0, 0, 1, 0, 1, 1};
const char *const kOptName[OPT_COUNT] = {"corr_mfma", "gwc_fused", "att_kernel", "att_waves", "tiles_per_wg",
                                         "prob_single_pass", "conv_rows", "corr_f16p", "corr_bwd_fused"};
std::atomic<int> g_opt[OPT_COUNT] = {{1}, {1}, {0}, {0}, {0}, {1}, {0}, {1}, {1}};
}  // namespace

int stereo_opt(int which) { return g_opt[which].load(std::memory_order_relaxed); }

extern "C" int stereo_test_option(const char *name, int value)
{
    if (name == nullptr) return STEREO_ERR_NULL_POINTER;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, kOptName[i]) == 0) {
            g_opt[i].store(value < 0 ? kOptDefault[i] : value, std::memory_order_relaxed);
            return STEREO_OK;
        }
    return STEREO_ERR_BAD_SHAPE;
}

// ---- content fingerprints of parameter tensors ------------------------------------------------------------------
// A packed copy of a weight tensor (conv_split.hip) must be rebuilt when the weights change -- and an in-place update through
// `tensor.data` changes them without touching anything the host can see (same storage, same version counter).  The host layer
// therefore compares a CONTENT fingerprint once per forward of a loop: out[k] = sum over the 32-bit words w_i of tensor k of
// mix(w_i, i) mod 2^64 (splitmix64 finaliser; position-dependent, so permutations change it too), all tensors in one launch.
namespace {
constexpr int FP_MAX = 32;
struct FpArgs {
    const unsigned *p[FP_MAX];
    long long words[FP_MAX];
};
__global__ __launch_bounds__(256) void fingerprint_kernel(FpArgs a, unsigned long long *__restrict__ out)
{
    const int k = blockIdx.y;
    const unsigned *__restrict__ w = a.p[k];
    const long long n = a.words[k];
    unsigned long long h = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned long long v = (unsigned long long)w[i] + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        v ^= v >> 30;
        v *= 0xBF58476D1CE4E5B9ull;
        v ^= v >> 27;
        v *= 0x94D049BB133111EBull;
        v ^= v >> 31;
        h += v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) h += __shfl_xor(h, o, 64);
    if ((threadIdx.x & 63) == 0 && n > 0) atomicAdd(out + k, h);
}
}  // namespace

extern "C" int stereo_fingerprint(const void *const *tensors, const long long *nbytes, int n, unsigned long long *out, void *stream)
{
    if (n < 0) return STEREO_ERR_BAD_SHAPE;
    if (n == 0) return STEREO_OK;
    if (tensors == nullptr || nbytes == nullptr) return STEREO_ERR_NULL_POINTER;
    STEREO_REQUIRE_PTR(out);
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(out, 0, sizeof(unsigned long long) * (size_t)n, st) != hipSuccess) return STEREO_ERR_LAUNCH;
    for (int k0 = 0; k0 < n; k0 += FP_MAX) {
        const int m = n - k0 < FP_MAX ? n - k0 : FP_MAX;
        FpArgs a{};
        long long most = 0;
        for (int k = 0; k < m; ++k) {
            if (nbytes[k0 + k] < 0 || nbytes[k0 + k] % 4 != 0) return STEREO_ERR_BAD_SHAPE;
            if (nbytes[k0 + k] > 0 && tensors[k0 + k] == nullptr) return STEREO_ERR_NULL_POINTER;
            if ((reinterpret_cast<uintptr_t>(tensors[k0 + k]) & 3u) != 0) return STEREO_ERR_UNSUPPORTED;
            a.p[k] = static_cast<const unsigned *>(tensors[k0 + k]);
            a.words[k] = nbytes[k0 + k] / 4;
            if (a.words[k] > most) most = a.words[k];
        }
        long long chunks = (most + 256 * 16 - 1) / (256 * 16);
        if (chunks < 1) chunks = 1;
        if (chunks > 1024) chunks = 1024;
        hipLaunchKernelGGL(fingerprint_kernel, dim3((unsigned)chunks, (unsigned)m), dim3(256), 0, st, a, out + k0);
        STEREO_LAUNCH_CHECK();
    }
    return STEREO_OK;
}
