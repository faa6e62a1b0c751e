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
const int kOptDefault[OPT_COUNT] = {1, 1, 0, 0, 0, 1};
const char *const kOptName[OPT_COUNT] = {"corr_mfma", "gwc_fused", "att_kernel", "att_waves", "tiles_per_wg",
                                         "prob_single_pass"};
std::atomic<int> g_opt[OPT_COUNT] = {{1}, {1}, {0}, {0}, {0}, {1}};
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
