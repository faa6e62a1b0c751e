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
