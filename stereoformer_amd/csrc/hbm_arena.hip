// HBM arena of separately mapped physical chunks (host code only; no kernel).
//
// The headline launch (volume.hip, 28.7 GB of row-contiguous stores) runs at 4.65-5.9 ms depending on WHERE in device memory
// its 25.5-GB volume lies (profiles/r03_place_window_probe.txt): inside one large allocation the time is a
// smooth function of the window's offset, and a window re-created as an allocation of its own lands elsewhere.  So the search
// has to keep the very pages it timed -- but not the rest.  This file builds the arena from `nchunks` physical allocations of
// `chunk_bytes` mapped back to back into one reserved address range (HIP virtual-memory management); once the caller has timed
// its windows, stereo_hbm_arena_keep() unmaps and releases every chunk outside the chosen one: the window keeps both its
// addresses and its pages, the driver gets everything else back.
//
// The reference has no counterpart (torch's allocator places build_gwc_volume's output, utils/build_cost_volume.py:5-54).
#include "common.h"

#include <vector>

namespace {

struct HbmArena {
    int dev = 0;
    size_t chunk = 0;
    char *base = nullptr;
    size_t reserved = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<char> live;  // chunk i is created + mapped
};

void drop_chunk(HbmArena *a, size_t i)
{
    if (!a->live[i]) return;
    (void)hipMemUnmap(a->base + i * a->chunk, a->chunk);
    (void)hipMemRelease(a->handles[i]);
    a->live[i] = 0;
}

}  // namespace

extern "C" int stereo_hbm_arena_granularity(long long *bytes)
{
    STEREO_REQUIRE_PTR(bytes);
    int dev = 0, ok = 0;
    if (hipGetDevice(&dev) != hipSuccess) return STEREO_ERR_LAUNCH;
    if (hipDeviceGetAttribute(&ok, hipDeviceAttributeVirtualMemoryManagementSupported, dev) != hipSuccess || !ok)
        return STEREO_ERR_UNSUPPORTED;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0)
        return STEREO_ERR_UNSUPPORTED;
    *bytes = (long long)g;
    return STEREO_OK;
}

extern "C" int stereo_hbm_arena_create(long long chunk_bytes, int nchunks, void **arena_out, void **base_out)
{
    STEREO_REQUIRE_PTR(arena_out);
    STEREO_REQUIRE_PTR(base_out);
    *arena_out = nullptr;
    *base_out = nullptr;
    if (chunk_bytes <= 0 || nchunks <= 0) return STEREO_ERR_BAD_SHAPE;
    long long gran = 0;
    int rc = stereo_hbm_arena_granularity(&gran);
    if (rc != STEREO_OK) return rc;
    if (chunk_bytes % gran) return STEREO_ERR_BAD_SHAPE;

    HbmArena *a = new HbmArena;
    (void)hipGetDevice(&a->dev);
    a->chunk = (size_t)chunk_bytes;
    a->reserved = a->chunk * (size_t)nchunks;
    a->handles.resize(nchunks);
    a->live.assign(nchunks, 0);
    void *base = nullptr;
    if (hipMemAddressReserve(&base, a->reserved, (size_t)gran, nullptr, 0) != hipSuccess || base == nullptr) {
        (void)hipGetLastError();
        delete a;
        return STEREO_ERR_LAUNCH;
    }
    a->base = static_cast<char *>(base);

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = a->dev;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    bool failed = false;
    for (int i = 0; i < nchunks && !failed; ++i) {
        if (hipMemCreate(&a->handles[i], a->chunk, &prop, 0) != hipSuccess) {
            failed = true;
            break;
        }
        if (hipMemMap(a->base + (size_t)i * a->chunk, a->chunk, 0, a->handles[i], 0) != hipSuccess) {
            (void)hipMemRelease(a->handles[i]);
            failed = true;
            break;
        }
        a->live[i] = 1;
        if (hipMemSetAccess(a->base + (size_t)i * a->chunk, a->chunk, &acc, 1) != hipSuccess) failed = true;
    }
    if (failed) {
        (void)hipGetLastError();
        for (size_t i = 0; i < a->live.size(); ++i) drop_chunk(a, i);
        (void)hipMemAddressFree(a->base, a->reserved);
        delete a;
        return STEREO_ERR_LAUNCH;
    }
    *arena_out = a;
    *base_out = a->base;
    return STEREO_OK;
}

// Everything outside chunks [first, first + count) goes back to the driver.  The device is synchronised first: no launch may
// still touch a page that is about to be unmapped.  Addresses inside the kept chunks stay valid; the rest of the range stays
// RESERVED (nothing else can be mapped there) but any access to it faults -- the host layer only ever hands out kept windows.
extern "C" int stereo_hbm_arena_keep(void *arena, int first, int count)
{
    STEREO_REQUIRE_PTR(arena);
    HbmArena *a = static_cast<HbmArena *>(arena);
    const int n = (int)a->live.size();
    if (first < 0 || count < 0 || first + count > n) return STEREO_ERR_BAD_SHAPE;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(a->dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return STEREO_ERR_LAUNCH;
    for (int i = 0; i < n; ++i)
        if (i < first || i >= first + count) drop_chunk(a, (size_t)i);
    (void)hipSetDevice(prev);
    return STEREO_OK;
}

extern "C" int stereo_hbm_arena_destroy(void *arena)
{
    if (arena == nullptr) return STEREO_OK;
    HbmArena *a = static_cast<HbmArena *>(arena);
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(a->dev);
    (void)hipDeviceSynchronize();
    for (size_t i = 0; i < a->live.size(); ++i) drop_chunk(a, i);
    (void)hipMemAddressFree(a->base, a->reserved);
    (void)hipSetDevice(prev);
    delete a;
    return STEREO_OK;
}
