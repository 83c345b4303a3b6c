// antsrl_mem.hip — antsrl_mem_alloc / antsrl_mem_free (include/antsrl.h): device memory for the step's big streams, built
// from physical PIECES of at most ANTSRL_MEM_PIECE_BYTES mapped into one virtual range (HIP virtual-memory management).
//
// Why the library offers an allocator at all (it never allocates inside a step): on MI355X the same kernels on the same
// inputs run 15 % apart depending on WHERE the two big buffers lie — the observation tensor (a 0.7 GB streaming write) and
// the workspace's cell records (scattered gathers).  The device's memory falls into a few large zones, and k_perceive is
// slow (0.197 against 0.167 ms at c3) exactly when the two lie in the same one (profiles/r05/two_colour.txt, region_map.txt;
// DESIGN.md section 2).  In a fresh process hipMalloc and the virtual-memory allocator used here draw from DIFFERENT zones, so
// a workspace from hipMalloc / torch.empty and an observation tensor from antsrl_mem_alloc are a fast pair, two hipMalloc
// buffers the slow one — until tens of GB are in use and either allocator has wandered into the other's zone, which is why
// the host side measures (BatchedAntsEnv.tune_placement).  Offsets inside an allocation and the stream's pitches change
// nothing (profiles/history/r04/placement_probe.txt, profiles/r05/env_pitch_probe.txt).
// Freed blocks are POOLED, not unmapped (round 5): antsrl_mem_free parks the block — range and pieces, still mapped — on a
// per-device free list keyed by its size, antsrl_mem_alloc hands a parked block of the same device and size back before it
// creates anything.  No hipMemUnmap on any path a running program takes, so no address is ever translated to anything but
// the pieces it was first mapped to, and the reserved address space is bounded by the largest set of blocks that were live
// (or parked) at once.  Why this matters: on ROCm 7.2 / gfx950 a range that was unmapped, freed and reserved again could
// still be translated to its OLD physical pieces — two live buffers then aliased each other's memory, silently
// (profiles/r04/vmm_stress.py: 95 of 300 allocate / fill / check / free rounds).  Round 4 retired every freed range for the
// life of the process (no reuse, so no stale translation — and unbounded address-space growth: ADVICE r4); the pool needs
// neither.  Physical memory goes back to the device in antsrl_mem_trim() only (explicit; ranges unmapped there ARE
// retired), or with the process.
// Host code only; no kernel here.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/antsrl.h"

namespace {
struct Block {
    size_t size = 0, piece = 0;
    int device = 0; // the device the pieces live on (and the one a free / trim synchronises)
    std::vector<hipMemGenericAllocationHandle_t> pieces;
};
std::mutex g_mu;
std::unordered_map<void *, Block> g_live;                        // handed out
std::multimap<std::pair<int, size_t>, std::pair<void *, Block>> g_pool; // parked, still mapped: (device, size) -> block
size_t g_reserved = 0, g_retired = 0, g_pooled = 0;              // bytes of address space reserved / retired, bytes parked

// Waits until nothing enqueued on the BLOCK's device can still touch it (the caller's current device may be another one).
void sync_device(int device)
{
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return;
    (void)hipDeviceSynchronize();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
}

// Physical pieces back to the device.  The range itself stays reserved and is never handed out again (retired): a range
// that is never reused cannot meet a stale translation.  Only antsrl_mem_trim and a failed allocation come here.
void unmap_and_retire(void *base, Block &b, size_t mapped)
{
    sync_device(b.device);
    for (size_t off = 0; off < mapped; off += b.piece) (void)hipMemUnmap((char *)base + off, b.piece);
    for (auto h : b.pieces) (void)hipMemRelease(h);
    if (base && mapped == 0) {
        (void)hipMemAddressFree(base, b.size); // (nothing was ever mapped there: no translation exists)
        g_reserved -= b.size;
    } else if (base) {
        g_retired += b.size;
    }
}
} // namespace

extern "C" int antsrl_mem_alloc(size_t bytes, int device, void **ptr)
{
    if (!ptr || bytes == 0) return ANTSRL_E_INVALID;
    *ptr = nullptr;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return ANTSRL_E_DEVICE;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleType = hipMemHandleTypeNone;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    int rc = ANTSRL_E_DEVICE;
    Block b{};
    b.device = device;
    void *base = nullptr;
    size_t mapped = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    do {
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) break;
        const size_t piece = (ANTSRL_MEM_PIECE_BYTES + gran - 1) / gran * gran;
        const size_t n = (bytes + piece - 1) / piece;
        b.size = n * piece;
        b.piece = piece;
        auto it = g_pool.find({device, b.size});
        if (it != g_pool.end()) { // a parked block of this device and size: the same range on the same pieces, nothing to map
            base = it->second.first;
            b = std::move(it->second.second);
            g_pool.erase(it);
            g_pooled -= b.size;
            rc = ANTSRL_OK;
            break;
        }
        if (hipMemAddressReserve(&base, b.size, piece, nullptr, 0) != hipSuccess) { base = nullptr; break; }
        g_reserved += b.size;
        bool ok = true;
        for (size_t i = 0; i < n && ok; ++i) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) { ok = false; break; }
            b.pieces.push_back(h);
            if (hipMemMap((char *)base + i * piece, piece, 0, h, 0) != hipSuccess) { ok = false; break; }
            mapped += piece;
        }
        if (!ok) { rc = ANTSRL_E_NOMEM; break; }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(base, b.size, &acc, 1) != hipSuccess) break;
        rc = ANTSRL_OK;
    } while (false);
    if (rc != ANTSRL_OK) {
        if (base) unmap_and_retire(base, b, mapped);
        else for (auto h : b.pieces) (void)hipMemRelease(h);
    } else {
        g_live.emplace(base, std::move(b));
        *ptr = base;
    }
    (void)hipSetDevice(prev);
    return rc;
}

extern "C" int antsrl_mem_free(void *ptr)
{
    if (!ptr) return ANTSRL_OK;
    Block b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(ptr);
        if (it == g_live.end()) return ANTSRL_E_INVALID;
        b = std::move(it->second);
        g_live.erase(it);
    }
    // Nothing enqueued may still touch the block when its next owner gets it (hipFree synchronises, too) — on the block's
    // own device, whichever one is current.
    sync_device(b.device);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pooled += b.size;
    const std::pair<int, size_t> key{b.device, b.size};
    g_pool.emplace(key, std::make_pair(ptr, std::move(b)));
    return ANTSRL_OK;
}

extern "C" int antsrl_mem_trim(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_pool) unmap_and_retire(kv.second.first, kv.second.second, kv.second.second.size);
    g_pool.clear();
    g_pooled = 0;
    return ANTSRL_OK;
}

extern "C" int antsrl_mem_stats(size_t *live_bytes, size_t *pooled_bytes, size_t *reserved_va_bytes, size_t *retired_va_bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    size_t live = 0;
    for (auto &kv : g_live) live += kv.second.size;
    if (live_bytes) *live_bytes = live;
    if (pooled_bytes) *pooled_bytes = g_pooled;
    if (reserved_va_bytes) *reserved_va_bytes = g_reserved;
    if (retired_va_bytes) *retired_va_bytes = g_retired;
    return ANTSRL_OK;
}
