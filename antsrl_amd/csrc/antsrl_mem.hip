// antsrl_mem.hip — antsrl_mem_alloc / antsrl_mem_free (include/antsrl.h): device memory for the step's big streams, built
// from physical PIECES of at most ANTSRL_MEM_PIECE_BYTES mapped into one virtual range (HIP virtual-memory management).
//
// Why the library offers an allocator at all (it never allocates inside a step): on MI355X the same kernels on the same
// inputs run 15 % apart depending on the PHYSICAL layout of the two big buffers — the observation tensor (a 0.7 GB write
// stream of 1372-byte rows, 702 464 bytes from one environment to the next) and the workspace's cell records (the
// gathers).  When both lie in physically contiguous ranges of 128 MiB or more — what hipMalloc hands a fresh process —
// the streams alias on the memory channels: k_perceive 0.197 ms at c3.  With the observation tensor in pieces of <= 32 MiB
// it is 0.167-0.174 ms on nearly every allocation (profiles/r04/placement_probe4*.txt: pieces of 2 / 8 / 32 MiB fast,
// 128 / 256 / 512 / 1024 MiB slow; hipExtMallocWithFlags' fully contiguous memory was 25 % slower still,
// profiles/r03/box_state_probe3.txt).  Not a law: BOTH buffers pieced was slower again, and some processes' first
// allocations stay slow either way, which is why the host side measures (BatchedAntsEnv.tune_placement,
// profiles/r04/ROUND_NOTES.md).  Offsets INSIDE an allocation change nothing (placement_probe.txt).
// Host code only; no kernel here.
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/antsrl.h"

namespace {
struct Block {
    size_t size, piece;
    std::vector<hipMemGenericAllocationHandle_t> pieces;
};
std::mutex g_mu;
std::unordered_map<void *, Block> g_blocks;

// The physical pieces go back to the device; the VIRTUAL range is never handed out again (`keep_va`: it stays reserved
// for the life of the process).  On ROCm 7.2 / gfx950 a range that was unmapped, freed and reserved again can still be
// translated to its OLD physical pieces by the GPU: two live buffers then alias each other's memory — silently
// (profiles/r04/vmm_stress.py: 95 of 300 allocate / fill / check / free rounds, with one hipMemUnmap per range or one per
// piece, with the device idle, with 100 ms of waiting).  A virtual address that is never reused cannot meet a stale
// translation; address space is not a scarce resource (a c3 observation tensor is 0.7 GB of a 128 TB space).
void release(void *base, Block &b, size_t mapped, bool keep_va)
{
    (void)hipDeviceSynchronize();
    for (size_t off = 0; off < mapped; off += b.piece) (void)hipMemUnmap((char *)base + off, b.piece);
    for (auto h : b.pieces) (void)hipMemRelease(h);
    if (base && !keep_va) (void)hipMemAddressFree(base, b.size);
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
    void *base = nullptr;
    size_t mapped = 0;
    do {
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) break;
        const size_t piece = (ANTSRL_MEM_PIECE_BYTES + gran - 1) / gran * gran;
        const size_t n = (bytes + piece - 1) / piece;
        b.size = n * piece;
        b.piece = piece;
        if (hipMemAddressReserve(&base, b.size, piece, nullptr, 0) != hipSuccess) { base = nullptr; break; }
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
        release(base, b, mapped, mapped != 0); // (a range that was mapped at all is retired, too)
    } else {
        std::lock_guard<std::mutex> lk(g_mu);
        g_blocks.emplace(base, std::move(b));
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
        auto it = g_blocks.find(ptr);
        if (it == g_blocks.end()) return ANTSRL_E_INVALID;
        b = std::move(it->second);
        g_blocks.erase(it);
    }
    release(ptr, b, b.size, true);
    return ANTSRL_OK;
}
