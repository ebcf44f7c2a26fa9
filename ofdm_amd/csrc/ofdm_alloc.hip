// libofdmtx, placement of a large output buffer.  On MI355X the device memory behaves as three
// parts of 96 GiB: streaming stores that go to one part run at 5.6-5.7 TB/s, stores that go to two
// or three parts at the same time at 7.1 TB/s.  A plain allocation of tens of GB is one run of
// physical memory and so lies in one part (two, by luck).  ofdmtx_stream_alloc builds the buffer
// out of 64 MiB pieces of physical memory taken in turn from three far-apart places, mapped back to
// back into one range of addresses (HIP's virtual-memory API), so that whatever is being written at
// one moment -- a few hundred MB of neighbouring packets -- is spread over the parts.  (Two parts
// give the same rate for stores alone, and keeping the inputs in the remaining part buys k_tx
// nothing: 8.32-8.39 ms in every combination.)
//
// Where a piece of physical memory lies is not visible to a user process.  What can be relied on
// (measured, not documented): the driver hands out pieces of one size in ascending order.  So the
// call takes (nearly) all free memory of the device as 64 MiB pieces for a moment, uses pieces from
// the first, second and last third of that sequence in turn, and gives the rest back.  When there
// is too little free memory for that sequence to span the parts, or any step fails, the buffer is
// an ordinary allocation.  Callers that care verify with their own launch (TxPlan.alloc_iq_probed
// times it against ordinary allocations and keeps the fastest).

#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "ofdm_common.h"

namespace {

constexpr size_t PIECE = 64ull << 20;
constexpr size_t KEEP_FREE = 2ull << 30;        // never taken from the device, not even for a moment

struct StreamBuf {
    size_t size;
    bool mapped;                                  // false: plain hipMalloc
    int device;
};
std::mutex g_lock;
std::map<void *, StreamBuf> g_bufs;

}  // namespace

extern "C" {

int ofdmtx_stream_alloc(ofdmtx_ctx *ctx, uint64_t bytes, void **out, int *spread)
{
    if (!ctx || !out || bytes == 0) return OFDMTX_EINVAL;
    *out = nullptr;
    if (spread) *spread = 0;
    HIPCHK(hipSetDevice(ctx->device));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t pieces = ((size_t)bytes + PIECE - 1) / PIECE, per_third = (pieces + 2) / 3;
    const size_t pool = free_b > KEEP_FREE ? (free_b - KEEP_FREE) / PIECE : 0;
    // the sequence must span the memory's parts: at least three quarters of the device, and room
    // for the buffer's share inside every third
    const bool can_spread = pool * PIECE >= total_b / 4 * 3 && pool / 3 >= per_third + 2;
    void *va = nullptr;
    if (can_spread) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = ctx->device;
        std::vector<hipMemGenericAllocationHandle_t> h;
        h.reserve(pool);
        bool ok = true;
        for (size_t k = 0; k < pool && ok; k++) {
            hipMemGenericAllocationHandle_t one;
            ok = hipMemCreate(&one, PIECE, &prop, 0) == hipSuccess;
            if (ok) h.push_back(one);
        }
        size_t n_mapped = 0;
        if (ok) ok = hipMemAddressReserve(&va, pieces * PIECE, PIECE, nullptr, 0) == hipSuccess;
        if (ok) {
            const size_t third = h.size() / 3, skip = (third - per_third) / 2;     // the middle of every third
            for (size_t i = 0; i < pieces && ok; i++) {
                const size_t k = (i % 3) * third + skip + i / 3;
                ok = hipMemMap((char *)va + i * PIECE, PIECE, 0, h[k], 0) == hipSuccess;
                if (ok) n_mapped++;
            }
        }
        if (ok) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            ok = hipMemSetAccess(va, pieces * PIECE, &acc, 1) == hipSuccess;
        }
        for (hipMemGenericAllocationHandle_t one : h) (void)hipMemRelease(one);    // mapped pieces stay with their mapping
        if (!ok) {
            (void)hipGetLastError();
            if (va) {
                if (n_mapped) (void)hipMemUnmap(va, n_mapped * PIECE);
                (void)hipMemAddressFree(va, pieces * PIECE);
                va = nullptr;
            }
        }
    }
    StreamBuf rec = {pieces * PIECE, va != nullptr, ctx->device};
    if (!va) {
        rec.size = (size_t)bytes;
        HIPCHK(hipMalloc(&va, (size_t)bytes));
    }
    {
        std::lock_guard<std::mutex> g(g_lock);
        g_bufs[va] = rec;
    }
    *out = va;
    if (spread) *spread = rec.mapped ? 1 : 0;
    return OFDMTX_OK;
}

int ofdmtx_stream_free(ofdmtx_ctx *ctx, void *ptr)
{
    if (!ctx || !ptr) return OFDMTX_EINVAL;
    StreamBuf rec;
    {
        std::lock_guard<std::mutex> g(g_lock);
        auto it = g_bufs.find(ptr);
        if (it == g_bufs.end()) return OFDMTX_EINVAL;
        rec = it->second;
        g_bufs.erase(it);
    }
    // (the context is not looked into: the buffer may outlive it)
    HIPCHK(hipSetDevice(rec.device));
    HIPCHK(hipDeviceSynchronize());                 // nothing may still be writing to it
    if (rec.mapped) {
        HIPCHK(hipMemUnmap(ptr, rec.size));
        HIPCHK(hipMemAddressFree(ptr, rec.size));
    } else {
        HIPCHK(hipFree(ptr));
    }
    return OFDMTX_OK;
}

}  // extern "C"
