// libofdmtx, placement of a large output buffer.  On MI355X the device memory behaves as three
// parts of 96 GiB: streaming stores that go to one part run at 5.6-5.9 TB/s, stores that go to two
// or three parts at the same time at 6.8-7.1 TB/s.  (What the parts are is not documented; it is
// what eight stacks of 12-high HBM3E give if the stack-ID bits -- one of three groups of four dies,
// each with banks of its own -- are the top address bits: a stream confined to one part then has a
// third of the banks to work with.)  A plain allocation of tens of GB is one run of physical memory
// and so lies in one part (two, by luck).  ofdmtx_stream_alloc builds the buffer out of 64 MiB
// pieces of physical memory that alternate between parts, mapped back to back into one range of
// addresses (HIP's virtual-memory API), so that whatever is being written at one moment -- a few
// hundred MB of neighbouring packets -- is spread over the parts.
//
// Where a piece of physical memory lies is not visible to a user process, and the order in which
// the driver hands out memory depends on what was allocated and released before (taking the whole
// device in 64 MiB handles and trusting their order, or skipping ahead with large spacer handles,
// gave 7.1 TB/s on one box and 6.2 on the next).  So the parts are found by MEASUREMENT:
//   1. physical memory is taken in windows of 2 GiB (32 handles of 64 MiB, neighbours in the
//      allocation order), 16 GiB at a time, each window mapped as it comes;
//   2. every new window is timed with itself (its two halves written as a buffer of alternating
//      64 MiB pieces, the waves spread over all of it) and, as such a buffer, together with a
//      reference window.  The time of a slow window with itself (the median of those within 10 %
//      of the slowest, for single times scatter by several per cent while the driver clears what a
//      process before left) is the one-part time; a window that is faster than that with itself is
//      a mixture of parts (memory that was released in another order than it was taken) and is
//      left out.  Together with the reference (the window nearest to the one-part time), a window
//      of another part is 15-25 % faster than the one-part time, a window of the same part is
//      not; what falls in between is left out too;
//   3. the windows that differ from the reference are split once more, against the first of them;
//   4. taking stops as soon as the classes found can supply the buffer in equal shares -- two
//      classes, or all three if the caller wants them (stores alone run at the same rate either
//      way; with the transmit kernel's payload reads beside them three parts are 2-4 % faster, but
//      the third lies 192 GiB into an empty device) -- or when the free memory ends;
//   5. the chosen pieces are mapped in rotation, everything else goes back to the device.
// On a device that starts out empty the allocation order runs through the parts one after the
// other: a second part shows up about 96 GiB in, the third 192 GiB in, so a buffer on two parts is
// placed with about 130 GiB taken for a moment (0.8 s), on three with 220 GiB (2.3 s).  When only one class shows up (little free memory) or any step fails, the buffer is an
// ordinary allocation.  The call is for set-up time: it synchronises the device, and a second
// allocator at work on the same device meanwhile can run out of memory.

#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "ofdm_common.h"

namespace {

constexpr size_t PIECE = 64ull << 20;
constexpr size_t KEEP_FREE = 2ull << 30;        // never taken from the device, not even for a moment
constexpr size_t WINDOW = 32;                   // pieces per window: 2 GiB
constexpr size_t STEP_WINDOWS = 8;              // windows taken per step: 16 GiB
constexpr uint32_t UNIT_F4 = 2048;              // probe unit: 32 KiB = 2048 float4
constexpr uint32_t UNITS_PER_PIECE = (uint32_t)(PIECE / (UNIT_F4 * 16));

struct StreamBuf {
    size_t size;
    bool mapped;                                  // false: plain hipMalloc
    int device;
    size_t peak_held;                             // most physical memory held at one time while building
    uint32_t handles;                             // hipMemCreate calls
    uint32_t classes;                             // parts the pieces alternate between (0: ordinary allocation)
};
std::mutex g_lock;                                // serialises the build: two at once would starve each other
std::map<void *, StreamBuf> g_bufs;

// Writes two runs of pieces, at a and at b, as one buffer made of their pieces in alternation, every
// wave sweeping a contiguous run of 32 KiB units of its own: what is in flight at one moment is
// spread over the whole of both runs, the condition under which the parts show.  (2 GiB + 2 GiB
// written this way take 0.73-0.75 ms when both lie in one part and 0.59-0.62 ms when they lie in
// two; with the units handed out in logical order -- everything in flight within 128 MiB -- the
// same bytes take 0.66-0.68 ms wherever they lie.)  n_units: a multiple of the number of waves.
__global__ void __launch_bounds__(512) k_probe_pair(float4 *__restrict__ a, float4 *__restrict__ b, uint32_t n_units)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = n_units / (gridDim.x * 8u), u0 = (blockIdx.x * 8u + wave) * per;
    for (uint32_t u = u0; u < u0 + per; u++) {
        const uint32_t piece = u / UNITS_PER_PIECE, within = u % UNITS_PER_PIECE;
        float4 *dst = ((piece & 1u) ? b : a) + ((size_t)(piece >> 1) * UNITS_PER_PIECE + within) * UNIT_F4;
        for (uint32_t i = lane; i < UNIT_F4; i += 64) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

struct Built {
    void *va = nullptr;
    size_t peak = 0;
    uint32_t handles = 0, classes = 0;
};

struct Prober {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = true;
    Prober() { ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess; }
    ~Prober()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    // ms for writing `each` pieces at a and `each` pieces at b in alternation (the best of three timed
    // launches behind an untimed one that touches the pages)
    float pair(void *a, void *b, size_t each = WINDOW)
    {
        const uint32_t n_units = (uint32_t)(2 * each) * UNITS_PER_PIECE;
        float best = 1e30f;
        for (int rep = 0; rep < 4 && ok; rep++) {
            if (rep) ok = ok && hipEventRecord(e0, nullptr) == hipSuccess;
            hipLaunchKernelGGL(k_probe_pair, dim3(512), dim3(512), 0, nullptr, (float4 *)a, (float4 *)b, n_units);
            if (rep) {
                float ms = 0;
                ok = ok && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                     hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
                if (ok && ms < best) best = ms;
            }
        }
        ok = ok && hipGetLastError() == hipSuccess;
        return best;
    }
};

// The buffer of `pieces` pieces alternating between parts, or va == nullptr when the device cannot
// provide it.
Built build_spread(int device, size_t pieces, int want_parts)
{
    Built r;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return r;
    // The memory of a process that has just ended comes back over a few seconds: while the free
    // memory is still growing, wait for it (at most 5 s) rather than look at a device that seems full.
    for (int tries = 0; tries < 50 && free_b < total_b / 10 * 9; tries++) {
        usleep(100000);
        size_t f2 = 0;
        if (hipMemGetInfo(&f2, &total_b) != hipSuccess) return r;
        const bool growing = f2 > free_b + (256ull << 20);
        free_b = f2;
        if (!growing && tries >= 2) break;
    }
    if (free_b <= KEEP_FREE) return r;
    const size_t max_windows = (free_b - KEEP_FREE) / (PIECE * WINDOW);
    if (max_windows < 4 || pieces < 2) return r;

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    Prober probe;
    char *pool = nullptr;                                   // address range the windows are mapped into while they are timed
    const size_t pool_bytes = max_windows * WINDOW * PIECE;
    if (!probe.ok || hipMemAddressReserve((void **)&pool, pool_bytes, PIECE, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return r;
    }
    std::vector<hipMemGenericAllocationHandle_t> h;         // in allocation order; window w = handles [32 w, 32 w + 32)
    struct Win {
        float own = 0.f;          // ms for its own two halves in alternation, scaled to two windows
        float with_ref = 0.f;     // ms together with the reference
        int cls = -2;             // 0 = the reference's part, 1 / 2 = others, -1 = left out, -2 = not looked at yet
    };
    std::vector<Win> W;
    h.reserve(max_windows * WINDOW);
    size_t n_mapped = 0;
    bool ok = true;
    float t_one = 0.f;                                      // the one-part time (see below)
    int ref0 = -1, ref1 = -1, restarts = 0;                 // the reference, and the first window of another part
    size_t count[3] = {0, 0, 0};
    auto win = [&](size_t w) { return (void *)(pool + w * WINDOW * PIECE); };
    auto enough = [&](int parts) {
        // equal shares from the classes found: all three, or (parts == 2) the two largest
        size_t c[3] = {count[0], count[1], count[2]};
        std::sort(c, c + 3);
        const size_t want3 = (pieces + 2) / 3, want2 = (pieces + 1) / 2;
        return (c[0] * WINDOW >= want3) || (parts < 3 && c[1] * WINDOW >= want2);
    };

    bool exhausted = false;                                 // the device gave no more memory (less was free than it said)
    while (ok && !exhausted && !enough(want_parts) && W.size() < max_windows) {
        // ---- one step: STEP_WINDOWS more windows, mapped behind the ones already there
        const size_t w0 = W.size();
        size_t w1 = std::min(max_windows, w0 + STEP_WINDOWS);
        for (size_t k = w0 * WINDOW; k < w1 * WINDOW && ok; k++) {
            hipMemGenericAllocationHandle_t one;
            if (hipMemCreate(&one, PIECE, &prop, 0) != hipSuccess) {
                (void)hipGetLastError();
                exhausted = true;
                w1 = k / WINDOW;                            // whole windows only; the pieces behind them are released with the rest
                break;
            }
            h.push_back(one);
            ok = hipMemMap(pool + k * PIECE, PIECE, 0, one, 0) == hipSuccess;
            if (ok) n_mapped++;
        }
        if (!ok || w1 == w0) break;
        ok = hipMemSetAccess(win(w0), (w1 - w0) * WINDOW * PIECE, &acc, 1) == hipSuccess;
        if (!ok) break;
        // ---- every new window with itself.  A window that lies in one part takes the one-part time; on
        // a device with a history (memory released in another order than it was taken) a window can be
        // a mixture of parts, is faster with itself, and is left out.  The times of one-part windows
        // scatter by several per cent when the driver is still clearing a previous process's memory,
        // so the one-part time is not the largest time seen but the median of the step's slow windows
        // (those within 10 % of the slowest), and the reference is the window nearest to it.  Should a
        // later step's median be clearly slower still, the first step held mixtures only, and the
        // classification starts over from that step.
        W.resize(w1);
        float slowest = 0.f;
        for (size_t w = w0; w < w1 && ok; w++) {
            W[w].own = 2.f * probe.pair(win(w), (char *)win(w) + WINDOW / 2 * PIECE, WINDOW / 2);
            ok = probe.ok;
            slowest = std::max(slowest, W[w].own);
        }
        if (!ok) break;
        std::vector<float> slow;
        for (size_t w = w0; w < w1; w++)
            if (W[w].own >= 0.9f * slowest) slow.push_back(W[w].own);
        std::sort(slow.begin(), slow.end());
        const float step_one = slow[slow.size() / 2];
        if (ref0 < 0 || (step_one > 1.07f * t_one && restarts < 2)) {
            if (ref0 >= 0) restarts++;
            t_one = step_one;
            size_t nearest = w0;
            for (size_t w = w0; w < w1; w++)
                if (std::fabs(W[w].own - t_one) < std::fabs(W[nearest].own - t_one)) nearest = w;
            ref0 = (int)nearest;
            ref1 = -1;
            count[0] = count[1] = count[2] = 0;
            for (Win &x : W) x.cls = -2;
        }
        // ---- the windows not looked at yet: together with the reference, and those of another part
        // together with the first of them
        for (size_t w = 0; w < W.size() && ok; w++) {
            if (W[w].cls != -2) continue;
            int c;
            if (W[w].own < 0.93f * t_one) {
                c = -1;
            } else if ((int)w == ref0) {
                c = 0;
            } else {
                const float t = W[w].with_ref = probe.pair(win((size_t)ref0), win(w));
                c = t >= 0.93f * t_one ? 0 : t <= 0.88f * t_one ? 1 : -1;
                if (c == 1) {
                    if (ref1 < 0) {
                        ref1 = (int)w;
                    } else {
                        const float t2 = probe.pair(win((size_t)ref1), win(w));
                        c = t2 >= 0.93f * t_one ? 1 : t2 <= 0.88f * t_one ? 2 : -1;
                    }
                }
                ok = probe.ok;
            }
            W[w].cls = c;
            if (c >= 0) count[c]++;
        }
    }
    r.peak = h.size() * PIECE;
    r.handles = (uint32_t)h.size();
    if (getenv("OFDMTX_ALLOC_TRACE")) {
        // what the measurement saw: per window the time with itself, with the reference, and its class
        fprintf(stderr, "ofdmtx_stream_alloc: %zu pieces wanted, %zu windows of 2 GiB taken; one-part time %.3f ms (reference: window %d; "
                        "first window of another part: %d; restarts %d); classes %zu / %zu / %zu windows\n",
                pieces, W.size(), t_one, ref0, ref1, restarts, count[0], count[1], count[2]);
        for (size_t w = 0; w < W.size(); w++)
            fprintf(stderr, "  window %3zu  own %.3f ms  with the reference %.3f ms  class %d\n", w, W[w].own, W[w].with_ref, W[w].cls);
    }
    (void)hipDeviceSynchronize();

    // ---- the pieces of the result, taken from the classes in rotation
    std::vector<size_t> chosen;                             // handle indices in mapping order
    if (ok && enough(2)) {                                  // (three were wanted and the memory ended with two: two it is)
        int order[3] = {0, 1, 2};
        std::sort(order, order + 3, [&](int x, int y) { return count[x] > count[y]; });
        const int n_cls = (count[order[2]] * WINDOW >= (pieces + 2) / 3) ? 3 : 2;
        // (of each class the windows that took longest with themselves first: the least mixed ones)
        std::vector<size_t> by_time;
        for (size_t w = 0; w < W.size(); w++)
            if (W[w].cls >= 0) by_time.push_back(w);
        std::stable_sort(by_time.begin(), by_time.end(), [&](size_t x, size_t y) { return W[x].own > W[y].own; });
        std::vector<size_t> of[3];
        for (size_t w : by_time)
            for (size_t k = 0; k < WINDOW; k++) of[W[w].cls].push_back(w * WINDOW + k);
        size_t next[3] = {0, 0, 0};
        for (size_t i = 0; i < pieces; i++) {
            const int c = order[i % n_cls];
            chosen.push_back(of[c][next[c]++]);
        }
        r.classes = (uint32_t)n_cls;
    }
    // ---- the pool's mappings go; the chosen handles are mapped into the result's range
    if (n_mapped) (void)hipMemUnmap(pool, n_mapped * PIECE);
    (void)hipMemAddressFree(pool, pool_bytes);
    void *va = nullptr;
    size_t n_final = 0;
    if (!chosen.empty()) {
        ok = hipMemAddressReserve(&va, pieces * PIECE, PIECE, nullptr, 0) == hipSuccess;
        for (size_t i = 0; i < pieces && ok; i++) {
            ok = hipMemMap((char *)va + i * PIECE, PIECE, 0, h[chosen[i]], 0) == hipSuccess;
            if (ok) n_final++;
        }
        if (ok) ok = hipMemSetAccess(va, pieces * PIECE, &acc, 1) == hipSuccess;
    }
    for (hipMemGenericAllocationHandle_t one : h) (void)hipMemRelease(one);        // mapped pieces stay with their mapping
    if (chosen.empty() || !ok) {
        (void)hipGetLastError();
        if (va) {
            if (n_final) (void)hipMemUnmap(va, n_final * PIECE);
            (void)hipMemAddressFree(va, pieces * PIECE);
        }
        r.classes = 0;
        return r;
    }
    r.va = va;
    return r;
}

}  // namespace

extern "C" {

int ofdmtx_stream_alloc_parts(ofdmtx_ctx *ctx, uint64_t bytes, int want_parts, void **out, int *spread)
{
    if (!ctx || !out || bytes == 0 || want_parts < 2 || want_parts > 3) return OFDMTX_EINVAL;
    *out = nullptr;
    if (spread) *spread = 0;
    std::lock_guard<std::mutex> g(g_lock);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t pieces = ((size_t)bytes + PIECE - 1) / PIECE;
    const Built b = build_spread(ctx->device, pieces, want_parts >= 3 ? 3 : 2);
    void *va = b.va;
    StreamBuf rec = {pieces * PIECE, va != nullptr, ctx->device, b.peak, b.handles, b.classes};
    if (!va) {
        rec.size = (size_t)bytes;
        if (rec.peak_held < (size_t)bytes) rec.peak_held = (size_t)bytes;
        HIPCHK(hipMalloc(&va, (size_t)bytes));
    }
    g_bufs[va] = rec;
    *out = va;
    if (spread) *spread = rec.mapped ? 1 : 0;
    return OFDMTX_OK;
}

int ofdmtx_stream_alloc(ofdmtx_ctx *ctx, uint64_t bytes, void **out, int *spread)
{
    return ofdmtx_stream_alloc_parts(ctx, bytes, 2, out, spread);
}

int ofdmtx_stream_info(const void *ptr, uint64_t *bytes, uint64_t *peak_held_bytes, uint32_t *handles, uint32_t *parts)
{
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_bufs.find(const_cast<void *>(ptr));
    if (it == g_bufs.end()) return OFDMTX_EINVAL;
    if (bytes) *bytes = it->second.size;
    if (peak_held_bytes) *peak_held_bytes = it->second.peak_held;
    if (handles) *handles = it->second.handles;
    if (parts) *parts = it->second.classes;
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
