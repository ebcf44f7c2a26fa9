// Micro-benchmark: can an output buffer whose pieces alternate between far-apart places of the device memory be
// had WITHOUT taking the whole device as small handles first?  Physical memory is taken with the virtual-memory
// API in this order: a spacer handle (one hipMemCreate of many GiB, never mapped), the pieces of place 0, a spacer,
// the pieces of place 1, ... -- on the assumption (checked by the result: the store rate) that the driver hands
// out physical memory in ascending order, so that a spacer moves the next pieces that many GiB up.  The spacers
// are released before the buffer is used.  The wave-per-packet store pattern of k_tx (48 000-byte packets, no
// reads) is then timed into the buffer.
//
// usage: vmm_spacer <packets> <piece MiB> <place GiB> [<place GiB> ...]
//   places = offsets from the start of the free memory at which the pieces of each place begin (ascending);
//   pieces are mapped round robin over the places.  One place = an ordinary contiguous run.
// prints one line: piece size, places, seconds to build, handles created, ms per launch, TB/s.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(512) k_pkt(float4 *__restrict__ out, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t step = gridDim.x * 8;
    for (uint32_t p = blockIdx.x * 8 + wave; p < n_pkts; p += step) {
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
    }
}

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    if (argc < 4) { printf("usage: vmm_spacer <packets> <piece MiB> <place GiB> ...\n"); return 2; }
    const size_t MiB = 1ull << 20, GiB = 1ull << 30;
    const uint32_t n = (uint32_t)atol(argv[1]);
    const size_t piece = (size_t)atol(argv[2]) * MiB;
    std::vector<size_t> place;
    for (int i = 3; i < argc; i++) place.push_back((size_t)(atof(argv[i]) * GiB));
    const size_t n_places = place.size();
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    size_t free_b, total_b;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t pieces = ((size_t)n * 48000 + piece - 1) / piece;
    const size_t per_place = (pieces + n_places - 1) / n_places;

    const double t0 = now();
    std::vector<hipMemGenericAllocationHandle_t> spacers;
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> h(n_places);
    size_t frontier = 0, created = 0;
    double t_spacers = 0;
    for (size_t a = 0; a < n_places; a++) {
        if (place[a] > frontier) {
            // the spacer itself in units of 1 GiB handles at most 64 GiB each (one call per unit)
            size_t left = (place[a] - frontier) / piece * piece;
            const double ts = now();
            while (left) {
                const size_t sz = left;        // one handle for the whole distance
                hipMemGenericAllocationHandle_t s;
                CK(hipMemCreate(&s, sz, &prop, 0));
                spacers.push_back(s);
                created++;
                left -= sz;
            }
            t_spacers += now() - ts;
            frontier = place[a] / piece * piece;
        }
        for (size_t k = 0; k < per_place; k++) {
            hipMemGenericAllocationHandle_t one;
            CK(hipMemCreate(&one, piece, &prop, 0));
            h[a].push_back(one);
            created++;
        }
        frontier += per_place * piece;
    }
    const double t_created = now();
    for (auto s : spacers) CK(hipMemRelease(s));
    const double t_released = now();
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, pieces * piece, piece, nullptr, 0));
    for (size_t i = 0; i < pieces; i++) CK(hipMemMap((char *)va + i * piece, piece, 0, h[i % n_places][i / n_places], 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, pieces * piece, &acc, 1));
    const double t1 = now();

    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, (float4 *)va, n);
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    const int reps = 5;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(ev0));
        hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, (float4 *)va, n);
        CK(hipEventRecord(ev1));
        CK(hipEventSynchronize(ev1));
        float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("piece %5zu MiB  places", piece / MiB);
    for (size_t a = 0; a < n_places; a++) printf(" %.0f", place[a] / (double)GiB);
    printf(" GiB | build %.3f s (spacers %.3f s, create all %.3f, release spacers %.3f, map %.3f), %zu handles | %.3f ms mean, %.3f best = %.2f TB/s\n",
           t1 - t0, t_spacers, t_created - t0, t_released - t_created, t1 - t_released, created, sum / reps, best,
           (double)n * 48000 / (sum / reps) / 1e9);
    CK(hipMemUnmap(va, pieces * piece));
    for (auto &v : h) for (auto one : v) CK(hipMemRelease(one));
    CK(hipMemAddressFree(va, pieces * piece));
    return 0;
}
