// Micro-benchmark: can the part of the device memory a 2 GiB window lies in be told from a SHORT measurement?
// All free memory is taken as 64 MiB handles and mapped in allocation order; windows of 2 GiB are written
// together with window 0 as alternating 64 MiB pieces, with two store orders:
//   tight  every wave takes the next 32 KiB unit in logical order: what is in flight lies within 128 MiB
//   wide   every wave owns a contiguous run of units: what is in flight is spread over the whole 4 GiB
// usage: probe_pair [stride of windows tested, default 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr size_t PIECE = 64ull << 20, WINDOW = 32;
constexpr uint32_t UNIT_F4 = 2048, UNITS_PER_PIECE = (uint32_t)(PIECE / (UNIT_F4 * 16));

template <bool WIDE>
__global__ void __launch_bounds__(512) k_probe(float4 *__restrict__ a, float4 *__restrict__ b, uint32_t n_units)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_waves = gridDim.x * 8, g = blockIdx.x * 8 + wave, per = n_units / n_waves;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t u = WIDE ? g * per + j : j * n_waves + g;
        const uint32_t piece = u / UNITS_PER_PIECE, within = u % UNITS_PER_PIECE;
        float4 *dst = ((piece & 1u) ? b : a) + ((size_t)(piece >> 1) * UNITS_PER_PIECE + within) * UNIT_F4;
        for (uint32_t i = lane; i < UNIT_F4; i += 64) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

int main(int argc, char **argv)
{
    const int stride = argc > 1 ? atoi(argv[1]) : 4;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    CK(hipSetDevice(0));
    size_t free_b, total_b;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t n_win = (free_b - (4ull << 30)) / (PIECE * WINDOW), n_h = n_win * WINDOW;
    char *pool;
    CK(hipMemAddressReserve((void **)&pool, n_h * PIECE, PIECE, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n_h);
    for (size_t k = 0; k < n_h; k++) { CK(hipMemCreate(&h[k], PIECE, &prop, 0)); CK(hipMemMap(pool + k * PIECE, PIECE, 0, h[k], 0)); }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(pool, n_h * PIECE, &acc, 1));
    printf("%zu windows of 2 GiB\n", n_win);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t n_units = 2 * WINDOW * UNITS_PER_PIECE;
    auto win = [&](size_t w) { return (float4 *)(pool + w * WINDOW * PIECE); };
    auto time2 = [&](bool wide, float4 *a, float4 *b, uint32_t units) {
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0));
            if (wide) hipLaunchKernelGGL(k_probe<true>, dim3(512), dim3(512), 0, 0, a, b, units);
            else hipLaunchKernelGGL(k_probe<false>, dim3(512), dim3(512), 0, 0, a, b, units);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return best;
    };
    printf("window 0 halves: tight %.3f ms  wide %.3f ms (x2 for comparison)\n",
           2 * time2(false, win(0), win(0) + WINDOW / 2 * PIECE / 16, n_units / 2), 2 * time2(true, win(0), win(0) + WINDOW / 2 * PIECE / 16, n_units / 2));
    for (size_t w = 1; w < n_win; w += stride)
        printf("window %3zu (%5.1f GiB): tight %.3f ms  wide %.3f ms\n", w, w * 2.0, time2(false, win(0), win(w), n_units), time2(true, win(0), win(w), n_units));
    return 0;
}
