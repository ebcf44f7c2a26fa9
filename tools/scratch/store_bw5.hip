// Micro-benchmark: is it the number of stores a wave has in flight that separates 6.8 TB/s (one store per thread,
// then the wave ends) from 5.6 TB/s (a wave sweeping a packet)?  Waits for the stores' acknowledgements every
// EVERY stores (0 = never).  48 GB per launch.  Timing experiments only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

template <int WPB, int EVERY>
__global__ void __launch_bounds__(64 * WPB) k_wave_pkt(float4 *out, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t p = blockIdx.x * WPB + wave; p < n_pkts; p += gridDim.x * WPB) {
        float4 *dst = out + (size_t)p * 3000;
        int k = 0;
        for (int i = lane; i < 3000; i += 64) {
            dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
            if (EVERY > 0 && ++k % EVERY == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

// fixed grid, one 16-byte store per thread per round, the workgroup's stores adjacent
template <int T, int EVERY>
__global__ void __launch_bounds__(T) k_stride(float4 *out, size_t n16)
{
    const size_t step = (size_t)gridDim.x * T;
    int k = 0;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n16; i += step) {
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
        if (EVERY > 0 && ++k % EVERY == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <typename F>
void timed(const char *name, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    ms /= 5;
    printf("%-72s %.3f ms %.2f TB/s\n", name, ms, 48e9 / ms / 1e9);
    fflush(stdout);
}

int main()
{
    const uint32_t n = 1000000;
    const size_t n16 = (size_t)n * 3000;
    float4 *buf; CK(hipMalloc(&buf, n16 * 16));
#define L(K, G, T, ...) hipLaunchKernelGGL((K), dim3(G), dim3(T), 0, 0, __VA_ARGS__)
    timed("wave per packet, 8 waves/WG, grid 512, no waits", [&] { L((k_wave_pkt<8, 0>), 512, 512, buf, n); });
    timed("wave per packet, 8 waves/WG, grid 512, wait every store", [&] { L((k_wave_pkt<8, 1>), 512, 512, buf, n); });
    timed("wave per packet, 8 waves/WG, grid 512, wait every 2 stores", [&] { L((k_wave_pkt<8, 2>), 512, 512, buf, n); });
    timed("wave per packet, 8 waves/WG, grid 512, wait every 4 stores", [&] { L((k_wave_pkt<8, 4>), 512, 512, buf, n); });
    timed("wave per packet, 8 waves/WG, grid 1024, wait every store", [&] { L((k_wave_pkt<8, 1>), 1024, 512, buf, n); });
    timed("wave per packet, 8 waves/WG, grid 1024, wait every 2 stores", [&] { L((k_wave_pkt<8, 2>), 1024, 512, buf, n); });
    timed("wave per packet, 4 waves/WG, grid 512 (8 waves/CU), wait every store", [&] { L((k_wave_pkt<4, 1>), 512, 256, buf, n); });
    timed("wave per packet, 4 waves/WG, grid 512 (8 waves/CU), wait every 4", [&] { L((k_wave_pkt<4, 4>), 512, 256, buf, n); });
    timed("stride T=256 grid 2048, no waits", [&] { L((k_stride<256, 0>), 2048, 256, buf, n16); });
    timed("stride T=256 grid 2048, wait every store", [&] { L((k_stride<256, 1>), 2048, 256, buf, n16); });
    timed("stride T=256 grid 1024, wait every store", [&] { L((k_stride<256, 1>), 1024, 256, buf, n16); });
    timed("stride T=256 grid 512, wait every store", [&] { L((k_stride<256, 1>), 512, 256, buf, n16); });
    timed("stride T=256 grid 2048, wait every 2 stores", [&] { L((k_stride<256, 2>), 2048, 256, buf, n16); });
    timed("stride T=256 grid 1024, wait every 2 stores", [&] { L((k_stride<256, 2>), 1024, 256, buf, n16); });
    timed("stride T=256 grid 512, wait every 4 stores", [&] { L((k_stride<256, 4>), 512, 256, buf, n16); });
    timed("stride T=256 grid 256, wait every 4 stores", [&] { L((k_stride<256, 4>), 256, 256, buf, n16); });
    return 0;
}
