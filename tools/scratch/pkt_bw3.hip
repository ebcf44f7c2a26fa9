// Micro-benchmark: 10^6 packets of 48000 bytes written by (a) one wave per packet, (b) one workgroup per packet,
// each with a fixed grid striding through the packets or with one workgroup per unit of work (the dispatcher
// keeps the write window tight).  Optional 1500-byte payload read per packet.  Timing experiments only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float read_payload(const uint32_t *pay, uint32_t p, int lane, int n_lanes)
{
    const uint32_t *q = pay + (size_t)p * 375;
    float acc = 0.f;
    for (int i = lane; i < 375; i += n_lanes) acc += (float)q[i];
    return acc;
}

// one wave per packet: the wave sweeps its 3000 float4 in 47 stores
template <int WPB, bool READ>
__global__ void __launch_bounds__(64 * WPB) k_wave_pkt(float4 *out, const uint32_t *pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t p = blockIdx.x * WPB + wave; p < n_pkts; p += gridDim.x * WPB) {
        float acc = READ ? read_payload(pay, p, lane, 64) : 1.f;
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
    }
}

// one workgroup per packet: the T threads sweep the packet together
template <int T, bool READ>
__global__ void __launch_bounds__(T) k_wg_pkt(float4 *out, const uint32_t *pay, uint32_t n_pkts)
{
    for (uint32_t p = blockIdx.x; p < n_pkts; p += gridDim.x) {
        float acc = READ ? read_payload(pay, p, threadIdx.x, T) : 1.f;
        float4 *dst = out + (size_t)p * 3000;
        for (int i = threadIdx.x; i < 3000; i += T) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
    }
}

// one workgroup per K consecutive packets, written as one region of K*3000 float4
template <int T, int K>
__global__ void __launch_bounds__(T) k_wg_pkts(float4 *out, uint32_t n_pkts)
{
    for (uint32_t g = blockIdx.x; g < n_pkts / K; g += gridDim.x) {
        float4 *dst = out + (size_t)g * (3000 * K);
        for (int i = threadIdx.x; i < 3000 * K; i += T) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
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
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 64)); CK(hipMemset(pay, 1, (size_t)n * 1500));
#define L(K, G, T, ...) hipLaunchKernelGGL((K), dim3(G), dim3(T), 0, 0, __VA_ARGS__)
    timed("wave per packet, 8 waves/WG, fixed grid 512", [&] { L((k_wave_pkt<8, false>), 512, 512, buf, pay, n); });
    timed("wave per packet, 8 waves/WG, fixed grid 512, payload read", [&] { L((k_wave_pkt<8, true>), 512, 512, buf, pay, n); });
    timed("wave per packet, 8 waves/WG, one WG per 8 packets", [&] { L((k_wave_pkt<8, false>), n / 8, 512, buf, pay, n); });
    timed("wave per packet, 8 waves/WG, one WG per 8 packets, payload read", [&] { L((k_wave_pkt<8, true>), n / 8, 512, buf, pay, n); });
    timed("wave per packet, 4 waves/WG, one WG per 4 packets", [&] { L((k_wave_pkt<4, false>), n / 4, 256, buf, pay, n); });
    timed("wave per packet, 16 waves/WG, one WG per 16 packets", [&] { L((k_wave_pkt<16, false>), n / 16, 1024, buf, pay, n); });
    timed("WG(256) per packet, fixed grid 2048", [&] { L((k_wg_pkt<256, false>), 2048, 256, buf, pay, n); });
    timed("WG(256) per packet, one WG per packet", [&] { L((k_wg_pkt<256, false>), n, 256, buf, pay, n); });
    timed("WG(256) per packet, one WG per packet, payload read", [&] { L((k_wg_pkt<256, true>), n, 256, buf, pay, n); });
    timed("WG(512) per packet, one WG per packet", [&] { L((k_wg_pkt<512, false>), n, 512, buf, pay, n); });
    timed("WG(512) per packet, one WG per packet, payload read", [&] { L((k_wg_pkt<512, true>), n, 512, buf, pay, n); });
    timed("WG(1024) per packet, one WG per packet", [&] { L((k_wg_pkt<1024, false>), n, 1024, buf, pay, n); });
    timed("WG(1024) per packet, fixed grid 512", [&] { L((k_wg_pkt<1024, false>), 512, 1024, buf, pay, n); });
    timed("WG(512) per 8 packets as one region, one WG per region", [&] { L((k_wg_pkts<512, 8>), n / 8, 512, buf, n); });
    timed("WG(512) per 8 packets as one region, fixed grid 512", [&] { L((k_wg_pkts<512, 8>), 512, 512, buf, n); });
    timed("WG(1024) per 16 packets as one region, one WG per region", [&] { L((k_wg_pkts<1024, 16>), n / 16, 1024, buf, n); });
    return 0;
}
