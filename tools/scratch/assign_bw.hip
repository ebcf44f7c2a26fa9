// Micro-benchmark: which packets are written at the same time?  48 GB of wave-per-packet stores into five 48 GB
// buffers (allocation order; their rates differ, see region_bw.hip), with the packets dealt to the 4096 waves in
// different ways:
//   MODE 0  round robin (wave w: packets w, w + 4096, ...): the packets in flight are 4096 neighbours (196 MB)
//   MODE 1  blocked (wave w: packets 244 w .. 244 w + 243): the packets in flight are spread over all 48 GB
//   MODE 2  round robin over workgroups, blocked inside (workgroup b: packets 8 k G + 8 b .. + 7; as MODE 0) with the
//           waves of a workgroup on ONE packet at a time (each wave every 8th 1 KB piece)
//   MODE 3  round robin with a stride of 61 packets between neighbouring waves (in flight: 4096 packets over 12 GB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k_pkt(float4 *__restrict__ out, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_waves = gridDim.x * 8, w = blockIdx.x * 8 + wave;
    if (MODE == 0) {
        for (uint32_t p = w; p < n_pkts; p += n_waves) {
            float4 *dst = out + (size_t)p * 3000;
            for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
        }
    } else if (MODE == 1) {
        const uint32_t per = (n_pkts + n_waves - 1) / n_waves;
        for (uint32_t p = w * per; p < (w + 1) * per && p < n_pkts; p++) {
            float4 *dst = out + (size_t)p * 3000;
            for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
        }
    } else if (MODE == 2) {
        for (uint32_t p = blockIdx.x; p < n_pkts; p += gridDim.x) {
            float4 *dst = out + (size_t)p * 3000;
            for (int i = threadIdx.x; i < 3000; i += 512) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
        }
    } else {
        // wave w starts at packet (61 w) mod n_waves-block: a permutation of the round-robin order inside every
        // run of n_waves * 61 packets
        const uint32_t run = n_waves * 61u;
        for (uint32_t base = 0; base < n_pkts; base += run)
            for (uint32_t k = 0; k < 61u; k++) {
                const uint32_t p = base + w * 61u + k;
                if (p >= n_pkts) break;
                float4 *dst = out + (size_t)p * 3000;
                for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
            }
    }
}

template <int MODE>
float run(float4 *buf, uint32_t n)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_pkt<MODE>), dim3(512), dim3(512), 0, 0, buf, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_pkt<MODE>), dim3(512), dim3(512), 0, 0, buf, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / 3;
}

int main()
{
    const uint32_t n = 1000000;
    float4 *bufs[5];
    for (int i = 0; i < 5; i++) CK(hipMalloc(&bufs[i], (size_t)n * 48000));
    for (int i = 0; i < 5; i++) {
        printf("buffer %d: round robin %.3f ms | blocked %.3f | workgroup per packet %.3f | runs of 61 per wave %.3f\n", i,
               run<0>(bufs[i], n), run<1>(bufs[i], n), run<2>(bufs[i], n), run<3>(bufs[i], n));
        fflush(stdout);
    }
    return 0;
}
