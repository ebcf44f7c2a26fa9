// Micro-benchmark: is the price of the payload reads paid where loads and stores share a CU's memory
// pipeline, or further out (L2 / fabric)?  256 workgroups of 1024 threads, one per CU (each asks for
// 100 KB of LDS so that no two share a CU).  48 GB of stores (wave per packet as in pkt_bw4) and the
// 1.5 GB of payload reads, arranged three ways:
//   MODE 0  stores only
//   MODE 1  every wave reads its packet's payload, then stores (the real arrangement)
//   MODE 2  readers apart: READERS workgroups only read (all payloads, nothing stored), the others
//           only store (all packets) -- same bytes read and written overall, never by the same CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int READERS>
__global__ void __launch_bounds__(1024) k_pkt(float4 *__restrict__ out, const uint32_t *__restrict__ pay, uint32_t n_pkts, uint32_t *sink)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) lds[0] = 0;
    if (MODE == 2 && blockIdx.x < READERS) {
        // reader workgroup: its share of all payload words, dword per lane, 6 loads in flight per lane
        const size_t total = (size_t)n_pkts * 375;
        uint32_t t = 0;
        for (size_t i = ((size_t)blockIdx.x * 16 + wave) * 384 + lane; i < total; i += (size_t)READERS * 16 * 384) {
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (i + 64 * k < total) t += pay[i + 64 * k];
        }
        if (t == 0x12345678u) sink[0] = t;
        return;
    }
    const uint32_t first = MODE == 2 ? (blockIdx.x - READERS) * 16 + wave : blockIdx.x * 16 + wave;
    const uint32_t step = (MODE == 2 ? gridDim.x - READERS : gridDim.x) * 16;
    for (uint32_t p = first; p < n_pkts; p += step) {
        float acc = 1.f;
        if (MODE == 1) {
            const uint32_t *q = pay + (size_t)p * 375;
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k * 64 + lane < 375) acc += (float)q[k * 64 + lane];
        }
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
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
    printf("%-72s %.3f ms\n", name, ms / 5);
    fflush(stdout);
}

int main()
{
    const uint32_t n = 1000000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 4096)); CK(hipMemset(pay, 1, (size_t)n * 1500 + 4096));
    uint32_t *sink; CK(hipMalloc(&sink, 64));
    const size_t lds = 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pkt<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pkt<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pkt<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pkt<2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pkt<2, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; rep++) {
        timed("stores only (256 workgroups, one per CU)", [&] { hipLaunchKernelGGL((k_pkt<0, 0>), dim3(256), dim3(1024), lds, 0, buf, pay, n, sink); });
        timed("every wave reads its payload, then stores", [&] { hipLaunchKernelGGL((k_pkt<1, 0>), dim3(256), dim3(1024), lds, 0, buf, pay, n, sink); });
        timed("8 CUs only read, 248 only store", [&] { hipLaunchKernelGGL((k_pkt<2, 8>), dim3(256), dim3(1024), lds, 0, buf, pay, n, sink); });
        timed("16 CUs only read, 240 only store", [&] { hipLaunchKernelGGL((k_pkt<2, 16>), dim3(256), dim3(1024), lds, 0, buf, pay, n, sink); });
        timed("32 CUs only read, 224 only store", [&] { hipLaunchKernelGGL((k_pkt<2, 32>), dim3(256), dim3(1024), lds, 0, buf, pay, n, sink); });
    }
    return 0;
}
