// Micro-benchmark: does the HBM region matter?  One 240 GB allocation; the wave-per-packet store pattern of
// pkt_bw4 (48 GB, 10^6 packets of 48 000 bytes, no reads) into windows at different offsets, and the same with the
// 1.5 GB of payload reads from a buffer allocated first / last.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k_pkt(float4 *__restrict__ out, const uint32_t *__restrict__ pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t step = gridDim.x * 8;
    for (uint32_t p = blockIdx.x * 8 + wave; p < n_pkts; p += step) {
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
float timed(F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 4; r++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / 4;
}

int main()
{
    const uint32_t n = 1000000;
    size_t free_b, total_b;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("free %.1f GB of %.1f GB\n", free_b / 1e9, total_b / 1e9);
    uint32_t *pay_first; CK(hipMalloc(&pay_first, (size_t)n * 1500 + 4096)); CK(hipMemset(pay_first, 1, (size_t)n * 1500 + 4096));
    const size_t GB = 1000000000ull, span = 240 * GB;
    char *big; CK(hipMalloc(&big, span));
    uint32_t *pay_last; CK(hipMalloc(&pay_last, (size_t)n * 1500 + 4096)); CK(hipMemset(pay_last, 1, (size_t)n * 1500 + 4096));
    printf("payload first %p, arena %p, payload last %p\n", (void *)pay_first, (void *)big, (void *)pay_last);
    for (int rep = 0; rep < 2; rep++)
        for (size_t off = 0; off + 48 * GB <= span; off += 24 * GB) {
            float4 *w = reinterpret_cast<float4 *>(big + off);
            const float s = timed([&] { hipLaunchKernelGGL((k_pkt<0>), dim3(512), dim3(512), 0, 0, w, pay_first, n); });
            const float rf = timed([&] { hipLaunchKernelGGL((k_pkt<1>), dim3(512), dim3(512), 0, 0, w, pay_first, n); });
            const float rl = timed([&] { hipLaunchKernelGGL((k_pkt<1>), dim3(512), dim3(512), 0, 0, w, pay_last, n); });
            printf("window at %3zu GB: stores only %.3f ms | + reads (payload allocated first) %.3f | + reads (allocated last) %.3f\n",
                   off / GB, s, rf, rl);
            fflush(stdout);
        }
    return 0;
}
