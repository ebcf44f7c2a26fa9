// Micro-benchmark: a 48 GB window that contains the 96 GiB boundary of the allocation is written 11 % faster than
// one that does not (region_scan.hip).  Where does the time go?  The window [84 GB, 132 GB) of a 280 GiB allocation
// is written (a) in one launch, (b) as its two parts below and above 96 GiB in two launches, (c) in 12 launches of
// 4 GB, and for comparison the window [120 GB, 168 GB) the same three ways.  Wave-per-packet stores, no reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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

static float sweep(char *base, size_t bytes, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t n = (uint32_t)(bytes / 48000);
    hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, (float4 *)base, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, (float4 *)base, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t GiB = 1ull << 30, GB = 1000000000ull;
    char *big; CK(hipMalloc(&big, 280 * GiB));
    for (size_t start : {84 * GB, 120 * GB}) {
        const size_t end = start + 48 * GB, cut = start == 84 * GB ? 96 * GiB / 48000 * 48000 : start + 24 * GB;
        const float whole = sweep(big + start, 48 * GB, 3);
        const float lo = sweep(big + start, cut - start, 3), hi = sweep(big + cut, end - cut, 3);
        float parts = 0;
        for (size_t o = start; o < end; o += 4 * GB) parts += sweep(big + o, 4 * GB, 3);
        printf("window [%zu, %zu) GB: one launch %.3f ms (%.2f TB/s) | two launches split at %.1f GB: %.3f + %.3f = %.3f ms (%.2f / %.2f TB/s) | 12 launches of 4 GB: %.3f ms\n",
               start / GB, end / GB, whole, 48e9 / whole / 1e9, cut / 1e9, lo, hi, lo + hi, (cut - start) / lo / 1e9, (end - cut) / hi / 1e9, parts);
        fflush(stdout);
    }
    return 0;
}
