// Micro-benchmark: store rate of the wave-per-packet pattern (48 000-byte packets, no reads) into windows of a
// given size at given offsets of one 280 GiB allocation.  usage: region_scan <window bytes> <first offset> <step> <count>
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

int main(int argc, char **argv)
{
    const size_t GiB = 1ull << 30, span = 280 * GiB;
    const size_t win = strtoull(argv[1], 0, 0), first = strtoull(argv[2], 0, 0), step = strtoull(argv[3], 0, 0);
    const int count = atoi(argv[4]);
    char *big; CK(hipMalloc(&big, span));
    const uint32_t n = (uint32_t)(win / 48000);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int k = 0; k < count; k++) {
        const size_t off = first + (size_t)k * step;
        if (off + win > span) break;
        float4 *w = reinterpret_cast<float4 *>(big + off);
        hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, w, n);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, w, n);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ms /= 3;
        printf("window %6.2f GB at %8.3f GB (%#zx): %.3f ms %.2f TB/s\n", win / 1e9, off / 1e9, off, ms, (double)n * 48000 / ms / 1e9);
        fflush(stdout);
    }
    return 0;
}
