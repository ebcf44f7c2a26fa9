// Micro-benchmark: map of the store rate over one 280 GB allocation in windows of 4 GB: (a) linear fill (every
// workgroup sweeps a contiguous slice, 16 bytes per lane), (b) the wave-per-packet pattern (48 000-byte packets).
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

__global__ void __launch_bounds__(512) k_lin(float4 *__restrict__ out, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512)
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F>
float timed(F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 6; r++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / 6;
}

int main()
{
    const size_t GiB = 1ull << 30, span = 280 * GiB, win = 4 * GiB;
    char *big; CK(hipMalloc(&big, span));
    printf("arena %p\n", (void *)big);
    const uint32_t n = (uint32_t)(win / 48000);
    for (size_t off = 0; off + win <= span; off += win) {
        float4 *w = reinterpret_cast<float4 *>(big + off);
        const float a = timed([&] { hipLaunchKernelGGL(k_lin, dim3(2048), dim3(512), 0, 0, w, win / 16); });
        const float b = timed([&] { hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, w, n); });
        printf("%3zu GiB: linear %.3f ms %.2f TB/s | packets %.3f ms %.2f TB/s\n", off / GiB, a, win / a / 1e9, b,
               (double)n * 48000 / b / 1e9);
        fflush(stdout);
    }
    return 0;
}
