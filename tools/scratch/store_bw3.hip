// Micro-benchmark: does the rate of plain streaming stores depend on the size of the launch?  (A torch fill_ of 2^32
// bytes runs at 6.9 TB/s, of 6e9 bytes at 5.9 TB/s.)  One workgroup per chunk, as an elementwise library kernel
// does, against a fixed grid that strides through the buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

template <int U>
__global__ void __launch_bounds__(256) k_chunk(float4 *out, size_t n16)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) out[i] = make_float4(1.f, 2.f, 3.f, (float)u);
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_stride(float4 *out, size_t n16)
{
    const size_t step = (size_t)gridDim.x * (256 * U);
    for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n16; base += step) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 256;
            if (i < n16) out[i] = make_float4(1.f, 2.f, 3.f, (float)u);
        }
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
    for (int r = 0; r < 5; r++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / 5;
}

int main()
{
    const size_t cap = 52ull << 30;
    float4 *buf; CK(hipMalloc(&buf, cap));
    const size_t sizes[] = {1ull << 32, 4000000000ull, 5000000000ull, 6000000000ull, 3ull << 31, 8000000000ull, 1ull << 33,
                            1ull << 35, 48000000000ull, 45ull << 30, 48ull << 30};
    for (size_t bytes : sizes) {
        const size_t n16 = bytes / 16;
        float a = timed([&] { hipLaunchKernelGGL((k_chunk<1>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, buf, n16); });
        float b = timed([&] { hipLaunchKernelGGL((k_chunk<4>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, 0, buf, n16); });
        float c = timed([&] { hipLaunchKernelGGL((k_stride<4>), dim3(2048), dim3(256), 0, 0, buf, n16); });
        float d = timed([&] { hipLaunchKernelGGL((k_stride<4>), dim3(4096), dim3(256), 0, 0, buf, n16); });
        printf("%12zu bytes: chunk 4K %.3f ms %.2f TB/s | chunk 16K %.3f ms %.2f TB/s | stride g2048 %.3f ms %.2f TB/s | stride g4096 %.3f ms %.2f TB/s\n",
               bytes, a, bytes / a / 1e9, b, bytes / b / 1e9, c, bytes / c / 1e9, d, bytes / d / 1e9);
        fflush(stdout);
    }
    return 0;
}
