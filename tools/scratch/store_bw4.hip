// Micro-benchmark: which shapes of a plain streaming store reach the 6.8 TB/s of "one workgroup, one 16-byte store
// per thread, 4 KB per workgroup"?  48 GB written per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

// one workgroup per chunk of T*U float4; WAVE_CONTIG: a wave's U stores are adjacent (wave writes U KB contiguous),
// else the workgroup's stores of one round are adjacent (thread stride T)
template <int T, int U, bool WAVE_CONTIG>
__global__ void __launch_bounds__(T) k_chunk(float4 *out, size_t n16)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * (T * U);
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = base + (WAVE_CONTIG ? (size_t)wave * (64 * U) + u * 64 + lane : (size_t)u * T + threadIdx.x);
        if (i < n16) out[i] = make_float4(1.f, 2.f, 3.f, (float)u);
    }
}

// fixed grid striding through the buffer, T*U float4 per workgroup per round
template <int T, int U>
__global__ void __launch_bounds__(T) k_stride(float4 *out, size_t n16)
{
    const size_t step = (size_t)gridDim.x * (T * U);
    for (size_t base = (size_t)blockIdx.x * (T * U) + threadIdx.x; base < n16; base += step) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * T;
            if (i < n16) out[i] = make_float4(1.f, 2.f, 3.f, (float)u);
        }
    }
}

template <typename F>
void timed(const char *name, size_t bytes, F launch)
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
    printf("%-64s %.3f ms %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    fflush(stdout);
}

#define CHUNK(T, U, WC) timed("chunk: T=" #T " U=" #U " wave-contiguous=" #WC, bytes, [&] { \
    hipLaunchKernelGGL((k_chunk<T, U, WC>), dim3((unsigned)((n16 + (T * U) - 1) / (T * U))), dim3(T), 0, 0, buf, n16); })
#define STRIDE(T, U, G) timed("stride: T=" #T " U=" #U " grid=" #G, bytes, [&] { \
    hipLaunchKernelGGL((k_stride<T, U>), dim3(G), dim3(T), 0, 0, buf, n16); })

int main()
{
    const size_t bytes = 48000000000ull, n16 = bytes / 16;
    float4 *buf; CK(hipMalloc(&buf, bytes));
    CHUNK(256, 1, false);
    CHUNK(64, 1, false);
    CHUNK(128, 1, false);
    CHUNK(512, 1, false);
    CHUNK(1024, 1, false);
    CHUNK(256, 2, false);
    CHUNK(256, 2, true);
    CHUNK(256, 4, false);
    CHUNK(256, 4, true);
    CHUNK(64, 4, true);
    CHUNK(1024, 4, false);
    CHUNK(256, 16, false);
    CHUNK(256, 16, true);
    STRIDE(256, 1, 2048);
    STRIDE(256, 1, 4096);
    STRIDE(256, 1, 8192);
    STRIDE(1024, 1, 512);
    STRIDE(1024, 1, 1024);
    STRIDE(256, 4, 2048);
    STRIDE(1024, 4, 512);
    CHUNK(256, 1, false);
    return 0;
}
