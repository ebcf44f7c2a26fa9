// Micro-benchmark: 16-byte streaming stores over 48 GB with the cache-policy bits of the vector
// store instruction (timing experiment only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void store16(f4 *p, f4 v)
{
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

// a wave writes 5 KiB contiguous per iteration (the shape of k_tx's row groups), 48 KB per "packet"
template <int MODE>
__global__ void __launch_bounds__(512) k_store(f4 *out, size_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t p = (size_t)blockIdx.x * 8 + wave; p < n_pkts; p += (size_t)gridDim.x * 8) {
        f4 *dst = out + p * 3000;
        for (int g = 0; g < 9; g++) {
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int e = g * 320 + i * 64 + lane;
                if (e < 3000) store16<MODE>(dst + e, (f4){1.f, 2.f, (float)g, (float)i});
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
void run(const char *name, f4 *buf, size_t n_pkts)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_store<MODE>), dim3(512), dim3(512), 0, 0, buf, n_pkts);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_store<MODE>), dim3(512), dim3(512), 0, 0, buf, n_pkts);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-16s %7.3f ms  %7.1f GB/s\n", name, ms, n_pkts * 48000.0 / ms / 1e6);
}

int main()
{
    const size_t n_pkts = 1000000;
    f4 *buf;
    CK(hipMalloc(&buf, n_pkts * 48000));
    run<0>("plain", buf, n_pkts);
    run<1>("nt", buf, n_pkts);
    run<2>("sc0", buf, n_pkts);
    run<3>("sc1", buf, n_pkts);
    run<4>("sc0 sc1", buf, n_pkts);
    run<5>("sc1 nt", buf, n_pkts);
    run<6>("sc0 sc1 nt", buf, n_pkts);
    run<0>("plain", buf, n_pkts);
    return 0;
}
