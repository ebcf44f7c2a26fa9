// Micro-benchmark: what about "one workgroup, one 16-byte store per thread, done" makes it write at 6.9 TB/s?
// Variants: pages in a scrambled order; a delay before / after the store; two stores with a wait between.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

// MODE 0: page = workgroup index; 1: pages scrambled within blocks of 2^SPAN pages (multiplicative hash, a bijection)
template <int MODE, int SPAN>
__global__ void __launch_bounds__(256) k_page(float4 *out, size_t n_pages)
{
    size_t c = blockIdx.x;
    if (MODE == 1) {
        const size_t m = ((size_t)1 << SPAN) - 1;
        c = (c & ~m) | ((c * 40503u + 12345u) & m);          // odd multiplier: a permutation of the low SPAN bits
    }
    if (c < n_pages) out[c * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// DELAY: 0 none, 1 sleep before the store, 2 sleep after the store
template <int DELAY>
__global__ void __launch_bounds__(256) k_page_delay(float4 *out, size_t n_pages)
{
    const size_t c = blockIdx.x;
    if (DELAY == 1) __builtin_amdgcn_s_sleep(20);
    out[c * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    if (DELAY == 2) __builtin_amdgcn_s_sleep(20);
}

// two pages per workgroup; WAIT: acknowledgement of the first store awaited before the second
template <bool WAIT>
__global__ void __launch_bounds__(256) k_two(float4 *out, size_t n_pages)
{
    const size_t c = (size_t)blockIdx.x * 2;
    out[c * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(c + 1) * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 5.f);
}

// two pages per workgroup of 512 threads: the upper half writes the second page (one store per thread)
__global__ void __launch_bounds__(512) k_two_wide(float4 *out, size_t n_pages)
{
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// 8 bytes per lane: a workgroup of 256 writes 2 KB
__global__ void __launch_bounds__(256) k_half(float2 *out, size_t n)
{
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = make_float2(1.f, 2.f);
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
    printf("%-64s %.3f ms %.2f TB/s\n", name, ms, 48e9 / ms / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t n_pages = (48000000000ull >> 26) << 14, bytes = n_pages * 4096;     // multiple of 2^14 pages
    float4 *buf; CK(hipMalloc(&buf, bytes));
    printf("%zu bytes (rates below are quoted on 48e9)\n", bytes);
    const unsigned G = (unsigned)n_pages;
#define L(K, G_, T, ...) hipLaunchKernelGGL((K), dim3(G_), dim3(T), 0, 0, __VA_ARGS__)
    timed("page = workgroup", [&] { L((k_page<0, 0>), G, 256, buf, n_pages); });
    timed("pages scrambled within 32 KB (2^3)", [&] { L((k_page<1, 3>), G, 256, buf, n_pages); });
    timed("pages scrambled within 1 MB (2^8)", [&] { L((k_page<1, 8>), G, 256, buf, n_pages); });
    timed("pages scrambled within 8 MB (2^11)", [&] { L((k_page<1, 11>), G, 256, buf, n_pages); });
    timed("pages scrambled within 64 MB (2^14)", [&] { L((k_page<1, 14>), G, 256, buf, n_pages); });
    timed("sleep before the store", [&] { L((k_page_delay<1>), G, 256, buf, n_pages); });
    timed("sleep after the store", [&] { L((k_page_delay<2>), G, 256, buf, n_pages); });
    timed("two pages per workgroup of 256", [&] { L((k_two<false>), G / 2, 256, buf, n_pages); });
    timed("two pages per workgroup of 256, wait between", [&] { L((k_two<true>), G / 2, 256, buf, n_pages); });
    timed("two pages per workgroup of 512, one store per thread", [&] { L((k_two_wide), G / 2, 512, buf, n_pages); });
    timed("8 bytes per lane, 2 KB per workgroup of 256", [&] { L((k_half), G * 2, 256, (float2 *)buf, n_pages); });
    timed("page = workgroup", [&] { L((k_page<0, 0>), G, 256, buf, n_pages); });
    return 0;
}
