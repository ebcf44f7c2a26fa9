// Micro-benchmark: does the cache policy of the STORES change what the payload reads cost beside them?
// Wave per packet, 8 waves per workgroup, grid 512; each wave reads its 1500 payload bytes, then writes 48 000.
// POL 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0, 5 nt sc1.  READ_POL 0 plain, 1 nt, 2 sc1, 3 sc0 sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ void store16(float4 *p, f4 v)
{
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
}
template <int RP>
__device__ __forceinline__ uint32_t load4(const uint32_t *p)
{
    uint32_t v;
    if (RP == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (RP == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (RP == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (RP == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int POL, int RP, bool READ>
__global__ void __launch_bounds__(512) k_pkt(float4 *out, const uint32_t *pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t step = gridDim.x * 8;
    for (uint32_t p = blockIdx.x * 8 + wave; p < n_pkts; p += step) {
        float acc = 1.f;
        if (READ) {
            const uint32_t *q = pay + (size_t)p * 375;
            uint32_t w[6];
#pragma unroll
            for (int k = 0; k < 6; k++) w[k] = load4<RP>(q + (k * 64 + lane < 375 ? k * 64 + lane : 0));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 6; k++) acc += (float)w[k];
        }
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) {
            f4 v = {acc, 2.f, 3.f, (float)i};
            store16<POL>(dst + i, v);
        }
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
    printf("%-56s %.3f ms\n", name, ms / 5);
    fflush(stdout);
}

int main()
{
    const uint32_t n = 1000000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 4096)); CK(hipMemset(pay, 1, (size_t)n * 1500));
#define RUN(POL, RP, RD, NAME) timed(NAME, [&] { hipLaunchKernelGGL((k_pkt<POL, RP, RD>), dim3(512), dim3(512), 0, 0, buf, pay, n); })
    RUN(0, 0, false, "stores plain, no reads");
    RUN(0, 0, true, "stores plain, reads plain");
    RUN(1, 0, true, "stores nt, reads plain");
    RUN(2, 0, true, "stores sc1, reads plain");
    RUN(3, 0, true, "stores sc0 sc1, reads plain");
    RUN(4, 0, true, "stores sc0, reads plain");
    RUN(5, 0, true, "stores nt sc1, reads plain");
    RUN(0, 1, true, "stores plain, reads nt");
    RUN(0, 2, true, "stores plain, reads sc1");
    RUN(0, 3, true, "stores plain, reads sc0 sc1");
    RUN(1, 1, true, "stores nt, reads nt");
    RUN(3, 3, true, "stores sc0 sc1, reads sc0 sc1");
    RUN(1, 0, false, "stores nt, no reads");
    RUN(3, 0, false, "stores sc0 sc1, no reads");
    return 0;
}
