// Micro-benchmark: packets finished in LDS and committed in packet order.  One workgroup per packet (launched in
// order, 48 KB of LDS each, three to a CU): it "computes" for a while (sleep), waits until all but the last WINDOW
// packets before it have started their stores (a global counter), then writes its 48 000 bytes in one burst from
// all threads.  Does the chip then write at the 6.8 TB/s of a compact window, or at the 5.65 TB/s of independent
// packet streams?  10^6 packets.  Timing experiments only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

template <int T, bool ORDERED, bool READ>
__global__ void __launch_bounds__(T) k_commit(float4 *out, const uint32_t *pay, uint32_t n_pkts, uint32_t *started, uint32_t window,
                                              int sleep_units)
{
    extern __shared__ float4 s_iq[];                       // 3000 float4
    const uint32_t p = blockIdx.x;
    float acc = 1.f;
    if (READ) {
        const uint32_t *q = pay + (size_t)p * 375;
        for (int i = threadIdx.x; i < 375; i += T) acc += (float)q[i];
    }
    for (int i = threadIdx.x; i < 3000; i += T) s_iq[i] = make_float4(acc, 2.f, 3.f, (float)i);
    for (int k = 0; k < sleep_units; k++) __builtin_amdgcn_s_sleep(127);      // 127 * 64 cycles ~ 3.9 us each
    __syncthreads();
    if (ORDERED) {
        if (threadIdx.x == 0 && p > window) {
            const uint32_t need = p - window;
            while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
    }
    float4 *dst = out + (size_t)p * 3000;
    for (int i = threadIdx.x; i < 3000; i += T) dst[i] = s_iq[i];
    if (ORDERED && threadIdx.x == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename F>
void timed(const char *name, uint32_t *ctr, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(ctr, 0, 4));
    launch();
    CK(hipDeviceSynchronize());
    float tot = 0;
    for (int r = 0; r < 3; r++) {
        CK(hipMemset(ctr, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    CK(hipGetLastError());
    printf("%-72s %.3f ms %.2f TB/s\n", name, tot / 3, 48e9 / (tot / 3) / 1e9);
    fflush(stdout);
}

int main()
{
    const uint32_t n = 1000000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 4096)); CK(hipMemset(pay, 1, (size_t)n * 1500));
    uint32_t *ctr; CK(hipMalloc(&ctr, 256));
    const size_t lds = 48000;
#define RUN(T, ORD, RD, W, SL, NAME) timed(NAME, ctr, [&] { \
        hipLaunchKernelGGL((k_commit<T, ORD, RD>), dim3(n), dim3(T), lds, 0, buf, pay, n, ctr, (uint32_t)(W), SL); })
    RUN(1024, false, false, 0, 0, "T=1024, no order, no compute");
    RUN(1024, false, false, 0, 1, "T=1024, no order, 3.9 us compute");
    RUN(1024, true, false, 16, 0, "T=1024, window 16, no compute");
    RUN(1024, true, false, 32, 0, "T=1024, window 32, no compute");
    RUN(1024, true, false, 64, 0, "T=1024, window 64, no compute");
    RUN(1024, true, false, 128, 0, "T=1024, window 128, no compute");
    RUN(1024, true, false, 256, 0, "T=1024, window 256, no compute");
    RUN(1024, true, false, 32, 1, "T=1024, window 32, 3.9 us compute");
    RUN(1024, true, false, 64, 1, "T=1024, window 64, 3.9 us compute");
    RUN(1024, true, false, 128, 1, "T=1024, window 128, 3.9 us compute");
    RUN(1024, true, true, 64, 1, "T=1024, window 64, 3.9 us compute, payload read");
    RUN(1024, false, true, 0, 1, "T=1024, no order, 3.9 us compute, payload read");
    RUN(512, true, false, 64, 1, "T=512, window 64, 3.9 us compute");
    RUN(512, false, false, 0, 1, "T=512, no order, 3.9 us compute");
    RUN(256, true, false, 64, 1, "T=256, window 64, 3.9 us compute");
    return 0;
}
