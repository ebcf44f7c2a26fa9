// Micro-benchmark: does the KIND of memory the payloads live in change what their reads cost beside the 48 GB
// of stores?  Wave per packet as in pkt_bw4 (8 waves per workgroup, fixed grid of 512), each wave reads its
// packet's 1500 bytes and writes 48 000.
//   payload buffer: hipMalloc | hipExtMallocWithFlags(uncached) | hipExtMallocWithFlags(fine-grained)
//   reads: vector loads (dword per lane) | scalar loads (wave-uniform address, data arrives in SGPRs)
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
        if (MODE == 2) {
            // wave-uniform address: the compiler uses scalar loads
            const uint32_t pu = __builtin_amdgcn_readfirstlane(p);
            const uint32_t *q = pay + (size_t)pu * 375;
            uint32_t t = 0;
#pragma unroll
            for (int k = 0; k < 368; k++) t += q[k];
            acc += (float)t;
        }
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
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
    printf("%-72s %.3f ms\n", name, ms / 5);
    fflush(stdout);
}

int main()
{
    const uint32_t n = 1000000;
    const size_t pay_bytes = (size_t)n * 1500 + 4096;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay[3] = {nullptr, nullptr, nullptr};
    const char *kind[3] = {"hipMalloc", "uncached", "fine-grained"};
    CK(hipMalloc(&pay[0], pay_bytes));
    if (hipExtMallocWithFlags((void **)&pay[1], pay_bytes, hipDeviceMallocUncached) != hipSuccess) { pay[1] = nullptr; (void)hipGetLastError(); }
    if (hipExtMallocWithFlags((void **)&pay[2], pay_bytes, hipDeviceMallocFinegrained) != hipSuccess) { pay[2] = nullptr; (void)hipGetLastError(); }
    for (int i = 0; i < 3; i++)
        if (pay[i]) CK(hipMemset(pay[i], 1, pay_bytes));
    CK(hipDeviceSynchronize());
    char name[128];
    for (int rep = 0; rep < 2; rep++) {
        timed("no reads", [&] { hipLaunchKernelGGL((k_pkt<0>), dim3(512), dim3(512), 0, 0, buf, pay[0], n); });
        for (int i = 0; i < 3; i++) {
            if (!pay[i]) { printf("%s: allocation refused\n", kind[i]); continue; }
            snprintf(name, sizeof name, "vector loads, payloads in %s memory", kind[i]);
            timed(name, [&] { hipLaunchKernelGGL((k_pkt<1>), dim3(512), dim3(512), 0, 0, buf, pay[i], n); });
            snprintf(name, sizeof name, "scalar loads, payloads in %s memory", kind[i]);
            timed(name, [&] { hipLaunchKernelGGL((k_pkt<2>), dim3(512), dim3(512), 0, 0, buf, pay[i], n); });
        }
    }
    return 0;
}
