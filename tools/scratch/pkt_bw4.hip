// Micro-benchmark: what do the 1.5 GB of payload reads cost beside the 48 GB of stores, and does it help when the
// reads find their data in the memory-side cache?  Wave per packet, 8 waves per workgroup, fixed grid.
//   MODE 0  no reads
//   MODE 1  each wave reads its packet's 1500 bytes (the real thing)
//   MODE 2  the same reads folded onto a 32 MB region (cache-resident: what a perfect prefetch would give)
//   MODE 4  as 2 with a 1 MB region (L2-resident)
//   MODE 5  as 1, but the loaded words are first used after the packet's stores (no store waits for a load)
//   MODE 3  as 1, and one wave in every 16th workgroup first touches, in one burst, the 192 KB that those 16
//           workgroups will read in their NEXT round (one 64-byte line per lane and step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k_pkt(float4 *out, const uint32_t *pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t step = gridDim.x * 8;
    float carry = 0.f;
    for (uint32_t p = blockIdx.x * 8 + wave; p < n_pkts; p += step) {
        float acc = 1.f + carry;
        if (MODE == 3 && wave == 0 && blockIdx.x % 16 == 0) {
            // next round's payloads of workgroups blockIdx.x .. +15: packets p + step .. p + step + 127
            const uint32_t p0 = p + step;
            if (p0 + 128 <= n_pkts) {
                const uint32_t *q = pay + (size_t)p0 * 375;
                uint32_t t = 0;
                for (int i = lane * 16; i < 128 * 375; i += 64 * 16) t += __builtin_nontemporal_load(q + i);
                acc += (float)(t & 1);
            }
        }
        uint32_t w[6] = {0, 0, 0, 0, 0, 0};
        if (MODE >= 1) {
            const uint32_t pp = MODE == 2 ? p % 21845u : MODE == 4 ? p % 682u : p;
            const uint32_t *q = pay + (size_t)pp * 375;
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k * 64 + lane < 375) w[k] = q[k * 64 + lane];
            if (MODE != 5)
                for (int k = 0; k < 6; k++) acc += (float)w[k];
        }
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
        if (MODE == 5) {
            for (int k = 0; k < 6; k++) carry += (float)w[k];
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
    ms /= 5;
    printf("%-72s %.3f ms\n", name, ms);
    fflush(stdout);
}

int main()
{
    const uint32_t n = 1000000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 4096)); CK(hipMemset(pay, 1, (size_t)n * 1500));
    for (int g : {512, 256}) {
        printf("grid %d\n", g);
        timed("no reads", [&] { hipLaunchKernelGGL((k_pkt<0>), dim3(g), dim3(512), 0, 0, buf, pay, n); });
        timed("each wave reads its payload", [&] { hipLaunchKernelGGL((k_pkt<1>), dim3(g), dim3(512), 0, 0, buf, pay, n); });
        timed("reads folded onto 32 MB", [&] { hipLaunchKernelGGL((k_pkt<2>), dim3(g), dim3(512), 0, 0, buf, pay, n); });
        timed("reads folded onto 1 MB", [&] { hipLaunchKernelGGL((k_pkt<4>), dim3(g), dim3(512), 0, 0, buf, pay, n); });
        timed("payload words first used after the packet's stores", [&] { hipLaunchKernelGGL((k_pkt<5>), dim3(g), dim3(512), 0, 0, buf, pay, n); });
        if (g == 512) timed("next round's payloads touched in 192 KB bursts first", [&] { hipLaunchKernelGGL((k_pkt<3>), dim3(g), dim3(512), 0, 0, buf, pay, n); });
    }
    return 0;
}
