// Micro-benchmark: wave-per-packet store pattern (each wave sweeps its own 48000-byte region in
// 5 KiB pieces, with dummy ALU work in between) vs. the streaming pattern.  Timing experiments only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

// WPB waves per block; work = dummy FMA iterations per 5 KiB piece; pre = dummy iterations before the packet
template <int WPB, bool READ>
__global__ void __launch_bounds__(64 * WPB) k_pkt(float4 *out, const uint32_t *pay, uint32_t n_pkts, int work, int pre)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = (float)lane;
    for (uint32_t p = blockIdx.x * WPB + wave; p < n_pkts; p += gridDim.x * WPB) {
        if (READ) {
            const uint32_t *q = pay + (size_t)p * 375;
            for (int i = lane; i < 375; i += 64) acc += (float)q[i];
        }
        for (int k = 0; k < pre; k++) acc = fmaf(acc, 1.0001f, 0.5f);
        float4 *dst = out + (size_t)p * 3000;
        for (int c = 0; c < 10; c++) {
            for (int k = 0; k < work; k++) acc = fmaf(acc, 1.0001f, 0.5f);
            const int n = c < 9 ? 320 : 120;
#pragma unroll
            for (int i = 0; i < 5; i++)
                if (i * 64 + lane < n) dst[c * 320 + i * 64 + lane] = make_float4(acc, 2.f, 3.f, (float)i);
        }
    }
}

template <int WPB, bool READ>
void run(const char *name, float4 *buf, uint32_t *pay, uint32_t n, int grid, int work, int pre)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_pkt<WPB, READ>), dim3(grid), dim3(64 * WPB), 0, 0, buf, pay, n, work, pre);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_pkt<WPB, READ>), dim3(grid), dim3(64 * WPB), 0, 0, buf, pay, n, work, pre);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-34s grid %5d work %4d pre %5d: %7.3f ms  %7.1f GB/s\n", name, grid, work, pre, ms, n * 48000.0 / ms / 1e6);
}

int main()
{
    const uint32_t n = 1000000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 64)); CK(hipMemset(pay, 1, (size_t)n * 1500));
    run<4, false>("4 waves/block, no read", buf, pay, n, 1024, 0, 0);
    run<4, false>("4 waves/block, no read", buf, pay, n, 1024, 100, 1500);
    run<4, true>("4 waves/block, read", buf, pay, n, 1024, 0, 0);
    run<4, true>("4 waves/block, read", buf, pay, n, 1024, 100, 1500);
    run<8, true>("8 waves/block, read", buf, pay, n, 512, 100, 1500);
    run<8, true>("8 waves/block, read", buf, pay, n, 512, 200, 3000);
    run<8, false>("8 waves/block, no read", buf, pay, n, 512, 200, 3000);
    run<8, true>("8 waves/block, read, 8 w/CU", buf, pay, n, 256, 100, 1500);
    run<4, true>("4 w/b, read, 32 w/CU", buf, pay, n, 2048, 100, 1500);
    return 0;
}
