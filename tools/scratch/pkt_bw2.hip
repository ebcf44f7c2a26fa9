// Micro-benchmark: wave-per-packet store pattern with the NEXT packet's payload prefetched
// before the current packet's stores.  MODE 0: load at packet start and wait (baseline);
// MODE 1: prefetch, wait vmcnt(0) at the next packet start; MODE 2: prefetch, counted wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_pkt_x4(float4 *out, const uint32_t *pay, uint32_t n_pkts, int work)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = (float)lane;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t stride = gridDim.x * 4;
    for (uint32_t p = blockIdx.x * 4 + wave; p < n_pkts; p += stride) {
        // 1500 B = 93.75 x 16 B; packet base is only 4-byte aligned: round down to 16
        const uintptr_t a = (uintptr_t)(pay + (size_t)p * 375) & ~(uintptr_t)15;
        const u4 *q = (const u4 *)a + lane;
        u4 v0 = q[0], v1 = {0, 0, 0, 0};
        if (lane < 31) v1 = q[64];
        acc += (float)(v0.x ^ v0.y ^ v0.z ^ v0.w ^ v1.x ^ v1.y ^ v1.z ^ v1.w);
        float4 *dst = out + (size_t)p * 3000;
        for (int c = 0; c < 10; c++) {
            for (int k = 0; k < work; k++) acc = fmaf(acc, 1.0001f, 0.5f);
            if (c < 9) {
#pragma unroll
                for (int i = 0; i < 5; i++) dst[c * 320 + i * 64 + lane] = make_float4(acc, 2.f, 3.f, (float)i);
            } else {
                dst[c * 320 + lane] = make_float4(acc, 2.f, 3.f, 0.f);
                if (lane < 56) dst[c * 320 + 64 + lane] = make_float4(acc, 2.f, 3.f, 1.f);
            }
        }
    }
}

// one wave of the block loads the payloads of the block's 4 packets (6000 contiguous bytes) into LDS
__global__ void __launch_bounds__(256) k_pkt_blk(float4 *out, const uint32_t *pay, uint32_t n_pkts, int work)
{
    __shared__ uint32_t s_pay[2][1504];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = (float)lane;
    const uint32_t stride = gridDim.x * 4;
    int it = 0;
    for (uint32_t p0 = blockIdx.x * 4; p0 < n_pkts; p0 += stride, it ^= 1) {
        if (wave == 0) {
            const uint32_t *q = pay + (size_t)p0 * 375;
            for (int i = lane; i < 1500; i += 64) s_pay[it][i] = q[i];
        }
        __syncthreads();
        const uint32_t p = p0 + wave;
        for (int i = lane; i < 375; i += 64) acc += (float)s_pay[it][wave * 375 + i];
        float4 *dst = out + (size_t)p * 3000;
        for (int c = 0; c < 10; c++) {
            for (int k = 0; k < work; k++) acc = fmaf(acc, 1.0001f, 0.5f);
            if (c < 9) {
#pragma unroll
                for (int i = 0; i < 5; i++) dst[c * 320 + i * 64 + lane] = make_float4(acc, 2.f, 3.f, (float)i);
            } else {
                dst[c * 320 + lane] = make_float4(acc, 2.f, 3.f, 0.f);
                if (lane < 56) dst[c * 320 + 64 + lane] = make_float4(acc, 2.f, 3.f, 1.f);
            }
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_pkt(float4 *out, const uint32_t *pay, uint32_t n_pkts, int work)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = (float)lane;
    uint32_t r0, r1, r2, r3, r4, r5;
    const uint32_t stride = gridDim.x * 4;
    uint32_t p = blockIdx.x * 4 + wave;
#define LOADS(pp) do { const uint32_t *q = pay + (size_t)(pp) * 375 + lane;                                \
        asm volatile("global_load_dword %0, %1, off" : "=v"(r0) : "v"(q) : "memory");                       \
        asm volatile("global_load_dword %0, %1, off offset:256" : "=v"(r1) : "v"(q) : "memory");            \
        asm volatile("global_load_dword %0, %1, off offset:512" : "=v"(r2) : "v"(q) : "memory");            \
        asm volatile("global_load_dword %0, %1, off offset:768" : "=v"(r3) : "v"(q) : "memory");            \
        asm volatile("global_load_dword %0, %1, off offset:1024" : "=v"(r4) : "v"(q) : "memory");           \
        asm volatile("global_load_dword %0, %1, off offset:1280" : "=v"(r5) : "v"(q) : "memory"); } while (0)
    if (MODE != 0 && p < n_pkts) LOADS(p);
    for (; p < n_pkts; p += stride) {
        if (MODE == 0) LOADS(p);
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(47)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) :: "memory");
        acc += (float)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5);
        for (int k = 0; k < work * 10; k++) acc = fmaf(acc, 1.0001f, 0.5f);
        if (MODE != 0) { const uint32_t pn = p + stride < n_pkts ? p + stride : p; LOADS(pn); }
        float4 *dst = out + (size_t)p * 3000;
        for (int c = 0; c < 10; c++) {
            for (int k = 0; k < work; k++) acc = fmaf(acc, 1.0001f, 0.5f);
            if (c < 9) {
#pragma unroll
                for (int i = 0; i < 5; i++) dst[c * 320 + i * 64 + lane] = make_float4(acc, 2.f, 3.f, (float)i);
            } else {
                dst[c * 320 + lane] = make_float4(acc, 2.f, 3.f, 0.f);
                if (lane < 56) dst[c * 320 + 64 + lane] = make_float4(acc, 2.f, 3.f, 1.f);
                else asm volatile("" ::: "memory");
            }
        }
    }
}

template <int MODE>
void run(const char *name, float4 *buf, uint32_t *pay, uint32_t n, int grid, int work)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_pkt<MODE>), dim3(grid), dim3(256), 0, 0, buf, pay, n, work);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_pkt<MODE>), dim3(grid), dim3(256), 0, 0, buf, pay, n, work);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-34s grid %5d work %4d: %7.3f ms  %7.1f GB/s\n", name, grid, work, ms, n * 48000.0 / ms / 1e6);
}

int main()
{
    const uint32_t n = 1000000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 4096)); CK(hipMemset(pay, 1, (size_t)n * 1500));
    for (int r = 0; r < 2; r++) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        for (int k = 0; k < 3; k++) { if (r == 0) hipLaunchKernelGGL(k_pkt_x4, dim3(1024), dim3(256), 0, 0, buf, pay, n, 0); else hipLaunchKernelGGL(k_pkt_blk, dim3(1024), dim3(256), 0, 0, buf, pay, n, 0); }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("%s: %.3f ms\n", r == 0 ? "dwordx4 loads" : "block-batched loads", ms / 3);
    }
    for (int work : {0}) {
        run<0>("load at start, wait all", buf, pay, n, 1024, work);
        run<1>("prefetch, wait vmcnt(0)", buf, pay, n, 1024, work);
        run<2>("prefetch, counted wait", buf, pay, n, 1024, work);
    }
    return 0;
}
