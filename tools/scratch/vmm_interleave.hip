// Micro-benchmark: if stores into two of the memory's three 96 GiB parts at once are faster than into one, what
// about an output buffer whose consecutive pieces ALTERNATE between the parts?  The device memory is taken in
// physical handles of CHUNK MiB through the virtual-memory API (in the order the driver hands them out, assumed -- and
// checked by the result -- to be ascending), and three 48 GB buffers are mapped from them:
//   one part    consecutive handles of the first part
//   two parts   pieces alternately from the first and the second part
//   three parts pieces from the three parts in turn
// then the wave-per-packet store pattern (48 000-byte packets, no reads) is timed into each.
// usage: vmm_interleave <chunk MiB>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(512) k_pkt(float4 *__restrict__ out, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t step = gridDim.x * 8;
    for (uint32_t p = blockIdx.x * 8 + wave; p < n_pkts; p += step) {
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
    }
}

// the same with every wave reading its packet's 1500 payload bytes first
__global__ void __launch_bounds__(512) k_pkt_rd(float4 *__restrict__ out, const uint32_t *__restrict__ pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t step = gridDim.x * 8;
    for (uint32_t p = blockIdx.x * 8 + wave; p < n_pkts; p += step) {
        float acc = 1.f;
        const uint32_t *q = pay + (size_t)p * 375;
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (k * 64 + lane < 375) acc += (float)q[k * 64 + lane];
        float4 *dst = out + (size_t)p * 3000;
        for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
    }
}

// the same with the NEXT packet's payload requested before this packet's stores and a counted wait
// (47 stores may still be outstanding when the loads are needed)
__global__ void __launch_bounds__(512) k_pkt_ahead(float4 *__restrict__ out, const uint32_t *__restrict__ pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t stride = gridDim.x * 8;
    uint32_t r0, r1, r2, r3, r4, r5;
    uint32_t p = blockIdx.x * 8 + wave;
#define LOADS(pp) do { const uint32_t *q = pay + (size_t)(pp) * 375 + lane;                                \
        asm volatile("global_load_dword %0, %1, off" : "=v"(r0) : "v"(q) : "memory");                       \
        asm volatile("global_load_dword %0, %1, off offset:256" : "=v"(r1) : "v"(q) : "memory");            \
        asm volatile("global_load_dword %0, %1, off offset:512" : "=v"(r2) : "v"(q) : "memory");            \
        asm volatile("global_load_dword %0, %1, off offset:768" : "=v"(r3) : "v"(q) : "memory");            \
        asm volatile("global_load_dword %0, %1, off offset:1024" : "=v"(r4) : "v"(q) : "memory");           \
        asm volatile("global_load_dword %0, %1, off offset:1280" : "=v"(r5) : "v"(q) : "memory"); } while (0)
    if (p < n_pkts) LOADS(p);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) :: "memory");   // the first request: a full wait
    for (; p < n_pkts; p += stride) {
        asm volatile("s_waitcnt vmcnt(47)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) :: "memory");
        const float acc = (float)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5);
        { const uint32_t pn = p + stride < n_pkts ? p + stride : p; LOADS(pn); }
        float4 *dst = out + (size_t)p * 3000;
#pragma unroll
        for (int c = 0; c < 46; c++) dst[c * 64 + lane] = make_float4(acc, 2.f, 3.f, (float)c);
        if (lane < 56) dst[46 * 64 + lane] = make_float4(acc, 2.f, 3.f, 46.f);
        else asm volatile("" ::: "memory");
    }
}

// the NEXT packet's payload requested one load at a time between this packet's stores (a load after every
// eighth store) instead of six loads in one burst; counted wait (the last load is followed by 6 stores)
__global__ void __launch_bounds__(512) k_pkt_sprinkled(float4 *__restrict__ out, const uint32_t *__restrict__ pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t stride = gridDim.x * 8;
    uint32_t r0, r1, r2, r3, r4, r5;
    uint32_t p = blockIdx.x * 8 + wave;
    if (p < n_pkts) LOADS(p);
    // (the registers the loads write must not be touched before the data is there: a full wait after the
    //  first request; in the loop the last load is followed by seven stores)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) :: "memory");
    for (; p < n_pkts; p += stride) {
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5) :: "memory");
        const float acc = (float)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5);
        const uint32_t pn = p + stride < n_pkts ? p + stride : p;
        const uint32_t *q = pay + (size_t)pn * 375 + lane;
        float4 *dst = out + (size_t)p * 3000;
#define ST8(c0) _Pragma("unroll") for (int c = c0; c < c0 + 8 && c < 46; c++) dst[c * 64 + lane] = make_float4(acc, 2.f, 3.f, (float)c); asm volatile("" ::: "memory")
        ST8(0);  asm volatile("global_load_dword %0, %1, off" : "=v"(r0) : "v"(q) : "memory");
        ST8(8);  asm volatile("global_load_dword %0, %1, off offset:256" : "=v"(r1) : "v"(q) : "memory");
        ST8(16); asm volatile("global_load_dword %0, %1, off offset:512" : "=v"(r2) : "v"(q) : "memory");
        ST8(24); asm volatile("global_load_dword %0, %1, off offset:768" : "=v"(r3) : "v"(q) : "memory");
        ST8(32); asm volatile("global_load_dword %0, %1, off offset:1024" : "=v"(r4) : "v"(q) : "memory");
        asm volatile("global_load_dword %0, %1, off offset:1280" : "=v"(r5) : "v"(q) : "memory");
        ST8(40);
        if (lane < 56) dst[46 * 64 + lane] = make_float4(acc, 2.f, 3.f, 46.f);
        else asm volatile("" ::: "memory");
    }
}

// every wave reads the payloads of FOUR neighbouring packets at once (6000 contiguous bytes), then writes the four
__global__ void __launch_bounds__(512) k_pkt_rd4(float4 *__restrict__ out, const uint32_t *__restrict__ pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t step = gridDim.x * 8 * 4;
    for (uint32_t p = (blockIdx.x * 8 + wave) * 4; p + 3 < n_pkts; p += step) {
        float acc = 1.f;
        const uint32_t *q = pay + (size_t)p * 375;
#pragma unroll
        for (int k = 0; k < 24; k++)
            if (k * 64 + lane < 1500) acc += (float)q[k * 64 + lane];
        for (int j = 0; j < 4; j++) {
            float4 *dst = out + (size_t)(p + j) * 3000;
            for (int i = lane; i < 3000; i += 64) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
        }
    }
}

int main(int argc, char **argv)
{
    const size_t MiB = 1ull << 20, GiB = 1ull << 30;
    const size_t chunk = (size_t)atoi(argv[1]) * MiB;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu, chunk %zu MiB\n", gran, chunk / MiB);
    size_t free_b, total_b;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t n_handles = (free_b - 6 * GiB) / chunk;
    std::vector<hipMemGenericAllocationHandle_t> h(n_handles);
    for (size_t k = 0; k < n_handles; k++) CK(hipMemCreate(&h[k], chunk, &prop, 0));
    printf("%zu handles = %.1f GiB\n", n_handles, n_handles * chunk / (double)GiB);
    const size_t per_part = 96 * GiB / chunk;                 // handles per 96 GiB
    const uint32_t n = 1000000;
    const size_t pieces = ((size_t)n * 48000 + chunk - 1) / chunk;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const char *names[3] = {"one part", "two parts alternating", "three parts in turn"};
    void *bufs[3];
    size_t used[3] = {per_part / 8, per_part / 8, per_part / 8};   // next free handle of each part (skip the first 12 GiB)
    for (int b = 0; b < 3; b++) {
        CK(hipMemAddressReserve(&bufs[b], pieces * chunk, chunk, nullptr, 0));
        for (size_t i = 0; i < pieces; i++) {
            const int part = b == 0 ? 0 : (int)(i % (b + 1));
            const size_t k = part * per_part + used[part]++;
            if (k >= n_handles || (part < 2 && used[part] > per_part)) { printf("out of handles in part %d\n", part); return 1; }
            CK(hipMemMap((char *)bufs[b] + i * chunk, chunk, 0, h[k], 0));
        }
        CK(hipMemSetAccess(bufs[b], pieces * chunk, &acc, 1));
    }
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 4096)); CK(hipMemset(pay, 1, (size_t)n * 1500 + 4096));
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    for (int rep = 0; rep < 2; rep++)
        for (int b = 0; b < 3; b++) {
            hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], n);
            CK(hipEventRecord(ev1));
            CK(hipEventSynchronize(ev1));
            float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
            printf("%-24s %.3f ms %.2f TB/s", names[b], ms / 3, 48e9 / (ms / 3) / 1e9);
            hipLaunchKernelGGL(k_pkt_rd, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pkt_rd, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipEventRecord(ev1));
            CK(hipEventSynchronize(ev1));
            CK(hipEventElapsedTime(&ms, ev0, ev1));
            printf(" | with the payload reads %.3f ms", ms / 3);
            hipLaunchKernelGGL(k_pkt_ahead, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pkt_ahead, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipEventRecord(ev1));
            CK(hipEventSynchronize(ev1));
            CK(hipEventElapsedTime(&ms, ev0, ev1));
            printf(" | reads one packet ahead %.3f ms", ms / 3);
            hipLaunchKernelGGL(k_pkt_rd4, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pkt_rd4, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipEventRecord(ev1));
            CK(hipEventSynchronize(ev1));
            CK(hipEventElapsedTime(&ms, ev0, ev1));
            printf(" | four payloads per read %.3f ms", ms / 3);
            hipLaunchKernelGGL(k_pkt_sprinkled, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pkt_sprinkled, dim3(512), dim3(512), 0, 0, (float4 *)bufs[b], pay, n);
            CK(hipEventRecord(ev1));
            CK(hipEventSynchronize(ev1));
            CK(hipEventElapsedTime(&ms, ev0, ev1));
            printf(" | next payload one load per 8 stores %.3f ms\n", ms / 3);
            fflush(stdout);
        }
    return 0;
}
