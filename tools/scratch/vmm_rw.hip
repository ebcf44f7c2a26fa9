// Micro-benchmark: a read-and-write stream with the modulator's proportions (1536 bytes read, 2560 written per
// OFDM symbol of N = 256, cp = 64) by where the input and the output lie among the memory's three 96 GiB parts.
// The device memory is taken in 64 MiB handles through the virtual-memory API (handed out in ascending order) and
// buffers are mapped from chosen parts:
//   outputs: one part | parts 0 and 1 alternating | all three in turn
//   inputs : part 0 | part 2 | all three in turn
// usage: vmm_rw [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// one wave per group of 4 symbols: 4 x 1536 bytes in (96 float4 per symbol), 4 x 2560 bytes out (160 float4 per symbol)
__global__ void __launch_bounds__(256) k_rw(const float4 *__restrict__ in, float4 *__restrict__ out, uint32_t n_groups)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
        const float4 *src = in + (size_t)g * 384;
        float4 v[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = src[k * 64 + lane];
        float4 *dst = out + (size_t)g * 640;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const float4 a = v[k % 6];
            __builtin_nontemporal_store(a.x + (float)k, &dst[k * 64 + lane].x);
            __builtin_nontemporal_store(a.y, &dst[k * 64 + lane].y);
            __builtin_nontemporal_store(a.z, &dst[k * 64 + lane].z);
            __builtin_nontemporal_store(a.w, &dst[k * 64 + lane].w);
        }
    }
}

// the same bytes with the modulator's read pattern: 16 lanes per symbol, lane l reads the 8-byte values l + 16 m
// (m = 0..11) of its symbol -- 128 contiguous bytes per symbol and load instruction
__global__ void __launch_bounds__(256) k_rw8(const float2 *__restrict__ in, float4 *__restrict__ out, uint32_t n_groups)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, sym = lane >> 4;
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
        const float2 *src = in + ((size_t)g * 4 + sym) * 192;
        float2 v[12];
#pragma unroll
        for (int m = 0; m < 12; m++) v[m] = src[l + 16 * m];
        float4 *dst = out + (size_t)g * 640;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const float2 a = v[k], b = v[(k + 1) % 12];
            __builtin_nontemporal_store(a.x + (float)k, &dst[k * 64 + lane].x);
            __builtin_nontemporal_store(a.y, &dst[k * 64 + lane].y);
            __builtin_nontemporal_store(b.x, &dst[k * 64 + lane].z);
            __builtin_nontemporal_store(b.y + v[10].x + v[11].y, &dst[k * 64 + lane].w);
        }
    }
}

int main(int argc, char **argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 2048;                // workgroups of 4 waves: 2048 = 32 waves per CU
    printf("grid %d\n", grid);
    const size_t MiB = 1ull << 20, GiB = 1ull << 30, chunk = 64 * MiB;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t free_b, total_b;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t n_handles = (free_b - 6 * GiB) / chunk;
    std::vector<hipMemGenericAllocationHandle_t> h(n_handles);
    for (size_t k = 0; k < n_handles; k++) CK(hipMemCreate(&h[k], chunk, &prop, 0));
    printf("%zu handles = %.1f GiB\n", n_handles, n_handles * chunk / (double)GiB);
    const size_t per_part = 96 * GiB / chunk;
    const uint32_t n_groups = 1u << 20;                             // 4 Mi symbols: 6.4 GB in, 10.7 GB out
    const size_t in_bytes = (size_t)n_groups * 384 * 16, out_bytes = (size_t)n_groups * 640 * 16;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t used[3] = {per_part / 8, per_part / 8, per_part / 8};
    auto build = [&](size_t bytes, int n_parts, const int *parts) -> void * {
        const size_t pieces = (bytes + chunk - 1) / chunk;
        void *va;
        CK(hipMemAddressReserve(&va, pieces * chunk, chunk, nullptr, 0));
        for (size_t i = 0; i < pieces; i++) {
            const int part = parts[i % n_parts];
            const size_t k = part * per_part + used[part]++;
            if (k >= n_handles) { printf("out of handles in part %d\n", part); exit(1); }
            CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[k], 0));
        }
        CK(hipMemSetAccess(va, pieces * chunk, &acc, 1));
        return va;
    };
    const int p0[1] = {0}, p2[1] = {2}, p01[2] = {0, 1}, p012[3] = {0, 1, 2};
    const char *out_names[3] = {"output in part 0", "output in parts 0, 1 alternating", "output in parts 0, 1, 2 in turn"};
    const char *in_names[3] = {"input in part 0", "input in part 2", "input in parts 0, 1, 2 in turn"};
    void *outs[3] = {build(out_bytes, 1, p0), build(out_bytes, 2, p01), build(out_bytes, 3, p012)};
    void *ins[3] = {build(in_bytes, 1, p0), build(in_bytes, 1, p2), build(in_bytes, 3, p012)};
    for (int i = 0; i < 3; i++) CK(hipMemset(ins[i], 1, in_bytes));
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    for (int rep = 0; rep < 2; rep++)
        for (int o = 0; o < 3; o++)
            for (int i = 0; i < 3; i++) {
                hipLaunchKernelGGL(k_rw, dim3(grid), dim3(256), 0, 0, (const float4 *)ins[i], (float4 *)outs[o], n_groups);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(ev0));
                for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_rw, dim3(grid), dim3(256), 0, 0, (const float4 *)ins[i], (float4 *)outs[o], n_groups);
                CK(hipEventRecord(ev1));
                CK(hipEventSynchronize(ev1));
                float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
                printf("%-34s %-32s %.3f ms %.2f TB/s", out_names[o], in_names[i], ms / 3, (in_bytes + out_bytes) / (ms / 3) / 1e9);
                hipLaunchKernelGGL(k_rw8, dim3(grid), dim3(256), 0, 0, (const float2 *)ins[i], (float4 *)outs[o], n_groups);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(ev0));
                for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_rw8, dim3(grid), dim3(256), 0, 0, (const float2 *)ins[i], (float4 *)outs[o], n_groups);
                CK(hipEventRecord(ev1));
                CK(hipEventSynchronize(ev1));
                CK(hipEventElapsedTime(&ms, ev0, ev1));
                printf(" | 8-byte reads, 16 lanes per symbol: %.3f ms %.2f TB/s\n", ms / 3, (in_bytes + out_bytes) / (ms / 3) / 1e9);
                fflush(stdout);
            }
    return 0;
}
