// Micro-benchmark: streaming-store patterns on a 48 GB buffer (timing experiments only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

// each wave writes `CH` x 1 KiB contiguous per iteration; groups of consecutive chunks go to the
// waves of a block; MODE 0 plain, 1 nontemporal, 2 via LDS read
template <int CH, int MODE>
__global__ void __launch_bounds__(256) k_store(float4 *out, size_t n_chunks /* of CH KiB */)
{
    __shared__ float4 s[4][64 * CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < CH; i++) s[wave][i * 64 + lane] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t c = (size_t)blockIdx.x * 4 + wave; c < n_chunks; c += stride) {
        float4 *dst = out + c * (64 * CH);
#pragma unroll
        for (int i = 0; i < CH; i++) {
            float4 v = MODE == 2 ? s[wave][i * 64 + lane] : make_float4(1.f, 2.f, 3.f, (float)i);
            if (MODE == 1) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 w = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(w, reinterpret_cast<f4 *>(dst + i * 64 + lane));
            }
            else dst[i * 64 + lane] = v;
        }
        if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int CH, int MODE>
void run(const char *name, float4 *buf, size_t bytes, int grid)
{
    size_t n_chunks = bytes / (1024 * CH);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_store<CH, MODE>), dim3(grid), dim3(256), 0, 0, buf, n_chunks);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_store<CH, MODE>), dim3(grid), dim3(256), 0, 0, buf, n_chunks);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-28s grid %6d: %7.3f ms  %7.1f GB/s\n", name, grid, ms, bytes / ms / 1e6);
}

// occupancy-limited variant: PAD bytes of extra LDS per block; MODE2-like with LDS rewrite each iteration
template <int PADKB>
__global__ void __launch_bounds__(256) k_store_occ(float4 *out, size_t n_chunks)
{
    __shared__ float4 s[4][64 * 5];
    __shared__ float pad[PADKB * 256 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (out == nullptr) pad[threadIdx.x] = 1.f;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t c = (size_t)blockIdx.x * 4 + wave; c < n_chunks; c += stride) {
        float4 *dst = out + c * 320;
#pragma unroll
        for (int i = 0; i < 5; i++) s[wave][i * 64 + lane] = make_float4(1.f, 2.f, (float)c, (float)i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 5; i++) dst[i * 64 + lane] = s[wave][i * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (out == nullptr) out[0].x = pad[lane];
}
template <int PADKB>
void run_occ(float4 *buf, size_t bytes, int grid)
{
    size_t n_chunks = bytes / 5120;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_store_occ<PADKB>), dim3(grid), dim3(256), 0, 0, buf, n_chunks);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_store_occ<PADKB>), dim3(grid), dim3(256), 0, 0, buf, n_chunks);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("LDS-staged, +%2d KB LDS pad  grid %6d: %7.3f ms  %7.1f GB/s\n", PADKB, grid, ms, bytes / ms / 1e6);
}

int main()
{
    size_t bytes = 48000000000ull / 5120 * 5120;
    float4 *buf; CK(hipMalloc(&buf, bytes));
    for (int grid : {2048}) {
        run_occ<1>(buf, bytes, grid);
        run_occ<15>(buf, bytes, grid);   // 20 KB + 15 -> 4 blocks/CU
        run_occ<33>(buf, bytes, grid);   // -> 3 blocks/CU
        run_occ<60>(buf, bytes, grid);   // -> 2 blocks/CU
    }
    for (int grid : {2048}) {
        run<5, 0>("5KiB/wave plain", buf, bytes, grid);
        run<5, 1>("5KiB/wave nontemporal", buf, bytes, grid);
        run<5, 2>("5KiB/wave via LDS", buf, bytes, grid);
        run<1, 0>("1KiB/wave plain", buf, bytes, grid);
        run<20, 0>("20KiB/wave plain", buf, bytes, grid);
    }
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a)); for (int r = 0; r < 3; r++) CK(hipMemsetAsync(buf, 0, bytes)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("hipMemset: %.3f ms %.1f GB/s\n", ms / 3, bytes / (ms / 3) / 1e6);
    return 0;
}
