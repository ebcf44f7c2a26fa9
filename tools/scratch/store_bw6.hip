// Micro-benchmark: is the 6.8 TB/s of "workgroup i writes the 4 KB chunk i" an affinity between the XCD a workgroup
// runs on (i mod 8) and the 4 KB page it writes?  SHIFT moves the chunk a workgroup writes by SHIFT pages within
// its group of 8 (or 16, 32); a fixed-grid variant lets workgroup b keep to pages with page % 8 == (b + SHIFT) % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_chunk_shift(float4 *out, size_t n_chunks, int shift, int group)
{
    const size_t c = blockIdx.x;
    const size_t g = c / group * group, r = (c % group + shift) % group;
    if (g + r < n_chunks) out[(g + r) * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// fixed grid (multiple of 8): workgroup b writes pages 8*j + (b + shift) % 8 for j = b/8, b/8 + grid/8, ...
__global__ void __launch_bounds__(256) k_stride_affine(float4 *out, size_t n_chunks, int shift)
{
    const size_t r = (blockIdx.x + shift) % 8;
    for (size_t j = blockIdx.x / 8; j * 8 + r < n_chunks; j += gridDim.x / 8)
        out[(j * 8 + r) * 256 + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F>
void timed(const char *name, int arg, F launch)
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
    printf("%-56s %3d: %.3f ms %.2f TB/s\n", name, arg, ms, 48e9 / ms / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t bytes = 48000000000ull / 32768 * 32768, n_chunks = bytes / 4096;
    float4 *buf; CK(hipMalloc(&buf, bytes));
    printf("buffer %p\n", (void *)buf);
    for (int s = 0; s < 8; s++)
        timed("one WG per 4 KB page, page shifted within groups of 8 by", s,
              [&] { hipLaunchKernelGGL(k_chunk_shift, dim3((unsigned)n_chunks), dim3(256), 0, 0, buf, n_chunks, s, 8); });
    for (int s : {8, 16, 1, 9})
        timed("one WG per 4 KB page, shifted within groups of 32 by", s,
              [&] { hipLaunchKernelGGL(k_chunk_shift, dim3((unsigned)n_chunks), dim3(256), 0, 0, buf, n_chunks, s, 32); });
    for (int s = 0; s < 8; s++)
        timed("fixed grid 2048, WG b keeps to pages = b + s mod 8, s =", s,
              [&] { hipLaunchKernelGGL(k_stride_affine, dim3(2048), dim3(256), 0, 0, buf, n_chunks, s); });
    return 0;
}
