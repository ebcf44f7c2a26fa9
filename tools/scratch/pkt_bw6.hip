// Micro-benchmark: which packets does a wave take?  (a) packet = wave + k * all_waves: the chip sweeps the buffer
// front to back; (b) packet = wave * per_wave + k: every wave owns a contiguous range, 4096 ranges far apart.
// Wave per packet, 8 waves per workgroup, grid 512, payload read + 48 000 bytes of sc1 stores per packet.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MAP, bool READ>
__global__ void __launch_bounds__(512) k_pkt(float4 *out, const uint32_t *pay, uint32_t n_pkts)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * 8, me = blockIdx.x * 8 + wave;
    const uint32_t per = (n_pkts + n_waves - 1) / n_waves;
    // MAP 2: ranges per workgroup (its 8 waves interleave inside the workgroup's range)
    const uint32_t first = MAP == 0 ? me : MAP == 1 ? me * per : blockIdx.x * per * 8 + wave;
    const uint32_t step = MAP == 0 ? n_waves : MAP == 1 ? 1 : 8;
    const uint32_t last = MAP == 0 ? n_pkts : MAP == 1 ? min(n_pkts, (me + 1) * per) : min(n_pkts, (blockIdx.x + 1) * per * 8);
    for (uint32_t p = first; p < last; p += step) {
        float acc = 1.f;
        if (READ) {
            const uint32_t *q = pay + (size_t)p * 375;
            for (int i = lane; i < 375; i += 64) acc += (float)q[i];
        }
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)p * 3000, 0, 0x7FFFFFFF, 0x00020000);
        for (int i = lane; i < 3000; i += 64) {
            const u32x4 v = {__float_as_uint(acc), 2u, 3u, (unsigned)i};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, i * 16, 0, 16);
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
    printf("%-64s %.3f ms\n", name, ms / 5);
    fflush(stdout);
}

int main()
{
    const uint32_t n = 1000000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n * 48000));
    uint32_t *pay; CK(hipMalloc(&pay, (size_t)n * 1500 + 4096)); CK(hipMemset(pay, 1, (size_t)n * 1500));
#define RUN(MAP, RD, NAME) timed(NAME, [&] { hipLaunchKernelGGL((k_pkt<MAP, RD>), dim3(512), dim3(512), 0, 0, buf, pay, n); })
    for (int rep = 0; rep < 2; rep++) {
        RUN(0, false, "sweep, no reads");
        RUN(1, false, "range per wave, no reads");
        RUN(2, false, "range per workgroup, no reads");
        RUN(0, true, "sweep, reads");
        RUN(1, true, "range per wave, reads");
        RUN(2, true, "range per workgroup, reads");
    }
    return 0;
}
