// Micro-benchmark: is the slow / fast difference between 48 GB output buffers a matter of how the buffer is backed?
// Buffers allocated alternately with hipMalloc and through the virtual-memory API (physical handles of CHUNK bytes
// mapped back to back into one reserved range), the wave-per-packet store pattern timed into each.
// usage: vmm_bw <chunk MiB> <n buffers>
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

static void *vmm_alloc(size_t size, size_t chunk)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk % gran) { printf("chunk not a multiple of the granularity %zu\n", gran); exit(1); }
    size = (size + chunk - 1) / chunk * chunk;
    void *ptr = nullptr;
    CK(hipMemAddressReserve(&ptr, size, chunk, nullptr, 0));
    for (size_t off = 0; off < size; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)ptr + off, chunk, 0, h, 0));
        CK(hipMemRelease(h));                    // the mapping keeps the memory
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(ptr, size, &acc, 1));
    return ptr;
}

int main(int argc, char **argv)
{
    const size_t chunk = (size_t)atoi(argv[1]) << 20;
    const int nbuf = atoi(argv[2]);
    const uint32_t n = 1000000;
    const size_t bytes = (size_t)n * 48000;
    std::vector<float4 *> bufs;
    std::vector<const char *> kind;
    for (int i = 0; i < nbuf; i++) {
        if (i & 1) {
            bufs.push_back((float4 *)vmm_alloc(bytes, chunk));
            kind.push_back("virtual-memory API");
        } else {
            float4 *p; CK(hipMalloc(&p, bytes));
            bufs.push_back(p);
            kind.push_back("hipMalloc");
        }
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++)
        for (int i = 0; i < nbuf; i++) {
            hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, bufs[i], n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_pkt, dim3(512), dim3(512), 0, 0, bufs[i], n);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("buffer %d (%s, %p): %.3f ms %.2f TB/s\n", i, kind[i], (void *)bufs[i], ms / 3, bytes / (ms / 3) / 1e9);
            fflush(stdout);
        }
    return 0;
}
