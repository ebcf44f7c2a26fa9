// Micro-benchmark: the back end as short-lived workgroups launched in address order -- each reads the coded bits of
// R symbols (36 bytes each at 54 Mbit/s), "computes" briefly, writes R rows of 640 bytes and ends.  48 GB written.
// Would a modulator of this shape reach the rate of the compact write window (6.8 TB/s), and what do its reads cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

template <int T, int R, bool READ>
__global__ void __launch_bounds__(T) k_rows(float4 *out, const uint32_t *msg, int sleep_units)
{
    const size_t g = blockIdx.x;
    float acc = 1.f;
    if (READ) {
        const uint32_t *q = msg + g * (R * 9);                  // R * 36 bytes
        if (threadIdx.x < R * 9) acc += (float)q[threadIdx.x];
        acc += __shfl(acc, 0);
    }
    for (int k = 0; k < sleep_units; k++) __builtin_amdgcn_s_sleep(32);     // 32 * 64 cycles ~ 1 us
    float4 *dst = out + g * (R * 40);
    for (int i = threadIdx.x; i < R * 40; i += T) dst[i] = make_float4(acc, 2.f, 3.f, (float)i);
}

template <typename F>
void timed(const char *name, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; r++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    ms /= 3;
    printf("%-64s %.3f ms %.2f TB/s\n", name, ms, 48e9 / ms / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t rows = 75000000;                                // 48 GB / 640
    float4 *buf; CK(hipMalloc(&buf, rows * 640));
    uint32_t *msg; CK(hipMalloc(&msg, rows * 36 + 4096)); CK(hipMemset(msg, 1, rows * 36));
#define RUN(T, R, RD, SL, NAME) timed(NAME, [&] { hipLaunchKernelGGL((k_rows<T, R, RD>), dim3((unsigned)(rows / R)), dim3(T), 0, 0, buf, msg, SL); })
    RUN(256, 8, false, 0, "256 threads, 8 rows, no read, no compute");
    RUN(256, 8, false, 1, "256 threads, 8 rows, no read, 1 us");
    RUN(256, 8, false, 2, "256 threads, 8 rows, no read, 2 us");
    RUN(256, 8, true, 0, "256 threads, 8 rows, read, no compute");
    RUN(256, 8, true, 1, "256 threads, 8 rows, read, 1 us");
    RUN(256, 8, true, 2, "256 threads, 8 rows, read, 2 us");
    RUN(320, 8, true, 1, "320 threads, 8 rows (one store each), read, 1 us");
    RUN(320, 8, false, 1, "320 threads, 8 rows (one store each), no read, 1 us");
    RUN(256, 4, true, 1, "256 threads, 4 rows, read, 1 us");
    RUN(256, 4, false, 1, "256 threads, 4 rows, no read, 1 us");
    RUN(256, 16, true, 1, "256 threads, 16 rows, read, 1 us");
    RUN(64, 4, true, 1, "64 threads, 4 rows, read, 1 us");
    RUN(128, 8, true, 1, "128 threads, 8 rows, read, 1 us");
    return 0;
}
