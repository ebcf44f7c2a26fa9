// Check on hardware: a 4 x 4 transpose between (16-lane row of the wave, register index) with the gfx950
// swap instructions: y[a] of the lane in row r, column c  <-  y[r] of the lane in row a, column c.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(unsigned &d, unsigned &s) { const u32x2 r = __builtin_amdgcn_permlane32_swap(d, s, false, false); d = r.x; s = r.y; }
__device__ __forceinline__ void swap16(unsigned &d, unsigned &s) { const u32x2 r = __builtin_amdgcn_permlane16_swap(d, s, false, false); d = r.x; s = r.y; }
__device__ __forceinline__ void transpose4(unsigned (&y)[4])
{
    swap32(y[0], y[2]);   // rows 2,3 of y[0] <-> rows 0,1 of y[2]
    swap32(y[1], y[3]);
    swap16(y[0], y[1]);   // odd rows of y[0] <-> even rows of y[1]
    swap16(y[2], y[3]);
}
__global__ void k(unsigned *out)
{
    const unsigned lane = threadIdx.x, row = lane >> 4, col = lane & 15;
    unsigned y[4];
    for (int a = 0; a < 4; a++) y[a] = 1000u * row + 100u * a + col;     // (row, register, column)
    transpose4(y);
    for (int a = 0; a < 4; a++) out[4 * lane + a] = y[a];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (unsigned lane = 0; lane < 64; lane++)
        for (unsigned a = 0; a < 4; a++) {
            const unsigned row = lane >> 4, col = lane & 15, want = 1000u * a + 100u * row + col;
            if (h[4 * lane + a] != want) { if (bad < 8) printf("lane %u reg %u: got %u want %u\n", lane, a, h[4 * lane + a], want); bad++; }
        }
    printf("%s (%d mismatches)\n", bad ? "WRONG" : "transpose ok", bad);
    return bad != 0;
}
