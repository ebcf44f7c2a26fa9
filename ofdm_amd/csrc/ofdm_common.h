// Shared by the translation units of libofdmtx: error plumbing, the context object, complex
// arithmetic on packed float pairs and the wavefront-level LDS hand-over.
#ifndef OFDM_COMMON_H
#define OFDM_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "ofdmtx.h"

// last HIP error of the calling thread (ofdmtx_last_hip_error)
__attribute__((visibility("hidden"))) extern thread_local int ofdmtx_g_last_hip;

#define HIPCHK(expr)                                  \
    do {                                              \
        hipError_t e_ = (expr);                       \
        if (e_ != hipSuccess) {                       \
            ofdmtx_g_last_hip = (int)e_;              \
            return OFDMTX_EHIP;                       \
        }                                             \
    } while (0)

constexpr int LUT_SIZE = 128;         // constellation LUT slots, see ofdmtx.hip
constexpr int RX_TURNS = 64;          // launches of ofdmtx_rx_demod on ragged plans that may be in flight on one context

struct DevTables {
    const uint32_t *scr_word;   // 127
    const float2 *lut;          // LUT_SIZE
    const float2 *preamble;     // 320
    const float2 *tw64;         // 64
    const uint32_t *crc_tab;    // 4 x 256, slicing-by-4 tables of the reflected CRC-32
    const uint32_t *xpow8;      // x^(8n) mod P for n = 0..MAXLEN, reflected representation
    uint32_t pilot_neg[4];
};

struct ofdmtx_ctx {
    int device;
    uint32_t *d_scr;
    float2 *d_lut;
    float2 *d_pre;
    float2 *d_tw;
    uint32_t *d_crc;
    uint32_t *d_xpow8;
    float2 *d_gap;              // the 80-sample row ftw.zerogap repeats behind every packet (zeros unless set)
    bool gap_nonzero;
    uint8_t *d_static;          // this device's copy of the library's static work area (ofdmtx.hip)
    void *h_pinned;             // 256 page-locked bytes: totals read back by ofdmtx_tx_ragged_static
    uint32_t *d_turns;          // RX_TURNS packet counters, 64 bytes apart: one per ofdmtx_rx_demod launch in flight (ragged plans)
    std::atomic<uint32_t> turn; // the next launch's counter
    int n_cu;
    DevTables tables() const;   // defined in ofdmtx.hip
};


// Lanes of one wavefront exchange data through LDS without a workgroup barrier.  LDS
// operations of a wave execute in issue order, so the hand-over needs (a) the compiler kept
// from moving LDS accesses across this point ("memory" clobber) and (b) at most a wait for the
// wave's own LDS operations.  Deliberately NOT a fence: a release fence also drains the
// vector-memory counter, i.e. it would wait for every global store still in flight.
__device__ __forceinline__ void wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Complex samples are handled as 2-float vectors so that hipcc emits packed VALU operations
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two floats per instruction).
typedef float v2f __attribute__((ext_vector_type(2)));

// The packed-f32 operand modifiers (op_sel picks the half that feeds each result lane, neg_lo /
// neg_hi negate per lane) make multiplication by +-j free; hipcc does not use them for that (it
// emits an xor and a move per rotation), hence the inline assembly.
__device__ __forceinline__ v2f add_j(v2f a, v2f b)     // a + j*b = (a.x - b.y, a.y + b.x)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f sub_j(v2f a, v2f b)     // a - j*b = (a.x + b.y, a.y - b.x)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f cmul(v2f a, v2f b)      // a * b
{
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                 // a.x * (b.x, b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"          // + a.y * (-b.y, b.x)
        : "=v"(d) : "v"(a), "v"(b), "v"(t));
    return d;
}

// y[p] = sum_m x[m] e^{+j 2 pi m p / 8}, in place
__device__ __forceinline__ void idft8(v2f (&x)[8])
{
    const v2f h = {0.70710678118654752440f, 0.70710678118654752440f};
    const v2f a0 = x[0] + x[4], a1 = x[0] - x[4];
    const v2f a2 = x[2] + x[6], a3 = x[2] - x[6];
    const v2f a4 = x[1] + x[5], a5 = x[1] - x[5];
    const v2f a6 = x[3] + x[7], a7 = x[3] - x[7];
    const v2f b0 = a0 + a2, b2 = a0 - a2;
    const v2f b1 = add_j(a1, a3), b3 = sub_j(a1, a3);
    const v2f b4 = a4 + a6, b6 = a4 - a6;
    const v2f b5 = add_j(a5, a7), b7 = sub_j(a5, a7);
    const v2f t5 = add_j(b5, b5);                  // b5 * e^{+j pi/4}  =  h * t5
    const v2f t7 = sub_j(b7, b7);                  // b7 * e^{+j 3pi/4} = -h * t7
    x[0] = b0 + b4; x[4] = b0 - b4;
    x[1] = __builtin_elementwise_fma(h, t5, b1); x[5] = __builtin_elementwise_fma(-h, t5, b1);
    x[2] = add_j(b2, b6); x[6] = sub_j(b2, b6);
    x[3] = __builtin_elementwise_fma(-h, t7, b3); x[7] = __builtin_elementwise_fma(h, t7, b3);
}

// IQ stores.  A wave addresses a run of output rows through a buffer resource (base in
// SGPRs, 32-bit byte offsets per lane), which is also the way to give the stores a cache policy the
// compiler knows about: IQ_STORE_AUX is the instruction's policy field (1 = sc0, 2 = nt, 16 = sc1;
// a translation unit may set it before including this file).  The samples are
// written once and never read back by this kernel; with sc1 the payload reads issued beside them
// cost measurably less (48 GB of such stores next to 1.5 GB of reads: 9.60 against 9.75 ms), the
// stores themselves run at the same rate.
#ifndef IQ_STORE_AUX
#define IQ_STORE_AUX 16
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t iq_window(const void *base)
{
    const uint64_t a = reinterpret_cast<uint64_t>(base);          // the same in every lane of the wave
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    // raw buffer (stride 0): offsets are checked against the record count, which is left wide open --
    // the host has sized the IQ buffer for every row of the plan
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
}
// float4 number `idx` of the window
template <int AUX = IQ_STORE_AUX>
__device__ __forceinline__ void iq_store(__amdgpu_buffer_rsrc_t w, uint32_t idx, float4 v)
{
    const u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, w, (int)(idx * 16u), 0, AUX);
}

#endif
