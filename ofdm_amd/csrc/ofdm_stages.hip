// libofdmtx, single-message stage kernels: the bit-level functions of the reference's packet
// utilities as separately callable operators on ONE byte string (scrambler, convolutional
// encoder with puncturing, interleaver).  They exist so that the Python functions of the same
// names keep their per-stage signatures; batches go through ofdmtx_encode_bits / ofdmtx_tx.
// One thread per output byte; bytes are MSB-first bit strings.
//
// Citations: pu = ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py of the reference.

#include "ofdm_common.h"
#include "ofdm_plan.h"

namespace {

using namespace ofdmtx_detail;

struct StageRegime {
    int n_cbps, n_bpsc, punct;   // punct: 0 none, 1 = 3/4, 2 = 2/3
};

// pu:39-78
bool stage_regime(int regime, StageRegime *r)
{
    switch (regime) {
    case 1: *r = {48, 1, 0}; return true;
    case 2: *r = {48, 1, 1}; return true;
    case 3: *r = {96, 2, 0}; return true;
    case 4: *r = {96, 2, 1}; return true;
    case 5: *r = {192, 4, 0}; return true;
    case 6: *r = {192, 4, 1}; return true;
    case 7: *r = {288, 6, 2}; return true;
    case 8: *r = {288, 6, 1}; return true;
    default: return false;
    }
}

__device__ __forceinline__ uint32_t get_bit(const uint8_t *s, int64_t i)
{
    return i < 0 ? 0u : (uint32_t)(s[i >> 3] >> (7 - (i & 7))) & 1u;
}

// pu:344-367: bits 0..23 untouched, bit k >= 24 XOR sequence[(k - 24) % 127], then the six bits
// from 40 + 8*Length_data on are forced to zero.  scr_word[i] holds sequence bits i .. i+31, MSB first.
__global__ void k_stage_scramble(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t n_bytes,
                                 uint32_t length_data, const uint32_t *__restrict__ scr_word)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bytes) return;
    uint8_t v = in[b];
    if (b >= 3u) v ^= (uint8_t)(scr_word[(8u * b - 24u) % 127u] >> 24);
    if (b == 5u + length_data) v &= 0x03u;
    out[b] = v;
}

// pu:280-341: rate-1/2 encoder over the whole string (register starts at zero, SIGNAL included),
// output A (133 octal) before B (171 octal); the DATA part (coded bits from 48 on) is punctured
// and cut to n_out_bits - 48.
__global__ void k_stage_conv_encode(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t out_bytes,
                                    int punct)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= out_bytes) return;
    uint32_t v = 0;
    for (uint32_t i = 0; i < 8u; i++) {
        const uint32_t j = 8u * b + i;                 // index in the output stream
        uint32_t c = j;                                // index in the rate-1/2 stream
        if (j >= 48u) {
            const uint32_t q = j - 48u;
            c = 48u + (punct == 1 ? 6u * (q / 4u) + ((q % 4u) == 3u ? 5u : q % 4u)      // keep 0,1,2,5 of 6 (pu:316-328)
                                  : punct == 2 ? 4u * (q / 3u) + q % 3u                   // keep 0,1,2 of 4 (pu:330-341)
                                               : q);
        }
        const int64_t n = c >> 1;
        // A[n] = u[n]^u[n-2]^u[n-3]^u[n-5]^u[n-6], B[n] = u[n]^u[n-1]^u[n-2]^u[n-3]^u[n-6]  (pu:287-307)
        uint32_t bit = get_bit(in, n) ^ get_bit(in, n - 2) ^ get_bit(in, n - 3) ^ get_bit(in, n - 6);
        bit ^= (c & 1u) ? get_bit(in, n - 1) : get_bit(in, n - 5);
        v |= bit << (7u - i);
    }
    out[b] = (uint8_t)v;
}

// pu:222-277: per OFDM symbol, bit k of the input goes to i = (N/16)(k%16) + k/16 and on to
// j = s*(i/s) + (i + N - 16i/N) % s; the first 48 bits (SIGNAL) use N = 48 and the first step only.
// The thread computes the inverse: which input bit lands on each of its output bits.
__global__ void k_stage_interleave(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t n_bytes,
                                   int n_cbps, int n_bpsc)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bytes) return;
    const int R = 3 * n_bpsc, S = n_bpsc / 2 > 1 ? n_bpsc / 2 : 1;
    uint32_t v = 0;
    for (uint32_t i = 0; i < 8u; i++) {
        const uint32_t o = 8u * b + i;
        uint32_t src;
        if (o < 48u) {
            src = 16u * (o % 3u) + o / 3u;
        } else {
            const uint32_t sym = (o - 48u) / (uint32_t)n_cbps, j = (o - 48u) % (uint32_t)n_cbps;
            const uint32_t c = j / (uint32_t)R, u = j % (uint32_t)R;
            src = 48u + sym * (uint32_t)n_cbps + 16u * ((uint32_t)S * (u / (uint32_t)S) + ((u % (uint32_t)S) + c) % (uint32_t)S) + c;
        }
        v |= get_bit(in, src) << (7u - i);
    }
    out[b] = (uint8_t)v;
}

// gr.multiply_const_cc: out = in * k (complex), 16 bytes per thread
__global__ void k_scale_cc(const float2 *__restrict__ in, float2 *__restrict__ out, uint64_t n, float kr, float ki)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2u;
    if (i + 1 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(in + i);
        float4 o;
        o.x = v.x * kr - v.y * ki;
        o.y = v.x * ki + v.y * kr;
        o.z = v.z * kr - v.w * ki;
        o.w = v.z * ki + v.w * kr;
        *reinterpret_cast<float4 *>(out + i) = o;
    } else if (i < n) {
        const float2 v = in[i];
        out[i] = make_float2(v.x * kr - v.y * ki, v.x * ki + v.y * kr);
    }
}

// ftw/ftw_repetition.cc:54-76: the block never consumes its input, so its output is the first
// `period` input rows over and over.  16 bytes per thread; blockIdx.y walks the copies, so no
// thread computes a remainder.
__global__ void k_repeat(const float4 *__restrict__ in, float4 *__restrict__ out, uint64_t period4, uint64_t total4)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= period4) return;
    const float4 v = in[j];
    for (uint64_t o = (uint64_t)blockIdx.y * period4 + j; o < total4; o += (uint64_t)gridDim.y * period4) out[o] = v;
}

// The taps behind the transform, cut out of a finished IQ stream (amplitude 1): per packet the
// symbol rows (SIGNAL + DATA) sit behind the preamble rows.  scale = the rows themselves
// (examples/ftw_ofdm.py:126 multiplies by 1/8), cp = the rows times 8 (the cyclic prefixer's output,
// examples/ftw_ofdm.py:121-123), ifft = samples 16..79 of those (the transform's output,
// examples/ftw_ofdm.py:118), pre = preamble rows + symbol rows (ftw/ftw_ofdm_preamble.cc:79-95).
// Multiplying by 8 is exact in float32.  One workgroup per packet, 16 bytes per thread.
__global__ void __launch_bounds__(256) k_symbol_taps(PlanParams P, const float4 *__restrict__ iq, float4 *__restrict__ ifft,
                                                     float4 *__restrict__ cp, float4 *__restrict__ scale,
                                                     float4 *__restrict__ pre)
{
    const uint32_t pre_rows = (P.flags & OFDMTX_F_PREAMBLE)
                                  ? ((P.flags & (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B)) ? PNC_PRE_ROWS : PRE_ROWS) : 0;
    constexpr uint32_t Q = SYM_LEN / 2;                       // float4 per row
    for (uint32_t p = blockIdx.x; p < P.n_pkts; p += gridDim.x) {
        const size_t row0 = P.uniform ? (size_t)p * P.rows_per_pkt : (size_t)P.row_start[p];
        const uint32_t n_sym = P.uniform ? P.n_sym : P.recs[p].n_sym;
        const size_t sym0 = P.uniform ? (size_t)p * (P.n_sym + 1u) : (size_t)P.recs[p].sym_off;
        const uint32_t rows = n_sym + 1u;
        const float4 *src = iq + (row0 + pre_rows) * Q;
        for (uint32_t i = threadIdx.x; i < rows * Q; i += blockDim.x) {
            const uint32_t r = i / Q, c = i % Q;
            const float4 v = src[i];
            const float4 v8 = make_float4(v.x * 8.f, v.y * 8.f, v.z * 8.f, v.w * 8.f);
            if (scale) scale[sym0 * Q + i] = v;
            if (cp) cp[sym0 * Q + i] = v8;
            if (ifft && c >= 8u) ifft[(sym0 + r) * 32u + (c - 8u)] = v8;
        }
        if (pre) {
            float4 *dst = pre + (sym0 + (size_t)p * pre_rows) * Q;
            const float4 *s2 = iq + row0 * Q;
            for (uint32_t i = threadIdx.x; i < (pre_rows + rows) * Q; i += blockDim.x) dst[i] = s2[i];
        }
    }
}

}  // namespace

extern "C" {

int ofdmtx_symbol_taps(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const float *d_iq, float *d_ifft, float *d_cp,
                       float *d_scale, float *d_preamble, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    // the taps sit in front of tx_amplitude and of the I/Q exchange: only then are they the kernel's own samples
    if (pl->pp.amp != 1.0f || (pl->pp.flags & OFDMTX_F_SWAPIQ)) return OFDMTX_EINVAL;
    if (pl->pp.n_pkts == 0) return OFDMTX_OK;
    if (!d_iq) return OFDMTX_EINVAL;
    if (((uintptr_t)d_iq | (uintptr_t)d_ifft | (uintptr_t)d_cp | (uintptr_t)d_scale | (uintptr_t)d_preamble) & 15u)
        return OFDMTX_EINVAL;                                                 // 16-byte accesses
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = pl->pp.n_pkts;
    const uint32_t cap = (uint32_t)ctx->n_cu * 8u;
    hipLaunchKernelGGL(k_symbol_taps, dim3(n < cap ? n : cap), dim3(256), 0, (hipStream_t)stream, pl->pp,
                       reinterpret_cast<const float4 *>(d_iq), reinterpret_cast<float4 *>(d_ifft),
                       reinterpret_cast<float4 *>(d_cp), reinterpret_cast<float4 *>(d_scale),
                       reinterpret_cast<float4 *>(d_preamble));
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_scramble(ofdmtx_ctx *ctx, const uint8_t *d_in, uint32_t n_bytes, uint32_t length_data, uint8_t *d_out,
                    void *stream)
{
    if (!ctx || !d_in || !d_out) return OFDMTX_EINVAL;
    // the reference indexes the tail bits without a bound check (IndexError beyond the string)
    if (n_bytes < 3u || 5u + (uint64_t)length_data >= n_bytes) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_stage_scramble, dim3((n_bytes + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d_in,
                       d_out, n_bytes, length_data, ctx->d_scr);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_conv_encode(ofdmtx_ctx *ctx, const uint8_t *d_in, uint32_t n_bytes, int regime, uint32_t n_sym,
                       uint8_t *d_out, void *stream)
{
    if (!ctx || !d_in || !d_out) return OFDMTX_EINVAL;
    StageRegime r;
    if (!stage_regime(regime, &r)) return OFDMTX_EREGIME;
    const uint64_t out_bits = 48ull + (uint64_t)n_sym * (uint64_t)r.n_cbps;
    // last rate-1/2 bit the output needs must exist in the input
    uint64_t last = out_bits - 1;
    if (out_bits > 48) {
        const uint64_t q = out_bits - 49;
        last = 48 + (r.punct == 1 ? 6 * (q / 4) + ((q % 4) == 3 ? 5 : q % 4) : r.punct == 2 ? 4 * (q / 3) + q % 3 : q);
    }
    if (last / 2 >= 8ull * n_bytes || out_bits / 8 > 0xFFFFFFFFull) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t out_bytes = (uint32_t)(out_bits / 8);
    hipLaunchKernelGGL(k_stage_conv_encode, dim3((out_bytes + 255u) / 256u), dim3(256), 0, (hipStream_t)stream,
                       d_in, d_out, out_bytes, r.punct);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_interleave(ofdmtx_ctx *ctx, const uint8_t *d_in, uint32_t n_bytes, int regime, uint8_t *d_out,
                      void *stream)
{
    if (!ctx || !d_in || !d_out) return OFDMTX_EINVAL;
    StageRegime r;
    if (!stage_regime(regime, &r)) return OFDMTX_EREGIME;
    // whole symbols only: the reference leaves a trailing partial symbol as zeros
    if (n_bytes < 6u || ((uint64_t)(n_bytes - 6u) * 8u) % (uint64_t)r.n_cbps != 0) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_stage_interleave, dim3((n_bytes + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, d_in,
                       d_out, n_bytes, r.n_cbps, r.n_bpsc);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_scale_cc(ofdmtx_ctx *ctx, const float *d_in, uint64_t n, float k_re, float k_im, float *d_out,
                    void *stream)
{
    if (!ctx) return OFDMTX_EINVAL;
    if (n == 0) return OFDMTX_OK;
    if (!d_in || !d_out) return OFDMTX_EINVAL;
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15u) return OFDMTX_EINVAL;      // 16-byte accesses
    const uint64_t threads = (n + 1) / 2;
    if ((threads + 255) / 256 > 0x7FFFFFFFull) return OFDMTX_ETOOBIG;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_scale_cc, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2 *>(d_in), reinterpret_cast<float2 *>(d_out), n, k_re, k_im);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_repeat(ofdmtx_ctx *ctx, const float *d_in, uint64_t period, uint64_t n_out, float *d_out, void *stream)
{
    if (!ctx) return OFDMTX_EINVAL;
    if (n_out == 0) return OFDMTX_OK;
    if (!d_in || !d_out || period == 0) return OFDMTX_EINVAL;
    // whole 16-byte pieces: even sample counts (rows of 80 samples always are), aligned buffers
    if ((period & 1u) || (n_out & 1u) || (((uintptr_t)d_in | (uintptr_t)d_out) & 15u)) return OFDMTX_EINVAL;
    const uint64_t period4 = period / 2, total4 = n_out / 2;
    const uint64_t bx = (period4 + 255) / 256;
    if (bx > 0x7FFFFFFFull) return OFDMTX_ETOOBIG;
    uint64_t copies = (total4 + period4 - 1) / period4;
    // enough blocks to fill the GPU, at most 65535 in y
    const uint64_t want_y = ((uint64_t)ctx->n_cu * 16 + bx - 1) / bx;
    uint64_t by = copies < want_y ? copies : want_y;
    if (by > 65535) by = 65535;
    if (by < 1) by = 1;
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_repeat, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(d_in), reinterpret_cast<float4 *>(d_out), period4, total4);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

}  // extern "C"
