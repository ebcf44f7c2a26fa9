// libofdmtx, loop-back verifier: a receiver for the noiseless frames the transmit kernels
// produce, written from IEEE 802.11-2007 clause 17 (not from the transmitter's code paths):
//   k_rx_demod  : per packet, per OFDM symbol: 64-point DFT of the samples behind the cyclic
//                 prefix, nearest-point decision against the constellation table, pilot and
//                 null-carrier checks, de-interleaving (17.3.5.6, equations 17-18 / 17-19)
//                 -> the packet's hard coded bits in transmit order (the layout of the `coded`
//                 tap of ofdmtx_encode_bits)
//   k_rx_decode : per packet: the rate-1/2 K=7 code (133, 171) inverted bit by bit -- every
//                 data bit follows from one received coded bit and the six bits before it; where
//                 puncturing (17.3.5.5) left both coded bits, the second one is CHECKED against
//                 the decoded history -- SIGNAL field parsing (17.3.4), descrambling (17.3.5.4),
//                 PSDU bytes LSB first (17.3.5.1), FCS check, payload bytes out.
// A status word per packet says what did not hold.  With a clean loop-back every word is 0 and
// the payload buffer equals the transmitter's input: an end-to-end check that runs at batch
// sizes no CPU decoder can follow.
//
// Frame geometry (which rows are preamble / SIGNAL / DATA, where a packet's bytes live) comes
// from the transmit plan; rate and length are ALSO read from the decoded SIGNAL field and
// compared (status bits).  Citations: pu = ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py,
// ftw/ = ftw_80211_ofdm_tx/src/ftw/ of the reference, for the conventions the transmitter follows.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ofdm_common.h"
#include "ofdm_plan.h"

namespace {

using namespace ofdmtx_detail;

constexpr int RX_WAVES = 4;
constexpr float RX_TOL = 0.25f;        // distance a pilot / null carrier may be off its nominal value

// ---- pilots the transmit plan asked for: value at FFT bin `bin` of the symbol with running index
// pidx (ftw/ftw_ofdm_pilot_cc.cc:116-158), or the constant PNC pilots (ftw/ftw_pnc_ofdm_pilot_cc.cc:125-178)
__device__ __forceinline__ float expected_pilot(const PlanParams &P, const DevTables &T, int bin, uint32_t pidx)
{
    const bool first_or_third = bin == 43 || bin == 7;            // carriers -21 and +7
    if (P.pilot_mode == OFDMTX_PILOT_PNC_A) return first_or_third ? 1.f : 0.f;
    if (P.pilot_mode == OFDMTX_PILOT_PNC_B) return first_or_third ? 0.f : 1.f;
    if (P.pilot_mode == OFDMTX_PILOT_PNC_ALL) return 1.f;
    const uint32_t neg = (T.pilot_neg[pidx >> 5] >> (pidx & 31u)) & 1u;
    const float p = neg ? -1.f : 1.f;
    return bin == 21 ? -p : p;                                    // +21 carries -p
}

template <bool UNIFORM>
__global__ void __launch_bounds__(64 * RX_WAVES)
k_rx_demod(PlanParams P, DevTables T, const float2 *__restrict__ iq, uint8_t *__restrict__ coded,
           uint32_t *__restrict__ status, uint32_t wave_words)
{
    extern __shared__ uint32_t s_rxw[];                   // per wave: the packet's coded bits, MSB-first words
    __shared__ float2 s_w[64];                            // e^{-j 2 pi m / 64}
    __shared__ float2 s_lut[LUT_SIZE];
    __shared__ float2 s_x[RX_WAVES][64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64; i += 64 * RX_WAVES) s_w[i] = make_float2(T.tw64[i].x, -T.tw64[i].y);
    for (int i = tid; i < LUT_SIZE; i += 64 * RX_WAVES) s_lut[i] = T.lut[i];
    __syncthreads();

    uint32_t *W = s_rxw + (size_t)wave * wave_words;
    const bool swap_iq = (P.flags & OFDMTX_F_SWAPIQ) != 0;
    const uint32_t pre_rows = (P.flags & OFDMTX_F_PREAMBLE)
                                  ? ((P.flags & (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B)) ? 6u : 4u) : 0u;
    // the transmitter's rows are (amp / 8) * sum_k X[k] e^{+j 2 pi k n / 64}: the forward DFT
    // returns 8 * amp * X[k]
    const float inv = 1.0f / (8.0f * P.amp);
    const int code = bin_code(lane);                      // what this lane's FFT bin carries

    for (uint32_t p = blockIdx.x * RX_WAVES + wave; p < P.n_pkts; p += gridDim.x * RX_WAVES) {
        const Pkt k = load_pkt<UNIFORM>(P, p);
        const RegimeInfo ri = regime_info((int)k.regime);
        const size_t row0 = (UNIFORM ? (size_t)p * P.rows_per_pkt : (size_t)P.row_start[p]) + pre_rows;
        const uint32_t msg_len = 6u + k.n_sym * (uint32_t)ri.n_cbps / 8u;
        const uint32_t n_words = (msg_len + 3u) / 4u;
        for (uint32_t w = lane; w < n_words + 1u; w += 64) W[w] = 0u;
        uint32_t bad = 0;
        wave_sync();

        for (uint32_t s = 0; s <= k.n_sym; s++) {
            float2 x = iq[(row0 + s) * SYM_LEN + 16u + (uint32_t)lane];       // behind the cyclic prefix
            if (swap_iq) x = make_float2(x.y, x.x);
            s_x[wave][lane] = x;
            wave_sync();
            float ar = 0.f, ai = 0.f;
#pragma unroll 8
            for (int n = 0; n < 64; n++) {
                const float2 v = s_x[wave][n], w = s_w[(lane * n) & 63];
                ar += v.x * w.x - v.y * w.y;
                ai += v.x * w.y + v.y * w.x;
            }
            ar *= inv;
            ai *= inv;
            if (code >= 0) {
                // nearest constellation point; SIGNAL is BPSK whatever the rate (17.3.4)
                const int nb = s == 0 ? 1 : ri.n_bpsc;
                const int off = s == 0 ? LUT_PM1 : ri.lut_off;
                int best = 0;
                float best_d = 3.0e38f;
                for (int i = 0; i < (1 << nb); i++) {
                    const float2 c = s_lut[off + i];
                    const float d = (ar - c.x) * (ar - c.x) + (ai - c.y) * (ai - c.y);
                    if (d < best_d) { best_d = d; best = i; }
                }
                // 17.3.5.6: position j of the symbol came from coded bit k
                const uint32_t N = s == 0 ? 48u : (uint32_t)ri.n_cbps;
                const uint32_t sg = nb / 2 > 1 ? (uint32_t)nb / 2u : 1u;
                const uint32_t base = s == 0 ? 0u : 48u + (s - 1u) * N;
                for (int t = 0; t < nb; t++) {
                    const uint32_t j = (uint32_t)code * (uint32_t)nb + (uint32_t)t;
                    const uint32_t i = sg * (j / sg) + (j + 16u * j / N) % sg;        // (17-18)
                    const uint32_t kk = 16u * i - (N - 1u) * (16u * i / N);           // (17-19)
                    if ((best >> (nb - 1 - t)) & 1) {
                        const uint32_t g = base + kk;
                        atomicOr(&W[g >> 5], 1u << (31u - (g & 31u)));
                    }
                }
            } else if (code == -1) {
                if (ar * ar + ai * ai > RX_TOL * RX_TOL) bad |= OFDMTX_RX_NULLS;
            } else {
                const float e = expected_pilot(P, T, lane, (k.pilot_base + s) % 127u);
                if ((ar - e) * (ar - e) + ai * ai > RX_TOL * RX_TOL) bad |= OFDMTX_RX_PILOTS;
            }
            wave_sync();
        }
        // the coded bits as bytes, MSB first (the plan keeps messages 16-byte aligned, slack behind)
        uint32_t *dst = reinterpret_cast<uint32_t *>(coded + k.msg_off);
        for (uint32_t q = lane; q < n_words; q += 64) dst[q] = __builtin_bswap32(W[q]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bad |= __shfl_xor(bad, o);
        if (lane == 0) status[p] = bad;
        wave_sync();
    }
}

// MSB-first bit reader over the coded bytes of one packet
struct BitReader {
    const uint32_t *w;
    uint32_t cur, left;
    __device__ __forceinline__ explicit BitReader(const uint8_t *p) : w(reinterpret_cast<const uint32_t *>(p)), cur(0), left(0) {}
    __device__ __forceinline__ uint32_t get()
    {
        if (left == 0) {
            cur = __builtin_bswap32(*w++);
            left = 32;
        }
        const uint32_t b = cur >> 31;
        cur <<= 1;
        left--;
        return b;
    }
};

// One step of the decoder: hist bit d-1 = u[n-d].  g0 = 133 octal taps u[n] and the delays 2, 3, 5,
// 6; g1 = 171 octal taps u[n] and the delays 1, 2, 3, 6 (17.3.5.5).  have: bit 0 = A received,
// bit 1 = B received.  Returns u[n]; a redundant coded bit that contradicts it sets *mismatch.
__device__ __forceinline__ uint32_t decode_step(uint32_t &hist, uint32_t have, uint32_t a, uint32_t b, uint32_t *mismatch)
{
    const uint32_t pa = __popc(hist & 0x36u) & 1u, pb = __popc(hist & 0x27u) & 1u;
    uint32_t u;
    if (have & 1u) {
        u = a ^ pa;
        if ((have & 2u) && (b ^ pb) != u) *mismatch = 1u;
    } else {
        u = b ^ pb;
    }
    hist = ((hist << 1) | u) & 0x3Fu;
    return u;
}

template <bool UNIFORM>
__global__ void __launch_bounds__(64)
k_rx_decode(PlanParams P, DevTables T, const uint8_t *__restrict__ coded, uint8_t *__restrict__ payload_out,
            uint32_t *__restrict__ status)
{
    const uint32_t p = blockIdx.x * 64u + threadIdx.x;
    if (p >= P.n_pkts) return;
    const Pkt k = load_pkt<UNIFORM>(P, p);
    const RegimeInfo ri = regime_info((int)k.regime);
    BitReader in(coded + k.msg_off);
    uint32_t hist = 0, bad = 0, mism = 0;

    // ---- SIGNAL: 24 bits at rate 1/2 (17.3.4): RATE R1-R4, reserved, LENGTH (LSB first), parity, tail
    uint32_t sig = 0;                                     // bit 23 - n = SIGNAL bit n
    for (int n = 0; n < 24; n++) {
        const uint32_t a = in.get(), b = in.get();
        sig = (sig << 1) | decode_step(hist, 3u, a, b, &mism);
    }
    const uint32_t rate = sig >> 20;
    uint32_t length = 0;
    for (int i = 0; i < 12; i++) length |= ((sig >> (18 - i)) & 1u) << i;
    if (rate != (uint32_t)ri.rate_bits || ((sig >> 19) & 1u) || (__popc(sig >> 6) & 1u) || (sig & 0x3Fu)) bad |= OFDMTX_RX_SIGNAL;
    // MAC header + payload + FCS (pu:28-36,193); a raw-PSDU plan's bytes are the whole PSDU, which
    // comes back whole (FCS included; without one -- fewer than four bytes -- nothing is checked)
    const bool raw = (P.flags & OFDMTX_F_RAW_PSDU) != 0;
    const uint32_t hdr = raw ? 0u : 24u;
    const uint32_t want_len = raw ? k.pay_len : 24u + k.pay_len + 4u;
    const uint32_t fcs_at = want_len >= 4u ? want_len - 4u : want_len;   // first FCS byte
    if (length != want_len) bad |= OFDMTX_RX_LENGTH;

    // ---- DATA: SERVICE (16 bits), PSDU, six tail bits, pad (17.3.5)
    const uint32_t n_data = k.n_sym * (uint32_t)ri.n_dbps;
    const uint32_t psdu_end = 16u + 8u * want_len;
    uint32_t crc = 0xFFFFFFFFu, fcs = 0, byte = 0, phase = 0;
    uint8_t *pay = payload_out + k.pay_off;
    for (uint32_t n = 0; n < n_data; n++) {
        // 17.3.5.5, Figure 17-9: rate 3/4 sends A0 B0 A1 B2 of every three steps, rate 2/3 sends
        // A0 B0 A1 of every two
        uint32_t have = 3u;
        if (ri.punct == 1) have = phase == 0 ? 3u : phase == 1 ? 1u : 2u;
        if (ri.punct == 2) have = phase == 0 ? 3u : 1u;
        phase++;
        if ((ri.punct == 1 && phase == 3) || (ri.punct == 2 && phase == 2) || ri.punct == 0) phase = 0;
        const uint32_t a = (have & 1u) ? in.get() : 0u, b = (have & 2u) ? in.get() : 0u;
        const uint32_t scr = decode_step(hist, have, a, b, &mism);
        // 17.3.5.4: the scrambler's 127-bit sequence, restarted with every packet (pu:344-358)
        const uint32_t bit = scr ^ (T.scr_word[n % 127u] >> 31);
        if (n < 16u) {
            if (bit) bad |= OFDMTX_RX_SERVICE;
        } else if (n < psdu_end) {
            const uint32_t q = n - 16u, bi = q & 7u, by = q >> 3;
            byte |= bit << bi;                            // 17.3.5.1: LSB first
            if (by < fcs_at) {
                crc = (crc >> 1) ^ (0xEDB88320u & (0u - ((crc ^ bit) & 1u)));
            } else {
                fcs |= bit << (q - 8u * fcs_at);
            }
            if (bi == 7u) {
                if (by >= hdr && (raw || by < fcs_at)) pay[by - hdr] = (uint8_t)byte;
                byte = 0;
            }
        } else if (n < psdu_end + 6u) {
            if (scr) bad |= OFDMTX_RX_TAIL;               // tail bits are zero AFTER scrambling (17.3.5.2)
        }
    }
    if (fcs_at != want_len && (crc ^ 0xFFFFFFFFu) != fcs) bad |= OFDMTX_RX_FCS;
    if (mism) bad |= OFDMTX_RX_CODE;
    status[p] |= bad;
}

}  // namespace

extern "C" {

int ofdmtx_rx_demod(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const float *d_iq, uint8_t *d_coded, uint32_t *d_status,
                    void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    if (pl->pp.n_pkts == 0) return OFDMTX_OK;
    if (!d_iq || !d_coded || !d_status) return OFDMTX_EINVAL;
    if (pl->pp.amp == 0.0f) return OFDMTX_EINVAL;                 // nothing to demodulate in all-zero rows
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t wave_words = pl->tx_m_words + 4u;
    const size_t lds = (size_t)RX_WAVES * wave_words * sizeof(uint32_t);
    if (lds > 60000) return OFDMTX_ETOOBIG;
    uint64_t blocks = ((uint64_t)pl->pp.n_pkts + RX_WAVES - 1) / RX_WAVES;
    const uint64_t cap = (uint64_t)ctx->n_cu * 8u;
    if (blocks > cap) blocks = cap;
    const DevTables tab = ctx->tables();
    const float2 *iq = reinterpret_cast<const float2 *>(d_iq);
    if (pl->pp.uniform)
        hipLaunchKernelGGL((k_rx_demod<true>), dim3((unsigned)blocks), dim3(64 * RX_WAVES), lds, (hipStream_t)stream,
                           pl->pp, tab, iq, d_coded, d_status, wave_words);
    else
        hipLaunchKernelGGL((k_rx_demod<false>), dim3((unsigned)blocks), dim3(64 * RX_WAVES), lds, (hipStream_t)stream,
                           pl->pp, tab, iq, d_coded, d_status, wave_words);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_rx_decode(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const uint8_t *d_coded, uint8_t *d_payloads,
                     uint32_t *d_status, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    if (pl->pp.n_pkts == 0) return OFDMTX_OK;
    if (!d_coded || !d_status) return OFDMTX_EINVAL;
    if (!d_payloads && pl->sizes.payload_bytes) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const unsigned blocks = (unsigned)(((uint64_t)pl->pp.n_pkts + 63u) / 64u);
    const DevTables tab = ctx->tables();
    if (pl->pp.uniform)
        hipLaunchKernelGGL((k_rx_decode<true>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, pl->pp, tab, d_coded,
                           d_payloads, d_status);
    else
        hipLaunchKernelGGL((k_rx_decode<false>), dim3(blocks), dim3(64), 0, (hipStream_t)stream, pl->pp, tab, d_coded,
                           d_payloads, d_status);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

}  // extern "C"
