// libofdmtx, loop-back verifier: a receiver for the noiseless frames the transmit kernels
// produce, written from IEEE 802.11-2007 clause 17 (not from the transmitter's code paths):
//   k_rx_demod  : per packet, per OFDM symbol: 64-point DFT (8 x 8) of the samples behind the cyclic
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

// ---- 17.3.5.6 inverted for the NB bits one data carrier holds.  Equations 17-18 / 17-19 with N = 48 NB
// coded bits per symbol and s = max(NB / 2, 1) say: the interleaver writes the symbol's coded bits into
// 16 columns, row by row, and reads column by column, rotating each run of s bits by (column mod s).
// Backwards, for data carrier `code` = 3 c + d (c = column 0..15): the I bits and the Q bits are each
// rotated right by c mod s, and bit t of the result is coded bit 16 (d NB + t) + c of the symbol --
// bit c of the 16-bit unit number d NB + t.  `hw0` = 16-bit units of the packet before this symbol;
// the ones are ORed into the packet's MSB-first words.
// What a lane has to know about one of its FFT bins, packed: bits 0..7 the bin's code (bin_code, signed),
// and for a data carrier 3 c + d: bits 8..11 c, 12..13 d, 14 c mod 2, 15..16 c mod 3.
__host__ __device__ constexpr uint32_t bin_info(int bin)
{
    const int code = bin_code(bin);
    if (code < 0) return (uint32_t)code & 0xFFu;
    const uint32_t c = (uint32_t)code / 3u, d = (uint32_t)code % 3u;
    return (uint32_t)code | (c << 8) | (d << 12) | ((c % 2u) << 14) | ((c % 3u) << 15);
}

template <int NB>
__device__ __forceinline__ void deinterleave_carrier(uint32_t *W, uint32_t hw0, uint32_t info, uint32_t best)
{
    constexpr uint32_t SG = NB / 2 > 1 ? NB / 2 : 1, MASK = (1u << SG) - 1u;
    const uint32_t c = (info >> 8) & 15u;
    uint32_t v = best;
    if (SG > 1) {
        const uint32_t rot = SG == 2 ? (info >> 14) & 1u : (info >> 15) & 3u;           // c mod s
        const uint32_t hi = v >> SG, lo = v & MASK;
        v = ((((hi >> rot) | (hi << (SG - rot))) & MASK) << SG) | (((lo >> rot) | (lo << (SG - rot))) & MASK);
    }
    const uint32_t first = hw0 + ((info >> 12) & 3u) * NB;
    if constexpr (NB == 1) {
        if (v) atomicOr(&W[first >> 1], 1u << (((~first & 1u) << 4) + 15u - c));
    } else {
        // A symbol starts at an odd unit (three units of SIGNAL, then an even number per symbol) and so does
        // every carrier's run: bit 0 is the low half of a word, bits 1 and 2 are the next word, ..., the
        // last bit is a high half.  No branches: a word that gets no one is ORed with zero.
        uint32_t *w = W + (first >> 1);
        const uint32_t sh = 15u - c;
        atomicOr(w, (v >> (NB - 1)) << sh);
#pragma unroll
        for (int k = 1; k < NB / 2; k++) {
            const uint32_t x = (v >> (NB - 1 - 2 * k)) & 3u;                   // bits 2k - 1 (upper half) and 2k
            atomicOr(w + k, ((x | (x << 15)) & 0x10001u) << sh);
        }
        atomicOr(w + NB / 2, (v & 1u) << (16u + sh));
    }
}

// index of the level nearest to x among the first L: the low three bits of each distance are replaced by the
// level's number, so that one unsigned minimum finds both (distances are not negative: their bit patterns order
// as integers)
template <int L>
__device__ __forceinline__ uint32_t nearest_level(float x, const float (&lv)[8])
{
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int a = 0; a < L; a++) m = min(m, (__float_as_uint(fabsf(x - lv[a])) & ~7u) | (uint32_t)a);
    return m & 7u;
}

// The eight bins 8 q + l of one symbol whose data carriers hold NB bits: decisions against the table, the coded
// bits into W, pilot and null-carrier checks.  x[q] = (im, re) of the bin.  The table: (ti[a], tq[b]) for
// point (a << NB/2) | b where it is a product (per_axis); BPSK / QPSK: the points (ti[i], tq[i]); otherwise
// the points in LDS.  Returns the status bits.
template <int NB>
__device__ __forceinline__ uint32_t decide_bins(const v2f (&x)[8], int l, const uint32_t *s_info, const float (&ti)[8],
                                                const float (&tq)[8], bool per_axis, const float2 *points, uint32_t *W,
                                                uint32_t hw0, const PlanParams &P, const DevTables &T, uint32_t pidx)
{
    uint32_t bad = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int bin = 8 * q + l;
        const uint32_t info = s_info[bin];
        const int code = (int)(int8_t)(info & 0xFFu);
        const float ar = x[q].y, ai = x[q].x;
        if (code >= 0) {
            uint32_t best = 0;
            if (NB >= 4 && per_axis) {
                best = (nearest_level<(1 << (NB / 2))>(ar, ti) << (NB / 2)) | nearest_level<(1 << (NB / 2))>(ai, tq);
            } else if (NB <= 2) {
                float best_d = 3.0e38f;
#pragma unroll
                for (int i = 0; i < (1 << NB); i++) {
                    const float d = (ar - ti[i]) * (ar - ti[i]) + (ai - tq[i]) * (ai - tq[i]);
                    if (d < best_d) { best_d = d; best = (uint32_t)i; }
                }
            } else {
                float best_d = 3.0e38f;
                for (int i = 0; i < (1 << NB); i++) {
                    const float2 c = points[i];
                    const float d = (ar - c.x) * (ar - c.x) + (ai - c.y) * (ai - c.y);
                    if (d < best_d) { best_d = d; best = (uint32_t)i; }
                }
            }
            deinterleave_carrier<NB>(W, hw0, info, best);
        } else if (code == -1) {
            if (ar * ar + ai * ai > RX_TOL * RX_TOL) bad |= OFDMTX_RX_NULLS;
        } else {
            const float e = expected_pilot(P, T, bin, pidx);
            if ((ar - e) * (ar - e) + ai * ai > RX_TOL * RX_TOL) bad |= OFDMTX_RX_PILOTS;
        }
    }
    return bad;
}

constexpr int RX_TG = 72;              // float2 per symbol in the transposition layout (8 rows x 9)

// One wavefront per packet, eight OFDM symbols at a time, eight lanes per symbol (lane = 8 g + l).
// The forward DFT of a symbol is the inverse one with real and imaginary parts exchanged on the
// way in and on the way out; as 8 x 8: lane l transforms the samples l + 8 m over m, multiplies by
// W64^{l p}, the eight lanes of the symbol transpose through LDS, lane p transforms over l and
// holds the bins 8 q + p, q = 0..7.
template <bool UNIFORM>
__global__ void __launch_bounds__(64 * RX_WAVES, 5)
k_rx_demod(PlanParams P, DevTables T, const float2 *__restrict__ iq, uint8_t *__restrict__ coded,
           uint32_t *__restrict__ status, uint32_t wave_words, uint32_t *turn)
{
    extern __shared__ uint32_t s_rxw[];                   // per wave: the packet's coded bits, MSB-first words
    __shared__ v2f s_tw[64];                              // W64^{l p} / (8 amp) at 8 p + l
    __shared__ float2 s_lut[LUT_SIZE];
    __shared__ v2f s_t[RX_WAVES][8 * RX_TG];
    __shared__ uint32_t s_info[64];                       // what each FFT bin carries (bin_info)
    __shared__ int s_sep[2];                              // the 16-QAM / 64-QAM table is a product I[hi] x Q[lo]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = lane & 7, g = lane >> 3;
    // the transmitter's rows are (amp / 8) * sum_k X[k] e^{+j 2 pi k n / 64}: the forward DFT
    // returns 8 * amp * X[k]
    const float inv = 1.0f / (8.0f * P.amp);
    for (int i = tid; i < 64; i += 64 * RX_WAVES) {
        const float2 w = T.tw64[((i & 7) * (i >> 3)) & 63];
        s_tw[i] = (v2f){w.x * inv, w.y * inv};
        s_info[i] = bin_info(i);
    }
    for (int i = tid; i < LUT_SIZE; i += 64 * RX_WAVES) s_lut[i] = T.lut[i];
    if (tid < 2) s_sep[tid] = 1;
    __syncthreads();
    // A table whose point (a << h) | b is (I[a], Q[b]) is searched per axis: 2 * 2^h distances
    // instead of 4^h, the same nearest point.  The standard tables are of that kind (17.3.5.7: the
    // first half of a carrier's bits selects I, the second half Q); a caller's own table need not be.
    for (int i = tid; i < 16 + 64; i += 64 * RX_WAVES) {
        const int big = i >= 16, off = big ? 64 : 16, h = big ? 3 : 2, idx = big ? i - 16 : i;
        const float2 c = s_lut[off + idx];
        if (c.x != s_lut[off + (idx >> h << h)].x || c.y != s_lut[off + (idx & ((1 << h) - 1))].y) s_sep[big] = 0;
    }
    __syncthreads();

    uint32_t *W = s_rxw + (size_t)wave * wave_words;
    v2f *buf = s_t[wave];
    float sig_i[8] = {}, sig_q[8] = {};                   // the SIGNAL symbol's two points, in scalar registers
    for (int i = 0; i < 2; i++) {
        sig_i[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_lut[LUT_PM1 + i].x)));
        sig_q[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_lut[LUT_PM1 + i].y)));
    }
    const bool swap_iq = (P.flags & OFDMTX_F_SWAPIQ) != 0;
    const uint32_t pre_rows = (P.flags & OFDMTX_F_PREAMBLE)
                                  ? ((P.flags & (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B)) ? 6u : 4u) : 0u;

    // Packets of a uniform plan are dealt out in turn; those of a ragged plan (5 to 1400 symbols each) go to
    // whichever wave is free: `turn` counts them (zero at launch).
    auto next_packet = [&](uint32_t p) -> uint32_t {
        if (UNIFORM) return p + gridDim.x * RX_WAVES;
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(turn, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    for (uint32_t p = UNIFORM ? blockIdx.x * RX_WAVES + wave : next_packet(0u); p < P.n_pkts; p = next_packet(p)) {
        const Pkt k = load_pkt<UNIFORM>(P, p);
        const RegimeInfo ri = regime_info((int)k.regime);
        const size_t row0 = (UNIFORM ? (size_t)p * P.rows_per_pkt : (size_t)P.row_start[p]) + pre_rows;
        const uint32_t msg_len = 6u + k.n_sym * (uint32_t)ri.n_cbps / 8u;
        const uint32_t n_words = (msg_len + 3u) / 4u;
        for (uint32_t w = lane; w < n_words + 1u; w += 64) W[w] = 0u;
        uint32_t bad = 0;
        // the packet's DATA table in registers: by axis where it is a product (unused levels out of reach);
        // BPSK / QPSK: the points themselves, (lv_i[i], lv_q[i])
        const int h = ri.n_bpsc >> 1;
        const bool by_axis = ri.n_bpsc >= 4 && s_sep[ri.n_bpsc == 6];
        float lv_i[8], lv_q[8];
#pragma unroll
        for (int a = 0; a < 8; a++) {
            const bool used = by_axis ? a < (1 << h) : a < (1 << ri.n_bpsc) && ri.n_bpsc <= 2;
            const float2 ci = s_lut[ri.lut_off + (by_axis ? a << h : a)], cq = s_lut[ri.lut_off + a];
            lv_i[a] = used ? ci.x : 1.0e18f;
            lv_q[a] = used ? cq.y : 1.0e18f;
        }
        wave_sync();

        const uint32_t rows = k.n_sym + 1u;               // SIGNAL and the DATA symbols
        // a group's samples are fetched while the group before it is decided: (im, re) in; an
        // I/Q-exchanged transmission is exchanged back first, which leaves (re, im)
        const float2 *src = iq + row0 * SYM_LEN + 16u + (uint32_t)(g * SYM_LEN + l);     // behind the cyclic prefix
        float2 nx[8];
        if ((uint32_t)g < rows) {
#pragma unroll
            for (int m = 0; m < 8; m++) nx[m] = src[8 * m];
        }
        for (uint32_t r0 = 0; r0 < rows; r0 += 8u) {
            const uint32_t s = r0 + (uint32_t)g;
            const bool active = s < rows;
            v2f x[8];
            if (active) {
#pragma unroll
                for (int m = 0; m < 8; m++) x[m] = swap_iq ? (v2f){nx[m].x, nx[m].y} : (v2f){nx[m].y, nx[m].x};
            }
            if (s + 8u < rows) {
                const float2 *next = src + (size_t)(r0 + 8u) * SYM_LEN;
#pragma unroll
                for (int m = 0; m < 8; m++) nx[m] = next[8 * m];
            }
            if (active) {
                idft8(x);
#pragma unroll
                for (int q = 0; q < 8; q++) x[q] = cmul(x[q], s_tw[8 * q + l]);
#pragma unroll
                for (int q = 0; q < 8; q++) buf[g * RX_TG + 9 * q + l] = x[q];
            }
            wave_sync();
            if (active) {
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = buf[g * RX_TG + 9 * l + i];
                idft8(x);                                 // x[q] = (im, re) of bin 8 q + l

                const uint32_t pidx = (k.pilot_base + s) % 127u;
                if (s == 0) {
                    // SIGNAL is BPSK whatever the rate (17.3.4)
                    bad |= decide_bins<1>(x, l, s_info, sig_i, sig_q, false, s_lut, W, 0u, P, T, pidx);
                } else {
                    const uint32_t hw0 = 3u + (s - 1u) * (uint32_t)ri.n_cbps / 16u;
                    const float2 *points = s_lut + ri.lut_off;
                    switch (ri.n_bpsc) {
                    case 1: bad |= decide_bins<1>(x, l, s_info, lv_i, lv_q, false, points, W, hw0, P, T, pidx); break;
                    case 2: bad |= decide_bins<2>(x, l, s_info, lv_i, lv_q, false, points, W, hw0, P, T, pidx); break;
                    case 4: bad |= decide_bins<4>(x, l, s_info, lv_i, lv_q, by_axis, points, W, hw0, P, T, pidx); break;
                    default: bad |= decide_bins<6>(x, l, s_info, lv_i, lv_q, by_axis, points, W, hw0, P, T, pidx); break;
                    }
                }
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

// The coded bits of one packet, MSB first, through a 64-bit window (next bit = bit 63).  The words come
// from the packet's row of the LDS staging area, which holds words first .. end - 1 of the packet.
struct CodedWindow {
    const uint32_t *row;           // word `first` of the packet
    uint32_t first, end;           // staged at present
    uint32_t pos;                  // next word to take
    uint64_t win;
    uint32_t avail;
    __device__ __forceinline__ void refill()
    {
        if (avail <= 32u && pos < end) {
            win |= (uint64_t)__builtin_bswap32(row[pos - first]) << (32u - avail);
            avail += 32u;
            pos++;
        }
    }
};

// Six steps of the decoder.  `hist` holds the decoded bits, newest in bit 0: bit d-1 = u[n-d].
// g0 = 133 octal taps u[n] and the delays 2, 3, 5, 6; g1 = 171 octal taps u[n] and the delays 1, 2,
// 3, 6 (17.3.5.5).  `pattern` says for each step which coded bits were sent (two bits per step, step
// 0 lowest: bit 0 = A, bit 1 = B); u[n] follows from A where A was sent, from B otherwise, and where
// both were sent the second one is checked against the first.  The six decoded bits are the low six
// bits of hist afterwards; only bit 0 of `mismatch` means anything.
constexpr uint32_t PATTERN_1_2 = 0xFFFu;       // A B every step
constexpr uint32_t PATTERN_3_4 = 0x9E7u;       // A0 B0 A1 B2 of every three steps (17.3.5.5, Figure 17-9)
constexpr uint32_t PATTERN_2_3 = 0x777u;       // A0 B0 A1 of every two

__device__ __forceinline__ void decode_six(CodedWindow &in, uint32_t &hist, uint32_t pattern, uint32_t &mismatch)
{
    in.refill();                                            // at most 12 bits go
    const uint32_t bits = (uint32_t)(in.win >> 32);        // the next 32 coded bits, first one in bit 31
    const uint32_t both_of = pattern & (pattern >> 1);     // bit 2 i: both coded bits of step i were sent
    uint32_t at = 31u;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint32_t both = (both_of >> (2 * i)) & 1u;
        // what the step's first bit says if it is A, what its last bit says if it is B (bit 0 of each)
        const uint32_t ua = __popc(hist & 0x36u) + (bits >> at);
        const uint32_t ub = __popc(hist & 0x27u) + (bits >> (at - both));
        const uint32_t differ = ua ^ ub;
        mismatch |= differ & both;
        hist = (hist << 1) | ((ub ^ (differ & (pattern >> (2 * i)))) & 1u);      // A's word where A was sent
        at -= 1u + both;
    }
    in.win <<= 31u - at;
    in.avail -= 31u - at;
}

// The same with the pattern known to the compiler (every packet of a uniform plan has it): the group's coded
// positions in the window are constants then.
template <uint32_t PATTERN>
__device__ __forceinline__ void decode_six_fixed(CodedWindow &in, uint32_t &hist, uint32_t &mismatch)
{
    in.refill();
    const uint32_t bits = (uint32_t)(in.win >> 32);        // the next 32 coded bits, first one in bit 31
    int at = 31;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint32_t have = (PATTERN >> (2 * i)) & 3u;
        uint32_t u;
        if (have == 3u) {
            u = __popc(hist & 0x36u) + (bits >> at);
            mismatch |= u ^ (__popc(hist & 0x27u) + (bits >> (at - 1)));
            at -= 2;
        } else {
            u = __popc(hist & (have == 1u ? 0x36u : 0x27u)) + (bits >> at);
            at -= 1;
        }
        hist = (hist << 1) | (u & 1u);
    }
    in.win <<= 31 - at;
    in.avail -= (uint32_t)(31 - at);
}

constexpr int DEC_CHUNK = 16;          // coded words of every packet staged per round
constexpr int DEC_PER = 64 / DEC_CHUNK;  // packets a wave-wide load or store covers
constexpr int DEC_TURNS = 64 / DEC_PER;
constexpr int DEC_ROW = DEC_CHUNK + 1; // row pitch of the staging areas, in words (odd: lanes walk their own rows)

// One wavefront per 64 packets, one lane per packet.  A packet's coded bytes and its payload bytes lie apart
// from the next packet's, so lanes reading and writing their own packets word by word would touch 64
// memory lines per instruction.  Instead the wave works in rounds: the next DEC_CHUNK coded words of every
// packet are fetched by the whole wave (DEC_CHUNK lanes per packet, contiguous bytes) into LDS, every lane
// decodes what was staged for its packet and leaves the payload words in LDS, the wave stores them
// (DEC_CHUNK lanes per packet again; a round leaves fewer payload words than it takes coded ones).  The decoded bits are handled 32 at a time: descrambled with one XOR,
// bit-reversed (17.3.5.1 sends every byte LSB first, so the reversed word holds four bytes in order), the
// bytes fed to the CRC by table.
template <bool UNIFORM>
__global__ void __launch_bounds__(64)
k_rx_decode(PlanParams P, DevTables T, const uint8_t *__restrict__ coded, uint8_t *__restrict__ payload_out,
            uint32_t *__restrict__ status)
{
    __shared__ uint32_t s_crc[256];                       // reflected CRC-32 of one byte (polynomial 0xEDB88320)
    __shared__ uint32_t s_scr[127];                       // 32 bits of the scrambler's sequence from every phase
    __shared__ uint32_t s_in[64 * DEC_ROW], s_out[64 * DEC_ROW];
    __shared__ uint64_t s_src[64], s_dst[64];             // per packet: coded words, payload bytes
    __shared__ uint32_t s_words[64], s_count[64];         // coded words of the packet; payload words left this round
    __shared__ int s_store[64], s_at[64];                 // payload bytes of the packet; position of the first word left

    const int lane = threadIdx.x, part = lane / DEC_CHUNK, col = lane % DEC_CHUNK;
    for (uint32_t i = lane; i < 256u; i += 64u) {
        uint32_t c = i;
        for (int b = 0; b < 8; b++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        s_crc[i] = c;
    }
    for (uint32_t i = lane; i < 127u; i += 64u) s_scr[i] = T.scr_word[i];

    const uint32_t p = blockIdx.x * 64u + (uint32_t)lane;
    const bool valid = p < P.n_pkts;                      // a lane without a packet helps with the copies
    const Pkt k = load_pkt<UNIFORM>(P, valid ? p : 0u);
    const RegimeInfo ri = regime_info((int)k.regime);
    const uint32_t n_words = valid ? (6u + k.n_sym * (uint32_t)ri.n_cbps / 8u + 3u) / 4u : 0u;
    // MAC header + payload + FCS (pu:28-36,193); a raw-PSDU plan's bytes are the whole PSDU, which
    // comes back whole (FCS included; without one -- fewer than four bytes -- nothing is checked)
    const bool raw = (P.flags & OFDMTX_F_RAW_PSDU) != 0;
    const int hdr = raw ? 0 : 24;
    const uint32_t want_len = raw ? k.pay_len : 24u + k.pay_len + 4u;
    const uint32_t fcs_at = want_len >= 4u ? want_len - 4u : want_len;   // first FCS byte
    const int n_store = valid ? (int)k.pay_len : 0;
    // DATA: SERVICE (16 bits), PSDU, six tail bits, pad (17.3.5); N_dbps is a multiple of six
    const uint32_t groups = valid ? k.n_sym * (uint32_t)ri.n_dbps / 6u : 0u;
    const uint32_t pattern = ri.punct == 0 ? PATTERN_1_2 : ri.punct == 1 ? PATTERN_3_4 : PATTERN_2_3;
    s_src[lane] = reinterpret_cast<uint64_t>(coded + k.msg_off);
    s_dst[lane] = reinterpret_cast<uint64_t>(payload_out + k.pay_off);
    s_words[lane] = n_words;
    s_store[lane] = n_store;
    uint32_t most = n_words;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, o));
    __syncthreads();

    CodedWindow in = {s_in + lane * DEC_ROW, 0u, 0u, 0u, 0, 0u};
    uint32_t hist = 0, bad = 0, mism = 0;
    uint32_t crc = 0xFFFFFFFFu, fcs = 0, carry = 0, widx = 0, cnt = 0, g = 0, n_left = 0;
    int at_left = 0;
    bool finished = !valid;
    uint64_t acc = 0;

    // four payload bytes for position `at` of the packet: kept for the wave to store, unless none of them exists
    auto leave = [&](int at, uint32_t value) {
        if (at + 4 <= 0 || at >= n_store) return;
        if (n_left == 0u) at_left = at;
        s_out[lane * DEC_ROW + (int)n_left] = value;
        n_left++;
    };
    // decoded bits 32 widx .. 32 widx + 31, first one in bit 31: bytes 4 widx - 2 .. 4 widx + 1 of the PSDU
    auto word = [&](uint32_t u) {
        // 17.3.5.4: the scrambler's 127-bit sequence, restarted with every packet (pu:344-358)
        const uint32_t r = __brev(u ^ s_scr[(32u * widx) % 127u]);
        const int by0 = 4 * (int)widx - 2;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int by = by0 + b;
            const uint32_t byte = (r >> (8 * b)) & 0xFFu;
            if (by < 0) {
                if (byte) bad |= OFDMTX_RX_SERVICE;
            } else if ((uint32_t)by < fcs_at) {
                crc = s_crc[(crc ^ byte) & 0xFFu] ^ (crc >> 8);
            } else if ((uint32_t)by < want_len) {
                fcs |= byte << (8u * ((uint32_t)by - fcs_at));
            } else if ((uint32_t)by == want_len) {
                // the six tail bits are zero AFTER scrambling (17.3.5.2): looked at before the XOR
                if ((u >> (24 - 8 * b)) & 0xFCu) bad |= OFDMTX_RX_TAIL;
            }
        }
        // payload bytes 4 widx - 4 - hdr .. : the second half of the word before, the first half of this one
        leave(4 * (int)widx - 4 - hdr, carry | (r << 16));
        carry = r >> 16;
        widx++;
    };
    auto six_done = [&]() {
        acc = (acc << 6) | (hist & 63u);
        cnt += 6u;
        if (cnt >= 32u) {
            cnt -= 32u;
            word((uint32_t)(acc >> cnt));
        }
        g++;
    };
    // a group takes at most 12 coded bits: they are in the window, or the next word is staged, or there is
    // no next word (a packet's words hold all the bits its groups take)
    auto can_go = [&]() { return g < groups && (in.avail >= 12u || in.pos < in.end || in.pos >= n_words); };

    // words c0 .. c0 + DEC_CHUNK - 1 of every packet, DEC_PER packets per load: fetched one round ahead and
    // held in registers while the round before is decoded
    uint32_t ahead[DEC_TURNS] = {};
    auto fetch = [&](uint32_t c0) {
#pragma unroll
        for (int i = 0; i < DEC_TURNS; i++) {
            const int pk = DEC_PER * i + part;
            if (c0 + (uint32_t)col < s_words[pk]) ahead[i] = reinterpret_cast<const uint32_t *>(s_src[pk])[c0 + (uint32_t)col];
        }
    };
    fetch(0u);
    for (uint32_t c0 = 0; c0 < most; c0 += DEC_CHUNK) {
#pragma unroll
        for (int i = 0; i < DEC_TURNS; i++) s_in[(DEC_PER * i + part) * DEC_ROW + col] = ahead[i];
        wave_sync();
        if (c0 + DEC_CHUNK < most) fetch(c0 + DEC_CHUNK);
        in.first = c0;
        in.end = min(n_words, c0 + (uint32_t)DEC_CHUNK);
        n_left = 0;
        if (c0 == 0u && valid) {
            // ---- SIGNAL: 24 bits at rate 1/2 (17.3.4): RATE R1-R4, reserved, LENGTH (LSB first), parity, tail
            for (int i = 0; i < 4; i++) decode_six(in, hist, PATTERN_1_2, mism);
            const uint32_t sig = hist & 0xFFFFFFu;            // bit 23 - n = SIGNAL bit n
            const uint32_t rate = sig >> 20;
            const uint32_t length = (__brev(sig << 8) >> 5) & 0xFFFu;     // SIGNAL bits 5..16, LSB first
            if (rate != (uint32_t)ri.rate_bits || ((sig >> 19) & 1u) || (__popc(sig >> 6) & 1u) || (sig & 0x3Fu)) bad |= OFDMTX_RX_SIGNAL;
            if (length != want_len) bad |= OFDMTX_RX_LENGTH;
        }
        if (UNIFORM && ri.punct == 0) {
            while (can_go()) { decode_six_fixed<PATTERN_1_2>(in, hist, mism); six_done(); }
        } else if (UNIFORM && ri.punct == 1) {
            while (can_go()) { decode_six_fixed<PATTERN_3_4>(in, hist, mism); six_done(); }
        } else if (UNIFORM) {
            while (can_go()) { decode_six_fixed<PATTERN_2_3>(in, hist, mism); six_done(); }
        } else {
            // packets of a ragged plan differ from lane to lane
            while (can_go()) { decode_six(in, hist, pattern, mism); six_done(); }
        }
        if (g == groups && !finished) {
            if (cnt) word((uint32_t)((acc & ((1ull << cnt) - 1ull)) << (32u - cnt)));
            leave(4 * (int)widx - 4 - hdr, carry);
            finished = true;
        }
        s_count[lane] = n_left;
        s_at[lane] = at_left;
        wave_sync();
        // ---- the payload words left in LDS out to memory: whole words where the packet's bytes are
        // aligned and the word lies inside the payload, single bytes otherwise
#pragma unroll 4
        for (int i = 0; i < DEC_TURNS; i++) {
            const int pk = DEC_PER * i + part;
            if ((uint32_t)col < s_count[pk]) {
                uint8_t *pay = reinterpret_cast<uint8_t *>(s_dst[pk]);
                const int at = s_at[pk] + 4 * col, len = s_store[pk];
                const uint32_t value = s_out[pk * DEC_ROW + col];
                if (at >= 0 && at + 4 <= len && (s_dst[pk] & 3u) == 0) {
                    *reinterpret_cast<uint32_t *>(pay + at) = value;
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (at + b >= 0 && at + b < len) pay[at + b] = (uint8_t)(value >> (8 * b));
                }
            }
        }
        wave_sync();
    }
    if (valid) {
        if (fcs_at != want_len && (crc ^ 0xFFFFFFFFu) != fcs) bad |= OFDMTX_RX_FCS;
        if (mism & 1u) bad |= OFDMTX_RX_CODE;
        status[p] |= bad;
    }
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
    if (lds > 44000) return OFDMTX_ETOOBIG;                       // next to the kernel's 20 KB of static LDS: under 64 KB
    // one round of resident workgroups, each looping over packets: a second, partly filled round would
    // run at a fraction of the occupancy
    int per_cu = 0;
    if (pl->pp.uniform)
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rx_demod<true>, 64 * RX_WAVES, lds));
    else
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rx_demod<false>, 64 * RX_WAVES, lds));
    uint64_t blocks = ((uint64_t)pl->pp.n_pkts + RX_WAVES - 1) / RX_WAVES;
    const uint64_t cap = (uint64_t)ctx->n_cu * (uint64_t)(per_cu > 0 ? per_cu : 1);
    if (blocks > cap) blocks = cap;
    const DevTables tab = ctx->tables();
    const float2 *iq = reinterpret_cast<const float2 *>(d_iq);
    if (pl->pp.uniform) {
        hipLaunchKernelGGL((k_rx_demod<true>), dim3((unsigned)blocks), dim3(64 * RX_WAVES), lds, (hipStream_t)stream,
                           pl->pp, tab, iq, d_coded, d_status, wave_words, (uint32_t *)nullptr);
    } else {
        // this launch's packet counter: one of RX_TURNS, zeroed on the launch's stream in front of it
        uint32_t *turn = ctx->d_turns + 16u * (ctx->turn.fetch_add(1u) % (uint32_t)RX_TURNS);
        HIPCHK(hipMemsetAsync(turn, 0, sizeof(uint32_t), (hipStream_t)stream));
        hipLaunchKernelGGL((k_rx_demod<false>), dim3((unsigned)blocks), dim3(64 * RX_WAVES), lds, (hipStream_t)stream,
                           pl->pp, tab, iq, d_coded, d_status, wave_words, turn);
    }
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
