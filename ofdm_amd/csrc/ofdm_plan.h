// Shared between the translation units of libofdmtx: the 802.11a geometry (rate table, carrier
// layout), the plan as the kernels see it (PlanParams, PktRec) and the host-side plan object.
//
// Reference behaviour restated here is cited as file:line relative to the reference root
// (pu = ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py, ftw/ = ftw_80211_ofdm_tx/src/ftw/).
#pragma once

#include <stdint.h>

#include <mutex>
#include <vector>

#include "ofdm_common.h"

namespace ofdmtx_detail {

constexpr int MAXLEN = 3168;          // pu:119
constexpr int MAX_PLCP = 3264;        // 5 + 3168 + 28 tail/pad bytes, rounded up to 64
constexpr int SYM_LEN = 80;           // examples/ftw_ofdm.py:75
constexpr int PRE_ROWS = 4;           // ftw/ftw_ofdm_preamble.cc:79
constexpr int PNC_PRE_ROWS = 6;       // ftw/ftw_pnc_ofdm_preamble.cc:81-107
constexpr int GAP_ROWS = 13;          // ftw/ftw_zerogap.cc:88, ftw/ftw_pnc_zerogap.cc:88
// Constellation LUT slots (complex entries).  Every table starts at a power of two at least
// its size, so that slot = offset | index:
//   0 zero (null carriers) | 2 (-1,0), 3 (+1,0): SIGNAL BPSK and the pilots | 4,5 BPSK | 8..11 QPSK
//   | 16..31 16-QAM | 64..127 64-QAM
constexpr int LUT_PM1 = 2;            // slot of -1; +1 follows

struct RegimeInfo {
    int n_cbps, n_bpsc, n_dbps, rate_bits, punct, lut_off;   // punct: 0 none, 1 = 3/4, 2 = 2/3
};

// pu:39-78 (modulation / rate), pu:81 (N_dbps), pu:129-144 (RATE nibble)
__host__ __device__ __forceinline__ RegimeInfo regime_info(int regime)
{
    switch (regime) {
    case 1: return {48, 1, 24, 0x0D, 0, 4};
    case 2: return {48, 1, 36, 0x0F, 1, 4};
    case 3: return {96, 2, 48, 0x05, 0, 8};
    case 4: return {96, 2, 72, 0x07, 1, 8};
    case 5: return {192, 4, 96, 0x09, 0, 16};
    case 6: return {192, 4, 144, 0x0B, 1, 16};
    case 7: return {288, 6, 192, 0x01, 2, 64};
    default: return {288, 6, 216, 0x03, 1, 64};
    }
}

// Per-packet record of a ragged plan (device array).
struct PktRec {
    uint64_t pay_off;
    uint64_t plcp_off;
    uint64_t msg_off;
    uint32_t sym_off;      // index of the SIGNAL symbol among all symbols of the batch
    uint32_t pay_len;
    uint32_t n_sym;
    uint32_t pilot_base;
    uint16_t txtime;
    uint8_t regime;
    uint8_t pad_;
    uint32_t pad2_;
};

// Passed by value to the kernels.
struct PlanParams {
    uint32_t n_pkts;
    uint32_t uniform;
    // uniform batches: closed forms
    uint32_t regime, pay_len, n_sym, txtime;
    uint64_t pay_stride;
    uint32_t plcp_stride, msg_stride, rows_per_pkt;
    uint32_t pilot_mode, pilot_start;
    // ragged batches
    const PktRec *recs;
    const uint32_t *row_start;    // n_pkts + 1
    uint32_t *next_pkt;           // work counters (next packet, waves done): [0], [1] k_tx; [2], [3] k_encode_bits;
                                  // zero between launches (the kernels restore them)
    uint32_t grab_tx, grab_enc;   // packets per draw from the counters
    uint32_t total_rows;
    uint32_t flags;
    float amp;
};

struct Pkt {
    uint64_t pay_off, plcp_off, msg_off;
    uint32_t sym_off, pay_len, n_sym, pilot_base, txtime, regime;
};

template <bool UNIFORM>
__device__ __forceinline__ Pkt load_pkt(const PlanParams &P, uint32_t p)
{
    Pkt k;
    if (UNIFORM) {
        k.pay_off = (uint64_t)p * P.pay_stride;
        k.plcp_off = (uint64_t)p * P.plcp_stride;
        k.msg_off = (uint64_t)p * P.msg_stride;
        k.sym_off = p * (P.n_sym + 1);
        k.pay_len = P.pay_len;
        k.n_sym = P.n_sym;
        k.pilot_base = P.pilot_mode == OFDMTX_PILOT_REF_STREAM
                           ? (uint32_t)((P.pilot_start + (uint64_t)p * (P.n_sym + 1)) % 127u)
                           : P.pilot_start % 127u;
        k.txtime = P.txtime;
        k.regime = P.regime;
    } else {
        const PktRec r = P.recs[p];
        k.pay_off = r.pay_off;
        k.plcp_off = r.plcp_off;
        k.msg_off = r.msg_off;
        k.sym_off = r.sym_off;
        k.pay_len = r.pay_len;
        k.n_sym = r.n_sym;
        k.pilot_base = r.pilot_base;
        k.txtime = r.txtime;
        k.regime = r.regime;
    }
    return k;
}

// What FFT bin k carries: ftw/ftw_ofdm_pilot_cc.cc:119-155 then ftw/ftw_ofdm_cmap_cc.cc:102-107.
//   >= 0 : data carrier index 0..47;  -1 : null;  -2 : pilot with +p;  -3 : pilot with -p
__host__ __device__ constexpr int bin_code(int k)
{
    const int f = k < 32 ? k : k - 64;
    if (f == 0 || f > 26 || f < -26) return -1;
    if (f == -21 || f == -7 || f == 7) return -2;
    if (f == 21) return -3;
    const int t = f + 26;
    return t - (t > 5) - (t > 19) - (t > 26) - (t > 33) - (t > 47);
}

}  // namespace ofdmtx_detail

struct ofdmtx_plan {
    ofdmtx_ctx *ctx;
    int device;            // the context's device, kept so that destroying the plan never touches the context
    ofdmtx_detail::PlanParams pp;
    ofdmtx_plan_sizes sizes;
    // host copies of the layout (ragged); empty for uniform
    std::vector<ofdmtx_detail::PktRec> recs;
    std::vector<uint32_t> row_start;
    std::vector<uint32_t> plcp_len, msg_len;
    ofdmtx_detail::PktRec *d_recs;
    uint32_t *d_row_start;
    // plans built on the device: per-packet lengths stay there until the layout is asked for
    uint32_t *d_plcp_len, *d_msg_len;
    mutable bool layout_on_host;
    ofdmtx_pkt_info uni;   // uniform geometry
    uint32_t tx_a_words;   // LDS words per wave the fused kernel needs for the largest packet (first region)
    uint32_t tx_u_words;   // ... for the scrambled words of the longest packet (payload input)
    uint32_t tx_m_words;   // ... for the message words of the longest packet (message input)
    uint32_t max_sym_rows; // largest n_sym + 1 of the plan
    // Ragged plans share their device work counters between launches: an event is recorded behind
    // every launch, and a launch on another stream than the previous one is made to wait for it
    // (on the device).  launch_lock covers claim -> enqueue -> mark on the host.
    mutable hipEvent_t busy_ev = nullptr;
    mutable hipStream_t last_stream = nullptr;
    mutable bool launched = false;
    mutable std::mutex launch_lock;
};

// ofdmtx.hip: claim orders a launch about to be enqueued on `st` behind the plan's previous launch
// when that ran on another stream (uniform plans: nothing to do); mark_launched records the event
// behind the launch just enqueued.  Call both under pl->launch_lock.
int ofdmtx_plan_claim(const ofdmtx_plan *pl, hipStream_t st);
int ofdmtx_plan_mark_launched(const ofdmtx_plan *pl, hipStream_t st);
