// libofdmtx: batched 802.11a OFDM transmit chain for gfx950 (MI355X).
//
// Kernels (one wavefront per packet):
//   k_tx          : payload -> PLCP bits -> scrambler -> K=7 conv. encoder + puncture ->
//                   interleaver -> constellation + pilots + carrier map -> 64-point inverse FFT
//                   (radix 8x8, eight lanes per OFDM symbol) -> cyclic prefix, scale, preamble /
//                   zero-gap rows -> fully coalesced 16-byte stores; with FROM_MSG the same
//                   symbol-level part fed from mapper messages (ofdmtx_modulate)
//   k_encode_bits : the bit-level part alone, writing the mapper messages and the optional
//                   PLCP / scrambled / coded taps
// The output of a batch is one dense stream of 80-sample rows (640 B each); a wavefront
// produces eight consecutive rows = 5 KiB of contiguous IQ per iteration.
//
// Reference behaviour restated here is cited as file:line relative to the reference root
// (pu = ftw_80211_ofdm_tx/src/examples/ftw_packet_utils.py, ftw/ = ftw_80211_ofdm_tx/src/ftw/).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <new>
#include <utility>
#include <vector>

// Cache policy of this file's IQ stores: non-temporal + sc1 (18).  Against sc1 alone, which the other
// translation units keep: 7.79 against 7.93 ms per 10^6 packets into a spread output buffer, 8.27
// against 8.52 ms into an ordinary one (variants interleaved in one process).
#define IQ_STORE_AUX 18
#include "ofdm_common.h"
#include "ofdm_plan.h"
#include "ofdm_tables.h"

// Section markers in the generated assembly (tools/asm_sections.py); no code is emitted.
#ifdef OFDMTX_ASM_MARKERS
#define SEC(name) asm volatile("; SEC " name)
#else
#define SEC(name) do { } while (0)
#endif

namespace {

using namespace ofdmtx_detail;

// ---- static work area of ofdmtx_tx_ragged_static --------------------------------------------
// The plan tables of one ragged batch of up to OFDMTX_STATIC_MAX_PKTS packets, in a file-scope
// device array: every GPU the library runs on holds its own copy (at its own address, found once
// per context with hipGetSymbolAddress) until the process ends.  Host code never uses the array's
// name as an address -- only the pointer ofdmtx_create stored in the context.
constexpr size_t up256(size_t v) { return (v + 255u) & ~(size_t)255u; }
constexpr int PLAN_COLS = 5;          // scanned per packet: payload bytes, PLCP stride, message stride, symbols, rows
constexpr int SCAN_ITEMS = 8;         // per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;
struct StaticLayout {
    static constexpr size_t N = OFDMTX_STATIC_MAX_PKTS;
    static constexpr size_t n_blocks = (N + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK;
    static constexpr size_t recs = 0;
    static constexpr size_t row_start = recs + up256(N * sizeof(PktRec));
    static constexpr size_t counters = row_start + up256((N + 1) * sizeof(uint32_t));     // a 256-byte line of its own
    static constexpr size_t plcp_len = counters + 256;
    static constexpr size_t msg_len = plcp_len + up256(N * sizeof(uint32_t));
    static constexpr size_t col = msg_len + up256(N * sizeof(uint32_t));
    static constexpr size_t sums = col + up256(PLAN_COLS * (N + 1) * sizeof(unsigned long long));
    static constexpr size_t out = sums + up256(PLAN_COLS * n_blocks * sizeof(unsigned long long));
    static constexpr size_t total = out + 256;
};
static __device__ __align__(256) unsigned char g_static_area[StaticLayout::total];    // 6.6 MB

// Every context on a device shares that device's copy of the area, so calls of
// ofdmtx_tx_ragged_static are ordered per DEVICE: the host part runs under the device's lock, and
// an event recorded behind every call's kernels makes a call on another stream start behind them.
// The events are made by ofdmtx_create (the entry point itself creates nothing).
struct StaticGuard {
    std::mutex lock;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool launched = false;
};
constexpr int MAX_DEVICES = 64;
StaticGuard g_static_guard[MAX_DEVICES];

// Work counters of a ragged plan: ctr[0] is the next packet to draw, ctr[1] counts the waves
// that have left the packet loop.  The last wave out puts both back to zero, so a launch finds
// the counters as the plan's creation left them and nothing has to be cleared between launches
// (every wave draws at least once and draws no more after it has signed off; the reset is
// ordered behind all draws by the sign-off atomics).
__device__ __forceinline__ void sign_off(uint32_t *ctr, uint32_t n_waves, int lane)
{
    if (lane == 0) {
        __threadfence();
        if (atomicAdd(ctr + 1, 1u) == n_waves - 1u) {
            atomicExch(ctr, 0u);
            atomicExch(ctr + 1, 0u);
        }
    }
}

// ======================================================================================
// Bit-level kernel
// ======================================================================================
//
// One wavefront builds one packet.  The bit stream is kept as MSB-first 32-bit words
// (stream bit t of a word at bit 31-t) so that scrambling, encoding and puncturing are word
// operations:
//   1. natural-order MPDU (MAC header + payload) as dwords in LDS
//   2. CRC-32: every lane folds a chunk of dwords (slicing-by-4), multiplies its remainder
//      by x^(8*bytes behind the chunk) mod P, and the lanes' results are XOR-reduced
//   3. PLCP stream words = bit-reversed MPDU dwords + SIGNAL + FCS, XOR scrambler words,
//      tail bits cleared  -> U[] in LDS
//   4. SIGNAL symbol: 48 coded bits, one lane each
//   5. DATA symbols: one lane per OFDM symbol; the symbol's N_dbps(+6) scrambled bits are
//      pulled into registers, encoded + punctured piecewise with shift/XOR and bit-spread
//      tricks, and interleaved with compile-time bit moves -> LDS
//   6. the message is copied out with coalesced dword stores.

constexpr uint32_t CRC_POLY = 0xEDB88320u;   // ftw/ftw_crc32.cc:45

// b(x) * x mod P(x) in the reflected representation (x^0 = bit 31)
__host__ __device__ constexpr uint32_t crc_times_x(uint32_t b) { return (b >> 1) ^ (CRC_POLY & (0u - (b & 1u))); }

// Shared tables of the kernels that scramble: the slicing-by-4 CRC tables, then v(x) * x^4 mod P
// for the 16 values v of a word's low four bits (crc_mulmod); the scrambler's 127 windows of 32
// sequence bits, then in entry 127 the window that meets stream word 0 (the SIGNAL bits stay).
constexpr int CRC_X4 = 1024, CRC_TAB_WORDS = 1024 + 16, SCR_WORD0 = 127, SCR_TAB_WORDS = 128;

__device__ __forceinline__ void fill_scramble_tables(const DevTables &T, uint32_t *s_crc, uint32_t *s_scr, int tid, int n_thr)
{
    for (int i = tid; i < 127; i += n_thr) s_scr[i] = T.scr_word[i];
    if (tid == 0) s_scr[SCR_WORD0] = T.scr_word[0] >> 24;      // bits 24..31 of word 0 are SERVICE bits 0..7
    for (int i = tid; i < 1024; i += n_thr) s_crc[i] = T.crc_tab[i];
    if (tid < 16) s_crc[CRC_X4 + tid] = crc_times_x(crc_times_x(crc_times_x(crc_times_x((uint32_t)tid))));
}

// a(x) * b(x) mod P(x): sum over the bits of a (bit 31 = x^0) of b x^i, four bits of a per step
// from the high powers down -- p <- p x^4 + (four selected multiples b, b x, b x^2, b x^3) -- with
// the bits a shift by four pushes out folded back in through the 16-entry table.
__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b, const uint32_t *s_crc)
{
    const uint32_t b1 = crc_times_x(b), b2 = crc_times_x(b1), b3 = crc_times_x(b2);
    uint32_t p = 0;
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
        if (s) p = (p >> 4) ^ s_crc[CRC_X4 + (p & 15u)];
        p ^= (b & (0u - ((a >> (s + 3)) & 1u))) ^ (b1 & (0u - ((a >> (s + 2)) & 1u))) ^
             (b2 & (0u - ((a >> (s + 1)) & 1u))) ^ (b3 & (0u - ((a >> s) & 1u)));
    }
    return p;
}

// Coded bit with absolute index P (0 = first SIGNAL coded bit) from the scrambled stream
// held as MSB-first 32-bit words: pu:287-307, g0 = 133o -> even index, g1 = 171o -> odd.
__device__ __forceinline__ uint32_t coded_bit(const uint32_t *U, uint32_t P)
{
    uint32_t n = P >> 1;
    uint32_t w = n >> 5;
    uint32_t sh = 31u - (n & 31u);
    uint64_t X = ((uint64_t)(w ? U[w - 1] : 0u) << 32) | U[w];
    uint32_t win = (uint32_t)(X >> sh) & 0x7Fu;          // bit d = u[n-d]
    return __popc(win & ((P & 1u) ? 0x4Fu : 0x6Du)) & 1u;
}

// Index into the coded stream (relative to the symbol's first coded bit) of punctured index
// q (relative to the symbol's first punctured bit): pu:309-341.  Symbols hold whole
// puncturing periods, so the relation is the same in every symbol.
template <int PUNCT>
__device__ constexpr int unpuncture(int q)
{
    return PUNCT == 1 ? 6 * (q / 4) + ((q % 4) == 3 ? 5 : (q % 4))      // keep coded index mod 6 in {0,1,2,5}
         : PUNCT == 2 ? 4 * (q / 3) + q % 3                                // keep coded index mod 4 in {0,1,2}
                      : q;
}

// One bit of the per-symbol bit table, with every index a compile-time constant (a plain
// unrolled loop leaves hipcc free to keep the loop and index the register arrays through
// scratch memory): dst bit J <- coded bit unpuncture(Q), i.e. bit n of stream A (even) or B (odd).
template <int PUNCT, int J, int Q, int NA, int NW>
__device__ __forceinline__ void move_coded_bit(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA],
                                               uint32_t (&dst)[NW])
{
    constexpr int p = unpuncture<PUNCT>(Q), n = p >> 1;
    const uint32_t w = (p & 1) ? Bw[n >> 5] : Aw[n >> 5];
    dst[J >> 5] |= ((w >> (31 - (n & 31))) & 1u) << (31 - (J & 31));
}

// punctured index feeding interleaved position J of a symbol: k = 16r + c with J = R*c + u,
// r = S*(u/S) + ((u%S + c) % S)  (inverse of pu:241-253 followed by pu:267-273)
template <int NBPSC>
__device__ constexpr int deinterleave(int j)
{
    constexpr int R = 3 * NBPSC, S = NBPSC / 2 > 1 ? NBPSC / 2 : 1;
    const int c = j / R, u = j % R;
    return 16 * (S * (u / S) + ((u % S) + c) % S) + c;
}

// Bit t of interleaved word WORD (t = 0 is the MSB); bits beyond the symbol's N contribute 0.
template <int NBPSC, int PUNCT, int WORD, int T, int NA>
__device__ __forceinline__ uint32_t interleaved_bit(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA])
{
    constexpr int J = 32 * WORD + T;
    if constexpr (J >= 48 * NBPSC) {
        return 0u;
    } else {
        constexpr int p = unpuncture<PUNCT>(deinterleave<NBPSC>(J)), n = p >> 1;
        const uint32_t w = (p & 1) ? Bw[n >> 5] : Aw[n >> 5];
        return ((w >> (31 - (n & 31))) & 1u) << (31 - T);
    }
}

template <int NBPSC, int PUNCT, int WORD, int NA, int... T>
__device__ __forceinline__ uint32_t interleaved_word(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA],
                                                     std::integer_sequence<int, T...>)
{
    return (interleaved_bit<NBPSC, PUNCT, WORD, T>(Aw, Bw) | ...);
}

// All words of the symbol, one after the other: the scheduling barrier keeps hipcc from hoisting
// the bit extracts of every word to the front (which costs > 200 VGPRs and with them occupancy).
template <int NBPSC, int PUNCT, int NA, int NW, int... WORD>
__device__ __forceinline__ void interleave_bits(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA],
                                                uint32_t (&out)[NW], std::integer_sequence<int, WORD...>)
{
    ((out[WORD] = interleaved_word<NBPSC, PUNCT, WORD>(Aw, Bw, std::make_integer_sequence<int, 32>{}),
      __builtin_amdgcn_sched_barrier(0)), ...);
}

template <int PUNCT, int NA, int NW, int... Q>
__device__ __forceinline__ void punctured_bits(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA],
                                               uint32_t (&out)[NW], std::integer_sequence<int, Q...>)
{
    (move_coded_bit<PUNCT, Q, Q>(Aw, Bw, out), ...);
}

// --------------------------------------------------------------------------------------
// Interleaving by multiplication instead of bit by bit (QPSK, 16-QAM and 64-QAM at rates 1/2 and 3/4:
// regimes 3, 4, 5, 6 and 8).
//
// A symbol's streams A and B are R = 3 N_bpsc units of U bits (U = 8 at rate 1/2, 12 at rate 3/4); unit
// r holds the encoder outputs of U input bits, of which puncturing keeps all (rate 1/2) or A bits 0,1
// and B bits 0,2 of every three (pu:316-328): 16 bits, the columns c = 0..15 of row r of the
// interleaver's R x 16 block (pu:241-253), even columns from A and odd ones from B.  With
// s = max(N_bpsc / 2, 1), row r = N_bpsc g + s h + i0 feeds bit s h + i of the constellation index of
// carrier 3c + g, i = (i0 - c) mod s (second permutation, pu:267-273): the 48 index bytes of a symbol
// are the transposed block.  Half a unit of one stream (4 to 6 bits, four of them kept) times a
// constant with four bits set puts each kept bit into a byte of its own, at the bit position the
// row has in that byte; the other partial products fall between those places (checked at compile
// time for every value of the field, with the dropped bits cleared beforehand).  Per group g of 16
// carriers this gives four words -- (stream, half) = columns {8q + sigma, 8q + sigma + 2, ..} --
// built from N_bpsc rows each with a bit-field extract, a 24-bit multiply and an and-or per row: for
// 64-QAM 216 + 12 operations for 288 index bits (two per bit otherwise).  The fused kernel's slot words
// then gather their bytes with v_perm_b32; the mapper message takes them column by column.
namespace spread {
template <int NB, int P>
struct Geo {
    static_assert((NB == 2 || NB == 4 || NB == 6) && (P == 0 || P == 1), "QPSK .. 64-QAM, rates 1/2 and 3/4");
    static constexpr int S = NB / 2;                        // rows that rotate among themselves
    static constexpr int U = P == 0 ? 8 : 12;               // stream bits per unit
    static constexpr int NDBPS = 3 * NB * U;                // bits of a stream per symbol
    static constexpr int NA = (NDBPS + 31) / 32;
    static constexpr int NW = (48 * NB + 31) / 32;          // words of the symbol's interleaved bits
    static constexpr int len(int sigma) { return P == 0 ? 4 : 5 + sigma; }     // bits of a field
    // value bit of a field (bit 0 = its last stream bit) that feeds byte b of the word
    static constexpr int src_bit(int sigma, int b)
    {
        return P == 0 ? b : sigma == 0 ? (b < 2 ? b : b + 1) : (b == 0 ? 0 : b == 3 ? 5 : b + 1);
    }
    static constexpr int column(int sigma, int q, int b) { return 8 * q + 2 * (3 - b) + sigma; }
    static constexpr int byte_pos(int sigma, int q, int i0, int b) { return S - 1 - ((i0 - column(sigma, q, b)) % S + S) % S; }
    static constexpr uint32_t mult(int sigma, int q, int i0)
    {
        uint32_t k = 0;
        for (int b = 0; b < 4; b++) k |= 1u << (8 * b + byte_pos(sigma, q, i0, b) - src_bit(sigma, b));
        return k;
    }
    static constexpr uint32_t mask(int sigma, int q, int i0)
    {
        uint32_t m = 0;
        for (int b = 0; b < 4; b++) m |= 1u << (8 * b + byte_pos(sigma, q, i0, b));
        return m;
    }
    // the masks differ only by the rotation of the rows over the bytes: S of them
    static constexpr int mask_class(int sigma, int q, int i0)
    {
        for (int k = 0; k < S; k++)
            if (mask(sigma, q, i0) == mask(0, 0, k)) return k;
        return -1;
    }
    static constexpr bool ok(int sigma, int q, int i0)
    {
        uint32_t live = 0;
        for (int b = 0; b < 4; b++) live |= 1u << src_bit(sigma, b);
        if (mult(sigma, q, i0) >= (1u << 24) || mask_class(sigma, q, i0) < 0) return false;
        for (uint32_t f = 0; f < (1u << len(sigma)); f++) {
            if (f & ~live) continue;
            uint32_t want = 0;
            for (int b = 0; b < 4; b++) want |= ((f >> src_bit(sigma, b)) & 1u) << (8 * b + byte_pos(sigma, q, i0, b));
            if (((f * mult(sigma, q, i0)) & mask(sigma, q, i0)) != want) return false;
        }
        return true;
    }
    static constexpr bool all_ok()
    {
        for (int sigma = 0; sigma < 2; sigma++)
            for (int q = 0; q < 2; q++)
                for (int i0 = 0; i0 < S; i0++)
                    if (!ok(sigma, q, i0)) return false;
        return true;
    }
    static_assert(all_ok(), "a partial product collides with a kept bit");
    // stream bits that puncturing drops, cleared: at rate 3/4 A keeps n % 3 in {0, 1}, B keeps n % 3 in {0, 2}
    static constexpr uint32_t keep_mask(int sigma, int word)
    {
        uint32_t m = 0;
        for (int j = 0; j < 32; j++)
            if (P == 0 || (32 * word + j) % 3 != (sigma == 0 ? 2 : 1)) m |= 1u << (31 - j);
        return m;
    }
};

// bits n0 .. n0 + LEN - 1 of a stream held as MSB-first words, right-aligned
template <int N0, int LEN, int NA>
__device__ __forceinline__ uint32_t field(const uint32_t (&W)[NA])
{
    constexpr int w = N0 >> 5, o = N0 & 31;
    if constexpr (o + LEN <= 32)
        return (W[w] >> (32 - o - LEN)) & ((1u << LEN) - 1u);
    else
        return __funnelshift_l(W[w + 1], W[w], o) >> (32 - LEN);
}

// The masks, held in scalar registers so that "(product & mask) | word so far" is one v_and_or_b32.
struct Masks { uint32_t m[3]; };

// one row's contribution to the word of (group G, stream SIGMA, half Q), on top of acc
template <int NB, int P, int G, int SIGMA, int Q, int RHO, int NA>
__device__ __forceinline__ uint32_t row_bits(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA], const Masks &mk, uint32_t acc)
{
    using geo = Geo<NB, P>;
    constexpr int n0 = geo::U * (NB * G + RHO) + (geo::U / 2) * Q;
    const uint32_t f = SIGMA == 0 ? field<n0, geo::len(0)>(Aw) : field<n0, geo::len(1)>(Bw);
    const uint32_t prod = __umul24(f, geo::mult(SIGMA, Q, RHO % geo::S));
    uint32_t d;
    // (written out: the compiler regroups a chain of these into ands and a v_or3_b32)
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(prod), "s"(mk.m[geo::mask_class(SIGMA, Q, RHO % geo::S)]), "v"(acc));
    return d;
}

// the rows RHO0 + I of one half (first index bits: RHO0 = 0, last: RHO0 = S)
template <int NB, int P, int G, int SIGMA, int Q, int RHO0, int NA, int... I>
__device__ __forceinline__ uint32_t half_word(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA], const Masks &mk,
                                              std::integer_sequence<int, I...>)
{
    uint32_t acc = 0;
    ((acc = row_bits<NB, P, G, SIGMA, Q, RHO0 + I>(Aw, Bw, mk, acc)), ...);
    return acc;
}

// index bytes of carriers 3c + G, c = column(SIGMA, Q, b), in byte b
template <int NB, int P, int G, int SIGMA, int Q, int NA>
__device__ __forceinline__ uint32_t group_word(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA], const Masks &mk)
{
    constexpr int S = Geo<NB, P>::S;
    const uint32_t hi = half_word<NB, P, G, SIGMA, Q, 0>(Aw, Bw, mk, std::make_integer_sequence<int, S>{});
    const uint32_t lo = half_word<NB, P, G, SIGMA, Q, S>(Aw, Bw, mk, std::make_integer_sequence<int, S>{});
    return (hi << S) | lo;
}

// where the index byte of data carrier m lies: word 4g + 2 sigma + q of the twelve, byte b
constexpr int word_of(int m) { return 4 * (m % 3) + 2 * ((m / 3) & 1) + (m / 3) / 8; }
constexpr int byte_of(int m) { return 3 - ((m / 3) % 8) / 2; }

// The twelve words of a symbol: the index byte of carrier m in byte byte_of(m) of cw[word_of(m)].
// Clears the dropped bits of Aw / Bw on the way.
template <int NB, int P, int NA>
__device__ __forceinline__ void carrier_words(uint32_t (&Aw)[NA], uint32_t (&Bw)[NA], uint32_t (&cw)[12])
{
    using geo = Geo<NB, P>;
    static_assert(NA == geo::NA, "stream words");
    if constexpr (P != 0) {
        // (the empty asm keeps the compiler from folding the clearing into every field extract,
        // which would turn each one-instruction bit-field extract into a shift and an and)
#pragma unroll
        for (int i = 0; i < NA; i++) {
            Aw[i] &= geo::keep_mask(0, i);
            Bw[i] &= geo::keep_mask(1, i);
            asm volatile("" : "+v"(Aw[i]), "+v"(Bw[i]));
        }
    }
    Masks mk = {{geo::mask(0, 0, 0), geo::mask(0, 0, geo::S > 1 ? 1 : 0), geo::mask(0, 0, geo::S > 2 ? 2 : 0)}};
    asm volatile("" : "+s"(mk.m[0]), "+s"(mk.m[1]), "+s"(mk.m[2]));
#define SPREAD_GROUP(G_)                                                                                  \
    cw[4 * G_ + 0] = group_word<NB, P, G_, 0, 0>(Aw, Bw, mk); __builtin_amdgcn_sched_barrier(0);           \
    cw[4 * G_ + 1] = group_word<NB, P, G_, 0, 1>(Aw, Bw, mk); __builtin_amdgcn_sched_barrier(0);           \
    cw[4 * G_ + 2] = group_word<NB, P, G_, 1, 0>(Aw, Bw, mk); __builtin_amdgcn_sched_barrier(0);           \
    cw[4 * G_ + 3] = group_word<NB, P, G_, 1, 1>(Aw, Bw, mk); __builtin_amdgcn_sched_barrier(0);
    SPREAD_GROUP(0) SPREAD_GROUP(1) SPREAD_GROUP(2)
#undef SPREAD_GROUP
}

// The symbol's interleaved bits as MSB-first words: carrier m's index bits are message bits
// N_bpsc m .. (ftw/ftw_ofdm_mapper.cc:262-275 reads them MSB first), i.e. column c of the block
// contributes 3 N_bpsc bits -- carriers 3c, 3c + 1, 3c + 2 -- at bit 3 N_bpsc c.
template <int NB, int C, int NW>
__device__ __forceinline__ void put_column(const uint32_t (&cw)[12], uint32_t (&out)[NW])
{
    constexpr int w = word_of(3 * C) & 3, sh = 8 * byte_of(3 * C);      // the same place in each group's four words
    constexpr uint32_t m = (1u << NB) - 1u;
    const uint32_t v = (((cw[w] >> sh) & m) << (2 * NB)) | (((cw[4 + w] >> sh) & m) << NB) | ((cw[8 + w] >> sh) & m);
    constexpr int bits = 3 * NB, pos = bits * C, k = pos >> 5, o = pos & 31;   // `bits` bits from bit o of word k on
    if constexpr (o + bits <= 32) {
        out[k] |= v << (32 - bits - o);
    } else {
        out[k] |= v >> (o + bits - 32);
        out[k + 1] |= v << (64 - bits - o);
    }
}
template <int NB, int NW, int... C>
__device__ __forceinline__ void message_words(const uint32_t (&cw)[12], uint32_t (&out)[NW], std::integer_sequence<int, C...>)
{
#pragma unroll
    for (int i = 0; i < NW; i++) out[i] = 0;
    (put_column<NB, C>(cw, out), ...);
}

// regimes this form covers
constexpr bool covers(int nbpsc, int punct) { return (nbpsc == 2 || nbpsc == 4 || nbpsc == 6) && (punct == 0 || punct == 1); }
}   // namespace spread


// DATA symbols of one packet, one lane per symbol.  OUTW: MSB-first words, word 2 holds the
// first 32 interleaved DATA bits (words 0,1 carry the SIGNAL symbol).
//
// The encoder outputs of a symbol are produced at full rate as two word streams
//   A[n] = u[n]^u[n-2]^u[n-3]^u[n-5]^u[n-6]   (133 octal, pu:287-307)
//   B[n] = u[n]^u[n-1]^u[n-2]^u[n-3]^u[n-6]   (171 octal)
// with funnel shifts and 3-input XORs; puncturing (pu:309-341) and both interleaver
// permutations (pu:231-277) then collapse into ONE compile-time table "output bit j comes from
// bit n of A or of B", i.e. one bit-field extract and one shift-or per output bit.
template <int NBPSC, int PUNCT>
__device__ __forceinline__ void encode_data_symbols(const uint32_t *U, uint32_t *OUTW, uint32_t n_sym,
                                           uint8_t *cod_out, int lane)
{
    constexpr int N = 48 * NBPSC;                                  // coded bits per symbol
    constexpr int NDBPS = PUNCT == 0 ? N / 2 : (PUNCT == 1 ? N * 3 / 4 : N * 2 / 3);
    constexpr int NW = (N + 31) / 32;
    constexpr int NA = (NDBPS + 31) / 32;                          // words of the A / B streams
    constexpr int NX = NA + 1;                                     // scrambled bits incl. 6 bits of history

    for (uint32_t s = lane; s < n_sym; s += 64) {
        // scrambled bits [24 + s*NDBPS - 6, ...), left-aligned in X
        const uint32_t b0 = 18u + s * NDBPS;
        const uint32_t w0 = b0 >> 5, o = b0 & 31u;
        uint32_t X[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) X[i] = __funnelshift_l(U[w0 + i + 1], U[w0 + i], o);

        // A/B bit t (t = 0 .. NDBPS-1) needs X bits t+6 (u[n]) down to t (u[n-6])
        uint32_t Aw[NA], Bw[NA];
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const uint32_t y0 = X[i];
            const uint32_t y1 = __funnelshift_l(X[i + 1], X[i], 1);
            const uint32_t y3 = __funnelshift_l(X[i + 1], X[i], 3);
            const uint32_t y4 = __funnelshift_l(X[i + 1], X[i], 4);
            const uint32_t y5 = __funnelshift_l(X[i + 1], X[i], 5);
            const uint32_t y6 = __funnelshift_l(X[i + 1], X[i], 6);
            const uint32_t common = y6 ^ y3 ^ y0;                  // u[n] ^ u[n-3] ^ u[n-6]
            Aw[i] = common ^ y4 ^ y1;                              // ^ u[n-2] ^ u[n-5]
            Bw[i] = common ^ y5 ^ y4;                              // ^ u[n-1] ^ u[n-2]
        }

        if (cod_out) {                                             // tap: punctured stream in order
            uint32_t Pw[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) Pw[i] = 0;
            punctured_bits<PUNCT>(Aw, Bw, Pw, std::make_integer_sequence<int, N>{});
#pragma unroll
            for (int t = 0; t < N / 8; t++)
                cod_out[6u + s * (N / 8) + t] = (uint8_t)(Pw[t >> 2] >> (24 - 8 * (t & 3)));
        }

        uint32_t out[NW];
        if constexpr (spread::covers(NBPSC, PUNCT)) {
            uint32_t cw[12];
            spread::carrier_words<NBPSC, PUNCT>(Aw, Bw, cw);
            spread::message_words<NBPSC>(cw, out, std::make_integer_sequence<int, 16>{});
        } else {
            interleave_bits<NBPSC, PUNCT>(Aw, Bw, out, std::make_integer_sequence<int, NW>{});
        }

        if (N % 32 == 0) {
#pragma unroll
            for (int i = 0; i < NW; i++) OUTW[2u + s * (N / 32) + i] = out[i];
        } else {
            // N = 48: three halfwords; halfword h (stream order) lives at u16 index h^1
            uint16_t *H = reinterpret_cast<uint16_t *>(OUTW + 2);
            const uint32_t h0 = 3u * s;
            H[(h0 + 0) ^ 1u] = (uint16_t)(out[0] >> 16);
            H[(h0 + 1) ^ 1u] = (uint16_t)out[0];
            H[(h0 + 2) ^ 1u] = (uint16_t)(out[1] >> 16);
        }
    }
}

// Steps 1-3 for one packet, executed by one wavefront: the scrambled PLCP stream as MSB-first
// words in U[0 .. n_words].  A: >= n_words + 3 words of LDS (natural-order MPDU, dead afterwards);
// U: >= n_words + 2 words.  plcp_out / scr_out are optional taps (global memory).  K: rounds of
// 64 payload words loaded before the first is used.
// One round of a packet's payload words, requested ahead of their use: the aligned dwords
// q[lane + 64 kk], kk < K, of the payload that starts at pp (0 beyond it).
// A packet's bytes as a buffer resource: aligned dwords from the one that holds the first byte on,
// as many as hold bytes of the packet -- a load beyond them returns 0 without a compare or a select
// in the wave's instruction stream, and the address is one 32-bit offset per lane.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t payload_window(const uint8_t *pp, uint32_t pay_len)
{
    const uint64_t a = reinterpret_cast<uint64_t>(pp);                   // the same in every lane of the wave
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    const uint32_t al = lo & 3u;
    const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)((al + pay_len + 3u) & ~3u));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>((((uint64_t)hi << 32) | lo) - al), 0, (int)bytes, 0x00020000);
}

template <int K>
struct PayloadAhead {
    uint32_t w[K];
    __device__ __forceinline__ void fetch(const uint8_t *pp, uint32_t pay_len, int lane)
    {
        const __amdgpu_buffer_rsrc_t win = payload_window(pp, pay_len);
#pragma unroll
        // (the row's offset goes into the per-lane offset, which the range check covers; the scalar
        //  offset operand is outside it)
        for (int kk = 0; kk < K; kk++) w[kk] = __builtin_amdgcn_raw_buffer_load_b32(win, 4 * lane + 256 * kk, 0, 0);
    }
};

// AHEAD: `ahead` holds the first round of this packet's payload words on entry; once they are in
// LDS the first round of the packet whose payload starts at next_pp (nullptr: none) is requested.
// RAW (plans with OFDMTX_F_RAW_PSDU): the packet's bytes are the whole PSDU -- MPDU and FCS as they go
// on air, in natural byte order -- so no MAC header is built and no CRC computed; LENGTH is the byte
// count itself.  Everything behind step 2 is the same code.
// FAST: payloads that start on a dword boundary (a branch the whole wave takes together) are staged
// row by row -- 64 dwords per load, fetched through a buffer resource that answers 0 beyond the
// packet, stored with immediate offsets -- instead of word by word with compares and selects.  Used
// where the instruction count sets the time (k_encode_bits: 3 % there); the fused kernel, bound by its
// stores, keeps the form its register budget was tuned with.
template <int K, bool AHEAD = false, bool RAW = false, bool FAST = false>
__device__ __forceinline__ void scramble_packet(const Pkt &k, const RegimeInfo &ri, const DevTables &T,
                                                const uint8_t *__restrict__ payloads, uint32_t *A, uint32_t *U,
                                                const uint32_t *s_crc, const uint32_t *s_scr, int lane_in,
                                                uint8_t *plcp_out, uint8_t *scr_out, PayloadAhead<K> *ahead = nullptr,
                                                const uint8_t *next_pp = nullptr, uint32_t next_len = 0)
{
    // Opaque copy of the lane number: in a uniform plan every per-lane address below is invariant
    // over the caller's packet loop; hoisted out of it, they would take registers that the
    // encoder needs and come back as scratch reloads, which wait for every store in flight.
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const uint32_t mpdu_len = RAW ? 0u : 24u + k.pay_len;
    const uint32_t length = RAW ? k.pay_len : mpdu_len + 4u;                 // pu:193
    const uint32_t n_pad = RAW ? k.n_sym * ri.n_dbps - (22u + 8u * length)
                               : k.n_sym * ri.n_dbps - (54u + 8u * mpdu_len);   // pu:97
    const uint32_t plcp_len = 5u + length + (n_pad + 13u) / 8u;              // pu:160-171
    const uint32_t n_words = (plcp_len + 3u) / 4u;                           // stream words
    const uint32_t n_dw = RAW ? (length + 3u) / 4u : (mpdu_len + 3u) / 4u;   // MPDU (RAW: PSDU) dwords
    constexpr uint32_t HDR = RAW ? 2u : 8u;                                  // words in front of the caller's bytes

    SEC("e1_stage");
    // ---- 1. MPDU in natural byte order (pu:28-36,100), Dw[-2], Dw[-1] and the tail zero
    uint32_t *Dw = A + 2;
    {
        // A[0] byte 3 and A[1] = the SIGNAL field and the SERVICE bits (0, pu:183), every byte with its
        // bits reversed like the MPDU's, so that step 3 treats the whole PLCP stream alike;
        // A[2..7] = MAC header; A[8 + j] = payload dword j; zeros up to A[n_words + 2]
        if (lane < (int)HDR) {
            const uint32_t rev = __brev(length) >> 20;                           // pu:147-150
            const uint32_t par = (__popc(rev & 0xFFFu) + __popc((uint32_t)ri.rate_bits)) & 1u;   // pu:153-157
            const uint32_t field = ((uint32_t)ri.rate_bits << 20) | ((rev & 0xFFFu) << 7) | (par << 6);   // pu:174-180
            const uint32_t sig = __brev(field) >> 8;                             // stream word 0 = __brev(sig)
            uint32_t v = lane == 0 ? sig << 24 : sig >> 8;
            if constexpr (!RAW) {
                v = lane == 2 ? (0x00000008u | ((k.txtime >> 8) & 0xFFu) << 16 | (k.txtime & 0xFFu) << 24) : v;   // pu:91
                v = lane == 3 ? 0xcd086000u : v;
                v = lane == 4 ? 0x2000a637u : v;
                v = lane == 5 ? 0xf13c01d6u : v;
                v = lane == 6 ? 0xad086000u : v;
                v = lane == 7 ? 0x0000af3bu : v;
            }
            A[lane] = v;
        }
        // The payload may start at any byte: aligned dwords q[j], q[j + 1] are funnelled together.
        // All loads of a round are issued before the first is used (a loop of load -> LDS write
        // would wait for each load in turn: one HBM round trip per 64 words).
        const uint8_t *pp = payloads + k.pay_off;
        const uint32_t al = (uint32_t)(reinterpret_cast<uintptr_t>(pp) & 3u);
        const uint32_t *q = reinterpret_cast<const uint32_t *>(pp - al);
        const uint32_t n_pw = n_dw - (HDR - 2u);                             // payload dwords, the last may be partial
        const uint32_t total = n_words + 3u - HDR;                           // words behind the header to write
        bool staged = false;
        if constexpr (FAST) {
            if (__builtin_amdgcn_readfirstlane((int)al) == 0) {
                staged = true;
                const __amdgpu_buffer_rsrc_t win = payload_window(pp, k.pay_len);
                const uint32_t rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)((total + 63u) / 64u));
                uint32_t *dst = A + HDR + lane;
                for (uint32_t r0 = 0; r0 < rows; r0 += K) {
                    uint32_t w[K];
#pragma unroll
                    for (int kk = 0; kk < K; kk++) {
                        if (AHEAD && r0 == 0) w[kk] = ahead->w[kk];
                        else w[kk] = __builtin_amdgcn_raw_buffer_load_b32(win, 4 * lane + 256 * (int)(r0 + kk), 0, 0);
                    }
#pragma unroll
                    for (int kk = 0; kk < K; kk++) {
                        if (r0 + kk + 1u < rows) dst[64u * (r0 + kk)] = w[kk];                     // a whole row
                        else if (r0 + kk < rows && (uint32_t)lane + 64u * (r0 + kk) < total) dst[64u * (r0 + kk)] = w[kk];
                    }
                }
                // the last payload word may hold bytes of whatever follows the packet: cleared by the lane that wrote it
                const uint32_t part = k.pay_len & 3u;
                if (part && (uint32_t)lane == ((n_pw - 1u) & 63u)) A[HDR + n_pw - 1u] &= (1u << (8u * part)) - 1u;
            }
        }
        if (!staged)
        for (uint32_t j0 = (uint32_t)lane; j0 < total; j0 += 64u * K) {
            uint32_t lo[K], hi[K];
#pragma unroll
            for (int kk = 0; kk < K; kk++) {
                const uint32_t j = j0 + 64u * kk;
                if (AHEAD && j0 < 64u) lo[kk] = ahead->w[kk];
                else lo[kk] = j < n_pw ? q[j] : 0u;
            }
            if (al) {
#pragma unroll
                for (int kk = 0; kk < K; kk++) {
                    const uint32_t j = j0 + 64u * kk;
                    const uint32_t nb = k.pay_len - 4u * j < 4u ? k.pay_len - 4u * j : 4u;
                    hi[kk] = (j < n_pw && al + nb > 4u) ? q[j + 1] : 0u;
                }
            }
#pragma unroll
            for (int kk = 0; kk < K; kk++) {
                const uint32_t j = j0 + 64u * kk;
                if (j < total) {
                    uint32_t v = al ? (uint32_t)((((uint64_t)hi[kk] << 32) | lo[kk]) >> (8u * al)) : lo[kk];
                    const uint32_t nb = k.pay_len - 4u * j;                  // bytes left from this word on
                    if (j < n_pw && nb < 4u) v &= (1u << (8u * nb)) - 1u;
                    A[HDR + j] = j < n_pw ? v : 0u;
                }
            }
        }
        if constexpr (AHEAD)
            if (next_pp) ahead->fetch(next_pp, next_len, lane);
    }
    wave_sync();

    SEC("e2_crc");
    // ---- 2. CRC-32 of the MPDU (ftw/ftw_crc32.cc:52-57); a raw PSDU brings its FCS along
    if constexpr (!RAW) {
        const uint32_t c4 = (n_dw + 63u) / 64u;
        const uint32_t j0 = lane * c4;
        uint32_t st = lane == 0 ? 0xFFFFFFFFu : 0u;
        for (uint32_t j = j0; j < j0 + c4; j++) {
            if (4u * j + 4u <= mpdu_len) {
                st ^= Dw[j];
                st = s_crc[768 + (st & 0xFFu)] ^ s_crc[512 + ((st >> 8) & 0xFFu)] ^
                     s_crc[256 + ((st >> 16) & 0xFFu)] ^ s_crc[st >> 24];
            } else if (4u * j < mpdu_len) {
                const uint32_t d = Dw[j];
                for (uint32_t b = 0; b < mpdu_len - 4u * j; b++)
                    st = (st >> 8) ^ s_crc[(st ^ (d >> (8u * b))) & 0xFFu];
            }
        }
        const uint32_t endb = 4u * (j0 + c4) < mpdu_len ? 4u * (j0 + c4) : mpdu_len;
        uint32_t part = crc_mulmod(st, T.xpow8[mpdu_len - endb], s_crc);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part ^= __shfl_xor(part, off);
        // The FCS is the state bit-reversed, then ^~0 (ftw/ftw_crc32.cc:58-63,75), appended big-endian
        // behind the bit-reversed MPDU (pu:372): in the natural-order buffer that is ~state, low
        // byte first, at byte mpdu_len (the words there are still zero).
        const uint32_t fsh = 8u * (mpdu_len & 3u);
        if (lane == 0) Dw[mpdu_len >> 2] |= ~part << fsh;
        if (lane == 1 && fsh) Dw[(mpdu_len >> 2) + 1u] = ~part >> (32u - fsh);
    }
    wave_sync();

    SEC("e3_words");
    // ---- 3. PLCP stream words (pu:128-205) and scrambler (pu:344-367): stream word w = buffer bytes
    // 4w-5 .. 4w-2 with the bit order reversed inside every byte (pu:208-218), XOR the scrambler
    // window (32w - 24) mod 127, which advances by 16 (mod 127) from one round to the next.
    {
        const uint32_t t0 = ((uint32_t)lane >> 2) + 32u * ((uint32_t)lane & 3u) + 103u;     // 32 lane - 24 (mod 127)
        uint32_t idx = t0 < 127u ? t0 : t0 - 127u;
        uint32_t cur = lane == 0 ? (uint32_t)SCR_WORD0 : idx;
        for (uint32_t w = lane; w <= n_words; w += 64) {
            uint32_t v = __brev(__builtin_amdgcn_alignbyte(A[w + 1u], A[w], 3));
            if (plcp_out && 4u * w < plcp_len)
                *reinterpret_cast<uint32_t *>(plcp_out + k.plcp_off + 4u * w) = __builtin_bswap32(v);
            v ^= s_scr[cur];
            U[w] = v;
            if (scr_out && 4u * w < plcp_len)
                *reinterpret_cast<uint32_t *>(scr_out + k.plcp_off + 4u * w) = __builtin_bswap32(v);
            idx += 16u;
            idx = idx < 127u ? idx : idx - 127u;
            cur = idx;
        }
        // six tail bits to 0 (pu:361-362), by the lane that wrote their word
        const uint32_t tz = 40u + 8u * length;                               // first tail bit
        if ((uint32_t)lane == ((tz >> 5) & 63u)) {
            const uint32_t v = U[tz >> 5] & ~(0x3Fu << (26u - (tz & 31u)));
            U[tz >> 5] = v;
            if (scr_out) *reinterpret_cast<uint32_t *>(scr_out + k.plcp_off + 4u * (tz >> 5)) = __builtin_bswap32(v);
        }
    }
    wave_sync();
}

// Steps 1-5 for one packet, executed by one wavefront.  A: >= max(n_words + 3, message words + 4)
// words of LDS (first the natural-order MPDU, finally the message words OUTW: OUTW[0] low half
// and OUTW[1] = the 48 SIGNAL bits, OUTW[2..] = interleaved DATA bits, all MSB first);
// U: >= n_words + 2 words.  plcp_out / scr_out / cod_out are optional taps (global memory).
template <int REGIME, bool RAW = false>
__device__ __forceinline__ void encode_packet(const Pkt &k, const RegimeInfo &ri, const DevTables &T,
                                     const uint8_t *__restrict__ payloads, uint32_t *A, uint32_t *U,
                                     const uint32_t *s_crc, const uint32_t *s_scr, int lane,
                                     uint8_t *plcp_out, uint8_t *scr_out, uint8_t *cod_out,
                                     PayloadAhead<8> *ahead = nullptr, const uint8_t *next_pp = nullptr, uint32_t next_len = 0)
{
    // (row-wise staging everywhere but in the 16-QAM r1/2 instantiation, where its few extra registers
    //  -- 126 -> 129 -- would cost a wave per SIMD: 2.37 -> 2.61 ms per 10^6 packets; with it the headline
    //  regime runs 1.98 -> 1.92 ms, the others within 1 %)
    constexpr bool FAST = REGIME != 5;
    if constexpr (RAW) scramble_packet<8, false, true, FAST>(k, ri, T, payloads, A, U, s_crc, s_scr, lane, plcp_out, scr_out);
    else if (ahead) scramble_packet<8, true, false, FAST>(k, ri, T, payloads, A, U, s_crc, s_scr, lane, plcp_out, scr_out, ahead, next_pp, next_len);
    else scramble_packet<8, false, false, FAST>(k, ri, T, payloads, A, U, s_crc, s_scr, lane, plcp_out, scr_out);

    SEC("e4_signal");
    // ---- 4. SIGNAL symbol: 48 coded bits, N = 48 first permutation only (pu:245-249)
    uint32_t *OUTW = A;
    uint8_t *co = cod_out ? cod_out + k.msg_off : nullptr;
    {
        const uint32_t t = 63u - lane;                // ballot bit `lane` = stream bit 63 - lane
        uint32_t bi = 0, bc = 0;
        if (t < 48u) {
            bi = coded_bit(U, 16u * (t % 3u) + t / 3u);
            bc = coded_bit(U, t);
        }
        const unsigned long long mi = __ballot(bi);
        const unsigned long long mc = __ballot(bc);
        const uint32_t ndw = (k.n_sym * ri.n_cbps) >> 5;      // whole DATA words
        if (lane == 0) {
            OUTW[0] = (uint32_t)(mi >> 48);
            OUTW[1] = (uint32_t)(mi >> 16);
            OUTW[2u + ndw] = 0;                       // partly filled / trailing words
            OUTW[3u + ndw] = 0;
            if (co) {
#pragma unroll
                for (int b = 0; b < 6; b++) co[b] = (uint8_t)(mc >> (56 - 8 * b));
            }
        }
    }

    SEC("e5_data");
    // ---- 5. DATA symbols
    // REGIME != 0: the batch has one regime, known when the kernel was chosen, and only that
    // instantiation is compiled in; REGIME == 0 (mixed-rate batches) dispatches per packet.
    switch (REGIME ? (uint32_t)REGIME : k.regime) {
    case 1: encode_data_symbols<1, 0>(U, OUTW, k.n_sym, co, lane); break;
    case 2: encode_data_symbols<1, 1>(U, OUTW, k.n_sym, co, lane); break;
    case 3: encode_data_symbols<2, 0>(U, OUTW, k.n_sym, co, lane); break;
    case 4: encode_data_symbols<2, 1>(U, OUTW, k.n_sym, co, lane); break;
    case 5: encode_data_symbols<4, 0>(U, OUTW, k.n_sym, co, lane); break;
    case 6: encode_data_symbols<4, 1>(U, OUTW, k.n_sym, co, lane); break;
    case 7: encode_data_symbols<6, 2>(U, OUTW, k.n_sym, co, lane); break;
    default: encode_data_symbols<6, 1>(U, OUTW, k.n_sym, co, lane); break;
    }
    wave_sync();

}

// --------------------------------------------------------------------------------------
// Slot rows (fused kernel): instead of message bits, the encoder can emit for every OFDM symbol
// the constellation-LUT slot of each of its 52 occupied FFT bins, one byte per bin.  The bytes
// are grouped by l = bin % 8 -- the lane that feeds the bin to the first radix-8 stage -- and
// ascend by bin inside a group, so lane l finds its bins l, l+8, .. l+56 (nulls left out) in
// one short byte window.  A row is 52 bytes = 13 words; with that odd word stride the
// lane-per-symbol row writes are free of LDS bank conflicts.
constexpr int SLOT_ROW_WORDS = 13;

constexpr int slot_row_bin(int rb)            // FFT bin of row byte rb
{
    int n = 0;
    for (int l = 0; l < 8; l++)
        for (int m = 0; m < 8; m++)
            if (bin_code(l + 8 * m) != -1) {
                if (n == rb) return l + 8 * m;
                n++;
            }
    return 0;                                 // not reached for rb < 52; bin 0 is a null
}

constexpr int slot_seg_off(int l)             // first row byte of lane l's group
{
    int n = 0;
    for (int ll = 0; ll < l; ll++)
        for (int m = 0; m < 8; m++)
            if (bin_code(ll + 8 * m) != -1) n++;
    return n;
}

template <int RB>
struct SlotCode { static constexpr int value = bin_code(slot_row_bin(RB)); };

// bin_code of the bin behind row byte `rb` (rb >= 52: -1), by compile-time selects
template <int... RB>
__device__ __forceinline__ int slot_code_of(int rb, std::integer_sequence<int, RB...>)
{
    int c = -1;
    ((c = rb == RB ? SlotCode<RB>::value : c), ...);
    return c;
}

constexpr int lut_off_for(int nbpsc) { return nbpsc == 1 ? 4 : nbpsc == 2 ? 8 : nbpsc == 4 ? 16 : 64; }

// ftw/ftw_pnc_ofdm_pilot_cc.cc:125-131,146-178: the PNC pilot block writes constant pilots, no
// polarity sequence.  As a "pilot pack": one slot byte (0 = zero, LUT_PM1 + 1 = +1) per pilot
// in the order +7, +21, -21, -7 (FFT bins 7, 21, 43, 57).
constexpr int pilot_bin(int i) { return i == 0 ? 7 : i == 1 ? 21 : i == 2 ? 43 : 57; }
constexpr uint32_t PILOTS_PNC_A = ((LUT_PM1 + 1u) << 0) | ((LUT_PM1 + 1u) << 16);      // user 1: -21 and +7 carry +1
constexpr uint32_t PILOTS_PNC_B = ((LUT_PM1 + 1u) << 8) | ((LUT_PM1 + 1u) << 24);      // user 2: -7 and +21
constexpr uint32_t PILOTS_PNC_ALL = PILOTS_PNC_A | PILOTS_PNC_B;                       // user 0

constexpr int slot_row_byte_of(int bin)       // inverse of slot_row_bin
{
    for (int rb = 0; rb < 4 * SLOT_ROW_WORDS; rb++)
        if (slot_row_bin(rb) == bin) return rb;
    return 0;
}

// Overwrites the four pilot bytes of one slot row with a pilot pack.
template <int... I>
__device__ __forceinline__ void set_row_pilots(uint32_t *row, uint32_t pack, std::integer_sequence<int, I...>)
{
    uint8_t *b = reinterpret_cast<uint8_t *>(row);
    ((b[std::integral_constant<int, slot_row_byte_of(pilot_bin(I))>::value] = (uint8_t)(pack >> (8 * I))), ...);
}
__device__ __forceinline__ void set_row_pilots(uint32_t *row, uint32_t pack)
{
    set_row_pilots(row, pack, std::make_integer_sequence<int, 4>{});
}

// ftw/ftw_ofdm_pilot_cc.cc:116,159: polarity of the pilots of the symbol with running index pidx
__device__ __forceinline__ uint32_t pilot_negative(uint32_t pidx, uint64_t pneg_lo, uint64_t pneg_hi)
{
    return (uint32_t)((pidx < 64u ? pneg_lo : pneg_hi) >> (pidx & 63u)) & 1u;
}

// PUNCT value that makes the slot tables read an already interleaved mapper message (stream A,
// bit j of the symbol) instead of the encoder's output streams
constexpr int SRC_MESSAGE = 3;

// Bit I of slot word WORD: byte I / NBPSC, constellation index bit t = I % NBPSC (t = 0 is the
// index MSB: ftw/ftw_ofdm_mapper.cc:262-275 consumes the message MSB first).
template <int NBPSC, int PUNCT, int WORD, int I, int NA>
__device__ __forceinline__ uint32_t slot_bit(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA])
{
    constexpr int b = I / NBPSC, t = I % NBPSC;
    constexpr int c = SlotCode<4 * WORD + b>::value;
    if constexpr (c < 0) {
        return 0u;
    } else if constexpr (PUNCT == SRC_MESSAGE) {
        constexpr int n = NBPSC * c + t;
        return ((Aw[n >> 5] >> (31 - (n & 31))) & 1u) << (8 * b + NBPSC - 1 - t);
    } else {
        constexpr int p = unpuncture<PUNCT>(deinterleave<NBPSC>(NBPSC * c + t)), n = p >> 1;
        const uint32_t w = (p & 1) ? Bw[n >> 5] : Aw[n >> 5];
        return ((w >> (31 - (n & 31))) & 1u) << (8 * b + NBPSC - 1 - t);
    }
}

// constant part of slot word WORD (LUT offsets of data bytes, pilots for polarity +1) and the
// mask of the bits that flip when the polarity is -1
template <int NBPSC, int WORD>
struct SlotConst {
    static constexpr uint32_t byte_const(int b)
    {
        const int c = bin_code(slot_row_bin(4 * WORD + b));
        return (uint32_t)(c >= 0 ? lut_off_for(NBPSC) : c == -2 ? LUT_PM1 + 1 : c == -3 ? LUT_PM1 : 0) << (8 * b);
    }
    static constexpr uint32_t byte_flip(int b)
    {
        const int c = bin_code(slot_row_bin(4 * WORD + b));
        return (uint32_t)(c == -2 || c == -3 ? 1 : 0) << (8 * b);
    }
    static constexpr uint32_t value = byte_const(0) | byte_const(1) | byte_const(2) | byte_const(3);
    static constexpr uint32_t flip = byte_flip(0) | byte_flip(1) | byte_flip(2) | byte_flip(3);
};

template <int NBPSC, int PUNCT, int WORD, int NA, int... I>
__device__ __forceinline__ uint32_t slot_word(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA], uint32_t neg,
                                              std::integer_sequence<int, I...>)
{
    uint32_t v = (slot_bit<NBPSC, PUNCT, WORD, I>(Aw, Bw) | ...) | SlotConst<NBPSC, WORD>::value;
    if constexpr (SlotConst<NBPSC, WORD>::flip != 0u) v ^= neg ? SlotConst<NBPSC, WORD>::flip : 0u;
    return v;
}

// Every word goes to LDS as soon as it is complete (dst[WORD]): holding the 13 words back
// for one burst of stores costs a dozen registers the encoder does not have.
template <int NBPSC, int PUNCT, int NA, int... WORD>
__device__ __forceinline__ void slot_words(const uint32_t (&Aw)[NA], const uint32_t (&Bw)[NA], uint32_t neg,
                                           uint32_t *dst, std::integer_sequence<int, WORD...>)
{
    ((dst[WORD] = slot_word<NBPSC, PUNCT, WORD>(Aw, Bw, neg, std::make_integer_sequence<int, 4 * NBPSC>{}),
      __builtin_amdgcn_sched_barrier(0)), ...);
}


// The fused kernel's slot rows of a symbol from its twelve carrier words.
namespace spread {
// v_perm_b32 selector that moves into slot word WORD the bytes it takes from carrier word SRC and
// keeps the others (selector bytes 0-3: the word so far, 4-7: the carrier word); 0 if it takes none
template <int WORD, int SRC>
constexpr uint32_t gather_sel()
{
    uint32_t sel = 0;
    bool any = false;
    for (int b = 0; b < 4; b++) {
        const int c = bin_code(slot_row_bin(4 * WORD + b));
        if (c >= 0 && word_of(c) == SRC) {
            sel |= (uint32_t)(4 + byte_of(c)) << (8 * b);
            any = true;
        } else {
            sel |= (uint32_t)b << (8 * b);
        }
    }
    return any ? sel : 0u;
}

template <int WORD, int SRC>
__device__ __forceinline__ void gather_from(uint32_t &v, const uint32_t (&cw)[12])
{
    constexpr uint32_t sel = gather_sel<WORD, SRC>();
    if constexpr (sel != 0u) v = __builtin_amdgcn_perm(cw[SRC], v, sel);
}

template <int NB, int WORD, int... SRC>
__device__ __forceinline__ uint32_t slot_word(const uint32_t (&cw)[12], uint32_t neg, std::integer_sequence<int, SRC...>)
{
    uint32_t v = 0;
    (gather_from<WORD, SRC>(v, cw), ...);
    v |= SlotConst<NB, WORD>::value;
    if constexpr (SlotConst<NB, WORD>::flip != 0u) v ^= neg ? SlotConst<NB, WORD>::flip : 0u;
    return v;
}

template <int NB, int P, int NA, int... WORD>
__device__ __forceinline__ void slot_words(uint32_t (&Aw)[NA], uint32_t (&Bw)[NA], uint32_t neg, uint32_t *dst,
                                           std::integer_sequence<int, WORD...>)
{
    uint32_t cw[12];
    carrier_words<NB, P>(Aw, Bw, cw);
    ((dst[WORD] = slot_word<NB, WORD>(cw, neg, std::make_integer_sequence<int, 12>{})), ...);
}
}   // namespace spread

// Slot rows of the DATA symbols with 0-based index d0 + lane (the SIGNAL symbol is not handled here), one
// lane per symbol, into the row SI points at + lane.  Encoder, puncturing and interleaver as in
// encode_data_symbols; the pilot polarity of DATA symbol d is entry (pilot_base + 1 + d) % 127.
// MUL: by multiplication where that form exists (spread::covers); the PNC instantiations of the
// kernel, which dispatch every regime per packet and have no register to spare, keep the bit-by-bit form.
template <int NBPSC, int PUNCT, bool MUL = true>
__device__ __forceinline__ void symbol_slots(const uint32_t *U, uint32_t *SI, uint32_t d0, uint32_t n_sym,
                                             uint32_t pilot_base, uint64_t pneg_lo, uint64_t pneg_hi, int lane)
{
    constexpr int N = 48 * NBPSC;
    constexpr int NDBPS = PUNCT == 0 ? N / 2 : (PUNCT == 1 ? N * 3 / 4 : N * 2 / 3);
    constexpr int NA = (NDBPS + 31) / 32;
    constexpr int NX = NA + 1;

    const uint32_t d = d0 + (uint32_t)lane;
    if (d < n_sym) {
        const uint32_t b0 = 18u + d * NDBPS;
        const uint32_t w0 = b0 >> 5, o = b0 & 31u;
        uint32_t X[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) X[i] = __funnelshift_l(U[w0 + i + 1], U[w0 + i], o);
        uint32_t Aw[NA], Bw[NA];
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const uint32_t y0 = X[i];
            const uint32_t y1 = __funnelshift_l(X[i + 1], X[i], 1);
            const uint32_t y3 = __funnelshift_l(X[i + 1], X[i], 3);
            const uint32_t y4 = __funnelshift_l(X[i + 1], X[i], 4);
            const uint32_t y5 = __funnelshift_l(X[i + 1], X[i], 5);
            const uint32_t y6 = __funnelshift_l(X[i + 1], X[i], 6);
            const uint32_t common = y6 ^ y3 ^ y0;
            Aw[i] = common ^ y4 ^ y1;
            Bw[i] = common ^ y5 ^ y4;
        }
        const uint32_t neg = pilot_negative((pilot_base + 1u + d) % 127u, pneg_lo, pneg_hi);
        if constexpr (MUL && spread::covers(NBPSC, PUNCT))
            spread::slot_words<NBPSC, PUNCT>(Aw, Bw, neg, SI + SLOT_ROW_WORDS * lane, std::make_integer_sequence<int, SLOT_ROW_WORDS>{});
        else
            slot_words<NBPSC, PUNCT>(Aw, Bw, neg, SI + SLOT_ROW_WORDS * lane, std::make_integer_sequence<int, SLOT_ROW_WORDS>{});
    }
}

// The same slot rows from a mapper message held as MSB-first words MW (bit 0 of the message =
// bit 31 of MW[0]; the 48 SIGNAL bits lead): DATA symbol d occupies message bits
// 48 + d*N_cbps .. (ftw/ftw_ofdm_mapper.cc:184-260).
template <int NBPSC>
__device__ __forceinline__ void message_slots(const uint32_t *MW, uint32_t *SI, uint32_t d0, uint32_t n_sym,
                                              uint32_t pilot_base, uint64_t pneg_lo, uint64_t pneg_hi, int lane)
{
    constexpr int N = 48 * NBPSC;
    constexpr int NW = (N + 31) / 32;
    const uint32_t d = d0 + (uint32_t)lane;
    if (d < n_sym) {
        const uint32_t b0 = 48u + d * N;
        const uint32_t w0 = b0 >> 5, o = b0 & 31u;
        uint32_t Mw[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) Mw[i] = __funnelshift_l(MW[w0 + i + 1], MW[w0 + i], o);
        const uint32_t neg = pilot_negative((pilot_base + 1u + d) % 127u, pneg_lo, pneg_hi);
        slot_words<NBPSC, SRC_MESSAGE>(Mw, Mw, neg, SI + SLOT_ROW_WORDS * lane,
                                       std::make_integer_sequence<int, SLOT_ROW_WORDS>{});
    }
}

constexpr int ENC_WAVES = 4;

// (The instantiation that dispatches the regime per packet is held to 128 VGPRs = 4 waves per SIMD: the
//  bit table of 64-QAM r2/3 alone would take 200, which every packet of a mixed batch then paid for in
//  occupancy; with the cap that table spills instead -- 1.59 -> 1.29 ms on a mixed-rate batch of 4 10^5
//  packets.  The same cap on the uniform 64-QAM r2/3 instantiation, where every packet runs the table,
//  doubles its time.)
//  RAW: the instantiations for plans with OFDMTX_F_RAW_PSDU (regime dispatched per packet).
template <bool UNIFORM, int REGIME, bool RAW = false>
__global__ void __launch_bounds__(64 * ENC_WAVES) __attribute__((amdgpu_waves_per_eu(REGIME == 0 ? 4 : 1)))
k_encode_bits(PlanParams P, DevTables T, const uint8_t *__restrict__ payloads,
              uint8_t *__restrict__ plcp_out, uint8_t *__restrict__ scr_out,
              uint8_t *__restrict__ cod_out, uint8_t *__restrict__ itl_out, uint32_t a_words, uint32_t u_words)
{
    // per wave: a_words for the natural-order MPDU, later the message words; u_words for the scrambled
    // words -- both sized by the launcher for the plan's longest packet, so that batches of ordinary
    // packets leave room for more workgroups per CU than the longest packet would
    extern __shared__ __align__(16) uint32_t s_enc[];
    __shared__ uint32_t s_crc[CRC_TAB_WORDS];
    __shared__ uint32_t s_scr[SCR_TAB_WORDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill_scramble_tables(T, s_crc, s_scr, tid, 64 * ENC_WAVES);
    __syncthreads();

    uint32_t *A = s_enc + (size_t)wave * (a_words + u_words);
    uint32_t *U = A + a_words;

    // ragged plans: packets are drawn from the plan's second work counter (their cost follows
    // their length); uniform plans use a fixed stride
    uint32_t p_end = 0;
    auto next_packet = [&](uint32_t prev, bool first) -> uint32_t {
        if constexpr (UNIFORM) {
            return first ? blockIdx.x * ENC_WAVES + wave : prev + gridDim.x * ENC_WAVES;
        } else {
            if (!first && prev + 1u < p_end) return prev + 1u;
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(P.next_pkt + 2, P.grab_enc);
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            p_end = v + P.grab_enc;
            return v;
        }
    };
    // uniform plans: the next packet's payload is requested as soon as this one's is in LDS
    [[maybe_unused]] PayloadAhead<8> ahead;
    if constexpr (UNIFORM && !RAW) {
        const uint32_t p0 = next_packet(0, true);
        if (p0 < P.n_pkts) ahead.fetch(payloads + load_pkt<UNIFORM>(P, p0).pay_off, P.pay_len, lane);
    }
    for (uint32_t p = next_packet(0, true); p < P.n_pkts; p = next_packet(p, false)) {
        const Pkt k = load_pkt<UNIFORM>(P, p);
        const RegimeInfo ri = regime_info(REGIME ? REGIME : (int)k.regime);
        const uint32_t msg_len = 6u + k.n_sym * ri.n_cbps / 8u;
        if constexpr (UNIFORM && !RAW) {
            const uint32_t pn = next_packet(p, false);
            encode_packet<REGIME>(k, ri, T, payloads, A, U, s_crc, s_scr, lane, plcp_out, scr_out, cod_out, &ahead,
                                  pn < P.n_pkts ? payloads + load_pkt<UNIFORM>(P, pn).pay_off : nullptr, P.pay_len);
        } else {
            encode_packet<REGIME, RAW>(k, ri, T, payloads, A, U, s_crc, s_scr, lane, plcp_out, scr_out, cod_out);
        }
        uint32_t *OUTW = A;

        // ---- 6. message bytes = OUTW stream shifted by 16 bits (6 SIGNAL bytes lead)
        {
            uint32_t *dst = reinterpret_cast<uint32_t *>(itl_out + k.msg_off);
            const uint32_t n_mw = (msg_len + 3u) / 4u;
            for (uint32_t q = lane; q < n_mw; q += 64)
                dst[q] = __builtin_bswap32((OUTW[q] << 16) | (OUTW[q + 1] >> 16));
        }
        wave_sync();
    }
    if constexpr (!UNIFORM) sign_off(P.next_pkt + 2, gridDim.x * ENC_WAVES, lane);
}

// ======================================================================================
// Symbol-level kernel
// ======================================================================================

constexpr int TG = 72;   // float2 per symbol in the transposition layout (8 rows x 9)
constexpr int RS = 88;   // float2 per row in the staging layout

// A group of up to 8 consecutive OFDM symbol rows by one wavefront, 8 lanes per row
// (lane = 8*g + l): 64-point inverse DFT -> cyclic prefix -> coalesced stores.
//   active  the lane's row exists
//   x       (active) the lane's decimated carriers X[l + 8m], m = 0..7
//   buf     the wave's LDS area of 8*RS float2; dst = float4 number, in the packet's window `win`,
//           of the first of the group's `rows_here` rows
__device__ __forceinline__ void ifft64_emit(bool active, v2f (&x)[8], int l, int g, int lane, v2f *buf,
                                            const v2f *s_tw, bool swap_iq, __amdgpu_buffer_rsrc_t win,
                                            uint32_t dst, uint32_t rows_here)
{
    SEC("fft1");
    // ---- 64-point inverse DFT as 8 x 8: x[8q+p] = sum_l W8^{lq} W64^{lp} sum_m X[l+8m] W8^{mp}
    // (s_tw holds W64^{lp} times the output scale, so no separate scaling pass is needed)
    if (active) {
        idft8(x);
#pragma unroll
        for (int p = 0; p < 8; p++) x[p] = cmul(x[p], s_tw[8 * p + l]);
#pragma unroll
        for (int p = 0; p < 8; p++) buf[g * TG + 9 * p + l] = x[p];
    }
    wave_sync();
    if (active) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = buf[g * TG + 9 * l + i];   // lane now plays p = l
    }
    wave_sync();

    SEC("fft2");
    // ---- the row's 80 samples into the staging area.  Lane l owns samples l + 8i, i = 0..9:
    // out[o] = x[(o + 48) % 64] (cyclic prefix, examples/ftw_ofdm.py:121-123), so i = 0,1 repeat
    // q = 6,7 and i >= 2 is q = i - 2.  I/Q exchange (examples/ftw_ofdm.py:155-163) sits in a branch
    // the compiler must keep (the empty asm): turned into selects it would cost every launch 16
    // instructions per group.
    if (active) {
        idft8(x);                                                     // x[q] = sample 8q + l
        if (swap_iq) {
            asm volatile("");
#pragma unroll
            for (int q = 0; q < 8; q++) x[q] = (v2f){x[q].y, x[q].x};
        }
#pragma unroll
        for (int i = 0; i < 10; i++) buf[g * RS + l + 8 * i] = x[(i + 6) & 7];
    }
    wave_sync();

    SEC("store");
    // ---- 8 rows = 5120 contiguous bytes, 16 B per lane per store; lanes beyond the last row are
    // masked, which happens in the last, partial group of a chunk only
    if (rows_here == 8u) {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int e = i * 64 + lane;              // float4 index, 40 per row
            iq_store(win, dst + (uint32_t)e, *reinterpret_cast<const float4 *>(buf + (e / 40) * RS + (e % 40) * 2));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int e = i * 64 + lane;
            if ((uint32_t)e < rows_here * 40u)
                iq_store(win, dst + (uint32_t)e, *reinterpret_cast<const float4 *>(buf + (e / 40) * RS + (e % 40) * 2));
        }
    }
}

// ======================================================================================
// Fused kernel: payload -> IQ, one wavefront per packet
// ======================================================================================
//
// Per packet the only HBM traffic is the payload read and the IQ rows written.  The wavefront
//   1. scrambles its packet into LDS (scramble_packet),
//   2. copies the preamble rows and zeroes the gap rows (coalesced stores that drain while the
//      wave goes on computing),
//   3. per chunk of up to 64 DATA symbols: builds the symbols' slot rows, one lane per symbol
//      (SIGNAL: one lane per carrier), then
//   4. emits the chunk's rows 8 at a time, 8 lanes per symbol: the lane fetches its 8 decimated
//      carriers straight from the LUT by slot byte and runs ifft64_emit.
// Dynamic LDS per wave: [0, a_words) natural MPDU, later the slot rows; then 8*RS float2 of FFT
// transposition / row staging; the scrambled words U live in the staging area when the plan's
// packets have at most 64 DATA symbols (one chunk: U is dead before the first group), otherwise
// behind it (u_off).

constexpr int TX_WAVES = 8;          // waves per block when two blocks fit a CU's LDS
constexpr size_t LDS_PER_CU = 160 * 1024;   // gfx950: one workgroup may use all of it
constexpr int TX_MAX_WAVES = 16;     // launch bound: one block of up to 16 waves otherwise
constexpr int TX_B_WORDS = 8 * RS * 2;

// (launch bound of 16 waves = 4 per SIMD: caps the kernel at 128 VGPRs; uncapped, hipcc spends
//  ~250 on the encoder's bit table and halves the occupancy of the whole kernel.  The block size
//  itself is the launcher's choice: it picks the number of waves the per-wave LDS need allows.)
//
// FROM_MSG: the input is not the payloads but the mapper messages k_encode_bits wrote (the
// two-kernel path, with its taps): the message words take the place of the scrambled words and
// the slot rows come from message_slots; `mapped` (optional) receives the mapper block's output.
//
// PNC: the instantiations for plans with a PNC role (six-row preamble patterns, constant pilots).
// They are separate so that the standard ones keep their register allocation; they dispatch
// the regime per packet (REGIME = 0).
//
// RAW: the payload-input instantiations for plans with OFDMTX_F_RAW_PSDU (the packet's bytes are the
// whole PSDU); separate for the same reason, regime dispatched per packet.
template <bool UNIFORM, int REGIME, bool FROM_MSG, bool PNC = false, bool RAW = false>
__global__ void __launch_bounds__(64 * TX_MAX_WAVES)
k_tx(PlanParams P, DevTables T, const uint8_t *__restrict__ in, float2 *__restrict__ mapped,
     float2 *__restrict__ pilot_tap, float2 *__restrict__ cmap_tap, float2 *__restrict__ iq, uint32_t a_words,
     uint32_t u_off, uint32_t wave_words)
{
    extern __shared__ __align__(16) uint32_t s_dyn[];
    __shared__ uint32_t s_crc[FROM_MSG ? 1 : CRC_TAB_WORDS];
    __shared__ uint32_t s_scr[FROM_MSG ? 1 : SCR_TAB_WORDS];
    __shared__ float2 s_lut[LUT_SIZE];
    __shared__ __align__(16) float2 s_pre[(PNC ? PNC_PRE_ROWS : PRE_ROWS) * SYM_LEN];   // preamble rows, amplitude applied
    __shared__ float2 s_tw[64];
    __shared__ int s_sigc[64];            // SIGNAL row: bin_code of the bin behind row byte i (-1: none)
    __shared__ uint32_t s_win[8][4];      // per l = lane % 8: slot-row window (see below)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_thr = blockDim.x, n_wv = n_thr >> 6;
    const float amp = P.amp;
    const bool swap_iq = (P.flags & OFDMTX_F_SWAPIQ) != 0;
    if (tid < 64) s_sigc[tid] = slot_code_of(tid, std::make_integer_sequence<int, 4 * SLOT_ROW_WORDS>{});
    if (tid < 8) {
        // lane l's window into a slot row: three words from word seg_off / 4, shifted right by
        // seg_off % 4 bytes, hold the lane's bins in ascending order; sel_lo / sel_hi (v_perm_b32
        // selectors) spread them over m = 0..3 / 4..7 and put slot 0 (zero) where bin l + 8m is a null
        uint32_t sel_lo = 0, sel_hi = 0, seg_off = 0, cnt = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const bool null_bin = bin_code(tid + 8 * m) == -1;
            const uint32_t sb = null_bin ? 0x0Cu : cnt;
            cnt += null_bin ? 0u : 1u;
            if (m < 4) sel_lo |= sb << (8 * m); else sel_hi |= sb << (8 * (m - 4));
        }
#pragma unroll
        for (int ll = 1; ll < 8; ll++) seg_off = tid == ll ? (uint32_t)slot_seg_off(ll) : seg_off;
        s_win[tid][0] = sel_lo;
        s_win[tid][1] = sel_hi;
        s_win[tid][2] = seg_off >> 2;
        s_win[tid][3] = seg_off & 3u;
    }
    // 1/sqrt(64) (examples/ftw_ofdm.py:126) is a power of two: (x*0.125)*amp == x*(0.125*amp)
    const float scale = 0.125f * amp;
    if constexpr (!FROM_MSG) fill_scramble_tables(T, s_crc, s_scr, tid, n_thr);
    for (int i = tid; i < LUT_SIZE; i += n_thr) s_lut[i] = T.lut[i];
    if constexpr (PNC) {
        // Row r of the packet's preamble = table row (pre_map >> 4r) & 15, or zeros (15).  The PNC
        // roles (ftw/ftw_pnc_ofdm_preamble.cc:80-103) send the two short-training rows, then the
        // two long-training rows followed (A) or preceded (B) by two zero rows.
        const uint32_t pre_map = (P.flags & OFDMTX_F_PNC_B) ? 0x32FF10u : 0xFF3210u;
        for (int i = tid; i < PNC_PRE_ROWS * SYM_LEN; i += n_thr) {
            const uint32_t r = (uint32_t)i / SYM_LEN, src = (pre_map >> (4u * r)) & 15u;
            float2 v = make_float2(0.f, 0.f);
            if (src < (uint32_t)PRE_ROWS) v = T.preamble[src * SYM_LEN + (uint32_t)i - r * SYM_LEN];
            s_pre[i] = swap_iq ? make_float2(v.y * amp, v.x * amp) : make_float2(v.x * amp, v.y * amp);
        }
    } else {
        for (int i = tid; i < PRE_ROWS * SYM_LEN; i += n_thr) {
            const float2 v = T.preamble[i];               // pu:508 literals as float32 times tx_amplitude,
            s_pre[i] = swap_iq ? make_float2(v.y * amp, v.x * amp) : make_float2(v.x * amp, v.y * amp);
        }                                                 // I and Q exchanged if asked (examples/ftw_ofdm.py:155-163)
    }
    for (int i = tid; i < 64; i += n_thr) {
        const float2 w = T.tw64[((i & 7) * (i >> 3)) & 63];
        s_tw[i] = make_float2(w.x * scale, w.y * scale);
    }
    __syncthreads();

    uint32_t *A = s_dyn + (size_t)wave * wave_words;      // natural MPDU, then the slot rows
    uint32_t *SI = A;
    v2f *buf = reinterpret_cast<v2f *>(A + a_words);
    uint32_t *U = A + u_off;

    uint64_t pneg_lo = T.pilot_neg[0] | ((uint64_t)T.pilot_neg[1] << 32);
    uint64_t pneg_hi = T.pilot_neg[2] | ((uint64_t)T.pilot_neg[3] << 32);
    asm volatile("" : "+s"(pneg_lo), "+s"(pneg_hi));

    uint32_t pre_rows = (P.flags & OFDMTX_F_PREAMBLE) ? PRE_ROWS : 0;
    uint32_t pilots_fixed = 0;
    if constexpr (PNC) {
        if (pre_rows && (P.flags & (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B))) pre_rows = PNC_PRE_ROWS;
        pilots_fixed = P.pilot_mode == OFDMTX_PILOT_PNC_A ? PILOTS_PNC_A
                     : P.pilot_mode == OFDMTX_PILOT_PNC_B ? PILOTS_PNC_B
                     : P.pilot_mode == OFDMTX_PILOT_PNC_ALL ? PILOTS_PNC_ALL : 0u;
    }
    const uint32_t gap_rows = (P.flags & OFDMTX_F_ZEROGAP) ? GAP_ROWS : 0;
    const v2f *lut = reinterpret_cast<const v2f *>(s_lut);

    // Uniform plans: packets cost the same, a fixed stride balances the waves.  Ragged plans: a
    // packet's cost follows its symbol count (1 .. 1366 rows), so each wave draws its next packet
    // from a counter instead (lane 0 asks, the wave reads the answer from lane 0).
    // A draw takes P.grab_tx consecutive packets: one address sustains only ~10^8 atomics per
    // second, which a batch of short packets would exceed at one draw per packet.
    // Uniform batches of mapper messages: the first AHEAD_WORDS * 64 words of the NEXT packet's message are
    // requested before this packet's rows are emitted and wait in registers.  Without the preamble
    // and gap rows (the map + IFFT + CP stage on its own) nothing else covers the time between a
    // wave's last row and the arrival of its next message.
    // (Tried for the payloads as well: no gain -- there the preamble and gap rows, written before the
    // encoder starts, already cover that time.  The PNC instantiation has no registers to spare.)
    constexpr bool AHEAD = UNIFORM && FROM_MSG && !PNC;
    constexpr int AHEAD_WORDS = 9;                        // 576 words = 2304 message bytes
    [[maybe_unused]] uint32_t ahead[AHEAD_WORDS];
    [[maybe_unused]] auto fetch_ahead = [&](uint32_t pn) {
        if (pn < P.n_pkts) {
            const Pkt kn = load_pkt<UNIFORM>(P, pn);
            const RegimeInfo rn = regime_info(REGIME ? REGIME : (int)kn.regime);
            const uint32_t n_mw = (6u + kn.n_sym * (uint32_t)rn.n_cbps / 8u + 3u) / 4u;
            const uint32_t *mp = reinterpret_cast<const uint32_t *>(in + kn.msg_off);
#pragma unroll
            for (int kk = 0; kk < AHEAD_WORDS; kk++)
                ahead[kk] = (uint32_t)lane + 64u * kk < n_mw ? mp[(uint32_t)lane + 64u * kk] : 0u;
        }
    };

    uint32_t p_end = 0;
    auto next_packet = [&](uint32_t prev, bool first) -> uint32_t {
        if constexpr (UNIFORM) {
            return first ? blockIdx.x * n_wv + wave : prev + gridDim.x * n_wv;
        } else {
            if (!first && prev + 1u < p_end) return prev + 1u;
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(P.next_pkt, P.grab_tx);
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            p_end = v + P.grab_tx;
            return v;
        }
    };
    if constexpr (AHEAD) fetch_ahead(next_packet(0, true));
    for (uint32_t p = next_packet(0, true); p < P.n_pkts; p = next_packet(p, false)) {
        const Pkt k = load_pkt<UNIFORM>(P, p);
        const RegimeInfo ri = regime_info(REGIME ? REGIME : (int)k.regime);
        const size_t row0 = UNIFORM ? (size_t)p * P.rows_per_pkt : (size_t)P.row_start[p];
        const __amdgpu_buffer_rsrc_t win = iq_window(iq + row0 * SYM_LEN);     // the packet's rows
        const uint32_t sym_rows = k.n_sym + 1u;                    // SIGNAL + DATA symbols

        if constexpr (FROM_MSG) {
            // message bytes -> MSB-first words (the plan keeps messages 16-byte aligned, with slack
            // behind the last one); two zero words behind them for the funnel shifts
            const uint32_t msg_len = 6u + k.n_sym * (uint32_t)ri.n_cbps / 8u;
            const uint32_t n_mw = (msg_len + 3u) / 4u;
            const uint32_t *mp = reinterpret_cast<const uint32_t *>(in + k.msg_off);
            // (all loads of a round first: a loop of load -> LDS write waits for every load in turn)
            constexpr int K = 8;
            if constexpr (AHEAD) {
#pragma unroll
                for (int kk = 0; kk < AHEAD_WORDS; kk++)
                    if ((uint32_t)lane + 64u * kk < n_mw + 2u) U[(uint32_t)lane + 64u * kk] = __builtin_bswap32(ahead[kk]);
            }
            for (uint32_t w0 = (uint32_t)lane + (AHEAD ? 64u * AHEAD_WORDS : 0u); w0 < n_mw + 2u; w0 += 64u * K) {
                uint32_t mv[K];
#pragma unroll
                for (int kk = 0; kk < K; kk++) mv[kk] = w0 + 64u * kk < n_mw ? mp[w0 + 64u * kk] : 0u;
#pragma unroll
                for (int kk = 0; kk < K; kk++)
                    if (w0 + 64u * kk < n_mw + 2u) U[w0 + 64u * kk] = __builtin_bswap32(mv[kk]);
            }
            wave_sync();
        } else {
            scramble_packet<8, false, RAW>(k, ri, T, in, A, U, s_crc, s_scr, lane, nullptr, nullptr);
        }

        SEC("copy");
        // ---- preamble rows (ftw/ftw_ofdm_preamble.cc:79-95) and zero gap (ftw/ftw_zerogap.cc:88-99)
        if (pre_rows) {
            const float4 *pre4 = reinterpret_cast<const float4 *>(s_pre);
            for (uint32_t i = lane; i < PRE_ROWS * SYM_LEN / 2; i += 64) iq_store(win, i, pre4[i]);
            if constexpr (PNC)
                if (pre_rows > (uint32_t)PRE_ROWS)        // PNC roles: two more rows
                    for (uint32_t i = PRE_ROWS * SYM_LEN / 2 + lane; i < PNC_PRE_ROWS * SYM_LEN / 2; i += 64) iq_store(win, i, pre4[i]);
        }
        if (gap_rows) {
            const uint32_t gp = (pre_rows + sym_rows) * (SYM_LEN / 2);
            for (uint32_t i = lane; i < GAP_ROWS * SYM_LEN / 2; i += 64) iq_store(win, gp + i, make_float4(0.f, 0.f, 0.f, 0.f));
        }

        // chunks of up to 64 DATA symbols (one per lane); the first chunk also carries the SIGNAL
        // symbol, in slot row 0, so its DATA rows sit one row further down
        for (uint32_t d0 = 0; d0 == 0 || d0 < k.n_sym; d0 += 64u) {
            const uint32_t shift = d0 == 0 ? 1u : 0u;
            SEC("e4_signal");
            // ---- SIGNAL row: 48 coded bits (N = 48: first permutation only, pu:245-249), BPSK
            if (d0 == 0) {
                unsigned long long mi;                    // bit 63 - d = interleaved SIGNAL bit d
                if constexpr (FROM_MSG) {
                    mi = ((unsigned long long)U[0] << 32) | U[1];
                } else {
                    const uint32_t t = 63u - lane;        // ballot bit `lane` = carrier 63 - lane
                    uint32_t bi = 0;
                    if (t < 48u) bi = coded_bit(U, 16u * (t % 3u) + t / 3u);
                    mi = __ballot(bi);
                }
                const uint32_t neg0 = pilot_negative(k.pilot_base % 127u, pneg_lo, pneg_hi);
                if (lane < 4 * SLOT_ROW_WORDS) {
                    const int sig_c = s_sigc[lane];
                    const uint32_t v = sig_c >= 0 ? (uint32_t)LUT_PM1 | ((uint32_t)(mi >> (63 - sig_c)) & 1u)
                                                  : ((sig_c == -3 ? (uint32_t)LUT_PM1 : LUT_PM1 + 1u) ^ neg0);
                    reinterpret_cast<uint8_t *>(SI)[lane] = (uint8_t)v;
                }
            }
            SEC("e5_data");
            // ---- DATA rows.  REGIME != 0: the batch has one regime, known when the kernel was
            // chosen, and only that instantiation is compiled in; 0 dispatches per packet.
            uint32_t *rows = SI + SLOT_ROW_WORDS * shift;
            if constexpr (FROM_MSG) {
                switch (ri.n_bpsc) {
                case 1: message_slots<1>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 2: message_slots<2>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 4: message_slots<4>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                default: message_slots<6>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                }
            } else {
                switch (REGIME ? (uint32_t)REGIME : k.regime) {
                case 1: symbol_slots<1, 0, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 2: symbol_slots<1, 1, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 3: symbol_slots<2, 0, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 4: symbol_slots<2, 1, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 5: symbol_slots<4, 0, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 6: symbol_slots<4, 1, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                case 7: symbol_slots<6, 2, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                default: symbol_slots<6, 1, !PNC>(U, rows, d0, k.n_sym, k.pilot_base, pneg_lo, pneg_hi, lane); break;
                }
            }
            if (PNC && pilots_fixed) {
                // PNC pilot block: the rows just built carry the polarity-sequence pilots;
                // each lane replaces them in its own DATA row, lane 0 also in the SIGNAL row
                int lane_p = lane;                        // opaque: keeps these addresses out of the encoder's way
                asm volatile("" : "+v"(lane_p));
                if (d0 + (uint32_t)lane_p < k.n_sym) set_row_pilots(rows + SLOT_ROW_WORDS * lane_p, pilots_fixed);
                if (d0 == 0 && lane_p == 0) set_row_pilots(SI, pilots_fixed);
            }
            wave_sync();

            // Lane-dependent addressing of the row phase, derived here from an opaque copy of the
            // lane number: hoisted out of the packet loop these ~25 registers would stay live
            // through the encoder, whose bit table needs the register file to itself.
            if constexpr (AHEAD)
                if (d0 == 0) fetch_ahead(next_packet(p, false));
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
            const int l = lane_r & 7, g = lane_r >> 3;
            const uint32_t sel_lo = s_win[l][0], sel_hi = s_win[l][1], slot_sh = s_win[l][3];
            const uint32_t slot_word0 = SLOT_ROW_WORDS * (uint32_t)g + s_win[l][2];
            const uint32_t rows_chunk = (k.n_sym - d0 < 64u ? k.n_sym - d0 : 64u) + shift;
            const uint32_t row_base = d0 == 0 ? 0u : d0 + 1u;           // first output row of the chunk
            // The row phase runs at raised wave priority: a wave that has rows to emit is issued ahead
            // of the waves still encoding, so the memory side is fed whenever anything can feed it.  With
            // the output in one part of the device memory (stores at 5.65 TB/s) this changes nothing
            // measurable; with the output spread over the parts (7.1 TB/s) it is worth 4.6 %
            // (8.33 -> 7.94 ms per 10^6 packets, variants interleaved in one process).
            __builtin_amdgcn_s_setprio(3);
            for (uint32_t r0 = 0; r0 < rows_chunk; r0 += 8u) {
                SEC("map");
                const bool active = r0 + (uint32_t)g < rows_chunk;
                v2f x[8];
                if (active) {
                    const uint32_t *rp = SI + SLOT_ROW_WORDS * r0 + slot_word0;
                    const uint32_t d0 = rp[0], d1 = rp[1], d2 = rp[2];
                    const uint32_t wlo = __builtin_amdgcn_alignbyte(d1, d0, slot_sh);
                    const uint32_t whi = __builtin_amdgcn_alignbyte(d2, d1, slot_sh);
                    const uint32_t s_lo = __builtin_amdgcn_perm(whi, wlo, sel_lo);
                    const uint32_t s_hi = __builtin_amdgcn_perm(whi, wlo, sel_hi);
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        x[m] = lut[(s_lo >> (8 * m)) & 0xFFu];
                        x[m + 4] = lut[(s_hi >> (8 * m)) & 0xFFu];
                    }
                }
                if constexpr (FROM_MSG) {
                    // tap: the mapper block's 48 values per symbol, in carrier order
                    if (mapped && active) {
                        float2 *mrow = mapped + ((size_t)k.sym_off + row_base + r0 + (uint32_t)g) * 48u;
#pragma unroll
                        for (int m = 0; m < 8; m++) {
                            const int c = bin_code(l + 8 * m);
                            if (c >= 0) mrow[c] = make_float2(x[m].x, x[m].y);
                        }
                    }
                    // taps: the pilot block's 53 carriers (-26..+26, ftw/ftw_ofdm_pilot_cc.cc:119-155) and
                    // the carrier map's 64 bins (ftw/ftw_ofdm_cmap_cc.cc:102-107), i.e. the transform's input
                    if ((pilot_tap || cmap_tap) && active) {
                        const size_t sym = (size_t)k.sym_off + row_base + r0 + (uint32_t)g;
#pragma unroll
                        for (int m = 0; m < 8; m++) {
                            const int bin = l + 8 * m;
                            const int f = bin < 32 ? bin : bin - 64;
                            if (cmap_tap) cmap_tap[sym * 64u + (uint32_t)bin] = make_float2(x[m].x, x[m].y);
                            if (pilot_tap && f >= -26 && f <= 26) pilot_tap[sym * 53u + (uint32_t)(f + 26)] = make_float2(x[m].x, x[m].y);
                        }
                    }
                }
                ifft64_emit(active, x, l, g, lane_r, buf, reinterpret_cast<const v2f *>(s_tw), swap_iq, win,
                            (pre_rows + row_base + r0) * (SYM_LEN / 2),
                            rows_chunk - r0 < 8u ? rows_chunk - r0 : 8u);
                SEC("end");
                wave_sync();
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    if constexpr (!UNIFORM) sign_off(P.next_pkt, gridDim.x * (uint32_t)n_wv, lane);
}

// ======================================================================================
// Plan building on the device (ragged plans from device arrays)
// ======================================================================================

// Geometry of one packet: the arithmetic of get_info / ftw_make (pu:81-97,160-171), shared by the
// host entry point and the device-side plan builder.
struct PktGeom {
    int n_sym, n_pad, txtime, length, plcp_len, msg_len, n_rows;
};
__host__ __device__ __forceinline__ PktGeom pkt_geometry(const RegimeInfo &ri, int payload_len, double symbol_time,
                                                          unsigned flags)
{
    PktGeom g;
    // OFDMTX_F_RAW_PSDU: the caller's bytes are MPDU + FCS already, LENGTH is their count
    const int mpdu = (flags & OFDMTX_F_RAW_PSDU) ? payload_len - 4 : 24 + payload_len;
    const int nbits = 16 + 32 + 6 + 8 * mpdu;
    g.n_sym = (nbits + ri.n_dbps - 1) / ri.n_dbps;                       // pu:85
    g.n_pad = g.n_sym * ri.n_dbps - nbits;                               // pu:94-97
    g.txtime = (int)(5 * symbol_time + symbol_time * g.n_sym) + 2;       // pu:90
    g.length = mpdu + 4;
    g.plcp_len = 5 + g.length + (g.n_pad + 6 + 7) / 8;                   // pu:160-171,205
    g.msg_len = 6 + g.n_sym * ri.n_cbps / 8;
    g.n_rows = ((flags & OFDMTX_F_PREAMBLE) ? ((flags & (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B)) ? PNC_PRE_ROWS : PRE_ROWS) : 0) +
               1 + g.n_sym + ((flags & OFDMTX_F_ZEROGAP) ? GAP_ROWS : 0);
    return g;
}

struct PlanDevOut {                   // totals and maxima read back by the host
    unsigned long long max_plcp_len, max_msg_len, max_n_sym, pay_extent;
    uint32_t error;                   // bit 0: regime outside 1..8, bit 1: payload too long
};

// per packet: geometry into the record and the PLAN_COLS columns col[c * pitch + i]
__global__ void k_plan_geom(uint32_t n, const uint8_t *__restrict__ regimes, const uint32_t *__restrict__ lens,
                            const uint64_t *__restrict__ pay_off, double symbol_time, unsigned flags,
                            PktRec *__restrict__ recs, uint32_t *__restrict__ plcp_len, uint32_t *__restrict__ msg_len,
                            unsigned long long *__restrict__ col, size_t pitch, PlanDevOut *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int regime = regimes[i];
    const uint32_t len = lens[i];
    uint32_t err = 0;
    if (regime < 1 || regime > 8) err |= 1u;
    if (len > (uint32_t)MAXLEN || (!(flags & OFDMTX_F_RAW_PSDU) && 24u + len + 4u > (uint32_t)MAXLEN)) err |= 2u;
    if (err) {
        atomicOr(&out->error, err);
        return;
    }
    const PktGeom g = pkt_geometry(regime_info(regime), (int)len, symbol_time, flags);
    PktRec r;
    r.pay_off = pay_off ? pay_off[i] : 0;
    r.plcp_off = 0;
    r.msg_off = 0;
    r.sym_off = 0;
    r.pay_len = len;
    r.n_sym = (uint32_t)g.n_sym;
    r.pilot_base = 0;
    r.txtime = (uint16_t)((uint32_t)g.txtime & 0xFFFFu);
    r.regime = (uint8_t)regime;
    r.pad_ = 0;
    r.pad2_ = 0;
    recs[i] = r;
    plcp_len[i] = (uint32_t)g.plcp_len;
    msg_len[i] = (uint32_t)g.msg_len;
    col[0 * pitch + i] = len;
    col[1 * pitch + i] = ((uint32_t)g.plcp_len + 15u) & ~15u;
    col[2 * pitch + i] = ((uint32_t)g.msg_len + 15u) & ~15u;
    col[3 * pitch + i] = (uint32_t)g.n_sym + 1u;
    col[4 * pitch + i] = (uint32_t)g.n_rows;
    atomicMax(&out->max_plcp_len, (unsigned long long)g.plcp_len);
    atomicMax(&out->max_msg_len, (unsigned long long)g.msg_len);
    atomicMax(&out->max_n_sym, (unsigned long long)g.n_sym);
    if (pay_off) atomicMax(&out->pay_extent, (unsigned long long)(pay_off[i] + len));
}

// exclusive scan of every column, in three steps: inside blocks of SCAN_BLOCK items, over the
// block totals, and the block offsets added back (element n of a column receives the total)
__global__ void __launch_bounds__(256) k_scan_local(unsigned long long *col, size_t pitch, uint32_t n,
                                                    unsigned long long *block_sums, uint32_t n_blocks)
{
    __shared__ unsigned long long s[256];
    unsigned long long *c = col + (size_t)blockIdx.y * pitch;
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = base + k < n ? c[base + k] : 0;
        sum += v[k];
    }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const unsigned long long t = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned long long run = s[threadIdx.x] - sum;                // exclusive prefix of this thread
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k <= n) c[base + k] = run;                     // (index n: written again by the last step)
        run += v[k];
    }
    if (threadIdx.x == 255) block_sums[(size_t)blockIdx.y * n_blocks + blockIdx.x] = s[255];
}

__global__ void __launch_bounds__(256) k_scan_sums(unsigned long long *block_sums, uint32_t n_blocks)
{
    __shared__ unsigned long long s[256];
    unsigned long long *b = block_sums + (size_t)blockIdx.x * n_blocks;
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_blocks ? b[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned long long t = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) b[i] = carry + s[threadIdx.x] - v;
        carry += s[255];
        __syncthreads();
    }
}

// adds the block offsets and builds the records: offsets, row table, pilot index
__global__ void k_plan_fill(uint32_t n, unsigned long long *__restrict__ col, size_t pitch,
                            const unsigned long long *__restrict__ block_sums, uint32_t n_blocks, int have_pay_off,
                            int pilot_mode, uint32_t pilot_start, PktRec *__restrict__ recs,
                            uint32_t *__restrict__ row_start)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    unsigned long long v[PLAN_COLS];
#pragma unroll
    for (int c = 0; c < PLAN_COLS; c++) {
        v[c] = col[c * pitch + i] + block_sums[(size_t)c * n_blocks + i / SCAN_BLOCK];
        col[c * pitch + i] = v[c];
    }
    row_start[i] = (uint32_t)v[4];
    if (i == n) return;
    PktRec r = recs[i];
    if (!have_pay_off) r.pay_off = v[0];
    r.plcp_off = v[1];
    r.msg_off = v[2];
    r.sym_off = (uint32_t)v[3];
    r.pilot_base = pilot_mode == OFDMTX_PILOT_REF_STREAM ? (uint32_t)((pilot_start + v[3]) % 127u) : pilot_start;
    recs[i] = r;
}

// ftw/ftw_zerogap.cc:88-92 with a gap row that is not all zeros: the 13 rows behind every packet
// are copies of the context's row, times the plan's amplitude (I and Q exchanged if asked).  Runs
// behind k_tx, which has written zeros there.  One workgroup per packet.
__global__ void __launch_bounds__(256) k_gap_rows(PlanParams P, const float2 *__restrict__ gap, float2 *__restrict__ iq)
{
    __shared__ __align__(16) float2 s_gap[SYM_LEN];
    const bool swap_iq = (P.flags & OFDMTX_F_SWAPIQ) != 0;
    for (int i = threadIdx.x; i < SYM_LEN; i += blockDim.x) {
        const float2 v = gap[i];
        s_gap[i] = swap_iq ? make_float2(v.y * P.amp, v.x * P.amp) : make_float2(v.x * P.amp, v.y * P.amp);
    }
    __syncthreads();
    const float4 *g4 = reinterpret_cast<const float4 *>(s_gap);
    for (uint32_t p = blockIdx.x; p < P.n_pkts; p += gridDim.x) {
        const size_t row_end = P.uniform ? (size_t)(p + 1u) * P.rows_per_pkt : (size_t)P.row_start[p + 1u];
        float4 *dst = reinterpret_cast<float4 *>(iq + (row_end - GAP_ROWS) * SYM_LEN);
        for (uint32_t i = threadIdx.x; i < GAP_ROWS * SYM_LEN / 2; i += blockDim.x) dst[i] = g4[i % (SYM_LEN / 2)];
    }
}

}  // namespace

thread_local int ofdmtx_g_last_hip = 0;

// ======================================================================================
// Host side
// ======================================================================================

DevTables ofdmtx_ctx::tables() const
{
    DevTables t;
    t.scr_word = d_scr;
    t.lut = d_lut;
    t.preamble = d_pre;
    t.tw64 = d_tw;
    t.crc_tab = d_crc;
    t.xpow8 = d_xpow8;
    for (int i = 0; i < 4; i++) t.pilot_neg[i] = OFDMTX_PILOT_NEG[i];
    return t;
}

extern "C" {

const char *ofdmtx_strerror(int code)
{
    switch (code) {
    case OFDMTX_OK: return "ok";
    case OFDMTX_EINVAL: return "invalid argument";
    case OFDMTX_EREGIME: return "regime must be 1..8";
    case OFDMTX_ELENGTH: return "MPDU-length must be in [0, 3168]";
    case OFDMTX_EHIP: return "HIP runtime error";
    case OFDMTX_ENOMEM: return "out of memory";
    case OFDMTX_ESHAPE: return "block shape not supported (802.11a needs 48 data carriers, 64-point FFT)";
    case OFDMTX_ETOOBIG: return "batch too large for one launch";
    case OFDMTX_EBUSY: return "busy (reserved; not returned)";
    case OFDMTX_ECOMM: return "RCCL not available or an RCCL call failed";
    default: return "unknown error";
    }
}

int ofdmtx_last_hip_error(void) { return ofdmtx_g_last_hip; }

uint32_t ofdmtx_crc32(const uint8_t *buf, size_t len)
{
    // ftw/ftw_crc32.cc:33-82
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < len; i++) {
        crc ^= buf[i];
        for (int j = 0; j < 8; j++) crc = (crc >> 1) ^ (CRC_POLY & (0u - (crc & 1u)));
    }
    uint32_t rev = 0;
    for (int j = 0; j < 32; j++) rev |= ((crc >> j) & 1u) << (31 - j);
    return rev ^ 0xFFFFFFFFu;
}

uint32_t ofdmtx_update_crc32(uint32_t crc, const uint8_t *buf, size_t len)
{
    (void)crc;   // ftw/ftw_crc32.cc:53 restarts from all ones whatever it is given
    return ofdmtx_crc32(buf, len) ^ 0xFFFFFFFFu;
}

int ofdmtx_packet_info(int regime, int payload_len, double symbol_time, unsigned flags,
                       ofdmtx_pkt_info *out)
{
    if (!out || payload_len < 0) return OFDMTX_EINVAL;
    if (regime < 1 || regime > 8) return OFDMTX_EREGIME;
    const RegimeInfo ri = regime_info(regime);
    const PktGeom g = pkt_geometry(ri, payload_len, symbol_time, flags);
    out->n_cbps = ri.n_cbps;
    out->n_bpsc = ri.n_bpsc;
    out->n_dbps = ri.n_dbps;
    out->n_sym = g.n_sym;
    out->n_data = g.n_sym * ri.n_dbps;                                  // pu:94
    out->n_pad = g.n_pad;
    out->txtime = g.txtime;
    out->length = g.length;
    out->plcp_len = g.plcp_len;
    out->msg_len = g.msg_len;
    out->n_rows = g.n_rows;
    // the length limit is ftw_make's (pu:119-121), get_info itself has none: *out is complete either way
    return g.length > MAXLEN ? OFDMTX_ELENGTH : OFDMTX_OK;
}

int ofdmtx_create(int device, ofdmtx_ctx **out)
{
    if (!out) return OFDMTX_EINVAL;
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    ofdmtx_ctx *c = new (std::nothrow) ofdmtx_ctx();
    if (!c) return OFDMTX_ENOMEM;
    c->device = device;
    // CRC helper tables: slicing-by-4 and the powers x^(8n) mod P used to merge lane chunks
    std::vector<uint32_t> tab(1024), xp(MAXLEN + 1);
    for (uint32_t b = 0; b < 256; b++) {
        uint32_t crc = b;
        for (int j = 0; j < 8; j++) crc = (crc >> 1) ^ (CRC_POLY & (0u - (crc & 1u)));   // ftw/ftw_crc32.cc:41-48
        tab[b] = crc;
    }
    for (int t = 1; t < 4; t++)
        for (uint32_t b = 0; b < 256; b++)
            tab[256 * t + b] = (tab[256 * (t - 1) + b] >> 8) ^ tab[tab[256 * (t - 1) + b] & 0xFFu];
    uint32_t v = 0x80000000u;
    for (int n = 0; n <= MAXLEN; n++) {
        xp[n] = v;
        for (int j = 0; j < 8; j++) v = (v >> 1) ^ (CRC_POLY & (0u - (v & 1u)));
    }
    float lut[2 * LUT_SIZE];
    memset(lut, 0, sizeof(lut));
    lut[2 * LUT_PM1] = -1.f; lut[2 * LUT_PM1 + 2] = 1.f;        // ftw/ftw_ofdm_mapper.cc:179, pilots
    memcpy(lut + 2 * 4, OFDMTX_CONST_DEFAULT + 2 * 0, 2 * 2 * sizeof(float));     // BPSK
    memcpy(lut + 2 * 8, OFDMTX_CONST_DEFAULT + 2 * 2, 2 * 4 * sizeof(float));     // QPSK
    memcpy(lut + 2 * 16, OFDMTX_CONST_DEFAULT + 2 * 6, 2 * 16 * sizeof(float));   // 16-QAM
    memcpy(lut + 2 * 64, OFDMTX_CONST_DEFAULT + 2 * 22, 2 * 64 * sizeof(float));  // 64-QAM

    // every step runs only while the ones before it succeeded; one exit frees whatever exists
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) c->n_cu = prop.multiProcessorCount;
    if (e == hipSuccess) e = hipMalloc(&c->d_scr, 127 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&c->d_lut, LUT_SIZE * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(&c->d_pre, 320 * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(&c->d_tw, 64 * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(&c->d_gap, SYM_LEN * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(&c->d_crc, tab.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&c->d_xpow8, xp.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_pinned, 256, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&c->d_turns, RX_TURNS * 64);
    if (e == hipSuccess) e = hipMemset(c->d_turns, 0, RX_TURNS * 64);
    // this device's copy of the static work area (the device is current: hipSetDevice above)
    if (e == hipSuccess) e = hipGetSymbolAddress(reinterpret_cast<void **>(&c->d_static), HIP_SYMBOL(g_static_area));
    if (e == hipSuccess && device >= 0 && device < MAX_DEVICES) {
        StaticGuard &g = g_static_guard[device];
        std::lock_guard<std::mutex> guard(g.lock);
        if (!g.done) e = hipEventCreateWithFlags(&g.done, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMemcpy(c->d_scr, OFDMTX_SCR_WORD, sizeof(OFDMTX_SCR_WORD), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_lut, lut, sizeof(lut), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_pre, OFDMTX_PREAMBLE, sizeof(OFDMTX_PREAMBLE), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_tw, OFDMTX_TW64, sizeof(OFDMTX_TW64), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c->d_gap, 0, SYM_LEN * sizeof(float2));       // pu:510: 80 zeros
    if (e == hipSuccess) e = hipMemcpy(c->d_crc, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->d_xpow8, xp.data(), xp.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ofdmtx_g_last_hip = (int)e;
        ofdmtx_destroy(c);
        return OFDMTX_EHIP;
    }
    *out = c;
    return OFDMTX_OK;
}

int ofdmtx_destroy(ofdmtx_ctx *c)
{
    if (!c) return OFDMTX_EINVAL;
    (void)hipSetDevice(c->device);
    if (c->d_scr) (void)hipFree(c->d_scr);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_pre) (void)hipFree(c->d_pre);
    if (c->d_tw) (void)hipFree(c->d_tw);
    if (c->d_gap) (void)hipFree(c->d_gap);
    if (c->d_crc) (void)hipFree(c->d_crc);
    if (c->d_xpow8) (void)hipFree(c->d_xpow8);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->d_turns) (void)hipFree(c->d_turns);
    delete c;
    return OFDMTX_OK;
}

int ofdmtx_set_preamble(ofdmtx_ctx *c, const float *re_im, int n_samples)
{
    if (!c || !re_im) return OFDMTX_EINVAL;
    if (n_samples < PRE_ROWS * SYM_LEN) return OFDMTX_EINVAL;   // the blocks copy four rows of 80 (ftw/ftw_ofdm_preamble.cc:79-83)
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());                             // kernels in flight still read the old table
    HIPCHK(hipMemcpy(c->d_pre, re_im, (size_t)PRE_ROWS * SYM_LEN * sizeof(float2), hipMemcpyHostToDevice));
    return OFDMTX_OK;
}

int ofdmtx_set_gap(ofdmtx_ctx *c, const float *re_im, int n_samples)
{
    if (!c || !re_im) return OFDMTX_EINVAL;
    if (n_samples < SYM_LEN) return OFDMTX_EINVAL;              // one row of 80 (ftw/ftw_zerogap.cc:88-92)
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(c->d_gap, re_im, (size_t)SYM_LEN * sizeof(float2), hipMemcpyHostToDevice));
    bool nz = false;
    for (int i = 0; i < 2 * SYM_LEN; i++) nz = nz || re_im[i] != 0.f;
    c->gap_nonzero = nz;
    return OFDMTX_OK;
}

int ofdmtx_set_constellation(ofdmtx_ctx *c, int n_bpsc, const float *re_im, int n)
{
    if (!c || !re_im) return OFDMTX_EINVAL;
    int off;
    switch (n_bpsc) {
    case 1: off = 4; break;
    case 2: off = 8; break;
    case 4: off = 16; break;
    case 6: off = 64; break;
    default: return OFDMTX_EINVAL;
    }
    if (n != (1 << n_bpsc)) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());                             // kernels in flight still read the old table
    HIPCHK(hipMemcpy(c->d_lut + off, re_im, (size_t)n * sizeof(float2), hipMemcpyHostToDevice));
    return OFDMTX_OK;
}

int ofdmtx_get_table(int which, float *out, int n_floats)
{
    const float *src;
    int n;
    switch (which) {
    case OFDMTX_TABLE_BPSK: src = OFDMTX_CONST_DEFAULT; n = 4; break;
    case OFDMTX_TABLE_QPSK: src = OFDMTX_CONST_DEFAULT + 4; n = 8; break;
    case OFDMTX_TABLE_QAM16: src = OFDMTX_CONST_DEFAULT + 12; n = 32; break;
    case OFDMTX_TABLE_QAM64: src = OFDMTX_CONST_DEFAULT + 44; n = 128; break;
    case OFDMTX_TABLE_PREAMBLE: src = OFDMTX_PREAMBLE; n = 640; break;
    default: return OFDMTX_EINVAL;
    }
    if (!out || n_floats < n) return OFDMTX_EINVAL;
    memcpy(out, src, (size_t)n * sizeof(float));
    return n;
}

static inline uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

// LDS words k_tx needs per wave for a packet: first region = natural MPDU (2 + n_words + 1) or
// the slot rows of one chunk (+3 words read past the last row's window), whichever is larger;
// u = the scrambled words (n_words + 2), or the message words when the input is messages
static inline uint32_t tx_words_for(const ofdmtx_pkt_info &pi)
{
    const uint32_t d = (uint32_t)(pi.plcp_len + 3) / 4 + 4;
    const uint32_t rows = (uint32_t)pi.n_sym + 1 < 65u ? (uint32_t)pi.n_sym + 1 : 65u;
    const uint32_t m = rows * SLOT_ROW_WORDS + 3;
    const uint32_t w = d > m ? d : m;
    return (w + 3u) & ~3u;
}
static inline uint32_t tx_u_words_for(const ofdmtx_pkt_info &pi)
{
    return ((uint32_t)(pi.plcp_len + 3) / 4 + 4 + 3u) & ~3u;
}
static inline uint32_t tx_m_words_for(const ofdmtx_pkt_info &pi)
{
    return ((uint32_t)(pi.msg_len + 3) / 4 + 4 + 3u) & ~3u;
}

static bool plan_modes_ok(int pilot_mode, unsigned flags)
{
    if (pilot_mode < OFDMTX_PILOT_REF_STREAM || pilot_mode > OFDMTX_PILOT_PNC_ALL) return false;
    const unsigned roles = flags & (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B);
    if (roles == (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B)) return false;       // one role at a time
    if (roles && !(flags & OFDMTX_F_PREAMBLE)) return false;            // the role only shapes the preamble
    // a raw PSDU with a PNC role or PNC pilots: no instantiation of the kernels covers the pair
    if ((flags & OFDMTX_F_RAW_PSDU) && (roles || pilot_mode >= OFDMTX_PILOT_PNC_A)) return false;
    return true;
}

// packets per draw from a ragged plan's work counters: about 128 output rows (80 KB of samples)
// for k_tx, eight times as many packets for the much cheaper k_encode_bits; 1 for batches of
// ordinary packets, and never so many that a small batch would be left to a few waves
static void size_draws(PlanParams &pp, uint64_t n_pkts, uint64_t rows)
{
    const uint64_t avg_rows = n_pkts ? (rows + n_pkts - 1) / n_pkts : 1;
    const uint64_t by_rows = avg_rows >= 128 ? 1 : 128 / (avg_rows ? avg_rows : 1);
    const uint64_t by_count = n_pkts / 16384u > 0 ? n_pkts / 16384u : 1;
    pp.grab_tx = (uint32_t)(by_rows < by_count ? by_rows : by_count);
    pp.grab_enc = (uint32_t)(8u * by_rows < by_count ? 8u * by_rows : by_count);
}

static void plan_common(ofdmtx_plan *pl, ofdmtx_ctx *ctx, uint64_t n, int pilot_mode,
                        uint32_t pilot_start, unsigned flags, float amp)
{
    memset(&pl->pp, 0, sizeof(pl->pp));
    pl->ctx = ctx;
    pl->device = ctx->device;
    pl->pp.n_pkts = (uint32_t)n;
    pl->pp.pilot_mode = (uint32_t)pilot_mode;
    pl->pp.pilot_start = pilot_start % 127u;
    pl->pp.flags = flags;
    pl->pp.amp = amp;
    pl->d_recs = nullptr;
    pl->d_row_start = nullptr;
    pl->d_plcp_len = nullptr;
    pl->d_msg_len = nullptr;
    pl->layout_on_host = true;
}

int ofdmtx_plan_uniform(ofdmtx_ctx *ctx, uint64_t n_pkts, int regime, uint32_t payload_len,
                        uint64_t payload_stride, double symbol_time, int pilot_mode,
                        uint32_t pilot_start, unsigned flags, float amp, ofdmtx_plan **out)
{
    if (!ctx || !out) return OFDMTX_EINVAL;
    *out = nullptr;
    if (!plan_modes_ok(pilot_mode, flags))
        return OFDMTX_EINVAL;
    if (payload_len > (uint32_t)MAXLEN) return OFDMTX_ELENGTH;
    ofdmtx_pkt_info pi;
    int rc = ofdmtx_packet_info(regime, (int)payload_len, symbol_time, flags, &pi);
    if (rc) return rc;
    if (payload_stride == 0) payload_stride = payload_len;
    if (payload_stride < payload_len) return OFDMTX_EINVAL;
    if (n_pkts * (uint64_t)pi.n_rows >= 0xFFFFFFF0ull || n_pkts >= 0xFFFFFFF0ull) return OFDMTX_ETOOBIG;

    ofdmtx_plan *pl = new (std::nothrow) ofdmtx_plan();
    if (!pl) return OFDMTX_ENOMEM;
    plan_common(pl, ctx, n_pkts, pilot_mode, pilot_start, flags, amp);
    PlanParams &pp = pl->pp;
    pp.uniform = 1;
    pp.regime = (uint32_t)regime;
    pp.pay_len = payload_len;
    pp.n_sym = (uint32_t)pi.n_sym;
    pp.txtime = (uint32_t)pi.txtime & 0xFFFFu;                           // pu:91 masks to 16 bits
    pp.pay_stride = payload_stride;
    pp.plcp_stride = align16((uint32_t)pi.plcp_len);
    pp.msg_stride = align16((uint32_t)pi.msg_len);
    pp.rows_per_pkt = (uint32_t)pi.n_rows;
    pp.total_rows = (uint32_t)(n_pkts * (uint64_t)pi.n_rows);
    pl->uni = pi;
    pl->tx_a_words = tx_words_for(pi);
    pl->tx_u_words = tx_u_words_for(pi);
    pl->tx_m_words = tx_m_words_for(pi);
    pl->max_sym_rows = (uint32_t)pi.n_sym + 1;
    ofdmtx_plan_sizes &sz = pl->sizes;
    sz.n_pkts = n_pkts;
    sz.payload_bytes = n_pkts ? (n_pkts - 1) * payload_stride + payload_len : 0;
    sz.plcp_bytes = n_pkts * (uint64_t)pp.plcp_stride + 16;
    sz.msg_bytes = n_pkts * (uint64_t)pp.msg_stride + 16;
    sz.n_symbols = n_pkts * (uint64_t)(pi.n_sym + 1);
    sz.n_rows = pp.total_rows;
    sz.n_samples = sz.n_rows * SYM_LEN;
    *out = pl;
    return OFDMTX_OK;
}

int ofdmtx_plan_ragged(ofdmtx_ctx *ctx, uint64_t n_pkts, const uint8_t *regimes,
                       const uint32_t *payload_len, const uint64_t *payload_off,
                       double symbol_time, int pilot_mode, uint32_t pilot_start, unsigned flags,
                       float amp, ofdmtx_plan **out)
{
    if (!ctx || !out || (n_pkts && (!regimes || !payload_len))) return OFDMTX_EINVAL;
    *out = nullptr;
    if (!plan_modes_ok(pilot_mode, flags))
        return OFDMTX_EINVAL;
    if (n_pkts >= 0xFFFFFFF0ull) return OFDMTX_ETOOBIG;
    ofdmtx_plan *pl = new (std::nothrow) ofdmtx_plan();
    if (!pl) return OFDMTX_ENOMEM;
    plan_common(pl, ctx, n_pkts, pilot_mode, pilot_start, flags, amp);
    PlanParams &pp = pl->pp;
    pp.uniform = 0;
    pl->tx_a_words = 16;
    pl->tx_u_words = 16;
    pl->tx_m_words = 16;
    pl->max_sym_rows = 1;
    pl->recs.resize(n_pkts);
    pl->row_start.resize(n_pkts + 1);
    pl->plcp_len.resize(n_pkts);
    pl->msg_len.resize(n_pkts);
    uint64_t pay = 0, plcp = 0, msg = 0, sym = 0, rows = 0, pay_extent = 0;
    uint64_t pilot = pp.pilot_start;
    for (uint64_t i = 0; i < n_pkts; i++) {
        ofdmtx_pkt_info pi;
        int rc = payload_len[i] > (uint32_t)MAXLEN
                     ? OFDMTX_ELENGTH
                     : ofdmtx_packet_info(regimes[i], (int)payload_len[i], symbol_time, flags, &pi);
        if (rc) {
            delete pl;
            return rc;
        }
        if (tx_words_for(pi) > pl->tx_a_words) pl->tx_a_words = tx_words_for(pi);
        if (tx_u_words_for(pi) > pl->tx_u_words) pl->tx_u_words = tx_u_words_for(pi);
        if (tx_m_words_for(pi) > pl->tx_m_words) pl->tx_m_words = tx_m_words_for(pi);
        if ((uint32_t)pi.n_sym + 1 > pl->max_sym_rows) pl->max_sym_rows = (uint32_t)pi.n_sym + 1;
        PktRec &r = pl->recs[i];
        memset(&r, 0, sizeof(r));
        r.pay_off = payload_off ? payload_off[i] : pay;
        r.plcp_off = plcp;
        r.msg_off = msg;
        r.sym_off = (uint32_t)sym;
        r.pay_len = payload_len[i];
        r.n_sym = (uint32_t)pi.n_sym;
        r.pilot_base = (uint32_t)(pilot_mode == OFDMTX_PILOT_REF_STREAM ? pilot % 127u : pp.pilot_start);
        r.txtime = (uint16_t)((uint32_t)pi.txtime & 0xFFFFu);
        r.regime = regimes[i];
        pl->row_start[i] = (uint32_t)rows;
        pl->plcp_len[i] = (uint32_t)pi.plcp_len;
        pl->msg_len[i] = (uint32_t)pi.msg_len;
        if (r.pay_off + payload_len[i] > pay_extent) pay_extent = r.pay_off + payload_len[i];
        pay += payload_len[i];
        plcp += align16((uint32_t)pi.plcp_len);
        msg += align16((uint32_t)pi.msg_len);
        sym += (uint64_t)pi.n_sym + 1;
        pilot += (uint64_t)pi.n_sym + 1;
        rows += (uint64_t)pi.n_rows;
        if (rows >= 0xFFFFFFF0ull || sym >= 0xFFFFFFF0ull) {
            delete pl;
            return OFDMTX_ETOOBIG;
        }
    }
    pl->row_start[n_pkts] = (uint32_t)rows;
    pp.total_rows = (uint32_t)rows;
    ofdmtx_plan_sizes &sz = pl->sizes;
    sz.n_pkts = n_pkts;
    sz.payload_bytes = pay_extent;
    sz.plcp_bytes = plcp + 16;
    sz.msg_bytes = msg + 16;
    sz.n_symbols = sym;
    sz.n_rows = rows;
    sz.n_samples = rows * SYM_LEN;

    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->busy_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&pl->d_recs, (n_pkts ? n_pkts : 1) * sizeof(PktRec));
    // the fused kernel's work counter sits behind the row table, on a 256-byte line of its own
    const size_t counter_at = (((size_t)n_pkts + 1) * sizeof(uint32_t) + 255u) & ~(size_t)255u;
    if (e == hipSuccess) e = hipMalloc(&pl->d_row_start, counter_at + 256);
    if (e == hipSuccess) e = hipMemset(reinterpret_cast<uint8_t *>(pl->d_row_start) + counter_at, 0, 256);
    if (e == hipSuccess && n_pkts)
        e = hipMemcpy(pl->d_recs, pl->recs.data(), n_pkts * sizeof(PktRec), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(pl->d_row_start, pl->row_start.data(), (n_pkts + 1) * sizeof(uint32_t),
                      hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ofdmtx_g_last_hip = (int)e;
        ofdmtx_plan_destroy(pl);
        return OFDMTX_EHIP;
    }
    pp.recs = pl->d_recs;
    pp.row_start = pl->d_row_start;
    pp.next_pkt = pl->d_row_start + counter_at / sizeof(uint32_t);
    size_draws(pp, n_pkts, rows);
    *out = pl;
    return OFDMTX_OK;
}

}  // extern "C" (interrupted: helpers of the device-built plans)

// Device tables of a plan that is built by kernels, wherever they live (allocations of a plan
// object, or the static work area).
struct PlanBuildBufs {
    PktRec *recs;
    uint32_t *row_start;              // n + 1 entries
    uint32_t *counters;               // 64 words on a 256-byte line of their own
    uint32_t *plcp_len, *msg_len;     // n entries each
    unsigned long long *col;          // PLAN_COLS x (n + 1)
    unsigned long long *sums;         // PLAN_COLS x n_blocks
    PlanDevOut *out;
};

// Enqueues the plan-building kernels on `st` and the copies of the totals into h_out / h_totals
// (host memory that stays valid until the stream has been synchronised).  Allocates nothing.
static hipError_t enqueue_plan_build(uint32_t n, const uint8_t *d_regimes, const uint32_t *d_payload_len,
                                     const uint64_t *d_payload_off, double symbol_time, int pilot_mode,
                                     uint32_t pilot_start, unsigned flags, const PlanBuildBufs &b, PlanDevOut *h_out,
                                     unsigned long long *h_totals, hipStream_t st)
{
    const size_t pitch = (size_t)n + 1;
    const uint32_t n_blocks = (uint32_t)((pitch + SCAN_BLOCK - 1) / SCAN_BLOCK);
    hipError_t e = hipMemsetAsync(b.counters, 0, 256, st);
    if (e == hipSuccess) e = hipMemsetAsync(b.out, 0, sizeof(PlanDevOut), st);
    if (e == hipSuccess) e = hipMemsetAsync(b.col, 0, PLAN_COLS * pitch * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    if (n)
        hipLaunchKernelGGL(k_plan_geom, dim3((n + 255u) / 256u), dim3(256), 0, st, n, d_regimes, d_payload_len,
                           d_payload_off, symbol_time, flags, b.recs, b.plcp_len, b.msg_len, b.col, pitch, b.out);
    hipLaunchKernelGGL(k_scan_local, dim3(n_blocks, PLAN_COLS), dim3(256), 0, st, b.col, pitch, n, b.sums, n_blocks);
    hipLaunchKernelGGL(k_scan_sums, dim3(PLAN_COLS), dim3(256), 0, st, b.sums, n_blocks);
    hipLaunchKernelGGL(k_plan_fill, dim3((n + 1u + 255u) / 256u), dim3(256), 0, st, n, b.col, pitch, b.sums, n_blocks,
                       d_payload_off ? 1 : 0, pilot_mode, pilot_start, b.recs, b.row_start);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, b.out, sizeof(PlanDevOut), hipMemcpyDeviceToHost, st);
    for (int c = 0; c < PLAN_COLS && e == hipSuccess; c++)
        e = hipMemcpyAsync(&h_totals[c], b.col + c * pitch + n, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    return e;
}

// Error code for what the plan kernels found, 0 if the batch is fine.
static int plan_build_verdict(const PlanDevOut &h, const unsigned long long *totals)
{
    if (h.error & 1u) return OFDMTX_EREGIME;
    if (h.error & 2u) return OFDMTX_ELENGTH;
    if (totals[4] >= 0xFFFFFFF0ull || totals[3] >= 0xFFFFFFF0ull) return OFDMTX_ETOOBIG;
    return OFDMTX_OK;
}

// Fills in what the launchers need from the totals and maxima of a device-built plan.
static void finish_device_plan(ofdmtx_plan *pl, uint64_t n_pkts, bool have_pay_off, const PlanBuildBufs &b,
                               const PlanDevOut &h, const unsigned long long *totals)
{
    // LDS needs of the kernels from the maxima (the same expressions as tx_words_for and friends)
    ofdmtx_pkt_info worst;
    memset(&worst, 0, sizeof(worst));
    worst.plcp_len = (int32_t)h.max_plcp_len;
    worst.msg_len = (int32_t)h.max_msg_len;
    worst.n_sym = (int32_t)h.max_n_sym;
    pl->tx_a_words = tx_words_for(worst) > 16u ? tx_words_for(worst) : 16u;
    pl->tx_u_words = tx_u_words_for(worst) > 16u ? tx_u_words_for(worst) : 16u;
    pl->tx_m_words = tx_m_words_for(worst) > 16u ? tx_m_words_for(worst) : 16u;
    pl->max_sym_rows = (uint32_t)h.max_n_sym + 1u;
    PlanParams &pp = pl->pp;
    pp.total_rows = (uint32_t)totals[4];
    ofdmtx_plan_sizes &sz = pl->sizes;
    sz.n_pkts = n_pkts;
    sz.payload_bytes = have_pay_off ? h.pay_extent : totals[0];
    sz.plcp_bytes = totals[1] + 16;
    sz.msg_bytes = totals[2] + 16;
    sz.n_symbols = totals[3];
    sz.n_rows = totals[4];
    sz.n_samples = totals[4] * SYM_LEN;
    pp.recs = b.recs;
    pp.row_start = b.row_start;
    pp.next_pkt = b.counters;
    size_draws(pp, n_pkts, totals[4]);
}

extern "C" {

int ofdmtx_plan_ragged_device(ofdmtx_ctx *ctx, uint64_t n_pkts, const uint8_t *d_regimes,
                              const uint32_t *d_payload_len, const uint64_t *d_payload_off,
                              double symbol_time, int pilot_mode, uint32_t pilot_start, unsigned flags,
                              float amp, void *stream, ofdmtx_plan **out)
{
    if (!ctx || !out || (n_pkts && (!d_regimes || !d_payload_len))) return OFDMTX_EINVAL;
    *out = nullptr;
    if (!plan_modes_ok(pilot_mode, flags)) return OFDMTX_EINVAL;
    if (n_pkts >= 0xFFFFFFF0ull) return OFDMTX_ETOOBIG;
    HIPCHK(hipSetDevice(ctx->device));
    ofdmtx_plan *pl = new (std::nothrow) ofdmtx_plan();
    if (!pl) return OFDMTX_ENOMEM;
    plan_common(pl, ctx, n_pkts, pilot_mode, pilot_start, flags, amp);
    pl->pp.uniform = 0;
    pl->layout_on_host = false;
    const uint32_t n = (uint32_t)n_pkts;
    hipStream_t st = (hipStream_t)stream;
    const size_t pitch = (size_t)n + 1;
    const uint32_t n_blocks = (uint32_t)((pitch + SCAN_BLOCK - 1) / SCAN_BLOCK);
    const size_t counter_at = (((size_t)n_pkts + 1) * sizeof(uint32_t) + 255u) & ~(size_t)255u;
    PlanBuildBufs b = {};
    PlanDevOut h = {};
    unsigned long long totals[PLAN_COLS] = {};
    hipError_t e = hipEventCreateWithFlags(&pl->busy_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&pl->d_recs, (n_pkts ? n_pkts : 1) * sizeof(PktRec));
    if (e == hipSuccess) e = hipMalloc(&pl->d_row_start, counter_at + 256);
    if (e == hipSuccess) e = hipMalloc(&pl->d_plcp_len, (n_pkts ? n_pkts : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&pl->d_msg_len, (n_pkts ? n_pkts : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&b.col, PLAN_COLS * pitch * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&b.sums, (size_t)PLAN_COLS * n_blocks * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&b.out, sizeof(PlanDevOut));
    if (e == hipSuccess) {
        b.recs = pl->d_recs;
        b.row_start = pl->d_row_start;
        b.counters = pl->d_row_start + counter_at / sizeof(uint32_t);
        b.plcp_len = pl->d_plcp_len;
        b.msg_len = pl->d_msg_len;
        e = enqueue_plan_build(n, d_regimes, d_payload_len, d_payload_off, symbol_time, pilot_mode, pl->pp.pilot_start,
                               flags, b, &h, totals, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (b.col) (void)hipFree(b.col);
    if (b.sums) (void)hipFree(b.sums);
    if (b.out) (void)hipFree(b.out);
    int rc = OFDMTX_OK;
    if (e != hipSuccess) {
        ofdmtx_g_last_hip = (int)e;
        rc = OFDMTX_EHIP;
    } else {
        rc = plan_build_verdict(h, totals);
    }
    if (rc) {
        ofdmtx_plan_destroy(pl);
        return rc;
    }
    finish_device_plan(pl, n_pkts, d_payload_off != nullptr, b, h, totals);
    *out = pl;
    return OFDMTX_OK;
}

int ofdmtx_plan_destroy(ofdmtx_plan *pl)
{
    if (!pl) return OFDMTX_EINVAL;
    (void)hipSetDevice(pl->device);
    if (pl->busy_ev) {
        if (pl->launched) (void)hipEventSynchronize(pl->busy_ev);   // the last launch still reads the tables
        (void)hipEventDestroy(pl->busy_ev);
    }
    if (pl->d_recs) (void)hipFree(pl->d_recs);
    if (pl->d_row_start) (void)hipFree(pl->d_row_start);
    if (pl->d_plcp_len) (void)hipFree(pl->d_plcp_len);
    if (pl->d_msg_len) (void)hipFree(pl->d_msg_len);
    delete pl;
    return OFDMTX_OK;
}

int ofdmtx_plan_get_sizes(const ofdmtx_plan *pl, ofdmtx_plan_sizes *out)
{
    if (!pl || !out) return OFDMTX_EINVAL;
    *out = pl->sizes;
    return OFDMTX_OK;
}

int ofdmtx_plan_get_layout(const ofdmtx_plan *pl, uint64_t *plcp_off, uint64_t *msg_off,
                           uint64_t *sym_off, uint64_t *sample_off, uint32_t *n_sym,
                           uint32_t *plcp_len, uint32_t *msg_len, uint32_t *pilot_base)
{
    if (!pl) return OFDMTX_EINVAL;
    const PlanParams &pp = pl->pp;
    if (!pl->layout_on_host) {
        // a plan built on the device: fetch its tables once
        ofdmtx_plan *m = const_cast<ofdmtx_plan *>(pl);
        const size_t n = pp.n_pkts;
        m->recs.resize(n);
        m->row_start.resize(n + 1);
        m->plcp_len.resize(n);
        m->msg_len.resize(n);
        HIPCHK(hipSetDevice(pl->device));
        HIPCHK(hipDeviceSynchronize());
        if (n) {
            HIPCHK(hipMemcpy(m->recs.data(), pl->d_recs, n * sizeof(PktRec), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(m->plcp_len.data(), pl->d_plcp_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(m->msg_len.data(), pl->d_msg_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        HIPCHK(hipMemcpy(m->row_start.data(), pl->d_row_start, (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        pl->layout_on_host = true;
    }
    for (uint64_t i = 0; i < pp.n_pkts; i++) {
        if (pp.uniform) {
            if (plcp_off) plcp_off[i] = i * pp.plcp_stride;
            if (msg_off) msg_off[i] = i * pp.msg_stride;
            if (sym_off) sym_off[i] = i * (pp.n_sym + 1);
            if (sample_off) sample_off[i] = i * (uint64_t)pp.rows_per_pkt * SYM_LEN;
            if (n_sym) n_sym[i] = pp.n_sym;
            if (plcp_len) plcp_len[i] = (uint32_t)pl->uni.plcp_len;
            if (msg_len) msg_len[i] = (uint32_t)pl->uni.msg_len;
            if (pilot_base)
                pilot_base[i] = pp.pilot_mode == OFDMTX_PILOT_REF_STREAM
                                    ? (uint32_t)((pp.pilot_start + i * (pp.n_sym + 1)) % 127u)
                                    : pp.pilot_start;
        } else {
            const PktRec &r = pl->recs[i];
            if (plcp_off) plcp_off[i] = r.plcp_off;
            if (msg_off) msg_off[i] = r.msg_off;
            if (sym_off) sym_off[i] = r.sym_off;
            if (sample_off) sample_off[i] = (uint64_t)pl->row_start[i] * SYM_LEN;
            if (n_sym) n_sym[i] = r.n_sym;
            if (plcp_len) plcp_len[i] = pl->plcp_len[i];
            if (msg_len) msg_len[i] = pl->msg_len[i];
            if (pilot_base) pilot_base[i] = r.pilot_base;
        }
    }
    return OFDMTX_OK;
}

}  // extern "C" (interrupted for two helpers shared with the other translation units)

int ofdmtx_plan_claim(const ofdmtx_plan *pl, hipStream_t st)
{
    if (pl->pp.uniform || !pl->launched || !pl->busy_ev || st == pl->last_stream) return OFDMTX_OK;
    // the previous launch still draws packets from the plan's counters: this one starts behind it
    HIPCHK(hipStreamWaitEvent(st, pl->busy_ev, 0));
    return OFDMTX_OK;
}

int ofdmtx_plan_mark_launched(const ofdmtx_plan *pl, hipStream_t st)
{
    if (pl->pp.uniform || !pl->busy_ev) return OFDMTX_OK;   // (no event: the transient plan of the allocation-free call)
    HIPCHK(hipEventRecord(pl->busy_ev, st));
    pl->last_stream = st;
    pl->launched = true;
    return OFDMTX_OK;
}

extern "C" {

static int grid_for(const ofdmtx_ctx *c, uint64_t work_items, int items_per_block, int blocks_per_cu)
{
    uint64_t need = (work_items + items_per_block - 1) / items_per_block;
    uint64_t cap = (uint64_t)c->n_cu * blocks_per_cu;
    uint64_t gsz = need < cap ? need : cap;
    return (int)(gsz ? gsz : 1);
}

int ofdmtx_encode_bits(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const uint8_t *d_payloads,
                       uint8_t *d_plcp, uint8_t *d_scrambled, uint8_t *d_coded,
                       uint8_t *d_interleaved, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    if (pl->pp.n_pkts == 0) return OFDMTX_OK;              // nothing to do: buffers may be NULL
    if (!d_interleaved) return OFDMTX_EINVAL;
    if (!d_payloads && pl->sizes.payload_bytes) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const int grid = grid_for(ctx, pl->pp.n_pkts, ENC_WAVES, 8);
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> guard(pl->launch_lock);
    if (int rc = ofdmtx_plan_claim(pl, st)) return rc;
    const PlanParams &pp = pl->pp;
    const DevTables tab = ctx->tables();
    const uint32_t a_words = pl->tx_u_words > pl->tx_m_words ? pl->tx_u_words : pl->tx_m_words;
    const uint32_t u_words = pl->tx_u_words;
    const size_t lds = (size_t)ENC_WAVES * (a_words + u_words) * sizeof(uint32_t);
#define LAUNCH_ENC(U_, R_, ...)                                                                      \
    hipLaunchKernelGGL((k_encode_bits<U_, R_, ##__VA_ARGS__>), dim3(grid), dim3(64 * ENC_WAVES), lds, st, pp, tab,  \
                       d_payloads, d_plcp, d_scrambled, d_coded, d_interleaved, a_words, u_words)
    if (pp.flags & OFDMTX_F_RAW_PSDU) {
        if (pp.uniform) LAUNCH_ENC(true, 0, true); else LAUNCH_ENC(false, 0, true);
    } else if (!pp.uniform) {
        LAUNCH_ENC(false, 0);
    } else {
        switch (pp.regime) {
        case 1: LAUNCH_ENC(true, 1); break;
        case 2: LAUNCH_ENC(true, 2); break;
        case 3: LAUNCH_ENC(true, 3); break;
        case 4: LAUNCH_ENC(true, 4); break;
        case 5: LAUNCH_ENC(true, 5); break;
        case 6: LAUNCH_ENC(true, 6); break;
        case 7: LAUNCH_ENC(true, 7); break;
        default: LAUNCH_ENC(true, 8); break;
        }
    }
#undef LAUNCH_ENC
    HIPCHK(hipGetLastError());
    return ofdmtx_plan_mark_launched(pl, st);
}

}  // extern "C"

// Launch of the wave-per-packet kernel for either input (payloads or mapper messages).
template <bool FROM_MSG>
static int launch_tx(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const uint8_t *d_in, float2 *mapped, float2 *pilot_tap,
                     float2 *cmap_tap, float2 *iq, hipStream_t st)
{
    std::lock_guard<std::mutex> guard(pl->launch_lock);
    if (int rc = ofdmtx_plan_claim(pl, st)) return rc;
    // one chunk: the scrambled / message words share the staging area; several chunks: they need
    // a region of their own
    const bool one_chunk = pl->max_sym_rows <= 65u;
    const uint32_t a_words = pl->tx_a_words;
    const uint32_t own_words = FROM_MSG ? pl->tx_m_words : pl->tx_u_words;
    // in one chunk the staging area must still hold them
    const uint32_t b_words = one_chunk && own_words > (uint32_t)TX_B_WORDS ? ((own_words + 3u) & ~3u) : (uint32_t)TX_B_WORDS;
    const uint32_t u_off = one_chunk ? a_words : a_words + b_words;
    const uint32_t wave_words = a_words + b_words + (one_chunk ? 0u : own_words);
    const PlanParams &pp = pl->pp;
    const DevTables tab = ctx->tables();
    // Waves per block: 8 when two such blocks fit the CU's LDS, otherwise as many (up to 16) as
    // one block can hold.  A workgroup may use more than the default 64 KiB only after asking.
#define LAUNCH_TX(U_, R_, ...)                                                                       \
    do {                                                                                             \
        const void *fn = reinterpret_cast<const void *>(&k_tx<U_, R_, FROM_MSG, ##__VA_ARGS__>);     \
        hipFuncAttributes fa;                                                                        \
        HIPCHK(hipFuncGetAttributes(&fa, fn));                                                       \
        const size_t wave_bytes = (size_t)wave_words * sizeof(uint32_t);                             \
        const size_t room = LDS_PER_CU > fa.sharedSizeBytes ? LDS_PER_CU - fa.sharedSizeBytes : 0; \
        int waves = TX_WAVES, per_cu = 2;                                                            \
        if (2 * (TX_WAVES * wave_bytes + fa.sharedSizeBytes) > LDS_PER_CU) {                    \
            waves = (int)(room / wave_bytes);                                                        \
            waves = waves > TX_MAX_WAVES ? TX_MAX_WAVES : waves;                                     \
            per_cu = 1;                                                                              \
        }                                                                                            \
        if (waves < 1) return OFDMTX_ETOOBIG;                                                        \
        const size_t lds = (size_t)waves * wave_bytes;                                               \
        const int grid = grid_for(ctx, pl->pp.n_pkts, waves, per_cu);                                \
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
        hipLaunchKernelGGL((k_tx<U_, R_, FROM_MSG, ##__VA_ARGS__>), dim3(grid), dim3(64 * waves), lds, st, pp, \
                           tab, d_in, mapped, pilot_tap, cmap_tap, iq, a_words, u_off, wave_words);  \
    } while (0)
    const bool pnc = (pp.flags & (OFDMTX_F_PNC_A | OFDMTX_F_PNC_B)) || pp.pilot_mode >= OFDMTX_PILOT_PNC_A;
    if (pnc) {
        if (pp.uniform) LAUNCH_TX(true, 0, true); else LAUNCH_TX(false, 0, true);
    } else if (!FROM_MSG && (pp.flags & OFDMTX_F_RAW_PSDU)) {
        // (the message input does not care where the bits came from: the standard instantiations below)
        if constexpr (!FROM_MSG) {
            if (pp.uniform) LAUNCH_TX(true, 0, false, true); else LAUNCH_TX(false, 0, false, true);
        }
    } else if (!pp.uniform) {
        LAUNCH_TX(false, 0);
    } else if (FROM_MSG) {
        // the message path does not depend on the code rate: one instantiation per constellation
        switch (regime_info((int)pp.regime).n_bpsc) {
        case 1: LAUNCH_TX(true, 1); break;
        case 2: LAUNCH_TX(true, 3); break;
        case 4: LAUNCH_TX(true, 5); break;
        default: LAUNCH_TX(true, 8); break;
        }
    } else {
        switch (pp.regime) {
        case 1: LAUNCH_TX(true, 1); break;
        case 2: LAUNCH_TX(true, 2); break;
        case 3: LAUNCH_TX(true, 3); break;
        case 4: LAUNCH_TX(true, 4); break;
        case 5: LAUNCH_TX(true, 5); break;
        case 6: LAUNCH_TX(true, 6); break;
        case 7: LAUNCH_TX(true, 7); break;
        default: LAUNCH_TX(true, 8); break;
        }
    }
#undef LAUNCH_TX
    HIPCHK(hipGetLastError());
    if (ctx->gap_nonzero && (pp.flags & OFDMTX_F_ZEROGAP)) {
        // a gap row other than zeros (ofdmtx_set_gap): written over the zeros k_tx has just stored
        hipLaunchKernelGGL(k_gap_rows, dim3(grid_for(ctx, pp.n_pkts, 1, 8)), dim3(256), 0, st, pp, ctx->d_gap, iq);
        HIPCHK(hipGetLastError());
    }
    return ofdmtx_plan_mark_launched(pl, st);
}

extern "C" {

int ofdmtx_modulate(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const uint8_t *d_msg, float *d_mapped,
                    float *d_iq, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    if (pl->pp.n_pkts == 0) return OFDMTX_OK;
    if (!d_msg || !d_iq) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    return launch_tx<true>(ctx, pl, d_msg, reinterpret_cast<float2 *>(d_mapped), nullptr, nullptr,
                           reinterpret_cast<float2 *>(d_iq), (hipStream_t)stream);
}

int ofdmtx_modulate_taps(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const uint8_t *d_msg, float *d_mapped,
                         float *d_pilot, float *d_cmap, float *d_iq, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    if (pl->pp.n_pkts == 0) return OFDMTX_OK;
    if (!d_msg || !d_iq) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    return launch_tx<true>(ctx, pl, d_msg, reinterpret_cast<float2 *>(d_mapped), reinterpret_cast<float2 *>(d_pilot),
                           reinterpret_cast<float2 *>(d_cmap), reinterpret_cast<float2 *>(d_iq), (hipStream_t)stream);
}

int ofdmtx_tx(ofdmtx_ctx *ctx, const ofdmtx_plan *pl, const uint8_t *d_payloads, uint8_t *d_work,
              float *d_iq, void *stream)
{
    (void)d_work;   // the fused kernel keeps the mapper messages in LDS
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    if (pl->pp.n_pkts == 0) return OFDMTX_OK;
    if (!d_iq) return OFDMTX_EINVAL;
    if (!d_payloads && pl->sizes.payload_bytes) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    return launch_tx<false>(ctx, pl, d_payloads, nullptr, nullptr, nullptr, reinterpret_cast<float2 *>(d_iq),
                            (hipStream_t)stream);
}

// The plan of ofdmtx_plan_ragged_device built into the context's static work area, then the fused
// kernel on it -- with no allocation on the way: the plan object lives on this stack frame (its
// vectors stay empty), the totals come back through the context's page-locked words.
int ofdmtx_tx_ragged_static(ofdmtx_ctx *ctx, uint64_t n_pkts, const uint8_t *d_regimes,
                            const uint32_t *d_payload_len, const uint64_t *d_payload_off,
                            double symbol_time, int pilot_mode, uint32_t pilot_start, unsigned flags,
                            float amp, const uint8_t *d_payloads, float *d_iq, uint64_t iq_capacity,
                            uint64_t *n_samples, void *stream)
{
    if (n_samples) *n_samples = 0;
    if (!ctx || !ctx->d_static || !ctx->h_pinned) return OFDMTX_EINVAL;
    if (n_pkts && (!d_regimes || !d_payload_len)) return OFDMTX_EINVAL;
    if (!plan_modes_ok(pilot_mode, flags)) return OFDMTX_EINVAL;
    if (n_pkts > OFDMTX_STATIC_MAX_PKTS) return OFDMTX_ETOOBIG;          // the static tables hold no more
    if (n_pkts == 0) return OFDMTX_OK;
    if (ctx->device < 0 || ctx->device >= MAX_DEVICES || !g_static_guard[ctx->device].done) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    // one call at a time per device on the host; on the device, behind the previous call's kernels
    StaticGuard &guard = g_static_guard[ctx->device];
    std::lock_guard<std::mutex> host_order(guard.lock);
    if (guard.launched && st != guard.last) HIPCHK(hipStreamWaitEvent(st, guard.done, 0));
    struct MarkDone {                                                    // whatever this call enqueued is what the next one waits for
        StaticGuard &g;
        hipStream_t st;
        ~MarkDone()
        {
            if (hipEventRecord(g.done, st) == hipSuccess) {
                g.last = st;
                g.launched = true;
            }
        }
    } mark_done{guard, st};
    uint8_t *base = ctx->d_static;                                       // never the array's name: see ofdmtx_create
    PlanBuildBufs b;
    b.recs = reinterpret_cast<PktRec *>(base + StaticLayout::recs);
    b.row_start = reinterpret_cast<uint32_t *>(base + StaticLayout::row_start);
    b.counters = reinterpret_cast<uint32_t *>(base + StaticLayout::counters);
    b.plcp_len = reinterpret_cast<uint32_t *>(base + StaticLayout::plcp_len);
    b.msg_len = reinterpret_cast<uint32_t *>(base + StaticLayout::msg_len);
    b.col = reinterpret_cast<unsigned long long *>(base + StaticLayout::col);
    b.sums = reinterpret_cast<unsigned long long *>(base + StaticLayout::sums);
    b.out = reinterpret_cast<PlanDevOut *>(base + StaticLayout::out);
    // page-locked landing area: [PlanDevOut][PLAN_COLS totals]
    static_assert(sizeof(PlanDevOut) + PLAN_COLS * sizeof(unsigned long long) <= 256, "landing area too small");
    PlanDevOut *h = reinterpret_cast<PlanDevOut *>(ctx->h_pinned);
    unsigned long long *totals = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(ctx->h_pinned) + 64);
    static_assert(sizeof(PlanDevOut) <= 64, "landing area layout");
    // (the memsets inside clear what must start at zero: counters, error word, scan columns -- the
    //  area keeps the previous call's contents otherwise)
    hipError_t e = enqueue_plan_build((uint32_t)n_pkts, d_regimes, d_payload_len, d_payload_off, symbol_time, pilot_mode,
                                      pilot_start % 127u, flags, b, h, totals, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        ofdmtx_g_last_hip = (int)e;
        return OFDMTX_EHIP;
    }
    if (int rc = plan_build_verdict(*h, totals)) return rc;
    ofdmtx_plan pl;                                                      // transient: no event, no device memory of its own
    plan_common(&pl, ctx, n_pkts, pilot_mode, pilot_start, flags, amp);
    pl.pp.uniform = 0;
    finish_device_plan(&pl, n_pkts, d_payload_off != nullptr, b, *h, totals);
    if (n_samples) *n_samples = pl.sizes.n_samples;
    if (pl.sizes.n_samples > iq_capacity) return OFDMTX_EINVAL;
    if (!d_iq || (!d_payloads && pl.sizes.payload_bytes)) return OFDMTX_EINVAL;
    return launch_tx<false>(ctx, &pl, d_payloads, nullptr, nullptr, nullptr, reinterpret_cast<float2 *>(d_iq), st);
}

}  // extern "C"
