/*
 * libofdmtx -- batched 802.11a/g/p OFDM transmit chain on AMD MI355X (gfx950).
 *
 * C ABI of the HIP library.  Every pointer named d_* is a DEVICE pointer (HBM of the
 * context's device); everything else is host memory.  No call allocates or frees caller
 * buffers.  All functions return 0 on success or a negative OFDMTX_E* code; no exception
 * crosses this boundary.  A context is bound to one device and is thread-compatible (use
 * one per device / per host thread).  Kernels are enqueued on the stream passed in
 * (a hipStream_t passed as void*; NULL = the default stream) and are NOT synchronised.
 * Payload bytes are fetched as aligned 32-bit words: the kernels may read up to 3 bytes on either
 * side of a payload (never written, never used) -- keep payload buffers inside one allocation.
 * With an empty plan (0 packets / symbols) the kernel entry points return 0 and touch no buffer.
 *
 * What each entry point stands in for in the reference (paths relative to the reference
 * root, ftw_80211_ofdm_tx/src/...):
 *   - the SWIG module generated from ftw/ftw.i:42-218 (python "gnuradio.ftw"), i.e. the C++
 *     blocks ofdm_mapper / ofdm_pilot_cc / ofdm_cmap_cc / ofdm_preamble / zerogap and the
 *     free function ftw_crc32, plus the GNU Radio blocks wired between them in
 *     examples/ftw_ofdm.py:98-179 (fft_vcc, ofdm_cyclic_prefixer, multiply_const_cc);
 *   - the per-packet Python of examples/ftw_packet_utils.py:25-372 that ofdm_mod.send_pkt
 *     (examples/ftw_ofdm.py:191-219) runs before handing a message to the mapper.
 */
#ifndef OFDMTX_H
#define OFDMTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFDMTX_OK 0
#define OFDMTX_EINVAL (-1)    /* bad argument (NULL, out of range)                          */
#define OFDMTX_EREGIME (-2)   /* regime not in 1..8 (reference: UnboundLocalError,
                                 examples/ftw_packet_utils.py:39-78 has no else branch)    */
#define OFDMTX_ELENGTH (-3)   /* MPDU+FCS > 3168 bytes (reference: ValueError,
                                 examples/ftw_packet_utils.py:119-121)                     */
#define OFDMTX_EHIP (-4)      /* a HIP runtime call failed; see ofdmtx_last_hip_error()     */
#define OFDMTX_ENOMEM (-5)
#define OFDMTX_ESHAPE (-6)    /* block parameter the reference's blocks reject with
                                 std::invalid_argument (ftw/ftw_ofdm_pilot_cc.cc:110-111,
                                 ftw/ftw_ofdm_cmap_cc.cc:97-99, ftw/ftw_ofdm_mapper.cc:61-62) */
#define OFDMTX_ETOOBIG (-7)   /* batch exceeds the 32-bit row index of the kernels, or the
                                 fixed capacity of an allocation-free entry point           */
#define OFDMTX_EBUSY (-8)     /* not returned any more (launches that share state are ordered
                                 on the device instead); the value stays reserved             */
#define OFDMTX_ECOMM (-9)     /* RCCL missing or an RCCL call failed; ofdmtx_last_comm_error() */

/* plan flags */
#define OFDMTX_F_PREAMBLE 1u  /* prepend the 4x80-sample preamble (ftw/ftw_ofdm_preamble.cc:79-83) */
#define OFDMTX_F_ZEROGAP 2u   /* append 13x80 zero samples (ftw/ftw_zerogap.cc:88-92)              */
#define OFDMTX_F_SWAPIQ 4u    /* swap re/im (examples/ftw_ofdm.py:155-163)                         */
/* PNC roles (examples/ftw_ofdm.py:133-138, ftw/ftw_pnc_ofdm_preamble.cc:80-103): with
 * OFDMTX_F_PREAMBLE, SIX preamble rows -- the two short-training rows, then the two long-training
 * rows followed (role A, user 1) or preceded (role B, user 2) by two rows of zeros.  At most one
 * of the two; the zero gap is the same 13 rows (ftw/ftw_pnc_zerogap.cc:88-92). */
#define OFDMTX_F_PNC_A 8u
#define OFDMTX_F_PNC_B 16u
/* The plan's "payload" of every packet is the whole PSDU -- MPDU and FCS as they go on air, bytes in
 * natural order (the first byte's least significant bit is sent first) -- instead of the body behind
 * the MAC header of examples/ftw_packet_utils.py:28-36: no header is built and no CRC computed, and
 * payload_len (0..3168) is the SIGNAL field's LENGTH.  Everything else (SIGNAL, SERVICE, scrambler,
 * tail, pad, encoder, ...) is the same code.  The reference's Python cannot send such a frame (its
 * header and FCS are fixed, :28-36,370-372); its MATLAB model can (matlab/encode.m:55-66 takes any MPDU,
 * :221-223 forces the FCS printed in the standard), and this flag is what lets the Annex G.8 frame of
 * matlab/802.11-2007-Annex-G.8.txt run through ofdmtx_tx / ofdmtx_encode_bits.  Not with a PNC role or
 * PNC pilots (OFDMTX_EINVAL).  ofdmtx_rx_decode returns the whole PSDU for such a plan. */
#define OFDMTX_F_RAW_PSDU 32u

/* pilot polarity index of symbol s of packet q */
#define OFDMTX_PILOT_REF_STREAM 0 /* running counter over the whole stream, as the reference's
                                     process-global counter does (ftw/ftw_ofdm_pilot_cc.cc:82,159):
                                     pilot_start + sum_{r<q}(N_sym_r+1) + s                        */
#define OFDMTX_PILOT_PER_PACKET 1 /* restart at pilot_start for every packet (the standard)        */
/* the PNC pilot block (ftw/ftw_pnc_ofdm_pilot_cc.cc:125-131,146-178): constant pilots at
 * -21, -7, +7, +21, no polarity sequence, no sign change at +21; pilot_start is ignored */
#define OFDMTX_PILOT_PNC_A 2      /* user 1: +1  0 +1  0                                           */
#define OFDMTX_PILOT_PNC_B 3      /* user 2:  0 +1  0 +1                                           */
#define OFDMTX_PILOT_PNC_ALL 4    /* user 0: +1 +1 +1 +1                                           */

typedef struct ofdmtx_ctx ofdmtx_ctx;
typedef struct ofdmtx_plan ofdmtx_plan;

/* Geometry of one packet -- the numbers get_info/ftw_make derive
 * (examples/ftw_packet_utils.py:81-97,160-171,193). */
typedef struct ofdmtx_pkt_info {
    int32_t n_cbps, n_bpsc, n_dbps;
    int32_t n_sym;        /* DATA symbols                                       */
    int32_t n_data, n_pad;
    int32_t txtime;       /* microseconds, goes into the MAC duration field     */
    int32_t length;       /* MPDU + FCS bytes = SIGNAL LENGTH                   */
    int32_t plcp_len;     /* bytes entering the scrambler                       */
    int32_t msg_len;      /* bytes of the mapper message = 6 + N_sym*N_cbps/8   */
    int32_t n_rows;       /* 80-sample rows with the plan flags applied         */
} ofdmtx_pkt_info;

/* Totals and strides of a plan.  Tap/message buffers use 16-byte aligned per-packet
 * strides chosen by the library; the IQ buffer is densely packed (rows of 80 complex64). */
typedef struct ofdmtx_plan_sizes {
    uint64_t n_pkts;
    uint64_t payload_bytes;   /* extent of the payload buffer the plan reads                 */
    uint64_t plcp_bytes;      /* size of a plcp / scrambled tap buffer                       */
    uint64_t msg_bytes;       /* size of a coded / interleaved (mapper message) buffer       */
    uint64_t n_symbols;       /* sum of (1 + N_sym): rows of a `mapped` tap (48 complex64)   */
    uint64_t n_rows;          /* 80-sample rows of the IQ buffer                             */
    uint64_t n_samples;       /* = 80 * n_rows complex64                                     */
} ofdmtx_plan_sizes;

const char *ofdmtx_strerror(int code);
int ofdmtx_last_hip_error(void);   /* hipError_t of the last failing HIP call on this thread */

/* ---- context ---------------------------------------------------------------------- */
int ofdmtx_create(int device, ofdmtx_ctx **out);
int ofdmtx_destroy(ofdmtx_ctx *ctx);

/* Replace the constellation used for DATA symbols with n_bpsc bits per carrier
 * (n = 2^n_bpsc complex points, re/im interleaved, index = bits MSB first) -- the
 * `constellation` argument of ftw.ofdm_mapper (ftw/ftw.i:157-163).  The SIGNAL symbol always
 * uses the fixed BPSK {-1,+1} (ftw/ftw_ofdm_mapper.cc:179,238-240). */
int ofdmtx_set_constellation(ofdmtx_ctx *ctx, int n_bpsc, const float *re_im, int n);

/* Replace the training table the preamble block emits in front of every packet: the `preamble`
 * argument of ftw.ofdm_preamble / ftw.pnc_ofdm_preamble (ftw/ftw.i:78-106), of which the blocks
 * copy the first four rows of symbol_length = 80 samples (ftw/ftw_ofdm_preamble.cc:79-83; the PNC
 * block takes its rows from the same four, ftw/ftw_pnc_ofdm_preamble.cc:81-103).  re_im: n_samples
 * complex values (re,im float pairs), host memory; n_samples >= 320 (the reference reads past a
 * shorter table: OFDMTX_EINVAL here).  The table is scaled by the plan's amplitude like every other
 * sample.  Takes effect for launches enqueued after the call returns (it synchronises the device). */
int ofdmtx_set_preamble(ofdmtx_ctx *ctx, const float *re_im, int n_samples);
/* Replace the row the zero-gap block appends 13 times behind every packet: the `zerogap` argument
 * of ftw.zerogap / ftw.pnc_zerogap (ftw/ftw.i:110-138), one row of symbol_length = 80 samples
 * (ftw/ftw_zerogap.cc:88-92).  The reference passes 80 zeros (examples/ftw_packet_utils.py:510).  A
 * row with non-zero samples is written by a second small kernel behind the fused one. */
int ofdmtx_set_gap(ofdmtx_ctx *ctx, const float *re_im, int n_samples);

/* ---- host helpers ----------------------------------------------------------------- */
/* The static tables of examples/ftw_packet_utils.py:497-508 as the device uses them (float32,
 * re/im interleaved): gray_const_bpsk / _qpsk / _qam16 / _qam64 and the 320-sample fft_preamble.
 * Returns the number of floats written (4, 8, 32, 128, 640) or a negative code. */
#define OFDMTX_TABLE_BPSK 1
#define OFDMTX_TABLE_QPSK 2
#define OFDMTX_TABLE_QAM16 4
#define OFDMTX_TABLE_QAM64 6
#define OFDMTX_TABLE_PREAMBLE 100
int ofdmtx_get_table(int which, float *out, int n_floats);
/* ftw.ftw_crc32 (ftw/ftw_crc32.cc:72-82): bit-reversed reflected CRC-32. */
uint32_t ofdmtx_crc32(const uint8_t *buf, size_t len);
/* ftw.ftw_update_crc32 (ftw/ftw.i:215-218, ftw/ftw_crc32.cc:33-64): the CRC register after `len`
 * bytes from the all-ones start value, bit-reversed, without the final inversion.  `crc` is
 * accepted and ignored, as in the reference (overwritten at ftw/ftw_crc32.cc:53). */
uint32_t ofdmtx_update_crc32(uint32_t crc, const uint8_t *buf, size_t len);
/* get_info (examples/ftw_packet_utils.py:25-105) for payload_len bytes behind the 24-byte
 * MAC header.  flags as for plans (they only affect n_rows).  Returns OFDMTX_ELENGTH when the
 * MPDU exceeds what ftw_make accepts (examples/ftw_packet_utils.py:119-121); *out is filled in
 * that case too, as get_info itself has no limit. */
int ofdmtx_packet_info(int regime, int payload_len, double symbol_time, unsigned flags,
                       ofdmtx_pkt_info *out);

/* ---- plans -------------------------------------------------------------------------
 * A plan fixes the batch geometry: per packet regime and payload length, where each
 * packet's bytes and samples live, pilot indexing, output framing and amplitude
 * (tx_amplitude of examples/ftw_ofdm_tx_uhd.py:82-86; 1/sqrt(64) of examples/ftw_ofdm.py:126
 * is always applied to symbol rows). */
int ofdmtx_plan_uniform(ofdmtx_ctx *ctx, uint64_t n_pkts, int regime, uint32_t payload_len,
                        uint64_t payload_stride, double symbol_time, int pilot_mode,
                        uint32_t pilot_start, unsigned flags, float amp, ofdmtx_plan **out);
/* Ragged / mixed-rate batch.  regimes[n], payload_len[n], payload_off[n] (byte offset of each
 * payload in the payload buffer; NULL = densely packed in order) are host arrays.
 * A ragged plan owns device-side work counters from which the waves of ofdmtx_tx /
 * ofdmtx_modulate / ofdmtx_encode_bits draw packets (each kernel leaves them at zero when it
 * ends): the same ragged plan cannot run in two launches at once.  Launches on ONE stream are
 * ordered anyway; a launch on a different stream than the plan's previous launch is made to start
 * behind it (an event recorded behind every launch, waited for on the device) -- for launches that
 * really overlap use one plan per stream.  Uniform plans have no such limit.
 * Lifetime: a plan must stay alive until the launches that use it have finished.
 * ofdmtx_plan_destroy waits for the last launch of a ragged plan before it frees the plan's device
 * tables; uniform plans own no device memory. */
int ofdmtx_plan_ragged(ofdmtx_ctx *ctx, uint64_t n_pkts, const uint8_t *regimes,
                       const uint32_t *payload_len, const uint64_t *payload_off,
                       double symbol_time, int pilot_mode, uint32_t pilot_start, unsigned flags,
                       float amp, ofdmtx_plan **out);
/* The same plan from DEVICE arrays (d_payload_off may be NULL), built by kernels: geometry per
 * packet, prefix sums for every offset, records -- no per-packet host work (the host loop of
 * ofdmtx_plan_ragged costs about 18 ns per packet).  Enqueues on `stream` and waits for it (the
 * totals come back to the host).  Lengths are uint32 as above; an invalid regime or length in any
 * packet fails the whole call (OFDMTX_EREGIME / OFDMTX_ELENGTH).  No counterpart in the reference,
 * whose flowgraph is sized for one packet length (examples/ftw_ofdm.py:129-130). */
int ofdmtx_plan_ragged_device(ofdmtx_ctx *ctx, uint64_t n_pkts, const uint8_t *d_regimes,
                              const uint32_t *d_payload_len, const uint64_t *d_payload_off,
                              double symbol_time, int pilot_mode, uint32_t pilot_start,
                              unsigned flags, float amp, void *stream, ofdmtx_plan **out);
int ofdmtx_plan_destroy(ofdmtx_plan *plan);
int ofdmtx_plan_get_sizes(const ofdmtx_plan *plan, ofdmtx_plan_sizes *out);
/* (on a plan from ofdmtx_plan_ragged_device the first call synchronises the device and copies the
 * tables to the host: not reentrant for such a plan)
 * Per-packet layout, host arrays of n_pkts entries each (any may be NULL):
 * plcp_off / msg_off byte offsets into tap / message buffers, sym_off index of the packet's
 * SIGNAL symbol in a `mapped` tap, sample_off index of the packet's first complex sample,
 * n_sym DATA symbols, pilot_base polarity index of the packet's SIGNAL symbol. */
int ofdmtx_plan_get_layout(const ofdmtx_plan *plan, uint64_t *plcp_off, uint64_t *msg_off,
                           uint64_t *sym_off, uint64_t *sample_off, uint32_t *n_sym,
                           uint32_t *plcp_len, uint32_t *msg_len, uint32_t *pilot_base);

/* ---- kernels ---------------------------------------------------------------------- */
/* Bit-level chain of ofdm_mod.send_pkt for every packet of the plan
 * (examples/ftw_ofdm.py:201-217): ftw_make -> scrambler -> conv_encoder -> interleaver.
 * d_interleaved receives the mapper messages; the other outputs are optional taps
 * (NULL = not written): d_plcp = ftw_make output, d_scrambled, d_coded. */
int ofdmtx_encode_bits(ofdmtx_ctx *ctx, const ofdmtx_plan *plan, const uint8_t *d_payloads,
                       uint8_t *d_plcp, uint8_t *d_scrambled, uint8_t *d_coded,
                       uint8_t *d_interleaved, void *stream);

/* Symbol-level flowgraph mapper -> pilot -> cmap -> IFFT -> CP -> scale -> preamble ->
 * zerogap -> amplitude (examples/ftw_ofdm.py:175-179, examples/ftw_ofdm_tx_uhd.py:86), fused.
 * d_msg: mapper messages in plan layout.  d_mapped (optional tap): n_symbols x 48 complex64,
 * the mapper block's output.  d_iq: n_samples complex64 (re,im float32 pairs). */
int ofdmtx_modulate(ofdmtx_ctx *ctx, const ofdmtx_plan *plan, const uint8_t *d_msg,
                    float *d_mapped, float *d_iq, void *stream);

/* ofdmtx_modulate with the taps of the two blocks behind the mapper (the `--log` files of
 * examples/ftw_ofdm.py:181-184): d_pilot = n_symbols x 53 complex64, the output of ofdm_pilot_cc
 * ("ofdm_pilot.dat", ftw/ftw_ofdm_pilot_cc.cc:119-155); d_cmap = n_symbols x 64 complex64, the
 * output of ofdm_cmap_cc = the input of the inverse FFT ("ofdm_cmap.dat",
 * ftw/ftw_ofdm_cmap_cc.cc:102-107).  Any tap may be NULL.  The values are the ones the fused kernel
 * feeds to its transform (read from the same registers). */
int ofdmtx_modulate_taps(ofdmtx_ctx *ctx, const ofdmtx_plan *plan, const uint8_t *d_msg,
                         float *d_mapped, float *d_pilot, float *d_cmap, float *d_iq, void *stream);
/* The taps behind the transform (examples/ftw_ofdm.py:185-188), cut out of the IQ stream d_iq that
 * ofdmtx_modulate / ofdmtx_tx wrote for `plan`.  The plan's amplitude must be 1 and OFDMTX_F_SWAPIQ
 * unset (the reference taps sit in front of tx_amplitude and of the I/Q exchange): then every tap is
 * the kernel's own output, at most rescaled by the exact factor 8 = sqrt(64):
 *   d_ifft     n_symbols x 64   gr.fft_vcc output            ("ofdm_ifft.dat")     = 8 * row[16..79]
 *   d_cp       n_symbols x 80   ofdm_cyclic_prefixer output  ("ofdm_cp_adder.dat") = 8 * row
 *   d_scale    n_symbols x 80   multiply_const_cc(1/8)       ("ofdm_scale.dat")    = row
 *   d_preamble per packet its preamble rows + symbol rows    ("ofdm_preamble.dat"); n_symbols +
 *              n_pkts * (4, or 6 with a PNC role) rows of 80 (no rows added without OFDMTX_F_PREAMBLE)
 * Any output may be NULL.  ("ofdm_zerogap.dat" is d_iq itself.) */
int ofdmtx_symbol_taps(ofdmtx_ctx *ctx, const ofdmtx_plan *plan, const float *d_iq, float *d_ifft,
                       float *d_cp, float *d_scale, float *d_preamble, void *stream);

/* payload -> IQ for the whole batch (send_pkt + flowgraph) in one fused kernel.  d_work is
 * not used any more (the mapper messages stay in LDS) and may be NULL. */
int ofdmtx_tx(ofdmtx_ctx *ctx, const ofdmtx_plan *plan, const uint8_t *d_payloads,
              uint8_t *d_work, float *d_iq, void *stream);

/* ofdmtx_plan_ragged_device + ofdmtx_tx in one call that allocates nothing (no hipMalloc / hipFree,
 * no host heap): for callers that cannot allocate -- tight serving loops, harnesses that forbid
 * it, code that must not disturb the device's memory pools.  It has no counterpart in the reference.
 * The plan tables (per-packet records, row table, scan columns, work counters) live in a static
 * device array of the library, one copy per GPU, held until the process ends; hence the fixed
 * capacity n_pkts <= OFDMTX_STATIC_MAX_PKTS (OFDMTX_ETOOBIG beyond it).  Inputs as for
 * ofdmtx_plan_ragged_device (device arrays; d_payload_off may be NULL) and ofdmtx_tx.
 * iq_capacity = complex samples d_iq can hold: if the batch needs more, nothing is written and
 * OFDMTX_EINVAL is returned with *n_samples set to the need.  *n_samples (may be NULL) receives the
 * samples written.  The call waits on `stream` once (the plan totals come back to the host) and
 * returns with the transmit kernel enqueued.  Every call on a GPU uses that GPU's one array, so the
 * library orders them: the host part runs under a lock per GPU, and a call on another stream than
 * the previous call's is made to start behind that call's kernels (an event wait on the device,
 * not a refusal).  Calls from other PROCESSES on the same GPU have arrays of their own. */
#define OFDMTX_STATIC_MAX_PKTS 65536u
int ofdmtx_tx_ragged_static(ofdmtx_ctx *ctx, uint64_t n_pkts, const uint8_t *d_regimes,
                            const uint32_t *d_payload_len, const uint64_t *d_payload_off,
                            double symbol_time, int pilot_mode, uint32_t pilot_start, unsigned flags,
                            float amp, const uint8_t *d_payloads, float *d_iq, uint64_t iq_capacity,
                            uint64_t *n_samples, void *stream);

/* ---- single-message stage operators ---------------------------------------------------
 * The per-packet functions of examples/ftw_packet_utils.py, each on ONE byte string in device
 * memory (MSB-first bits), for callers that keep the reference's stage-by-stage call sequence
 * (examples/ftw_ofdm.py:207-214).  Batches belong to ofdmtx_encode_bits / ofdmtx_tx. */
/* scrambler(pkt, Length_data), examples/ftw_packet_utils.py:344-367.  d_out: n_bytes.
 * OFDMTX_EINVAL if the six tail bits at 40 + 8*length_data lie outside the string (the
 * reference raises IndexError). */
int ofdmtx_scramble(ofdmtx_ctx *ctx, const uint8_t *d_in, uint32_t n_bytes, uint32_t length_data,
                    uint8_t *d_out, void *stream);
/* conv_encoder(pkt, Length, regime, N_cbps, N_bpsc, N_sym, N_rate),
 * examples/ftw_packet_utils.py:280-341.  d_out: 6 + n_sym*N_cbps/8 bytes.  OFDMTX_EINVAL if
 * the input is too short to yield that many punctured bits. */
int ofdmtx_conv_encode(ofdmtx_ctx *ctx, const uint8_t *d_in, uint32_t n_bytes, int regime,
                       uint32_t n_sym, uint8_t *d_out, void *stream);
/* interleaver(payload, regime, N_cbps, N_bpsc), examples/ftw_packet_utils.py:222-277.
 * n_bytes = 6 + k*N_cbps/8 (whole symbols); d_out: n_bytes. */
int ofdmtx_interleave(ofdmtx_ctx *ctx, const uint8_t *d_in, uint32_t n_bytes, int regime,
                      uint8_t *d_out, void *stream);

/* gr.multiply_const_cc(k) on n complex64 samples (the tx_amplitude block of
 * examples/ftw_ofdm_tx_uhd.py:82-86 when samples are replayed from a file, :75-78); d_in may equal
 * d_out; both 16-byte aligned. */
int ofdmtx_scale_cc(ofdmtx_ctx *ctx, const float *d_in, uint64_t n, float k_re, float k_im,
                    float *d_out, void *stream);

/* ftw.repetition (ftw/ftw.i:143-146, ftw/ftw_repetition.cc:54-76): the block copies rows from its
 * input without ever consuming them, i.e. it emits the first packet of its input (period =
 * (N_symbols + 5 + 13) * symbol_length samples) again and again.  d_out[i] = d_in[i % period]
 * for i < n_out (complex64 samples; period and n_out even, buffers 16-byte aligned). */
int ofdmtx_repeat(ofdmtx_ctx *ctx, const float *d_in, uint64_t period, uint64_t n_out, float *d_out,
                  void *stream);

/* ---- loop-back verifier ------------------------------------------------------------------
 * A receiver for the noiseless frames of ofdmtx_tx / ofdmtx_modulate, written from IEEE
 * 802.11-2007 clause 17 (the reference has no receiver for this chain; its rawofdm tests check
 * a loop-back, rawofdm/src/python/qa_ofdm.py:40-67).  Frame geometry comes from the plan; rate
 * and length are also read from the decoded SIGNAL field and compared.
 *
 * ofdmtx_rx_demod: d_iq (the plan's n_samples complex64) -> d_coded: per packet the hard
 *   decisions as coded bits in transmit order, at the plan's message offsets (msg_bytes bytes:
 *   the layout and, for a clean loop-back, the content of the d_coded tap of
 *   ofdmtx_encode_bits); d_status[n_pkts] is SET to the OFDMTX_RX_PILOTS / _NULLS bits.  (A ragged
 *   plan's packets are handed to the waves through a counter of the context's, one of 64 per launch
 *   in turn: more than 64 such launches in flight on one context at once would share one.)
 * ofdmtx_rx_decode: d_coded -> d_payloads (payload bytes at the plan's payload offsets,
 *   payload_bytes bytes); ORs the remaining bits into d_status (zero it first if
 *   ofdmtx_rx_demod did not write it). */
#define OFDMTX_RX_PILOTS 1u   /* a pilot carrier is not what the plan's pilot mode prescribes   */
#define OFDMTX_RX_NULLS 2u    /* energy on DC or a guard carrier                                */
#define OFDMTX_RX_SIGNAL 4u   /* SIGNAL field: RATE differs from the plan, reserved bit, parity or tail */
#define OFDMTX_RX_LENGTH 8u   /* SIGNAL field: LENGTH differs from 24 + payload + 4             */
#define OFDMTX_RX_SERVICE 16u /* SERVICE bits not zero after descrambling                       */
#define OFDMTX_RX_TAIL 32u    /* the six tail bits are not zero                                 */
#define OFDMTX_RX_FCS 64u     /* CRC-32 of the MPDU differs from the received FCS               */
#define OFDMTX_RX_CODE 128u   /* a coded bit contradicts the bits decoded before it             */
int ofdmtx_rx_demod(ofdmtx_ctx *ctx, const ofdmtx_plan *plan, const float *d_iq, uint8_t *d_coded,
                    uint32_t *d_status, void *stream);
int ofdmtx_rx_decode(ofdmtx_ctx *ctx, const ofdmtx_plan *plan, const uint8_t *d_coded,
                     uint8_t *d_payloads, uint32_t *d_status, void *stream);

/* ---- variable-size modulator (rawofdm) --------------------------------------------------
 * Stands in for raw.ofdm_mapper (rawofdm/src/lib/ofdm/raw_ofdm.i:18-20,
 * rawofdm/src/lib/ofdm/raw_ofdm_mapper.cc:22-80) and the GNU Radio blocks wired behind it in
 * rawofdm/src/python/raw_ofdm_mod.py:44-52: digital.ofdm_insert_preamble, gr.fft_vcc(N, False,
 * [], True), digital.ofdm_cyclic_prefixer(N, N + cp), gr.multiply_const_cc(1/sqrt(N)).
 *
 * carrier_map[fft_length]: 0 unused, 1 data, 2 pilot, DC in the middle (padded_carriers of
 * rawofdm/src/python/raw_ofdm_params.py:77-93); any other value -> OFDMTX_ESHAPE (the reference
 * throws std::invalid_argument, raw_ofdm_mapper.cc:37-38).  preambles: n_pre frequency-domain
 * symbols of fft_length complex values (re,im float pairs; padded_preambles of
 * raw_ofdm_params.py:60-75), host memory.  fft_length must be 64, 128, 256, 512 or 1024
 * (OFDMTX_ESHAPE otherwise).  Output rows have fft_length + cp_length samples. */
typedef struct ofdmtx_raw_plan ofdmtx_raw_plan;
typedef struct ofdmtx_raw_sizes {
    uint32_t fft_length, cp_length;
    uint32_t data_tones;      /* carriers marked 1: complex values consumed per data symbol  */
    uint32_t n_pre;           /* preamble symbols emitted in front of every frame            */
    uint64_t n_sym;           /* data symbols the plan reads                                 */
    uint64_t n_frames;
    uint64_t n_rows;          /* output rows = symbols written (preambles + kept data)       */
    uint64_t n_samples;       /* n_rows * (fft_length + cp_length) complex64                 */
} ofdmtx_raw_sizes;
/* n_frames frames of frame_size data symbols each, every frame starting with its flag set
 * (rawofdm/examples/ofdm_rxtx.py:81-87): no per-symbol table is needed. */
int ofdmtx_raw_plan_frames(ofdmtx_ctx *ctx, uint32_t fft_length, uint32_t cp_length,
                           const int32_t *carrier_map, const float *preambles, uint32_t n_pre,
                           uint64_t n_frames, uint32_t frame_size, ofdmtx_raw_plan **out);
/* General form: flags[n_sym] (host) is the start-of-frame input of ofdm_mod
 * (raw_ofdm_mod.py:25,51): a set flag emits the preamble symbols before that data symbol;
 * data symbols in front of the first set flag are dropped. */
int ofdmtx_raw_plan_flags(ofdmtx_ctx *ctx, uint32_t fft_length, uint32_t cp_length,
                          const int32_t *carrier_map, const float *preambles, uint32_t n_pre,
                          const uint8_t *flags, uint64_t n_sym, ofdmtx_raw_plan **out);
int ofdmtx_raw_plan_destroy(ofdmtx_raw_plan *plan);
int ofdmtx_raw_plan_get_sizes(const ofdmtx_raw_plan *plan, ofdmtx_raw_sizes *out);
/* d_data: n_sym x data_tones complex64.  d_mapped (optional tap, may be NULL): n_sym x
 * fft_length complex64, the mapper block's output ("tx-map.dat", raw_ofdm_mod.py:55-57).
 * d_iq: n_samples complex64. */
int ofdmtx_raw_modulate(ofdmtx_ctx *ctx, const ofdmtx_raw_plan *plan, const float *d_data,
                        float *d_mapped, float *d_iq, void *stream);
/* The other tap points of ofdm_mod's --log (raw_ofdm_mod.py:58-66), formed from d_mapped and d_iq
 * of a finished ofdmtx_raw_modulate on the same stream; each output may be NULL.
 * d_pre:  n_rows x fft_length, the preamble inserter's output ("tx-pre.dat": the mapped symbols with
 *         the preamble symbols in front of every frame, frequency domain);
 * d_ifft: n_rows x fft_length, gr.fft_vcc's output ("tx-ifft.dat");
 * d_cp:   n_rows x (fft_length + cp_length), the prefixer's output ("tx-cp.dat"), i.e. d_iq before
 *         the multiply_const block of :48. */
int ofdmtx_raw_taps(ofdmtx_ctx *ctx, const ofdmtx_raw_plan *plan, const float *d_mapped,
                    const float *d_iq, float *d_pre, float *d_ifft, float *d_cp, void *stream);

/* rawofdm's bit chain in front of its mapper: qam_tx of rawofdm/src/python/raw_qam.py:76-102, i.e.
 * the blocks raw.conv_punc(framebytes, framebytes+1) (one zero byte of padding per frame),
 * raw.conv_enc() (rawofdm/src/lib/qam/raw_conv.cc:33-55; K = 7, polynomials {121, 91} of
 * rawofdm/src/lib/spiral/defs.h:3-7), raw.conv_punc(nc, np) (raw_conv.cc:141-163: the first np of
 * every nc coded bits), raw.intrlv_bit(ncarriers, qambits, False) (raw_intrlv.cc:14-76) and
 * raw.qam_enc(qambits) (raw_qam.cc:63-74,113-150).  (qambits, nc, np) as in the `bitrates` table
 * of raw_qam.py:19-28.
 * ofdmtx_raw_framebytes: data bytes per frame of `size` = ncarriers*nsymbols complex values
 * (raw_qam.py:30-37 minus the padding byte); OFDMTX_ESHAPE where the reference's assert fails.
 * ofdmtx_raw_qam_tx: d_bytes = n_frames x framebytes bytes -> d_out = n_frames x nsymbols x
 * ncarriers complex64, the input of ofdmtx_raw_modulate.  BPSK values have imaginary part 0 (the
 * reference leaves that float unwritten). */
int ofdmtx_raw_framebytes(int qambits, int nc, int np, uint64_t size, uint32_t *out);
int ofdmtx_raw_qam_tx(ofdmtx_ctx *ctx, int qambits, int nc, int np, uint32_t ncarriers,
                      uint32_t nsymbols, const uint8_t *d_bytes, uint64_t n_frames, float *d_out,
                      void *stream);

/* Bytes -> IQ in one kernel: ofdmtx_raw_qam_tx feeding ofdmtx_raw_modulate without the carrier
 * values passing through memory (the flow of rawofdm/examples/ofdm_rxtx.py:63-87: qam_tx into
 * ofdm_mod).  `plan` must come from ofdmtx_raw_plan_frames; its frame_size is the bit chain's
 * nsymbols and its data_tones the bit chain's ncarriers (a multiple of 16 here).  d_bytes:
 * n_frames * framebytes bytes (less than 2 GiB); d_iq as for ofdmtx_raw_modulate, 16-byte aligned.
 * OFDMTX_ESHAPE for geometries or rates this kernel does not take (carrier counts off the 16 grid,
 * rates outside the table of rawofdm/src/python/raw_qam.py:19-28, and its entry (1, (6, 4)) -- marked
 * "DONT USE" there -- whose groups of 16 carriers do not start on byte boundaries): run the two calls
 * instead. */
int ofdmtx_raw_tx(ofdmtx_ctx *ctx, const ofdmtx_raw_plan *plan, int qambits, int nc, int np,
                  const uint8_t *d_bytes, float *d_iq, void *stream);

/* ---- placement of a large, long-lived output buffer ----------------------------------------------
 * No counterpart in the reference (its sinks own their buffers).  On MI355X streaming stores into a
 * buffer that lies in one of the memory's three 96 GiB parts run at 5.6-5.9 TB/s, into a buffer
 * whose pieces alternate between parts at 6.8-7.1 TB/s.  ofdmtx_stream_alloc returns `bytes` of
 * device memory made of 64 MiB pieces that alternate between two or three parts (HIP virtual-memory
 * API), usable like any device pointer for d_iq.  Which part a piece of physical memory lies in is
 * not visible to a process, so the call finds out by measurement: it takes memory 16 GiB at a time,
 * times every 2 GiB of it being written together with the first 2 GiB, stops when it has pieces of
 * two different parts in equal shares, and gives the rest back.  It therefore holds more than the
 * buffer for a moment (typically 100-140 GiB on an empty device, where the second part begins 96 GiB
 * into the allocation order; up to all free memory but 2 GiB), synchronises the device and runs
 * short kernels on the default stream: a call for set-up time, serialised inside the library.
 * ofdmtx_stream_alloc_parts with want_parts = 3 goes on until all three parts are there (about
 * 210 GiB into an empty device; it settles for two when the memory ends first): with the transmit
 * kernel's payload reads beside the stores three parts are 2-4 % faster than two.  *spread (may be NULL) tells whether the layout was
 * built; when only one part shows up within the free memory, or on any failure on the way, the
 * buffer is an ordinary hipMalloc allocation and *spread is 0.  ofdmtx_stream_info reports what a
 * buffer cost to build: its mapped size, the most physical memory held at one time while it was
 * built, the number of handles created and the number of parts its pieces alternate between (0 for
 * an ordinary allocation); each output may be NULL; OFDMTX_EINVAL for a pointer that is not such a
 * buffer.  With OFDMTX_ALLOC_TRACE set in the environment the measurement is printed to stderr.
 * ofdmtx_stream_free releases either kind (after a device synchronisation); the buffer may outlive
 * the context it was made with (any non-NULL ctx is accepted then). */
int ofdmtx_stream_alloc(ofdmtx_ctx *ctx, uint64_t bytes, void **out, int *spread);
int ofdmtx_stream_alloc_parts(ofdmtx_ctx *ctx, uint64_t bytes, int want_parts, void **out, int *spread);
int ofdmtx_stream_info(const void *ptr, uint64_t *bytes, uint64_t *peak_held_bytes, uint32_t *handles,
                       uint32_t *parts);
int ofdmtx_stream_free(ofdmtx_ctx *ctx, void *ptr);

/* ---- optional exchange: the ranks' IQ shards into one buffer on a root GPU -----------------------
 * Packets are sharded over GPUs by index and nothing on the transmit path needs another rank's
 * data; this is for a consumer that wants ONE stream, as the reference's flowgraph ends in one sink
 * (ftw_80211_ofdm_tx/src/examples/ftw_ofdm_tx_uhd.py:83-92: a single file sink or USRP sink behind
 * ofdm_mod).  One process per GPU; point-to-point over RCCL: every peer sends its shard to the root,
 * the root posts all receives as one group, so the peers' transfers run side by side on their own
 * xGMI links.  librccl is looked up at the first call (not linked): OFDMTX_ECOMM when it is absent.
 *
 * ofdmtx_comm_unique_id: rank 0 makes the 128-byte id and hands it to the other ranks by its own means
 *   (file, socket, MPI, a torch.distributed store); ofdmtx_comm_create is collective over the
 *   `world` ranks with that id, each on its own context/device.
 * ofdmtx_gather: collective, enqueued on `stream` (may be called per chunk of a shard, on a side
 *   stream, while the kernels produce the next chunk).  d_src: this rank's n_samples complex64.  On the root:
 *   d_dst receives rank r's samples at sample offset sample_off[r] (host array of world + 1
 *   ascending offsets; sample_off[root + 1] - sample_off[root] must equal n_samples); the root's own
 *   shard is copied unless d_src already is d_dst + 2 * sample_off[root].  Other ranks pass d_dst =
 *   sample_off = NULL.  Ranks with n_samples == 0 exchange nothing. */
#define OFDMTX_COMM_ID_BYTES 128
typedef struct ofdmtx_comm ofdmtx_comm;
int ofdmtx_comm_unique_id(void *id);
int ofdmtx_comm_create(ofdmtx_ctx *ctx, int rank, int world, const void *id, ofdmtx_comm **out);
int ofdmtx_comm_destroy(ofdmtx_comm *comm);
int ofdmtx_gather(ofdmtx_comm *comm, const float *d_src, uint64_t n_samples, float *d_dst,
                  const uint64_t *sample_off, int root, void *stream);
int ofdmtx_last_comm_error(void);   /* ncclResult_t of the last failing RCCL call on this thread */

#ifdef __cplusplus
}
#endif
#endif /* OFDMTX_H */
