/*
 * CPU ORACLE (test infrastructure, not product code): plain-C restatement of the FTW
 * 802.11a transmit chain, one packet at a time, scalar, one bit per byte internally.
 *
 * Used only by tests/ (as the checker at sizes numpy is too slow for), by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The product library
 * never links or calls it.
 *
 * The reference cannot be compiled or run here (Python 2 + GNU Radio; execution denied),
 * so this file is pinned the same way as oracle/ftw_oracle.py: the reference's literal
 * tables are passed in at run time from tests/golden/ftw_tables.json, and the Annex G.8
 * frame shipped with the reference is reproduced end to end (tests/test_oracle.py).
 * Bit-level results for regimes other than the Annex G one are parity-unpinned by any
 * reference-held vector.
 *
 * Paths below are relative to the reference root; "pu" = ftw_80211_ofdm_tx/src/examples/
 * ftw_packet_utils.py, "ftw/" = ftw_80211_ofdm_tx/src/ftw/.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXLEN 3168      /* pu:119 */
#define ORC_MAX_PLCP 3264    /* 5 + 3168 + tail/pad (<= 28 bytes), rounded up */
#define ORC_MAX_BITS (ORC_MAX_PLCP * 8)

typedef struct {
    uint8_t scr[127];        /* pu:346-349 */
    float pol[127];          /* ftw/ftw_ofdm_pilot_cc.cc:96-104 */
    float cons[86 * 2];      /* pu:497-506; complex index: bpsk@0 qpsk@2 qam16@6 qam64@22 */
    float preamble[640];     /* pu:508 */
} orc_tables;

typedef struct {
    int n_cbps, n_bpsc, n_dbps, rate_bits, punct; /* punct: 0 none, 1 = 3/4, 2 = 2/3 */
    int cons_off;
} orc_regime;

/* pu:39-78 (regime table), pu:81 (N_dbps = int(N_cbps*rate)), pu:129-144 (RATE nibble) */
static int orc_regime_info(int regime, orc_regime *r)
{
    static const orc_regime T[8] = {
        {48, 1, 24, 0x0D, 0, 0},   {48, 1, 36, 0x0F, 1, 0},
        {96, 2, 48, 0x05, 0, 2},   {96, 2, 72, 0x07, 1, 2},
        {192, 4, 96, 0x09, 0, 6},  {192, 4, 144, 0x0B, 1, 6},
        {288, 6, 192, 0x01, 2, 22}, {288, 6, 216, 0x03, 1, 22},
    };
    if (regime < 1 || regime > 8) return -1;
    *r = T[regime - 1];
    return 0;
}

/* pu:85-97: N_sym, N_pad; pu:160-171: tail+pad byte count; message length for the mapper */
int orc_packet_sizes(int regime, int payload_len, int *n_sym, int *n_pad, int *plcp_len, int *msg_len)
{
    orc_regime r;
    if (orc_regime_info(regime, &r)) return -1;
    int mpdu = 24 + payload_len;
    if (mpdu + 4 > ORC_MAXLEN) return -2;
    int nbits = 16 + 32 + 6 + 8 * mpdu;
    int ns = (nbits + r.n_dbps - 1) / r.n_dbps;
    int np = ns * r.n_dbps - nbits;
    if (n_sym) *n_sym = ns;
    if (n_pad) *n_pad = np;
    if (plcp_len) *plcp_len = 3 + 2 + mpdu + 4 + (np + 6 + 7) / 8;
    if (msg_len) *msg_len = 6 + ns * r.n_cbps / 8;
    return 0;
}

/* ftw/ftw_crc32.cc:33-82: reflected CRC-32, final state bit-reversed, then ^0xFFFFFFFF */
uint32_t orc_crc32(const uint8_t *d, size_t n)
{
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        crc ^= d[i];
        for (int j = 0; j < 8; j++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    uint32_t rev = 0;
    for (int j = 0; j < 32; j++) rev |= ((crc >> j) & 1u) << (31 - j);
    return rev ^ 0xFFFFFFFFu;
}

static uint8_t bitrev8(uint8_t b)
{
    uint8_t r = 0;
    for (int i = 0; i < 8; i++) r |= ((b >> i) & 1) << (7 - i);
    return r;
}

static void unpack_bits(const uint8_t *b, int nbytes, uint8_t *bits)
{
    for (int i = 0; i < nbytes; i++)
        for (int j = 0; j < 8; j++) bits[8 * i + j] = (b[i] >> (7 - j)) & 1;   /* pu:406-417 */
}

static void pack_bits(const uint8_t *bits, int nbits, uint8_t *b)
{
    for (int i = 0; i < nbits / 8; i++) {
        uint8_t v = 0;
        for (int j = 0; j < 8; j++) v = (uint8_t)((v << 1) | bits[8 * i + j]);  /* pu:440-448 */
        b[i] = v;
    }
}

/* get_info + ftw_make: pu:25-105,108-205.  Returns PLCP length, writes `Length` (MPDU+FCS). */
static int orc_build_plcp(const uint8_t *payload, int len, int regime, double symboltime,
                          uint8_t *plcp, int *length_out, int *n_sym_out)
{
    orc_regime r;
    int n_sym, n_pad, plcp_len;
    if (orc_regime_info(regime, &r)) return -1;
    if (orc_packet_sizes(regime, len, &n_sym, &n_pad, &plcp_len, NULL)) return -2;
    int mpdu_len = 24 + len, length = mpdu_len + 4;
    static const uint8_t hdr[24] = {0x08, 0x00, 0, 0, 0x00, 0x60, 0x08, 0xcd, 0x37, 0xa6,
                                    0x00, 0x20, 0xd6, 0x01, 0x3c, 0xf1, 0x00, 0x60, 0x08, 0xad,
                                    0x3b, 0xaf, 0x00, 0x00};               /* pu:28-36 */
    uint8_t mpdu[ORC_MAXLEN];
    memcpy(mpdu, hdr, 24);
    int txtime = (int)(5 * symboltime + symboltime * n_sym) + 2;           /* pu:90 */
    mpdu[2] = (uint8_t)((txtime >> 8) & 0xff);                              /* pu:91 */
    mpdu[3] = (uint8_t)(txtime & 0xff);
    memcpy(mpdu + 24, payload, (size_t)len);

    uint32_t rev = 0;
    for (int i = 0; i < 12; i++) rev |= (uint32_t)((length >> i) & 1) << (11 - i);   /* pu:147-150 */
    int par = 0;
    for (int k = 0; k < 12; k++) par += ((rev >> k) & 1) + ((r.rate_bits >> k) & 1);  /* pu:153-157 */
    uint32_t field = ((uint32_t)r.rate_bits << 20) | (rev << 7) | ((uint32_t)(par & 1) << 6);  /* pu:176 */
    plcp[0] = (uint8_t)(field >> 16);
    plcp[1] = (uint8_t)(field >> 8);
    plcp[2] = (uint8_t)field;
    plcp[3] = plcp[4] = 0;                                                  /* SERVICE pu:183 */
    for (int i = 0; i < mpdu_len; i++) plcp[5 + i] = bitrev8(mpdu[i]);      /* make_MPDU pu:208-218 */
    uint32_t fcs = orc_crc32(mpdu, (size_t)mpdu_len);                       /* pu:370-372 */
    plcp[5 + mpdu_len + 0] = (uint8_t)(fcs >> 24);
    plcp[5 + mpdu_len + 1] = (uint8_t)(fcs >> 16);
    plcp[5 + mpdu_len + 2] = (uint8_t)(fcs >> 8);
    plcp[5 + mpdu_len + 3] = (uint8_t)fcs;
    memset(plcp + 5 + length, 0, (size_t)(plcp_len - 5 - length));          /* pu:160-171 */
    *length_out = length;
    *n_sym_out = n_sym;
    return plcp_len;
}

/*
 * send_pkt bit chain (ftw_ofdm.py:201-217) for one payload.  Any of the output pointers may
 * be NULL.  plcp/scrambled: plcp_len bytes; coded/interleaved: msg_len bytes.
 */
int orc_encode_bits(const orc_tables *t, const uint8_t *payload, int len, int regime,
                    double symboltime, uint8_t *plcp_out, uint8_t *scr_out, uint8_t *cod_out,
                    uint8_t *itl_out)
{
    orc_regime r;
    if (orc_regime_info(regime, &r)) return -1;
    uint8_t plcp[ORC_MAX_PLCP];
    int length, n_sym;
    int plcp_len = orc_build_plcp(payload, len, regime, symboltime, plcp, &length, &n_sym);
    if (plcp_len < 0) return plcp_len;
    if (plcp_out) memcpy(plcp_out, plcp, (size_t)plcp_len);

    static __thread uint8_t u[ORC_MAX_BITS], enc[2 * ORC_MAX_BITS], pun[2 * ORC_MAX_BITS],
        itl[2 * ORC_MAX_BITS];
    int nb = 8 * plcp_len;
    unpack_bits(plcp, plcp_len, u);
    /* scrambler pu:344-367 */
    for (int k = 24; k < nb; k++) u[k] ^= t->scr[(k - 24) % 127];
    for (int k = 40 + 8 * length; k <= 45 + 8 * length; k++) u[k] = 0;
    if (scr_out) pack_bits(u, nb, scr_out);

    /* conv_encoder pu:280-307 */
    unsigned reg = 0;
    for (int i = 0; i < nb; i++) {
        reg = (reg >> 1) | ((unsigned)u[i] << 6);
        enc[2 * i] = (uint8_t)(__builtin_popcount(reg & 0x5b) & 1);
        enc[2 * i + 1] = (uint8_t)(__builtin_popcount(reg & 0x79) & 1);
    }
    /* puncturing pu:309-341 */
    int real = n_sym * r.n_cbps, nd = 2 * nb - 48, m = 0;
    memcpy(pun, enc, 48);
    for (int p = 0; p < nd && m < real; p++) {
        int keep = 1;
        if (r.punct == 1) { int q = p % 6; keep = (q <= 2 || q == 5); }
        else if (r.punct == 2) keep = (p % 4 != 3);
        if (keep) pun[48 + m++] = enc[48 + p];
    }
    if (m != real) return -3;
    int msg_bits = 48 + real;
    if (cod_out) pack_bits(pun, msg_bits, cod_out);

    /* interleaver pu:222-277 */
    for (int k = 0; k < 48; k++) itl[3 * (k % 16) + k / 16] = pun[k];
    int N = r.n_cbps, s = r.n_bpsc / 2 > 1 ? r.n_bpsc / 2 : 1;
    for (int sym = 0; sym < n_sym; sym++) {
        const uint8_t *in = pun + 48 + sym * N;
        uint8_t *out = itl + 48 + sym * N;
        for (int k = 0; k < N; k++) {
            int i = (N / 16) * (k % 16) + k / 16;
            int j = s * (i / s) + (i + N - (16 * i) / N) % s;
            out[j] = in[k];
        }
    }
    if (itl_out) pack_bits(itl, msg_bits, itl_out);
    return msg_bits / 8;
}

/* ---- symbol level ------------------------------------------------------------------ */

static void ifft64(const double *xr, const double *xi, double *yr, double *yi)
{
    /* unnormalised inverse DFT, radix-2 decimation in time; gr.fft_vcc(64,False,[],False)
       call of ftw_ofdm.py:118: x[n] = sum_k X[k] e^{+j 2 pi k n / 64} */
    static double wr[32], wi[32];
    static int init = 0;
    if (!init) {
        for (int k = 0; k < 32; k++) { wr[k] = cos(2 * M_PI * k / 64); wi[k] = sin(2 * M_PI * k / 64); }
        init = 1;
    }
    for (int i = 0; i < 64; i++) {
        int j = 0;
        for (int b = 0; b < 6; b++) j |= ((i >> b) & 1) << (5 - b);
        yr[j] = xr[i]; yi[j] = xi[i];
    }
    for (int len = 2; len <= 64; len <<= 1) {
        int half = len / 2, step = 64 / len;
        for (int i = 0; i < 64; i += len)
            for (int k = 0; k < half; k++) {
                double cr = wr[k * step], ci = wi[k * step];
                double tr = yr[i + k + half] * cr - yi[i + k + half] * ci;
                double ti = yr[i + k + half] * ci + yi[i + k + half] * cr;
                yr[i + k + half] = yr[i + k] - tr; yi[i + k + half] = yi[i + k] - ti;
                yr[i + k] += tr; yi[i + k] += ti;
            }
    }
}

/*
 * Mapper message -> IQ (ftw_ofdm.py:175-179 flowgraph).  flags: bit0 preamble, bit1 zero
 * gap, bit2 swap I/Q; bit3 / bit4: the six-row preamble of PNC role A / B
 * (ftw/ftw_pnc_ofdm_preamble.cc:80-103); bits 8-10: pilot block, 0 = ftw_ofdm_pilot_cc,
 * 2 / 3 / 4 = ftw_pnc_ofdm_pilot_cc with user 1 / 2 / 0 (ftw/ftw_pnc_ofdm_pilot_cc.cc:125-131).  mapped_out (optional): (1+N_sym)*48 complex64 (re,im floats).
 * iq_out: complex64 samples; iq64_out (optional): the same values before the float cast.
 * Returns the number of complex samples written.
 */
int orc_modulate(const orc_tables *t, const uint8_t *msg, int msg_len, int regime, int pilot_offset,
                 float amp, int flags, float *mapped_out, float *iq_out, double *iq64_out)
{
    orc_regime r;
    if (orc_regime_info(regime, &r)) return -1;
    int n_sym = (msg_len - 6) * 8 / r.n_cbps;
    if (6 + n_sym * r.n_cbps / 8 != msg_len) return -2;
    long o = 0;
    double a = (double)amp;
#define PUT(re_, im_) do { double re__ = (re_), im__ = (im_); \
        if (flags & 4) { double t__ = re__; re__ = im__; im__ = t__; } \
        if (iq_out) { iq_out[2 * o] = (float)re__; iq_out[2 * o + 1] = (float)im__; } \
        if (iq64_out) { iq64_out[2 * o] = re__; iq64_out[2 * o + 1] = im__; } o++; } while (0)
    if ((flags & 1) && (flags & (8 | 16))) {
        /* ftw/ftw_pnc_ofdm_preamble.cc:80-103: the two short-training rows, then the two
         * long-training rows before (user 1) or after (user 2) two rows of zeros */
        static const int rows_a[6] = {0, 1, 2, 3, -1, -1}, rows_b[6] = {0, 1, -1, -1, 2, 3};
        const int *rows = (flags & 8) ? rows_a : rows_b;
        for (int r = 0; r < 6; r++)
            for (int i = 0; i < 80; i++) {
                if (rows[r] < 0) PUT(0.0, 0.0);
                else PUT(t->preamble[2 * (80 * rows[r] + i)] * a, t->preamble[2 * (80 * rows[r] + i) + 1] * a);
            }
    } else if (flags & 1)                                     /* ftw/ftw_ofdm_preamble.cc:79-83 */
        for (int i = 0; i < 320; i++) PUT(t->preamble[2 * i] * a, t->preamble[2 * i + 1] * a);
    for (int sym = 0; sym <= n_sym; sym++) {
        float d[48][2];
        if (sym == 0) {                                       /* ftw/ftw_ofdm_mapper.cc:179-191,238-240 */
            for (int c = 0; c < 48; c++) {
                int bit = (msg[c / 8] >> (7 - c % 8)) & 1;
                d[c][0] = bit ? 1.f : -1.f; d[c][1] = 0.f;
            }
        } else {                                              /* ftw/ftw_ofdm_mapper.cc:210-260 */
            long base = 48 + (long)(sym - 1) * r.n_cbps;
            for (int c = 0; c < 48; c++) {
                int idx = 0;
                for (int b = 0; b < r.n_bpsc; b++) {
                    long k = base + (long)c * r.n_bpsc + b;
                    idx = (idx << 1) | ((msg[k / 8] >> (7 - k % 8)) & 1);
                }
                d[c][0] = t->cons[2 * (r.cons_off + idx)];
                d[c][1] = t->cons[2 * (r.cons_off + idx) + 1];
            }
        }
        if (mapped_out) memcpy(mapped_out + (long)sym * 96, d, sizeof d);
        /* pilots ftw/ftw_ofdm_pilot_cc.cc:113-158, carrier map ftw/ftw_ofdm_cmap_cc.cc:101-109 */
        double p = t->pol[(sym + pilot_offset) % 127];
        double pv[4] = {p, p, p, -p};                         /* carriers -21, -7, +7, +21 */
        const int pilot_block = (flags >> 8) & 7;
        if (pilot_block) {                                    /* ftw/ftw_pnc_ofdm_pilot_cc.cc:125-131,146-178 */
            const int a_ = pilot_block == 2, b_ = pilot_block == 3, all = pilot_block == 4;
            pv[0] = pv[2] = (a_ || all) ? 1.0 : 0.0;
            pv[1] = pv[3] = (b_ || all) ? 1.0 : 0.0;
        }
        double c53r[53], c53i[53];
        int di = 0;
        for (int i = 0; i < 53; i++) {
            if (i == 5 || i == 19 || i == 33 || i == 47) { c53r[i] = pv[i == 5 ? 0 : i == 19 ? 1 : i == 33 ? 2 : 3]; c53i[i] = 0; }
            else if (i == 26) { c53r[i] = 0; c53i[i] = 0; }
            else { c53r[i] = d[di][0]; c53i[i] = d[di][1]; di++; }
        }
        double xr[64] = {0}, xi[64] = {0}, yr[64], yi[64];
        for (int i = 0; i <= 26; i++) { xr[i] = c53r[i + 26]; xi[i] = c53i[i + 26]; }
        for (int i = 38; i < 64; i++) { xr[i] = c53r[i - 38]; xi[i] = c53i[i - 38]; }
        ifft64(xr, xi, yr, yi);
        /* CP ftw_ofdm.py:121-123, scale 1/sqrt(64) :126, amplitude ftw_ofdm_tx_uhd.py:82-86 */
        for (int n = 0; n < 80; n++) {
            int s = (n + 48) & 63;
            PUT(yr[s] * 0.125 * a, yi[s] * 0.125 * a);
        }
    }
    if (flags & 2)                                            /* ftw/ftw_zerogap.cc:88-92 */
        for (int i = 0; i < 13 * 80; i++) PUT(0.0, 0.0);
#undef PUT
    return (int)o;
}

/* ---- batch driver (cpu_baseline and large-size checks) ------------------------------ */

typedef struct {
    const orc_tables *t;
    const uint8_t *payloads; const int64_t *pay_off; const int32_t *pay_len; const uint8_t *regimes;
    const int32_t *pilot_off; const int64_t *msg_off; const int64_t *iq_off;
    uint8_t *msg_out; float *iq_out; float amp; int flags; double symboltime;
    long lo, hi; int rc;
} orc_job;

static void *orc_worker(void *arg)
{
    orc_job *j = (orc_job *)arg;
    uint8_t msg[ORC_MAX_PLCP * 2];
    for (long p = j->lo; p < j->hi; p++) {
        int ml = orc_encode_bits(j->t, j->payloads + j->pay_off[p], j->pay_len[p], j->regimes[p],
                                 j->symboltime, NULL, NULL, NULL, msg);
        if (ml < 0) { j->rc = ml; return NULL; }
        if (j->msg_out) memcpy(j->msg_out + j->msg_off[p], msg, (size_t)ml);
        if (j->iq_out) {
            int ns = orc_modulate(j->t, msg, ml, j->regimes[p], j->pilot_off ? j->pilot_off[p] : 0,
                                  j->amp, j->flags, NULL, j->iq_out + 2 * j->iq_off[p], NULL);
            if (ns < 0) { j->rc = ns; return NULL; }
        }
    }
    return NULL;
}

/*
 * n packets, payload -> mapper message (msg_out, optional) -> IQ (iq_out, optional), split
 * over `threads` pthreads by packet index.  Offsets are in bytes (payload, msg) and complex
 * samples (iq).
 */
int orc_tx_batch(const orc_tables *t, long n, const uint8_t *payloads, const int64_t *pay_off,
                 const int32_t *pay_len, const uint8_t *regimes, const int32_t *pilot_off,
                 const int64_t *msg_off, const int64_t *iq_off, uint8_t *msg_out, float *iq_out,
                 float amp, int flags, double symboltime, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    { double z[64] = {0}, y[64], w[64]; ifft64(z, z, y, w); }   /* build twiddles before threads start */
    orc_job jobs[256];
    pthread_t th[256];
    for (int i = 0; i < threads; i++) {
        orc_job j = {t, payloads, pay_off, pay_len, regimes, pilot_off, msg_off, iq_off,
                     msg_out, iq_out, amp, flags, symboltime, n * i / threads, n * (i + 1) / threads, 0};
        jobs[i] = j;
    }
    if (threads == 1) { orc_worker(&jobs[0]); return jobs[0].rc; }
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, orc_worker, &jobs[i]);
    int rc = 0;
    for (int i = 0; i < threads; i++) { pthread_join(th[i], NULL); if (jobs[i].rc) rc = jobs[i].rc; }
    return rc;
}
