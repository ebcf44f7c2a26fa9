// libofdmtx, variable-size OFDM modulator (the rawofdm transmit chain): carrier map -> N-point
// inverse FFT with the DC bin in the middle of the input -> cyclic prefix -> 1/sqrt(N), with the
// frame preambles inserted.  Up to N = 256, N = L x M is split into two in-register transforms with
// one LDS transposition and max(L, M) lanes work on one OFDM symbol; N = 512 and 1024 take a whole
// wave per symbol and three in-register transforms (N = A x V x V, V = N / 64 values per lane).
//
// Reference behaviour restated here is cited as file:line relative to the reference root.

#include <math.h>
#include <string.h>

#include <new>
#include <utility>
#include <vector>

#include "ofdm_common.h"

// Section markers in the generated assembly (tools/asm_sections.py); no code is emitted.
#ifdef OFDMTX_ASM_MARKERS
#define SEC(name) asm volatile("; SEC " name)
#else
#define SEC(name) do { } while (0)
#endif

namespace {

constexpr uint32_t ROW_DROPPED = 0xFFFFFFFFu;

// Passed by value to the kernel.
struct RawParams {
    uint32_t n, cp, data_tones, n_pre;   // FFT length, cyclic prefix, data carriers per symbol, preamble symbols per frame
    uint32_t n_sym;                      // data symbols of the batch
    uint32_t n_frames;                   // frames whose preamble rows are copied in
    uint32_t frame_size;                 // uniform frames: data symbols per frame; 0: use the tables below
    const uint32_t *sym_row;             // output row of every data symbol (ROW_DROPPED: none)
    const uint32_t *frame_row;           // first preamble row of every frame
    const int16_t *src;                  // per FFT bin k: >= 0 index into the symbol's data, -1 zero, -2 (+1,0), -3 (-1,0)
    const float2 *tw;                    // e^{+j 2 pi i / N} / sqrt(N), i = 0 .. N-1
    const float2 *pre_rows;              // n_pre rows of n + cp time-domain samples
};

// e^{+j 2 pi k / 32}, k = 0 .. 15
__device__ constexpr float W32C[16] = {1.f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f,
                                       0.382683432f, 0.195090322f, 0.f, -0.195090322f, -0.382683432f, -0.555570233f,
                                       -0.707106781f, -0.831469612f, -0.923879533f, -0.98078528f};
__device__ constexpr float W32S[16] = {0.f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f,
                                       0.923879533f, 0.98078528f, 1.f, 0.98078528f, 0.923879533f, 0.831469612f,
                                       0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};

// a * w for a wave-uniform w (kept in a scalar register pair)
__device__ __forceinline__ v2f cmul_s(v2f a, v2f w)
{
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}

// (a, b) <- (a + w b, a - w b), w = e^{+j 2 pi T / 32}
template <int T>
__device__ __forceinline__ void butterfly(v2f &a, v2f &b)
{
    const v2f t = a;
    if constexpr (T == 0) {
        a = t + b;
        b = t - b;
    } else if constexpr (T == 8) {
        a = add_j(t, b);
        b = sub_j(t, b);
    } else if constexpr (T == 4) {
        // w = (1 + j) / sqrt 2:  w b = h (b + j b), the factor folded into the two multiply-adds
        const v2f h = {0.70710678118654752440f, 0.70710678118654752440f};
        const v2f u = add_j(b, b);
        a = __builtin_elementwise_fma(h, u, t);
        b = __builtin_elementwise_fma(-h, u, t);
    } else if constexpr (T == 12) {
        // w = (-1 + j) / sqrt 2:  w b = -h (b - j b)
        const v2f h = {0.70710678118654752440f, 0.70710678118654752440f};
        const v2f u = sub_j(b, b);
        a = __builtin_elementwise_fma(-h, u, t);
        b = __builtin_elementwise_fma(h, u, t);
    } else {
        const v2f w = cmul_s(b, (v2f){W32C[T], W32S[T]});
        a = t + w;
        b = t - w;
    }
}

template <int R, int LEN, int I>
__device__ __forceinline__ void butterfly_at(v2f (&x)[R])
{
    constexpr int half = LEN / 2, base = (I / half) * LEN, k = I % half;
    butterfly<k * (32 / LEN)>(x[base + k], x[base + k + half]);
}

template <int R, int LEN, int... I>
__device__ __forceinline__ void dit_stage(v2f (&x)[R], std::integer_sequence<int, I...>)
{
    (butterfly_at<R, LEN, I>(x), ...);
}

// In-register inverse DFT of R = 4, 8, 16 or 32 points, decimation in time: x holds the input in
// bit-reversed order on entry and y[p] = sum_m in[m] e^{+j 2 pi m p / R} in natural order on exit.
template <int R>
__device__ __forceinline__ void idft_reg(v2f (&x)[R])
{
    using half_seq = std::make_integer_sequence<int, R / 2>;
    dit_stage<R, 2>(x, half_seq{});
    dit_stage<R, 4>(x, half_seq{});
    if constexpr (R >= 8) dit_stage<R, 8>(x, half_seq{});
    if constexpr (R >= 16) dit_stage<R, 16>(x, half_seq{});
    if constexpr (R >= 32) dit_stage<R, 32>(x, half_seq{});
}

constexpr int bit_reverse(int i, int n)
{
    int r = 0;
    for (int b = 1; b < n; b <<= 1) {
        r = (r << 1) | (i & 1);
        i >>= 1;
    }
    return r;
}

constexpr int RAW_WAVES = 4;
// Store policy of these kernels' output (the policy field of the buffer store: 1 = sc0, 2 = nt,
// 16 = sc1): non-temporal for the modulator's rows and for the bit chain's values.  They read 0.6
// bytes per byte written (the 802.11a kernel 0.03), and what suits them changed with where the
// output lies: into an ordinary allocation (one part of the device memory) plain stores were best
// for the rows (sc1: N=1024 1.07 against 0.97 ms); into a buffer spread over the parts -- variants
// interleaved in one process -- nt gives bytes -> IQ 0.61 -> 0.53 ms at N=256 and 0.78 -> 0.63 ms at
// N=1024, the modulator alone 1-2 %, the bit chain's values 0.293 -> 0.249 ms (nt + sc1 is worse than
// nt for the rows, the same for the values; non-temporal LOADS of the carrier values cost 25 %;
// raised wave priority for the store phase: nothing).
constexpr int RAW_ROW_AUX = 2, RAW_VAL_AUX = 2;

// float2 per symbol area: the padded transposition layout M x (L + 1), which also holds the N
// time-domain samples afterwards (the cyclic prefix is not staged: it is read from the tail again)
template <int L, int M>
constexpr int raw_area() { return L == 64 ? M * (64 + 64 / M) : (M * (L + 1) + 1) & ~1; }

// Factor on the plan's twiddle table e^{j 2 pi i / N} / sqrt(N) when it is copied to LDS: the
// three-step transform (L == 64) multiplies by two twiddles per sample, each carrying N^(-1/4).
template <int L, int M>
constexpr float raw_tw_factor() { return L != 64 ? 1.f : (M == 16 ? 5.656854249f : 4.756828460f); }

// Steps 1b-3 of the N = L * M point modulator for the S = max(L, M) lanes of one symbol (lane t of
// the symbol, symbol area T, wave's first area T_wave with SPW areas of raw_area<L, M>() float2 each):
//   bin k = l + L m of the (DC-centred) input:  x[M q + p] = sum_l W_L^{l q} W_N^{l p} sum_m X[l + L m] W_M^{m p}
// On entry lane t < L holds its M bins in x, bit-reversed.  step 1b: M-point transforms in
// registers, times W_N^{l p} / sqrt(N), to LDS; step 2: lane p < M reads the L partial results,
// transforms them and puts its samples M q + p back into the area in time order; step 3: the wave
// stores the rows with 16 bytes per lane, consecutive over the lanes -- output sample i of a row is
// time sample (i - cp) mod N (gr-digital ofdm_cyclic_prefixer, rawofdm/src/python/raw_ofdm_mod.py:47).
// All lanes of the wave must call it (wave-level hand-overs inside).
// s_tw1 (L == 64 only): the first step's twiddles W_N^{A b n3} as a V x V image [n3][b], so that the
// 16 (or 8) different values a wave reads at once lie side by side -- out of the full table the same
// read strides over it by 32 n3 bytes and meets 2- to 8-way bank conflicts.
template <int L, int M>
constexpr int raw_tw1_size() { return L == 64 ? M * M : 1; }
template <int L, int M>
__device__ __forceinline__ void raw_tw1_fill(v2f *s_tw1, const v2f *s_tw, int tid, int n_thr)
{
    if constexpr (L == 64) {
        constexpr int V = M, A = 64 / V, N = L * M;
        for (int i = tid; i < V * V; i += n_thr) s_tw1[i] = s_tw[(A * (i % V) * (i / V)) & (N - 1)];
    }
}
template <int L, int M>
__device__ __forceinline__ void raw_transform_store(bool active, int t, int lane, uint32_t row, v2f (&x)[M], v2f *T,
                                                    const v2f *T_wave, const v2f *s_tw, const v2f *s_tw1, uint32_t n,
                                                    uint32_t cp, float2 *__restrict__ iq)
{
    constexpr int N = L * M, S = L > M ? L : M, SPW = 64 / S, TS = raw_area<L, M>();
    const uint32_t row_len = n + cp;
    SEC("t_fft1");
    if constexpr (L == 64) {
        // ---- one symbol per wave, V = M bins per lane, N = A x V x V with A = 64 / V:
        //   bin k = a + A b + 64 c, sample n = V V n1 + V n2 + n3,
        //   x[n] = sum_a W_A^{a n1} W_N^{a (V n2 + n3)} sum_b W_V^{b n2} W_N^{A b n3} sum_c W_V^{c n3} X[k]
        // three in-register transforms of V, V and A points with an exchange through LDS after each
        // (areas of V x (64 + 32 / V), A x (V V + pad) and N values, one after the other in the same
        // memory; the paddings make every exchange free of bank conflicts: 8-byte reads are banked
        // over 256 bytes per 32 lanes, 8-byte writes over 128 bytes per 16 lanes); the lane is
        // (a, b) in the first, (a, n3) in the second and (n2, n3) in the third.
        constexpr int V = M, A = 64 / V, E1 = 64 + 32 / V, Q = V * V, E2 = Q + (V == 16 ? 16 : 8);
        static_assert(V == 8 || V == 16, "64 lanes per symbol: N = 512 or 1024");
        static_assert(V * E1 <= TS && A * E2 <= TS && N <= TS, "symbol area too small");
        idft_reg<V>(x);
        {
            const int b = lane / A;
#pragma unroll
            for (int n3 = 0; n3 < V; n3++) T[n3 * E1 + lane] = cmul(x[n3], s_tw1[n3 * V + b]);
        }
        wave_sync();
        const int n3 = lane & (V - 1), a2 = lane / V;
        v2f y[V];
#pragma unroll
        for (int b = 0; b < V; b++) y[bit_reverse(b, V)] = T[n3 * E1 + a2 + A * b];
        wave_sync();
        SEC("t_fft2");
        idft_reg<V>(y);
#pragma unroll
        for (int n2 = 0; n2 < V; n2++) T[a2 * E2 + V * n2 + n3] = cmul(y[n2], s_tw[(a2 * (V * n2 + n3)) & (N - 1)]);
        wave_sync();
        SEC("t_fft3");
#pragma unroll
        for (int i = 0; i < Q / 64; i++) {
            v2f z[A];
#pragma unroll
            for (int a = 0; a < A; a++) z[bit_reverse(a, A)] = T[a * E2 + lane + 64 * i];
            idft_reg<A>(z);
#pragma unroll
            for (int n1 = 0; n1 < A; n1++) x[A * i + n1] = z[n1];
        }
        wave_sync();
        // ---- rows straight from the registers: x[A i + n1] is time sample Q n1 + 64 i + lane, so every
        // (i, n1) is 64 consecutive samples = 512 contiguous bytes of the row, 8 bytes per lane; the
        // samples of the last cp are stored a second time at the row's start (output sample i of a row
        // is time sample (i - cp) mod N).  Against staging the row in LDS for 16-byte stores (16 LDS
        // stores and 10 wide LDS loads per symbol, in kernels whose LDS array is busy for 60 % of their
        // time): bytes -> IQ at N = 1024 0.61 -> 0.575 ms per 2.6 10^5 symbols, N = 512 and the
        // modulator alone unchanged; the same bits.
        SEC("t_store");
        const uint32_t row_w = __shfl(row, 0);
        if (row_w != ROW_DROPPED) {
            const __amdgpu_buffer_rsrc_t win = iq_window(iq + (size_t)row_w * row_len);
            const uint32_t tail = n - cp;
#pragma unroll
            for (int n1 = 0; n1 < A; n1++)
#pragma unroll
                for (int i = 0; i < Q / 64; i++) {
                    const uint32_t sm = (uint32_t)(Q * n1 + 64 * i) + (uint32_t)lane;
                    const v2f v = x[A * i + n1];
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 d = {__float_as_uint(v.x), __float_as_uint(v.y)};
                    __builtin_amdgcn_raw_buffer_store_b64(d, win, (int)((cp + sm) * 8u), 0, RAW_ROW_AUX);
                    if (sm >= tail) __builtin_amdgcn_raw_buffer_store_b64(d, win, (int)((sm - tail) * 8u), 0, RAW_ROW_AUX);
                }
        }
        return;
    } else {
    if (active && t < L) {
        idft_reg<M>(x);
#pragma unroll
        for (int p = 0; p < M; p++) T[p * (L + 1) + t] = cmul(x[p], s_tw[(t * p) & (N - 1)]);
    }
    wave_sync();
    // ---- step 2: L-point transforms
    v2f y[L];
    SEC("t_fft2");
    if (active && t < M) {
#pragma unroll
        for (int l = 0; l < L; l++) y[bit_reverse(l, L)] = T[t * (L + 1) + l];
        idft_reg<L>(y);
    }
    wave_sync();                                              // the partial results are consumed
    SEC("t_cp");
    if (active && t < M) {
#pragma unroll
        for (int q = 0; q < L; q++) T[M * q + t] = y[q];
    }
    wave_sync();
    }
    // ---- step 3: rows to memory (a row starts on a 16-byte boundary when row_len is even; cp is
    // even then as well, so a 16-byte piece never straddles the end of the prefix)
    SEC("t_store");
    const uint32_t tail = n - cp;                             // first time sample of the prefix
    // output sample 2 i of a row is time sample (2 i - cp) mod N: the lane's first piece, then 128 samples on
    const uint32_t k0 = (2u * (uint32_t)lane + tail) & (uint32_t)(N - 1);
#pragma unroll
    for (int sl = 0; sl < SPW; sl++) {
        const uint32_t row_s = __shfl(row, sl * S);           // the row of slot sl
        if (row_s == ROW_DROPPED) continue;
        const v2f *Ts = T_wave + sl * TS;
        float2 *o = iq + (size_t)row_s * row_len;
        if ((row_len & 1u) == 0u) {
            const __amdgpu_buffer_rsrc_t win = iq_window(o);
            uint32_t k = k0;
            for (uint32_t i = lane; i < row_len / 2; i += 64) {
                iq_store<RAW_ROW_AUX>(win, i, *reinterpret_cast<const float4 *>(Ts + k));
                k = (k + 128u) & (uint32_t)(N - 1);
            }
        } else {
            for (uint32_t i = lane; i < row_len; i += 64) {
                const uint32_t k = i < cp ? tail + i : i - cp;
                o[i] = make_float2(Ts[k].x, Ts[k].y);
            }
        }
    }
}

// One work item per wavefront and iteration: either 64 / max(L, M) OFDM symbols (items
// 0 .. n_groups-1) or the copy of one frame's preamble rows (the items after those).
//   bin k = l + L m of the (DC-centred) input:  x[M q + p] = sum_l W_L^{l q} W_N^{l p} sum_m X[l + L m] W_M^{m p}
// step 1: lane l < L transforms its M bins in registers, applies W_N^{l p} / sqrt(N) and writes
// to LDS; step 2: lane p < M reads the L partial results and transforms them, and puts its
// samples M q + p, cyclic prefix included, back into the symbol's LDS area as one output row;
// step 3: the wave stores the rows with 16 bytes per lane, consecutive over the lanes.
template <int L, int M, int W>
__global__ void __launch_bounds__(64 * W)
k_raw_modulate(RawParams R, const float2 *__restrict__ data, float2 *__restrict__ mapped, float2 *__restrict__ iq)
{
    constexpr int N = L * M, S = L > M ? L : M, SPW = 64 / S, TS = raw_area<L, M>();
    extern __shared__ __align__(16) unsigned char s_raw[];
    v2f *s_tw = reinterpret_cast<v2f *>(s_raw);                            // N
    v2f *s_t = s_tw + N;                                                   // W * SPW * TS
    int16_t *s_src = reinterpret_cast<int16_t *>(s_t + W * SPW * TS);   // N

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < N; i += 64 * W) {
        const float2 w = R.tw[i];
        s_tw[i] = (v2f){w.x, w.y} * raw_tw_factor<L, M>();
        s_src[i] = R.src[i];
    }
    __shared__ v2f s_tw1[raw_tw1_size<L, M>()];
    __syncthreads();
    raw_tw1_fill<L, M>(s_tw1, s_tw, tid, 64 * W);
    __syncthreads();

    const int t = lane % S, slot = lane / S;
    v2f *T = s_t + (wave * SPW + slot) * TS;
    const uint32_t row_len = R.n + R.cp;
    const uint32_t n_groups = (R.n_sym + SPW - 1) / SPW;
    const uint32_t n_items = n_groups + R.n_frames;

    for (uint32_t item = blockIdx.x * W + wave; item < n_items; item += gridDim.x * W) {
        if (item >= n_groups) {
            // ---- preamble rows of one frame: time-domain table -> output
            // (gr-digital ofdm_insert_preamble as wired at rawofdm/src/python/raw_ofdm_mod.py:45,50-52)
            const uint32_t f = item - n_groups;
            const size_t row = R.frame_size ? (size_t)f * (R.frame_size + R.n_pre) : (size_t)R.frame_row[f];
            const uint32_t cnt = R.n_pre * row_len;
            if ((row_len & 1u) == 0u) {
                const float4 *src4 = reinterpret_cast<const float4 *>(R.pre_rows);
                const __amdgpu_buffer_rsrc_t win = iq_window(iq + row * row_len);
                for (uint32_t i = lane; i < cnt / 2; i += 64) iq_store<RAW_ROW_AUX>(win, i, src4[i]);
            } else {
                float2 *dst = iq + row * row_len;
                for (uint32_t i = lane; i < cnt; i += 64) dst[i] = R.pre_rows[i];
            }
            continue;
        }
        const uint32_t sym = item * SPW + slot;
        const bool active = sym < R.n_sym;
        uint32_t row = ROW_DROPPED;
        if (active)
            row = R.frame_size ? (sym / R.frame_size) * (R.frame_size + R.n_pre) + R.n_pre + sym % R.frame_size
                               : R.sym_row[sym];

        // ---- step 1: carrier map (rawofdm/src/lib/ofdm/raw_ofdm_mapper.cc:53-78)
        v2f x[M];
        if (active && t < L) {
            const float2 *dsym = data + (size_t)sym * R.data_tones;
            // all loads first, nothing between them that could make the compiler wait for one
            // load before it issues the next (a tap store in this loop did exactly that)
            int code[M];
            float2 d[M];
#pragma unroll
            for (int m = 0; m < M; m++) code[m] = s_src[t + L * m];
#pragma unroll
            for (int m = 0; m < M; m++) d[m] = dsym[code[m] >= 0 ? code[m] : 0];
#pragma unroll
            for (int m = 0; m < M; m++) {
                v2f v = {d[m].x, d[m].y};
                if (code[m] < 0) v = (v2f){code[m] == -2 ? 1.f : (code[m] == -3 ? -1.f : 0.f), 0.f};
                x[bit_reverse(m, M)] = v;
            }
            if (mapped) {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const v2f v = x[bit_reverse(m, M)];
                    mapped[(size_t)sym * N + ((t + L * m + N / 2) & (N - 1))] = make_float2(v.x, v.y);
                }
            }
        }
        raw_transform_store<L, M>(active, t, lane, active ? row : ROW_DROPPED, x, T, s_t + wave * SPW * TS, s_tw, s_tw1,
                                  R.n, R.cp, iq);
        wave_sync();
    }
}

template <int L, int M, int W>
size_t raw_lds_bytes()
{
    constexpr int N = L * M, S = L > M ? L : M, SPW = 64 / S, TS = raw_area<L, M>();
    return (size_t)N * sizeof(float2) + (size_t)W * SPW * TS * sizeof(float2) + (size_t)N * sizeof(int16_t);
}

template <int L, int M, int W = RAW_WAVES>
int launch_raw(const ofdmtx_ctx *ctx, const RawParams &rp, const float2 *data, float2 *mapped, float2 *iq,
               hipStream_t st)
{
    constexpr int S = L > M ? L : M, SPW = 64 / S;
    const size_t lds = raw_lds_bytes<L, M, W>();
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_raw_modulate<L, M, W>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t items = ((uint64_t)rp.n_sym + SPW - 1) / SPW + rp.n_frames;
    uint64_t blocks = (items + W - 1) / W;
    const uint64_t per_cu = 160u * 1024u / lds ? 160u * 1024u / lds : 1;
    const uint64_t cap = (uint64_t)ctx->n_cu * (per_cu < 8 ? per_cu : 8);
    if (blocks > cap) blocks = cap;
    if (blocks == 0) return OFDMTX_OK;
    hipLaunchKernelGGL((k_raw_modulate<L, M, W>), dim3((unsigned)blocks), dim3(64 * W), lds, st, rp, data,
                       mapped, iq);
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int launch_raw_n(const ofdmtx_ctx *ctx, const RawParams &rp, const float2 *data, float2 *mapped, float2 *iq,
                 hipStream_t st)
{
    switch (rp.n) {
    case 64: return launch_raw<8, 8>(ctx, rp, data, mapped, iq, st);
    case 128: return launch_raw<8, 16>(ctx, rp, data, mapped, iq, st);
    case 256: return launch_raw<16, 16>(ctx, rp, data, mapped, iq, st);
    case 512: return launch_raw<64, 8>(ctx, rp, data, mapped, iq, st);
    case 1024: return launch_raw<64, 16>(ctx, rp, data, mapped, iq, st);
    default: return OFDMTX_ESHAPE;
    }
}

// -----------------------------------------------------------------------------------------
// TX bit chain in front of the mapper (rawofdm/src/python/raw_qam.py:76-102): per frame of
// frame_bytes bytes: one zero byte appended (conv_punc(framebytes, framebytes + 1)), K = 7
// rate-1/2 encoder, puncturing (the first np of every nc coded bits), bit interleaver per OFDM
// symbol, Gray QAM.  One thread per output carrier value walks the chain backwards for each of
// its bits: interleaved position -> punctured index -> coded index -> a 7-bit window of the
// frame's bytes.  Frames are independent: the zero byte returns the encoder to state 0.
struct QamTxParams {
    uint32_t frame_bytes;      // data bytes per frame (without the padding byte)
    uint32_t ncarriers, nsymbols, qambits, nc, np;
    uint32_t cbps;             // qambits * ncarriers
    uint64_t n_values;         // n_frames * nsymbols * ncarriers
    uint64_t total_bytes;      // n_frames * frame_bytes
    float scale;               // PAM level spacing / 2 (rawofdm/src/lib/qam/raw_qam.cc:39-52)
};

// rawofdm/src/lib/spiral/config.h + defs.h:3-7: CODE 121-91, K = 7, POLYS {121, 91} (decimal);
// rawofdm/src/lib/qam/raw_conv.cc:46-51: newest bit at the register's LSB, first polynomial first
__device__ __forceinline__ uint32_t raw_coded_bit(const uint8_t *frame, uint32_t frame_bytes, uint32_t c)
{
    const uint32_t n = c >> 1;                               // input bit index inside the padded frame
    // bits n-6 .. n as a window, bit d = u[n-d]; bytes past the data are the zero padding
    const int32_t b1 = (int32_t)(n >> 3), b0 = b1 - 1;
    const uint32_t hi = b0 >= 0 && (uint32_t)b0 < frame_bytes ? frame[b0] : 0u;
    const uint32_t lo = (uint32_t)b1 < frame_bytes ? frame[b1] : 0u;
    const uint32_t two = (hi << 8) | lo;                     // u[8*b0 ..] MSB first
    const uint32_t sh = 7u - (n & 7u);                       // u[n] sits at bit sh of `two`
    const uint32_t reg = (two >> sh) & 0x7Fu;                // bit 0 = u[n], bit d = u[n-d]
    return __popc(reg & ((c & 1u) ? 91u : 121u)) & 1u;
}

// One workgroup walks rows (one OFDM symbol of one frame each) with a grid stride; its threads take
// the row's carriers.  Per row the encoder runs word-parallel first: every thread turns 32 input
// bits into 64 coded bits (both polynomials as XORs of shifted windows, interleaved A0 B0 A1 B1 ..)
// in LDS; the carriers then pick their bits from there.  The inverse interleaver map of a symbol
// (interleaved position -> punctured index) is computed once per workgroup into LDS.  QB / NC / NP
// are the table entries of rawofdm/src/python/raw_qam.py:19-28 as compile-time constants (NC = 0:
// taken from Q at run time).
__device__ __forceinline__ uint64_t spread_bits(uint32_t v)      // bit i -> bit 2i
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// 32 bits of the padded frame starting at bit position `pos` (may be negative), MSB first
__device__ __forceinline__ uint32_t frame_bits32(const uint8_t *frame, uint32_t frame_bytes, int64_t pos)
{
    uint64_t acc = 0;
    const int64_t b0 = pos >> 3;                                  // floor for negatives too
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int64_t b = b0 + i;
        acc = (acc << 8) | ((b >= 0 && (uint64_t)b < frame_bytes) ? frame[b] : 0u);
    }
    return (uint32_t)(acc >> (8u - (uint32_t)(pos & 7)));
}

template <int QB, int NC, int NP>
__global__ void __launch_bounds__(256) k_raw_qam_tx(QamTxParams Q, const uint8_t *__restrict__ data,
                                                    float2 *__restrict__ out, uint32_t cod_words, uint32_t rows_per_iter)
{
    extern __shared__ uint32_t s_qam[];
    uint32_t *s_cod = s_qam;                                      // coded bits of the row, MSB first
    uint16_t *s_k = reinterpret_cast<uint16_t *>(s_qam + cod_words * rows_per_iter);
    const uint32_t qambits = QB ? (uint32_t)QB : Q.qambits;
    const uint32_t nc = NC ? (uint32_t)NC : Q.nc, np = NC ? (uint32_t)NP : Q.np;
    const uint32_t s = qambits / 2u > 1u ? qambits / 2u : 1u, col = Q.cbps / 16u;
    for (uint32_t j = threadIdx.x; j < Q.cbps; j += blockDim.x) {
        // invert rawofdm/src/lib/qam/raw_intrlv.cc:22-30: j = s*(i/s) + (i + cbps - 16*i/cbps) % s,
        // i = (cbps/16)*(k%16) + k/16
        uint32_t i = 0;
        for (uint32_t q = 0; q < s; q++) {
            const uint32_t cand = s * (j / s) + q;
            if ((cand + Q.cbps - 16u * cand / Q.cbps) % s == j % s) i = cand;
        }
        s_k[j] = (uint16_t)(16u * (i % col) + i / col);
    }

    const uint32_t axis_bits = qambits == 1u ? 1u : qambits / 2u;
    const uint64_t n_rows = Q.n_values / Q.ncarriers;
    const uint32_t cod_per_row = Q.cbps / np * nc;                // coded bits behind one symbol
    const uint32_t words_per_row = (cod_per_row + 63u) / 64u;     // 32-bit input words feeding one symbol
    // rows_per_iter consecutive rows per pass, so that short symbols still fill the workgroup
    for (uint64_t row0 = (uint64_t)blockIdx.x * rows_per_iter; row0 < n_rows; row0 += (uint64_t)gridDim.x * rows_per_iter) {
        const uint32_t rows_here = n_rows - row0 < rows_per_iter ? (uint32_t)(n_rows - row0) : rows_per_iter;
        __syncthreads();                                          // previous rows consumed (and s_k written)
        // ---- rawofdm/src/lib/qam/raw_conv.cc:46-51 for the input bits behind each symbol:
        // polynomial 121 taps u[n], u[n-3..n-6]; 91 taps u[n], u[n-1], u[n-3], u[n-4], u[n-6]
        for (uint32_t idx = threadIdx.x; idx < rows_here * words_per_row; idx += blockDim.x) {
            const uint32_t r = idx / words_per_row, w = idx - r * words_per_row;
            const uint64_t row = row0 + r, f = row / Q.nsymbols;
            const uint32_t sym = (uint32_t)(row - f * Q.nsymbols);
            const uint8_t *frame = data + f * Q.frame_bytes;
            const int64_t n0 = (int64_t)((sym * cod_per_row) >> 1) + 32 * (int64_t)w;
            const uint32_t prev = frame_bits32(frame, Q.frame_bytes, n0 - 32), cur = frame_bits32(frame, Q.frame_bytes, n0);
            const uint32_t y1 = __funnelshift_r(cur, prev, 1), y3 = __funnelshift_r(cur, prev, 3);
            const uint32_t y4 = __funnelshift_r(cur, prev, 4), y5 = __funnelshift_r(cur, prev, 5);
            const uint32_t y6 = __funnelshift_r(cur, prev, 6);
            const uint32_t aw = cur ^ y3 ^ y4 ^ y5 ^ y6, bw = cur ^ y1 ^ y3 ^ y4 ^ y6;
            const uint64_t both = (spread_bits(aw) << 1) | spread_bits(bw);     // A before B, MSB first
            s_cod[r * cod_words + 2u * w] = (uint32_t)(both >> 32);
            s_cod[r * cod_words + 2u * w + 1u] = (uint32_t)both;
        }
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < rows_here * Q.ncarriers; idx += blockDim.x) {
            const uint32_t r = idx / Q.ncarriers, car = idx - r * Q.ncarriers;
            const uint32_t *cod = s_cod + r * cod_words;
            float lv[2] = {0.f, 0.f};
#pragma unroll
            for (uint32_t a = 0; a < 2u; a++) {
                if (a * axis_bits >= qambits) break;
                int pt = 0, flip = 1;                             // rawofdm/src/lib/qam/raw_qam.cc:63-74
#pragma unroll
                for (uint32_t t = 0; t < 3u; t++) {
                    if (t >= axis_bits) break;
                    const uint32_t k = s_k[car * qambits + a * axis_bits + t];
                    const uint32_t c = (k / np) * nc + k % np;    // raw_conv.cc:150-159, relative to the symbol
                    const int bit = (int)((cod[c >> 5] >> (31u - (c & 31u))) & 1u) * 2 - 1;
                    pt = bit * flip + pt * 2;
                    flip *= -bit;
                }
                lv[a] = (float)pt * Q.scale;
            }
            out[(row0 + r) * Q.ncarriers + car] = make_float2(lv[0], lv[1]);
        }
    }
}

// ---- the table entries of rawofdm/src/python/raw_qam.py:19-28: structure instead of a per-bit map
//
// rawofdm/src/lib/qam/raw_conv.cc:150-159 keeps the first np of every nc coded bits; on MSB-first
// words that is a fixed compaction.
__device__ __forceinline__ uint32_t keep_3_of_4(uint32_t x)      // 32 bits -> 24 bits, left-aligned
{
    const uint32_t y = (x & 0xE0E0E0E0u) | ((x & 0x0E0E0E0Eu) << 1);      // 6 bits at the top of each byte
    const uint32_t z = (y & 0xFC00FC00u) | ((y & 0x00FC00FCu) << 2);      // 12 bits at the top of each half
    return (z & 0xFFF00000u) | ((z & 0x0000FFF0u) << 4);
}
__device__ __forceinline__ uint32_t keep_4_of_6(uint32_t c)      // 24 bits (right-aligned) -> 16 bits
{
    return ((c >> 8) & 0xF000u) | ((c >> 6) & 0x0F00u) | ((c >> 4) & 0x00F0u) | ((c >> 2) & 0x000Fu);
}

// The bit interleaver (rawofdm/src/lib/qam/raw_intrlv.cc:22-30) on a symbol of cbps punctured
// bits is a 16-column block interleaver followed by a rotation inside groups of s bits:
// with col = cbps / 16 and the punctured stream read as 16-bit units H[m] (MSB first),
// position i = col * b + m holds bit b of H[m]; position j of the output takes
// i = s*(j/s) + ((j % s) + b) % s, b = i / col being the same for a whole group because s
// divides col.  A group of s bits is one axis of one carrier (raw_qam.cc:63-74: I, then Q), so a
// carrier's axis reads s neighbouring units at ONE bit position: s 16-bit LDS reads and a shift
// each, against a table look-up, two divisions and a word read per bit in the general kernel.

// Walks idx = tid, tid + 256, .. as (row, column) pairs of rows `width` wide without dividing
// inside the loop.
struct RowWalk {
    uint32_t r, x, dr, dx, width;
    __device__ __forceinline__ RowWalk(uint32_t tid, uint32_t step, uint32_t w)
        : r(tid / w), x(tid % w), dr(step / w), dx(step % w), width(w) {}
    __device__ __forceinline__ void next()
    {
        r += dr;
        x += dx;
        if (x >= width) { x -= width; r++; }
    }
};

// Carrier counts that are a multiple of 16: a lane takes the QB units behind carriers c', c' + G,
// .. c' + 15 G (G = ncarriers / 16) -- they are the 16 bit positions of the same units -- so every
// bit position, rotation and shift below is a compile-time constant.  The units are never
// materialised: the encoder's two output streams A and B (one bit per input bit each) stay
// separate, and bit b of unit m is looked up where puncturing took it from:
//   rate 1/2 (keep 4 of 4): punctured bit k = coded bit k      -> stream b & 1, input bit 8 m + b / 2
//   rate 3/4 (keep 4 of 6): coded bit 6 (k / 4) + k % 4         -> stream b & 1, input bit 12 m + 3 (b / 4) + (b % 4) / 2
//   rate 2/3 (keep 3 of 4): coded bit 4 (k / 3) + k % 3, k = 16 m + b; three units span 32 input bits
//                            -> with j = 16 (m % 3) + b: stream (j % 3) & 1, input bit 32 (m / 3) + 2 (j / 3) + (j % 3) / 2
// (rawofdm/src/lib/qam/raw_conv.cc:150-159).  The lane keeps the stream pieces behind its QB units:
// QB bytes, QB 12-bit windows, or -- rate 2/3, where QB = 6 makes m % 3 = u % 3 -- two words.
template <int QB, int NC, int NP>
struct UnitBits {
    static constexpr int NPIECE = (NC == 4 && NP == 3) ? 2 : (QB == 1 ? 1 : QB);
    uint32_t a[NPIECE], b[NPIECE];

    // The QB units of one group of 16 carriers are punctured bits 16 QB c' .. of the symbol, i.e. a
    // run of IN_BITS = 8 QB NC / NP input bits that starts on a
    // byte boundary (for every rate of raw_qam.py:19-28 but (1, (6, 4)), marked "DONT USE" there): the
    // lane fetches those bytes and the byte in front (the encoder's six bits of history), encodes
    // them itself (raw_conv.cc:46-51) and cuts its pieces out of the two streams.
    static constexpr int IN_BITS = 8 * QB * NC / NP, IN_BYTES = IN_BITS / 8, IN_WORDS = (IN_BITS + 31) / 32;
    static constexpr int FULL = (IN_BYTES - 1) / 4, TAIL = IN_BYTES - 4 * FULL;      // dwords in front of the last byte's word
    // What a lane fetches for one group, untouched (so that nothing waits for the loads before the
    // values are needed): FULL dwords, the 1..4 bytes of the last word, the byte in front.
    struct Raw {
        uint32_t le[FULL > 0 ? FULL : 1];
        uint32_t tail[TAIL];
        uint32_t prev;
    };
    // p: the group's first input byte; have_prev: the byte in front belongs to the frame (else the
    // encoder starts from zero); have_last: so does the group's last byte (else it is the frame's
    // zero padding byte, raw_qam.py:83-85, which need not exist in memory).
    static __device__ __forceinline__ void fetch(Raw &r, const uint8_t *__restrict__ p, bool have_prev, bool have_last)
    {
        static_assert(IN_BITS % 8 == 0, "the group's input must be whole bytes");
        r.prev = 0;
        if (have_prev) r.prev = p[-1];
#pragma unroll
        for (int j = 0; j < FULL; j++) __builtin_memcpy(&r.le[j], p + 4 * j, 4);      // any alignment
#pragma unroll
        for (int i = 0; i < TAIL - 1; i++) r.tail[i] = p[4 * FULL + i];
        r.tail[TAIL - 1] = 0;
        if (have_last) r.tail[TAIL - 1] = p[IN_BYTES - 1];
    }
    __device__ __forceinline__ void encode(const Raw &r)
    {
        uint32_t A[IN_WORDS + 1], Bs[IN_WORDS + 1];
        uint32_t prev = r.prev;
#pragma unroll
        for (int j = 0; j < IN_WORDS; j++) {
            uint32_t cur;
            if (j < FULL) {
                cur = __builtin_bswap32(r.le[j]);
            } else {
                cur = 0;
#pragma unroll
                for (int i = 0; i < TAIL; i++) cur |= r.tail[i] << (24 - 8 * i);
            }
            const uint32_t y1 = __funnelshift_r(cur, prev, 1), y3 = __funnelshift_r(cur, prev, 3);
            const uint32_t y4 = __funnelshift_r(cur, prev, 4), y5 = __funnelshift_r(cur, prev, 5);
            const uint32_t y6 = __funnelshift_r(cur, prev, 6);
            A[j] = cur ^ y3 ^ y4 ^ y5 ^ y6;                       // polynomial 121: u[n], u[n-3..n-6]
            Bs[j] = cur ^ y1 ^ y3 ^ y4 ^ y6;                      // polynomial 91: u[n], u[n-1], u[n-3], u[n-4], u[n-6]
            prev = cur;
        }
        A[IN_WORDS] = 0;
        Bs[IN_WORDS] = 0;
        if constexpr (NC == 4 && NP == 4) {
#pragma unroll
            for (int u = 0; u < NPIECE; u++) {
                a[u] = (A[u >> 2] >> (24 - 8 * (u & 3))) & 0xFFu;
                b[u] = (Bs[u >> 2] >> (24 - 8 * (u & 3))) & 0xFFu;
            }
        } else if constexpr (NC == 6) {
#pragma unroll
            for (int u = 0; u < NPIECE; u++) {
                const int i = (12 * u) >> 5, sh = (12 * u) & 31;
                a[u] = (sh ? __funnelshift_l(A[i + 1], A[i], sh) : A[i]) >> 20;
                b[u] = (sh ? __funnelshift_l(Bs[i + 1], Bs[i], sh) : Bs[i]) >> 20;
            }
        } else {
            a[0] = A[0];
            a[1] = A[1];
            b[0] = Bs[0];
            b[1] = Bs[1];
        }
    }

    template <int U, int B>
    __device__ __forceinline__ uint32_t bit() const                       // bit B of unit U
    {
        if constexpr (NC == 4 && NP == 4) {
            return (((B & 1) ? b[U] : a[U]) >> (7 - B / 2)) & 1u;
        } else if constexpr (NC == 6) {
            constexpr int o = 3 * (B / 4) + (B % 4) / 2;
            return (((B & 1) ? b[U] : a[U]) >> (11 - o)) & 1u;
        } else {
            constexpr int j = 16 * (U % 3) + B, n_off = 2 * (j / 3) + (j % 3) / 2;
            return ((((j % 3) & 1) ? b[U / 3] : a[U / 3]) >> (31 - n_off)) & 1u;
        }
    }
};

// index of one axis (A = 0: I, 1: Q) of the carrier at bit position B: output bit t of the group
// is stream bit (t + B) % S of the group; first bit = MSB
template <int QB, int NC, int NP, int S, int B, int A, int... T>
__device__ __forceinline__ uint32_t axis_index(const UnitBits<QB, NC, NP> &ub, std::integer_sequence<int, T...>)
{
    return ((ub.template bit<A * S + (T + B) % S, B>() << (S - 1 - T)) | ...);
}

// one byte per carrier value: (I index << 3) | Q index
template <int QB, int NC, int NP, int S, int... B>
__device__ __forceinline__ void emit_unit_group(const UnitBits<QB, NC, NP> &ub, uint8_t *dst, uint32_t g,
                                                std::integer_sequence<int, B...>)
{
    using bits = std::make_integer_sequence<int, S>;
    ((dst[(uint32_t)B * g] = (uint8_t)((axis_index<QB, NC, NP, S, B, 0>(ub, bits{}) << 3) |
                                       (QB == 1 ? 0u : axis_index<QB, NC, NP, S, B, (QB == 1 ? 0 : 1)>(ub, bits{})))), ...);
}

// ---- the same through look-up tables (rates 1/2 and 3/4; LDS reads instead of a dozen VALU
// operations per carrier).  Bit position B of the group's units splits as B = 8 h + 2 kk + s: s picks
// the encoder stream the bit was punctured from, h the half of the unit's piece (a byte at rate 1/2,
// a 12-bit window at rate 3/4), kk = 0..3 the bit inside that half (at rate 3/4 a half is 6 bits of
// which positions 0, 1, 3, 4 survive the puncturing).  The 16 index bytes of a group are written as
// four words (s, h) of four bytes kk.  Unit U = S a + v (axis a, place v in the axis' group of S bits)
// feeds bit (S - 1 - ((v - B) mod S)) of its axis' index in byte B: for one word that is a fixed
// function of the half-piece, tabulated per (v, s, h): GROUP_LUT[((v * 2 + s) * 2 + h) << HB | half].
template <int QB, int NC, int NP>
struct GroupLut {
    static constexpr bool USED = !(NC == 4 && NP == 3);
    static constexpr int S = QB / 2 > 1 ? QB / 2 : 1, HB = NC == 6 ? 6 : 4;
    // byte kk of word (s, h) takes, from the unit at place v, the bit for index bit S - 1 - ((v - 8 h - s - 2 kk) mod S):
    // the table depends on (v, s, h) only through the class (v - 8 h - s) mod S
    static constexpr int WORDS = USED ? S * (1 << HB) : 1;
    static constexpr int cls(int v, int sbit, int h) { return (((v - 8 * h - sbit) % S) + S) % S; }

    static __device__ __forceinline__ void fill(uint32_t *lut, int tid, int nthr)
    {
        if constexpr (USED) {
            for (int e = tid; e < WORDS; e += nthr) {
                const int half = e & ((1 << HB) - 1), c = e >> HB;
                uint32_t w = 0;
                for (int kk = 0; kk < 4; kk++) {
                    const int pos = NC == 6 ? 3 * (kk / 2) + kk % 2 : kk;            // MSB-first place inside the half
                    const int t = (((c - 2 * kk) % S) + S) % S;
                    w |= (uint32_t)((half >> (HB - 1 - pos)) & 1) << (8 * kk + S - 1 - t);
                }
                lut[e] = w;
            }
        }
    }

    // the group's 16 index bytes (four words) from the pieces of its units
    static __device__ __forceinline__ void emit(const UnitBits<QB, NC, NP> &ub, const uint32_t *lut, uint32_t (&out)[4])
    {
        constexpr uint32_t MASK = ((1u << HB) - 1u) << 2;
#pragma unroll
        for (int sbit = 0; sbit < 2; sbit++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint32_t axis[2] = {0u, 0u};
#pragma unroll
                for (int U = 0; U < (QB == 1 ? 1 : QB); U++) {
                    const uint32_t piece = sbit ? ub.b[U] : ub.a[U];
                    // byte offset of the entry: (half-piece) * 4
                    const uint32_t off = h ? (piece << 2) & MASK : (piece >> (HB - 2)) & MASK;
                    const int v = U % S, a = U / S;
                    axis[a] |= *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(lut + (cls(v, sbit, h) << HB)) + off);
                }
                out[2 * sbit + h] = (axis[0] << 3) | axis[1];
            }
        }
    }

    // where carrier c of a symbol (c = B G + c') sits in the symbol's index row; byte kk of word (s, h) of
    // group c' is the carrier at B = 8 h + 2 kk + s.  PLANAR: the row is four planes (one per word (s, h))
    // of G dwords, so that neighbouring lanes write, and later read, neighbouring dwords (no bank
    // conflicts); else the four words of a group lie together (one 16-byte store, fewer registers).
    template <bool PLANAR>
    static __device__ __forceinline__ uint32_t offset_of(uint32_t c, uint32_t G)
    {
        if constexpr (USED) {
            const uint32_t B = c / G, cp = c - B * G, w = 2u * (B & 1u) + (B >> 3), kk = (B >> 1) & 3u;
            return PLANAR ? 4u * (w * G + cp) + kk : 16u * cp + 4u * w + kk;
        } else {
            return c;
        }
    }
};

// LDS areas of one pass of the bit chain (rows = OFDM symbols): staged frame words, coded words,
// punctured words (the coded ones again at rate 1/2), 4 words of bookkeeping per row.
struct QamBufs {
    uint32_t *in, *cod, *pun, *row;
    uint32_t in_stride, cod_stride, pun_stride;
};
struct QamWalks {
    RowWalk in, enc, pun;
};

// Frame bytes -> punctured coded bits of rows row0 .. row0 + rows_here - 1 in B.pun (stride: the coded
// stride at rate 1/2, else the punctured one).  Called by every thread of the workgroup; starts and
// ends with a barrier.
template <int NC, int NP>
__device__ __forceinline__ void qam_rows_to_punctured(const QamTxParams &Q, const uint8_t *__restrict__ data,
                                                      uint64_t row0, uint32_t rows_here, const QamBufs &B,
                                                      const QamWalks &wk, uint32_t in_per_row)
{
    constexpr uint32_t PIN = NC == 4 ? 4 : 3, POUT = NC == 4 ? 3 : 2;
    const uint32_t tid0 = threadIdx.x, nthr = blockDim.x;
    __syncthreads();                                       // previous rows consumed (and tables written)
    SEC("q_rows");
    for (uint32_t rr = tid0; rr < rows_here; rr += nthr) {
        // where the row's input starts: its first staged byte lies 4 bytes ahead of the row's first
        // input bit -- as an offset into the whole buffer (negative in front of the first frame;
        // the launchers keep the buffer below 2 GiB, so 32 bits hold it) and inside its frame
        const uint64_t row = row0 + rr, f = row / Q.nsymbols;
        const uint32_t sym = (uint32_t)(row - f * Q.nsymbols);
        const uint32_t n0 = sym * in_per_row;
        const int32_t fb0 = (int32_t)(n0 >> 3) - 4;
        B.row[4u * rr + 0] = (uint32_t)((int32_t)(f * Q.frame_bytes) + fb0);
        B.row[4u * rr + 1] = (uint32_t)fb0;
        B.row[4u * rr + 3] = 32u + (n0 & 7u);        // bit offset of the first input bit in the stage
    }
    __syncthreads();
    // ---- stage the frame bytes behind every row, zeros outside the frame (the padding byte
    // of raw_qam.py:85 and the encoder's zero start state), as MSB-first words.  Frames lie back
    // to back, so a word that straddles a frame boundary is loaded whole and masked; only at the
    // two ends of the buffer is the 4-byte window moved inside (and the bytes shifted back).  A
    // thread issues all its loads before it uses the first one.
    SEC("q_stage");
    const int32_t last_window = (int32_t)Q.total_bytes - 4;
    for (RowWalk w = wk.in; w.r < rows_here;) {
        constexpr int K = 4;
        uint32_t le[K], lds_at[K], mask[K];
        int shift[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            lds_at[k] = 0xFFFFFFFFu;
            if (w.r < rows_here) {
                const int32_t g = (int32_t)B.row[4u * w.r + 0] + 4 * (int32_t)w.x;      // buffer offset of the 4 bytes
                const int32_t fb = (int32_t)B.row[4u * w.r + 1] + 4 * (int32_t)w.x;     // the same inside the frame
                const int32_t gc = g < 0 ? 0 : (g > last_window ? last_window : g);
                __builtin_memcpy(&le[k], data + (uint32_t)gc, 4);                       // any alignment
                shift[k] = 8 * (g - gc);                      // bytes the window was moved by, as bits
                // bytes lo .. hi-1 of the four lie inside the frame (byte 0 = most significant after the swap)
                const int32_t lo = fb < 0 ? (fb < -4 ? 4 : -fb) : 0;
                const int32_t room = (int32_t)Q.frame_bytes - fb, hi = room > 4 ? 4 : (room < 0 ? 0 : room);
                const int32_t width = hi - lo;
                mask[k] = width >= 4 ? 0xFFFFFFFFu : width <= 0 ? 0u : (((1u << (8 * width)) - 1u) << (8 * (4 - hi)));
                lds_at[k] = w.r * B.in_stride + w.x;
            }
            w.next();
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (lds_at[k] != 0xFFFFFFFFu) {
                const uint32_t be = __builtin_bswap32(le[k]);
                const int sh = shift[k];
                // (|sh| >= 32 only where the mask is zero: the shift count may wrap)
                const uint32_t v = sh > 0 ? be << (sh & 31) : be >> ((-sh) & 31);
                B.in[lds_at[k]] = v & mask[k];
            }
        }
    }
    __syncthreads();
    // ---- rawofdm/src/lib/qam/raw_conv.cc:46-51, 32 input bits -> 64 coded bits per thread:
    // polynomial 121 taps u[n], u[n-3..n-6]; 91 taps u[n], u[n-1], u[n-3], u[n-4], u[n-6]
    SEC("q_enc");
    for (RowWalk w = wk.enc; w.r < rows_here; w.next()) {
        const uint32_t off = B.row[4u * w.r + 3] + 32u * w.x, q = off >> 5, sh = off & 31u;
        const uint32_t *W = B.in + w.r * B.in_stride;
        const uint32_t w0 = W[q - 1u], w1 = W[q], w2 = W[q + 1u];
        const uint32_t prev = sh ? __funnelshift_l(w1, w0, sh) : w0, cur = sh ? __funnelshift_l(w2, w1, sh) : w1;
        const uint32_t y1 = __funnelshift_r(cur, prev, 1), y3 = __funnelshift_r(cur, prev, 3);
        const uint32_t y4 = __funnelshift_r(cur, prev, 4), y5 = __funnelshift_r(cur, prev, 5);
        const uint32_t y6 = __funnelshift_r(cur, prev, 6);
        const uint32_t aw = cur ^ y3 ^ y4 ^ y5 ^ y6, bw = cur ^ y1 ^ y3 ^ y4 ^ y6;
        const uint64_t both = (spread_bits(aw) << 1) | spread_bits(bw);         // A before B, MSB first
        B.cod[w.r * B.cod_stride + 2u * w.x] = (uint32_t)(both >> 32);
        B.cod[w.r * B.cod_stride + 2u * w.x + 1u] = (uint32_t)both;
    }
    if constexpr (NP != NC) {
        __syncthreads();
        // ---- puncturing: 4 words -> 3 (keep 3 of 4) or 3 words -> 2 (keep 4 of 6) per thread
        for (RowWalk w = wk.pun; w.r < rows_here; w.next()) {
            const uint32_t *ci = B.cod + w.r * B.cod_stride + PIN * w.x;
            uint32_t *po = B.pun + w.r * B.pun_stride + POUT * w.x;
            if constexpr (NC == 4) {
                const uint32_t p0 = keep_3_of_4(ci[0]), p1 = keep_3_of_4(ci[1]), p2 = keep_3_of_4(ci[2]),
                               p3 = keep_3_of_4(ci[3]);
                po[0] = p0 | (p1 >> 24);
                po[1] = (p1 << 8) | (p2 >> 16);
                po[2] = (p2 << 16) | (p3 >> 8);
            } else {
                const uint32_t w0 = ci[0], w1 = ci[1], w2 = ci[2];
                po[0] = (keep_4_of_6(w0 >> 8) << 16) | keep_4_of_6(((w0 & 0xFFu) << 16) | (w1 >> 16));
                po[1] = (keep_4_of_6(((w1 & 0xFFFFu) << 8) | (w2 >> 24)) << 16) | keep_4_of_6(w2 & 0xFFFFFFu);
            }
        }
    }
    __syncthreads();
}

// rawofdm/src/lib/qam/raw_qam.cc:63-74: level of an axis index (first bit = MSB), S bits per axis
__device__ __forceinline__ float qam_level(uint32_t idx, uint32_t S, float scale)
{
    int pt = 0, flip = 1;
    for (uint32_t t = 0; t < S; t++) {
        const int bit = (int)((idx >> (S - 1u - t)) & 1u) * 2 - 1;
        pt = bit * flip + pt * 2;
        flip *= -bit;
    }
    return (float)pt * scale;
}

template <int QB, int NC, int NP>
__global__ void __launch_bounds__(256) k_raw_qam_tx_fast(QamTxParams Q, const uint8_t *__restrict__ data,
                                                         float2 *__restrict__ out, uint32_t in_stride,
                                                         uint32_t cod_stride, uint32_t pun_stride,
                                                         uint32_t rows_per_iter)
{
    constexpr uint32_t S = QB / 2 > 1 ? QB / 2 : 1;              // group size = bits per axis
    constexpr uint32_t AXES = QB == 1 ? 1 : 2;
    extern __shared__ uint32_t s_qam[];
    uint32_t *s_in = s_qam;                                       // frame bytes behind the rows, MSB-first words
    uint32_t *s_cod = s_in + in_stride * rows_per_iter;           // coded bits of the rows, MSB first
    constexpr bool ONE_AREA = NP == NC;
    uint32_t *s_pun = ONE_AREA ? s_cod : s_cod + cod_stride * rows_per_iter;
    const uint32_t p_stride = ONE_AREA ? cod_stride : pun_stride;
    uint32_t *s_row = s_cod + cod_stride * rows_per_iter + (ONE_AREA ? 0u : pun_stride * rows_per_iter);   // 4 words per row
    uint16_t *s_g = reinterpret_cast<uint16_t *>(s_row + 4u * rows_per_iter);
    __shared__ float s_lv[8];
    const uint32_t col = Q.cbps / 16u, n_groups = Q.cbps / S;
    for (uint32_t g = threadIdx.x; g < n_groups; g += blockDim.x) {
        const uint32_t i0 = S * g, b = i0 / col;
        s_g[g] = (uint16_t)(((i0 - b * col) << 4) | b);           // first unit of the group, bit position
    }
    if (threadIdx.x < (1u << S)) s_lv[threadIdx.x] = qam_level(threadIdx.x, S, Q.scale);

    const uint64_t n_rows = Q.n_values / Q.ncarriers;
    const uint32_t cod_per_row = Q.cbps / NP * NC;                // coded bits behind one symbol
    const uint32_t in_per_row = cod_per_row >> 1;                 // input bits behind one symbol
    const uint32_t words_per_row = (cod_per_row + 63u) / 64u;     // 32-bit input words feeding one symbol
    constexpr uint32_t PIN = NC == 4 ? 4 : 3;
    const uint32_t units = (cod_per_row + 32u * PIN - 1u) / (32u * PIN);
    const QamWalks walks = {RowWalk(threadIdx.x, blockDim.x, in_stride), RowWalk(threadIdx.x, blockDim.x, words_per_row),
                            RowWalk(threadIdx.x, blockDim.x, units)};
    const RowWalk walk_car(threadIdx.x, blockDim.x, Q.ncarriers);
    const QamBufs bufs = {s_in, s_cod, s_pun, s_row, in_stride, cod_stride, pun_stride};
    for (uint64_t row0 = (uint64_t)blockIdx.x * rows_per_iter; row0 < n_rows; row0 += (uint64_t)gridDim.x * rows_per_iter) {
        const uint32_t rows_here = n_rows - row0 < rows_per_iter ? (uint32_t)(n_rows - row0) : rows_per_iter;
        qam_rows_to_punctured<NC, NP>(Q, data, row0, rows_here, bufs, walks, in_per_row);
        // ---- interleaver + QAM, one thread per carrier value
        float2 *orow = out + row0 * Q.ncarriers;
        for (RowWalk w = walk_car; w.r < rows_here; w.next()) {
            // unit m of the MSB-first stream is the 16-bit element m ^ 1 of the little-endian words
            const uint16_t *H = reinterpret_cast<const uint16_t *>(s_pun + w.r * p_stride);
            float lv[2] = {0.f, 0.f};
#pragma unroll
            for (uint32_t a = 0; a < AXES; a++) {
                const uint32_t e = s_g[w.x * AXES + a];
                const uint32_t b = e & 15u, m0 = e >> 4;
                uint32_t n = 0;                                   // the group's bits in stream order
#pragma unroll
                for (uint32_t t = 0; t < S; t++) n = (n << 1) | (((uint32_t)H[(m0 + t) ^ 1u] >> (15u - b)) & 1u);
                // output bit t of the group is stream bit (t + b) % S: rotate left by b % S
                uint32_t rot = 0;
                if (S == 2) rot = b & 1u;
                if (S == 3) rot = b - 3u * ((b * 11u) >> 5);      // b < 16
                const uint32_t idx = S == 1 ? n : ((n << rot) | (n >> (S - rot))) & ((1u << S) - 1u);
                lv[a] = s_lv[idx];
            }
            orow[(uint64_t)w.r * Q.ncarriers + w.x] = make_float2(lv[0], lv[1]);
        }
    }
}

// t / d and t % d for small t (below 2^22) without an integer division: the quotient estimate from
// a float reciprocal is off by at most one, downwards, which one comparison repairs.
struct SmallDiv {
    uint32_t d;
    float inv;
    __device__ __forceinline__ explicit SmallDiv(uint32_t dd) : d(dd), inv(1.0f / (float)dd) {}
    __device__ __forceinline__ void divmod(uint32_t t, uint32_t &q, uint32_t &r) const
    {
        q = (uint32_t)((float)t * inv);
        r = t - q * d;
        if ((int32_t)r < 0) { q--; r += d; }
        if (r >= d) { q++; r -= d; }
    }
};

// ---- pieces shared by the fused bytes -> IQ kernels --------------------------------------------
// Index bytes of a pass: one row of IDX_PAD extra bytes behind the ncarriers level-index bytes of
// every symbol; the extra bytes hold the LUT slots of the three values a bin can have without
// carrying data (zero, pilot +1, pilot -1), so that the modulator fetches EVERY bin the same way:
// one byte from the row, one complex value from the LUT -- no per-bin case distinction.
constexpr uint32_t IDX_PAD = 16;          // keeps rows 16-byte aligned
constexpr uint32_t LUT_ZERO = 64, LUT_PLUS = 65, LUT_MINUS = 66, FUSED_LUT = 72;

// complex value of every index byte (I index << 3 | Q index) plus the three fixed slots
template <int QB>
__device__ __forceinline__ void fused_lut_fill(v2f *s_lut, float scale, int tid, int nthr)
{
    constexpr uint32_t QS = QB / 2 > 1 ? QB / 2 : 1;
    for (int i = tid; i < (int)FUSED_LUT; i += nthr) {
        v2f v = {0.f, 0.f};
        if (i < 64) {
            const uint32_t hi = (uint32_t)i >> 3, lo = (uint32_t)i & 7u;
            if (hi < (1u << QS) && lo < (1u << QS)) v = (v2f){qam_level(hi, QS, scale), QB == 1 ? 0.f : qam_level(lo, QS, scale)};
        } else if (i == (int)LUT_PLUS) {
            v.x = 1.f;
        } else if (i == (int)LUT_MINUS) {
            v.x = -1.f;
        }
        s_lut[i] = v;
    }
}

// byte offset, inside a symbol's index row, of the byte that selects the value of bin code `code`
// (s_src: >= 0 data carrier, -1 zero, -2 pilot +1, -3 pilot -1)
template <int QB, int NC, int NP, bool PLANAR>
__device__ __forceinline__ uint32_t fused_idx_offset(int code, uint32_t ncarriers)
{
    return code >= 0 ? GroupLut<QB, NC, NP>::template offset_of<PLANAR>((uint32_t)code, ncarriers / 16u)
                     : ncarriers + (uint32_t)(-1 - code);
}

// the three fixed slots behind a row's index bytes
__device__ __forceinline__ void fused_idx_fixed(uint8_t *row_pad)
{
    row_pad[0] = (uint8_t)LUT_ZERO;
    row_pad[1] = (uint8_t)LUT_PLUS;
    row_pad[2] = (uint8_t)LUT_MINUS;
}

// ---- bytes -> IQ in one kernel: the bit chain above feeding the modulator, the carrier values never
// leave the CU.  Every wave runs the bit chain of its own symbols inside its own LDS and modulates
// them, handing over inside the wave only -- no wave waits for another.  LDS per wave: SPW symbol
// areas, then the index rows of a pass (with one round per pass the SPW index rows sit at the END
// of the areas instead: they are in registers before the transform overwrites them).  A wave also
// copies the preamble rows of the frames that start in its symbols.  Needs whole frames of
// Q.nsymbols symbols (R.frame_size), R.data_tones == Q.ncarriers, a multiple of 16.
template <int L, int M, int QB>
__device__ __forceinline__ void fused_modulate_round(const RawParams &R, const uint8_t *irow, const uint32_t (&ioff)[M],
                                                     bool active, int slot, int t, int lane, v2f *T, const v2f *T_wave,
                                                     const v2f *s_tw, const v2f *s_tw1, const v2f *s_lut, uint32_t first_frame,
                                                     uint32_t first_in_frame, uint32_t syms_here, const SmallDiv &fdiv,
                                                     float2 *__restrict__ iq)
{
    constexpr int S = L > M ? L : M, SPW = 64 / S;
    const uint32_t row_len = R.n + R.cp;
    SEC("m_gather");
    // the round's first symbol is symbol first_in_frame of frame first_frame; this lane's is `slot` further on
    uint32_t row = ROW_DROPPED;
    if (active) {
        uint32_t df, sf;
        fdiv.divmod(first_in_frame + (uint32_t)slot, df, sf);
        row = (first_frame + df) * (R.frame_size + R.n_pre) + R.n_pre + sf;
    }
    // (every lane reads -- cheaper than masking the reads: lanes beyond L fetch the zero slot, slots
    //  beyond the pass's last symbol are given a row of the pass by the caller)
    v2f x[M];
#pragma unroll
    for (int m = 0; m < M; m++) x[bit_reverse(m, M)] = s_lut[irow[ioff[m]]];
    wave_sync();                                              // index bytes read: the areas may be overwritten
    raw_transform_store<L, M>(active, t, lane, row, x, T, T_wave, s_tw, s_tw1, R.n, R.cp, iq);
    SEC("m_pre");
    // ---- preamble rows of the frames that start in this wave's symbols
    // (gr-digital ofdm_insert_preamble as wired at rawofdm/src/python/raw_ofdm_mod.py:45,50-52)
#pragma unroll
    for (int sl = 0; sl < SPW; sl++) {
        uint32_t df, sf;
        fdiv.divmod(first_in_frame + (uint32_t)sl, df, sf);
        if ((uint32_t)sl >= syms_here || sf != 0u) continue;
        const size_t frow = (size_t)(first_frame + df) * (R.frame_size + R.n_pre);
        const uint32_t cnt = R.n_pre * row_len;
        if ((row_len & 1u) == 0u) {
            const float4 *src4 = reinterpret_cast<const float4 *>(R.pre_rows);
            const __amdgpu_buffer_rsrc_t win = iq_window(iq + frow * row_len);
            for (uint32_t i = lane; i < cnt / 2; i += 64) iq_store<RAW_ROW_AUX>(win, i, src4[i]);
        } else {
            float2 *dst = iq + frow * row_len;
            for (uint32_t i = lane; i < cnt; i += 64) dst[i] = R.pre_rows[i];
        }
    }
}

// groups of 16 carriers one lane works on per pass (the raw input of the NEXT pass waits in registers)
constexpr int FUSED_IT = 3;

template <int L, int M, int W, int QB, int NC, int NP>
__global__ void __launch_bounds__(64 * W)
k_raw_tx(RawParams R, QamTxParams Q, const uint8_t *__restrict__ data, float2 *__restrict__ iq, uint32_t rounds,
         uint32_t wave_bytes)
{
    constexpr int N = L * M, S = L > M ? L : M, SPW = 64 / S, TS = raw_area<L, M>();
    // layout of the index rows: planar from N = 256 on (2-4 % faster; not for BPSK, whose single unit
    // per group makes the four stores cost more than the reads gain); the small sizes run four
    // workgroups to a CU and have no registers to spare for it
    constexpr bool PLANAR = N >= 256 && QB > 1;
    using Units = UnitBits<QB, NC, NP>;
    using GLut = GroupLut<QB, NC, NP>;
    extern __shared__ __align__(16) unsigned char s_raw[];
    v2f *s_tw = reinterpret_cast<v2f *>(s_raw);                            // N
    int16_t *s_src = reinterpret_cast<int16_t *>(s_tw + N);                // N
    __shared__ v2f s_lut[FUSED_LUT];
    __shared__ uint32_t s_glut[GLut::WORDS];
    const uint32_t idx_stride = Q.ncarriers + IDX_PAD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < N; i += 64 * W) {
        const float2 w = R.tw[i];
        s_tw[i] = (v2f){w.x, w.y} * raw_tw_factor<L, M>();
        s_src[i] = R.src[i];
    }
    fused_lut_fill<QB>(s_lut, Q.scale, tid, 64 * W);
    GLut::fill(s_glut, tid, 64 * W);
    __shared__ v2f s_tw1[raw_tw1_size<L, M>()];
    if constexpr (L == 64) {
        __syncthreads();
        raw_tw1_fill<L, M>(s_tw1, s_tw, tid, 64 * W);
    }
    __syncthreads();                                              // the last workgroup barrier

    // this wave's LDS: SPW symbol areas, then (rounds > 1) the index rows of a pass; with one round
    // per pass the SPW index rows lie at the end of the areas instead (they are in registers before
    // the transform overwrites them)
    v2f *T_wave = reinterpret_cast<v2f *>(s_raw + (size_t)N * (sizeof(float2) + sizeof(int16_t)) + (size_t)wave * wave_bytes);
    uint8_t *idx_wave = rounds > 1u ? reinterpret_cast<uint8_t *>(T_wave + SPW * TS)
                                    : reinterpret_cast<uint8_t *>(T_wave + SPW * TS) - SPW * idx_stride;
    const uint32_t rows_wave = rounds * SPW;                      // symbols per pass of this wave
    const uint32_t G = Q.ncarriers / 16u;
    const uint32_t sym_bytes = G * Units::IN_BYTES;               // input bytes behind one symbol
    const int t = lane % S, slot = lane / S;
    v2f *T = T_wave + slot * TS;
    // where the index byte of each of the lane's M bins sits in a symbol's row: the same for every symbol
    uint32_t ioff[M];
#pragma unroll
    for (int m = 0; m < M; m++) ioff[m] = fused_idx_offset<QB, NC, NP, PLANAR>(t < L ? s_src[t + L * m] : -1, Q.ncarriers);
    // the lane's groups of a pass: (row, c') of item lane + 64 it -- the same in every pass
    const SmallDiv gdiv(G), fdiv(R.frame_size);                   // (R.frame_size == Q.nsymbols)
    uint32_t g_row[FUSED_IT], g_c[FUSED_IT];
#pragma unroll
    for (int it = 0; it < FUSED_IT; it++) gdiv.divmod((uint32_t)lane + 64u * it, g_row[it], g_c[it]);

    const uint32_t n_rows = R.n_sym;
    const uint32_t n_items = (n_rows + rows_wave - 1u) / rows_wave;
    const uint32_t step = gridDim.x * W;
    typename Units::Raw raw[FUSED_IT];
    // fetch of the pass that starts at symbol row0 (nothing is waited for here)
    auto fetch_pass = [&](uint32_t row0) {
        const uint32_t rows_here = n_rows - row0 < rows_wave ? n_rows - row0 : rows_wave;
        const uint32_t f0 = row0 / R.frame_size, s0 = row0 - f0 * R.frame_size;      // wave-uniform
#pragma unroll
        for (int it = 0; it < FUSED_IT; it++) {
            if (g_row[it] < rows_here) {
                uint32_t f, sf;
                fdiv.divmod(s0 + g_row[it], f, sf);
                const uint8_t *p = data + (size_t)(f0 + f) * Q.frame_bytes + sf * sym_bytes + g_c[it] * (uint32_t)Units::IN_BYTES;
                Units::fetch(raw[it], p, (sf | g_c[it]) != 0u, !(sf == Q.nsymbols - 1u && g_c[it] == G - 1u));
            }
        }
    };
    uint32_t item = blockIdx.x * W + wave;
    if (item < n_items) fetch_pass(item * rows_wave);
    for (; item < n_items; item += step) {
        const uint32_t row0 = item * rows_wave;
        const uint32_t rows_here = n_rows - row0 < rows_wave ? n_rows - row0 : rows_wave;
        const uint32_t f0 = row0 / R.frame_size, s0 = row0 - f0 * R.frame_size;
        wave_sync();                                              // the previous pass is done with the areas
        // ---- bit chain, one lane per group of 16 carriers (carriers c', c' + G, .. c' + 15 G of a symbol)
        SEC("q_idx");
#pragma unroll
        for (int it = 0; it < FUSED_IT; it++) {
            if (g_row[it] < rows_here) {
                Units ub;
                ub.encode(raw[it]);
                uint8_t *irow = idx_wave + g_row[it] * idx_stride;
                if constexpr (GLut::USED) {
                    uint32_t out[4];
                    GLut::emit(ub, s_glut, out);
                    if constexpr (PLANAR) {
                        uint32_t *plane = reinterpret_cast<uint32_t *>(irow) + g_c[it];
#pragma unroll
                        for (int w = 0; w < 4; w++) plane[(uint32_t)w * G] = out[w];
                    } else {
                        *reinterpret_cast<uint4 *>(irow + 16u * g_c[it]) = make_uint4(out[0], out[1], out[2], out[3]);
                    }
                } else {
                    emit_unit_group<QB, NC, NP, (QB / 2 > 1 ? QB / 2 : 1)>(ub, irow + g_c[it], G, std::make_integer_sequence<int, 16>{});
                }
                if (g_c[it] == 0) fused_idx_fixed(irow + Q.ncarriers);
            }
        }
        wave_sync();                                              // index bytes complete
        // the next pass's input is on its way while this one is modulated
        if (item + step < n_items) fetch_pass((item + step) * rows_wave);
        // ---- modulator, SPW symbols per round
        for (uint32_t sub = 0; sub < rows_here; sub += (uint32_t)SPW) {
            const uint32_t r_in = sub + (uint32_t)slot;
            fused_modulate_round<L, M, QB>(R, idx_wave + (r_in < rows_here ? r_in : sub) * idx_stride, ioff, r_in < rows_here, slot, t, lane, T, T_wave,
                                           s_tw, s_tw1, s_lut, f0, s0 + sub, rows_here - sub, fdiv, iq);
            wave_sync();                                          // the areas are rewritten in the next round
        }
    }
}

// ---- the bit chain alone in the wave-local form of k_raw_tx: bytes -> carrier values.  A lane takes a
// group of 16 carriers (its input bytes straight from memory, encoded in registers, index bytes by
// table), the wave's index bytes of a pass lie in LDS in carrier order, and the wave writes the values
// of the pass -- consecutive symbols, so one contiguous run -- with 16 bytes per lane.  No wave waits
// for another; the input of the next pass is fetched before the values of this one are written.
// Needs ncarriers % 16 == 0 and a rate whose groups start on byte boundaries (all but (1, (6, 4))).
template <int W, int QB, int NC, int NP>
__global__ void __launch_bounds__(64 * W)
k_raw_qam_wave(QamTxParams Q, const uint8_t *__restrict__ data, float2 *__restrict__ out, uint32_t rows_wave)
{
    using Units = UnitBits<QB, NC, NP>;
    using GLut = GroupLut<QB, NC, NP>;
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ v2f s_lut[FUSED_LUT];
    __shared__ uint32_t s_glut[GLut::WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fused_lut_fill<QB>(s_lut, Q.scale, tid, 64 * W);
    GLut::fill(s_glut, tid, 64 * W);
    __syncthreads();                                              // the only workgroup barrier

    uint8_t *idx_wave = s_raw + (size_t)wave * rows_wave * Q.ncarriers;
    const uint32_t G = Q.ncarriers / 16u;
    const uint32_t sym_bytes = G * Units::IN_BYTES;               // input bytes behind one symbol
    const SmallDiv gdiv(G), fdiv(Q.nsymbols);
    uint32_t g_row[FUSED_IT], g_c[FUSED_IT];                      // the lane's groups of a pass: (row, c')
#pragma unroll
    for (int it = 0; it < FUSED_IT; it++) gdiv.divmod((uint32_t)lane + 64u * it, g_row[it], g_c[it]);

    const uint32_t n_rows = (uint32_t)(Q.n_values / Q.ncarriers);
    const uint32_t n_items = (n_rows + rows_wave - 1u) / rows_wave;
    const uint32_t step = gridDim.x * W;
    typename Units::Raw raw[FUSED_IT];
    auto fetch_pass = [&](uint32_t row0) {
        const uint32_t rows_here = n_rows - row0 < rows_wave ? n_rows - row0 : rows_wave;
        const uint32_t f0 = row0 / Q.nsymbols, s0 = row0 - f0 * Q.nsymbols;          // wave-uniform
#pragma unroll
        for (int it = 0; it < FUSED_IT; it++) {
            if (g_row[it] < rows_here) {
                uint32_t f, sf;
                fdiv.divmod(s0 + g_row[it], f, sf);
                const uint8_t *p = data + (size_t)(f0 + f) * Q.frame_bytes + sf * sym_bytes + g_c[it] * (uint32_t)Units::IN_BYTES;
                Units::fetch(raw[it], p, (sf | g_c[it]) != 0u, !(sf == Q.nsymbols - 1u && g_c[it] == G - 1u));
            }
        }
    };
    uint32_t item = blockIdx.x * W + wave;
    if (item < n_items) fetch_pass(item * rows_wave);
    for (; item < n_items; item += step) {
        const uint32_t row0 = item * rows_wave;
        const uint32_t rows_here = n_rows - row0 < rows_wave ? n_rows - row0 : rows_wave;
        wave_sync();                                              // the previous pass has been written out
#pragma unroll
        for (int it = 0; it < FUSED_IT; it++) {
            if (g_row[it] < rows_here) {
                Units ub;
                ub.encode(raw[it]);
                uint8_t *irow = idx_wave + g_row[it] * Q.ncarriers + g_c[it];
                if constexpr (GLut::USED) {
                    uint32_t w4[4];
                    GLut::emit(ub, s_glut, w4);
                    // byte kk of word (s, h) is the carrier at bit position 8 h + 2 kk + s: carrier c' + G (8 h + 2 kk + s)
#pragma unroll
                    for (int sbit = 0; sbit < 2; sbit++)
#pragma unroll
                        for (int h = 0; h < 2; h++)
#pragma unroll
                            for (int kk = 0; kk < 4; kk++)
                                irow[(uint32_t)(8 * h + 2 * kk + sbit) * G] = (uint8_t)(w4[2 * sbit + h] >> (8 * kk));
                } else {
                    emit_unit_group<QB, NC, NP, (QB / 2 > 1 ? QB / 2 : 1)>(ub, irow, G, std::make_integer_sequence<int, 16>{});
                }
            }
        }
        wave_sync();                                              // index bytes complete
        if (item + step < n_items) fetch_pass((item + step) * rows_wave);
        const uint32_t n_pairs = rows_here * Q.ncarriers / 2u;    // (ncarriers is even)
        const __amdgpu_buffer_rsrc_t win = iq_window(out + (size_t)row0 * Q.ncarriers);
        for (uint32_t i = lane; i < n_pairs; i += 64u) {
            const uint32_t v = *reinterpret_cast<const uint16_t *>(idx_wave + 2u * i);
            const v2f a = s_lut[v & 0xFFu], b = s_lut[v >> 8];
            iq_store<RAW_VAL_AUX>(win, i, make_float4(a.x, a.y, b.x, b.y));
        }
    }
}

// The tap points behind the mapper (rawofdm/src/python/raw_ofdm_mod.py:58-66), formed from what the
// modulator wrote: the preamble inserter's output (mapped symbols and the given preamble symbols
// in row order), the inverse transform's output (a row without its prefix, before the 1/sqrt(N)
// block) and the prefixer's output (the rows before that block).  One workgroup per row.
__global__ void __launch_bounds__(256)
k_raw_taps(RawParams R, uint32_t n_rows, float unscale, const float2 *__restrict__ pre_freq,
           const float2 *__restrict__ mapped, const float2 *__restrict__ iq, float2 *__restrict__ pre,
           float2 *__restrict__ ifft, float2 *__restrict__ cp)
{
    const uint32_t row_len = R.n + R.cp;
    const uint64_t n_items = (uint64_t)n_rows + (pre ? (uint64_t)R.n_sym + (uint64_t)R.n_frames * R.n_pre : 0u);
    for (uint64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        if (item < n_rows) {
            const float2 *src = iq + item * row_len;
            for (uint32_t i = threadIdx.x; i < row_len; i += 256) {
                const float2 v = make_float2(src[i].x * unscale, src[i].y * unscale);
                if (cp) cp[item * row_len + i] = v;
                if (ifft && i >= R.cp) ifft[item * R.n + (i - R.cp)] = v;
            }
            continue;
        }
        uint64_t j = item - n_rows;
        const float2 *src;
        uint64_t row;
        if (j < R.n_sym) {
            row = R.frame_size ? (j / R.frame_size) * (R.frame_size + R.n_pre) + R.n_pre + j % R.frame_size
                               : (R.sym_row[j] == ROW_DROPPED ? ~0ull : (uint64_t)R.sym_row[j]);
            if (row == ~0ull) continue;
            src = mapped + j * R.n;
        } else {
            j -= R.n_sym;
            const uint64_t f = j / R.n_pre, k = j % R.n_pre;
            row = (R.frame_size ? f * (R.frame_size + R.n_pre) : (uint64_t)R.frame_row[f]) + k;
            src = pre_freq + k * R.n;
        }
        for (uint32_t i = threadIdx.x; i < R.n; i += 256) pre[row * R.n + i] = src[i];
    }
}

}  // namespace

struct ofdmtx_raw_plan {
    ofdmtx_ctx *ctx;
    int device;
    RawParams rp;
    ofdmtx_raw_sizes sizes;
    int16_t *d_src;
    float2 *d_tw;
    float2 *d_pre_rows;
    float2 *d_pre_freq;          // the preamble symbols as given (n_pre x N bins): the "tx-pre.dat" tap
    uint32_t *d_sym_row;
    uint32_t *d_frame_row;
};

namespace {

// Tables shared by both plan flavours; transforms the preamble symbols once.
int raw_plan_common(ofdmtx_ctx *ctx, uint32_t n, uint32_t cp, const int32_t *carrier_map, const float *preambles,
                    uint32_t n_pre, ofdmtx_raw_plan **out)
{
    if (!ctx || !out || !carrier_map || (n_pre && !preambles)) return OFDMTX_EINVAL;
    *out = nullptr;
    if (n != 64 && n != 128 && n != 256 && n != 512 && n != 1024) return OFDMTX_ESHAPE;
    if (cp > n) return OFDMTX_EINVAL;
    // rawofdm/src/lib/ofdm/raw_ofdm_mapper.cc:32-40: data carriers and pilots in index order;
    // :62-66 pilots alternate +1, -1, ... over the pilot carriers, the same in every symbol
    std::vector<int16_t> code(n), src(n);
    uint32_t n_data = 0, n_pil = 0;
    for (uint32_t i = 0; i < n; i++) {
        switch (carrier_map[i]) {
        case 0: code[i] = -1; break;
        case 1: code[i] = (int16_t)n_data++; break;
        case 2: code[i] = (n_pil++ & 1u) ? -3 : -2; break;
        default: return OFDMTX_ESHAPE;          // "carrier_map must include only 0,1,2"
        }
    }
    // gr.fft_vcc(N, False, [], True) (rawofdm/src/python/raw_ofdm_mod.py:46): the transform's bin k
    // is input element (k + N/2) mod N
    for (uint32_t k = 0; k < n; k++) src[k] = code[(k + n / 2) % n];
    std::vector<float2> tw(n);
    const double scale = (double)(float)(1.0 / sqrt((double)n));       // raw_ofdm_mod.py:48, a float32 constant
    for (uint32_t i = 0; i < n; i++) {
        const double a = 2.0 * M_PI * (double)i / (double)n;
        tw[i] = make_float2((float)(cos(a) * scale), (float)(sin(a) * scale));
    }

    ofdmtx_raw_plan *pl = new (std::nothrow) ofdmtx_raw_plan();
    if (!pl) return OFDMTX_ENOMEM;
    memset(pl, 0, sizeof(*pl));
    pl->ctx = ctx;
    pl->device = ctx->device;
    RawParams &rp = pl->rp;
    rp.n = n;
    rp.cp = cp;
    rp.data_tones = n_data;
    rp.n_pre = n_pre;
    const uint32_t row_len = n + cp;
    int16_t *d_ident = nullptr;
    float2 *d_freq = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(&pl->d_src, n * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc(&pl->d_tw, n * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc(&pl->d_pre_rows, (size_t)(n_pre ? n_pre : 1) * row_len * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(pl->d_src, src.data(), n * sizeof(int16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d_tw, tw.data(), n * sizeof(float2), hipMemcpyHostToDevice);
    int rc = OFDMTX_OK;
    if (e == hipSuccess && n_pre) {
        // the preamble symbols are full N-bin vectors: run them through the same kernel with an
        // identity carrier map (every bin a "data" carrier)
        std::vector<int16_t> ident(n);
        for (uint32_t k = 0; k < n; k++) ident[k] = (int16_t)((k + n / 2) % n);
        e = hipMalloc(&d_ident, n * sizeof(int16_t));
        if (e == hipSuccess) e = hipMalloc(&d_freq, (size_t)n_pre * n * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(d_ident, ident.data(), n * sizeof(int16_t), hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(d_freq, preambles, (size_t)n_pre * n * sizeof(float2), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            RawParams pp = rp;
            pp.data_tones = n;
            pp.n_pre = 0;
            pp.n_sym = n_pre;
            pp.n_frames = 0;
            pp.frame_size = n_pre;              // row = symbol index
            pp.src = d_ident;
            pp.tw = pl->d_tw;
            rc = launch_raw_n(ctx, pp, d_freq, nullptr, pl->d_pre_rows, nullptr);
            if (rc == OFDMTX_OK) e = hipDeviceSynchronize();
        }
    }
    if (d_ident) (void)hipFree(d_ident);
    pl->d_pre_freq = d_freq;
    if (e != hipSuccess || rc != OFDMTX_OK) {
        if (e != hipSuccess) {
            ofdmtx_g_last_hip = (int)e;
            rc = OFDMTX_EHIP;
        }
        ofdmtx_raw_plan_destroy(pl);
        return rc;
    }
    rp.src = pl->d_src;
    rp.tw = pl->d_tw;
    rp.pre_rows = pl->d_pre_rows;
    pl->sizes.fft_length = n;
    pl->sizes.cp_length = cp;
    pl->sizes.data_tones = n_data;
    pl->sizes.n_pre = n_pre;
    *out = pl;
    return OFDMTX_OK;
}

void raw_finish_sizes(ofdmtx_raw_plan *pl, uint64_t n_sym, uint64_t n_frames, uint64_t n_rows)
{
    pl->sizes.n_sym = n_sym;
    pl->sizes.n_frames = n_frames;
    pl->sizes.n_rows = n_rows;
    pl->sizes.n_samples = n_rows * (pl->rp.n + pl->rp.cp);
}

}  // namespace

// launch of the fused bytes -> IQ kernel for one FFT size; W waves per workgroup, `rounds` rounds of
// 64 / max(L, M) symbols per wave and pass
template <int L, int M, int W>
static int launch_raw_tx(const ofdmtx_ctx *ctx, const RawParams &rp, const QamTxParams &Q, int key, const uint8_t *d_bytes,
                         float2 *iq, uint32_t rounds, hipStream_t st)
{
    constexpr int N = L * M, S = L > M ? L : M, SPW = 64 / S, TS = raw_area<L, M>();
    const size_t wave_area = (size_t)SPW * TS * sizeof(float2), idx_row = Q.ncarriers + IDX_PAD;
    // as many rounds per pass as FUSED_IT groups per lane cover (64 FUSED_IT groups of 16 carriers per wave)
    const uint32_t G = Q.ncarriers / 16u, max_rounds = 64u * FUSED_IT / (SPW * G);
    if (max_rounds < 1u) return OFDMTX_ESHAPE;
    if (rounds < 1u || rounds > max_rounds) rounds = max_rounds;
    if (rounds == 1u && SPW * idx_row >= wave_area) return OFDMTX_ESHAPE;   // no room to alias the index rows
    const size_t wave_bytes = wave_area + (rounds > 1u ? (size_t)rounds * SPW * idx_row : 0u);
    const size_t lds = (size_t)N * (sizeof(float2) + sizeof(int16_t)) + (size_t)W * wave_bytes;
    if (lds > 160u * 1024u - 1024u) return OFDMTX_ESHAPE;
    const uint64_t n_items = ((uint64_t)rp.n_sym + rounds * SPW - 1) / (rounds * SPW);
    uint64_t blocks = (n_items + W - 1) / W;
    hipFuncAttributes fa;
#define LAUNCH_RTX(QB_, NC_, NP_)                                                                          \
    do {                                                                                                   \
        const void *fn = reinterpret_cast<const void *>(&k_raw_tx<L, M, W, QB_, NC_, NP_>);                \
        HIPCHK(hipFuncGetAttributes(&fa, fn));                                                             \
        const uint64_t per_cu = 160u * 1024u / (lds + fa.sharedSizeBytes) ? 160u * 1024u / (lds + fa.sharedSizeBytes) : 1; \
        const uint64_t cap = (uint64_t)ctx->n_cu * (per_cu < 8 ? per_cu : 8);                              \
        if (blocks > cap) blocks = cap;                                                                    \
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL((k_raw_tx<L, M, W, QB_, NC_, NP_>), dim3((unsigned)blocks), dim3(64 * W), lds,  \
                           st, rp, Q, d_bytes, iq, rounds, (uint32_t)wave_bytes);                          \
    } while (0)
    switch (key) {
    case 144: LAUNCH_RTX(1, 4, 4); break;
    case 244: LAUNCH_RTX(2, 4, 4); break;
    case 264: LAUNCH_RTX(2, 6, 4); break;
    case 444: LAUNCH_RTX(4, 4, 4); break;
    case 464: LAUNCH_RTX(4, 6, 4); break;
    case 643: LAUNCH_RTX(6, 4, 3); break;
    case 664: LAUNCH_RTX(6, 6, 4); break;
    default: return OFDMTX_ESHAPE;      // (1, (6, 4)): a group's input is a byte and a half
    }
#undef LAUNCH_RTX
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

// launch of the wave-local bit chain; false if this rate cannot take it
template <int QB, int NC, int NP>
static bool launch_qam_wave(const ofdmtx_ctx *ctx, const QamTxParams &Q, const uint8_t *d_bytes, float2 *out, hipStream_t st)
{
    if constexpr ((8 * QB * NC / NP) % 8 != 0) {
        return false;                                             // (1, (6, 4)): a group's input is a byte and a half
    } else {
        constexpr int W = 8;
        const uint32_t G = Q.ncarriers / 16u;
        const uint32_t rows_wave = 64u * FUSED_IT / G;
        const size_t lds = (size_t)W * rows_wave * Q.ncarriers;
        const uint64_t n_rows = Q.n_values / Q.ncarriers;
        const uint64_t n_items = (n_rows + rows_wave - 1) / rows_wave;
        uint64_t blocks = (n_items + W - 1) / W;
        const uint64_t cap = (uint64_t)ctx->n_cu * 4u;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((k_raw_qam_wave<W, QB, NC, NP>), dim3((unsigned)blocks), dim3(64 * W), lds, st, Q, d_bytes, out,
                           rows_wave);
        return true;
    }
}

// fills the bit chain's parameters (shared by ofdmtx_raw_qam_tx and ofdmtx_raw_tx)
static int qam_params(int qambits, int nc, int np, uint32_t ncarriers, uint32_t nsymbols, uint64_t n_frames, QamTxParams *Q)
{
    uint32_t fb;
    int rc = ofdmtx_raw_framebytes(qambits, nc, np, (uint64_t)ncarriers * nsymbols, &fb);
    if (rc) return rc;
    Q->frame_bytes = fb;
    Q->ncarriers = ncarriers;
    Q->nsymbols = nsymbols;
    Q->qambits = (uint32_t)qambits;
    Q->nc = (uint32_t)nc;
    Q->np = (uint32_t)np;
    Q->cbps = (uint32_t)qambits * ncarriers;
    // the interleaver is a permutation only for whole columns (raw_intrlv.cc:26: cbps / 16) and whole groups of s
    const uint32_t s = qambits / 2 > 1 ? (uint32_t)qambits / 2 : 1u;
    if (Q->cbps % 16u != 0 || (Q->cbps / 16u) % s != 0) return OFDMTX_ESHAPE;
    // a symbol must start on a whole puncturing group (true for every table entry of raw_qam.py:19-28)
    if (Q->cbps % (uint32_t)np != 0) return OFDMTX_ESHAPE;
    Q->n_values = n_frames * (uint64_t)ncarriers * nsymbols;
    Q->total_bytes = n_frames * (uint64_t)fb;
    // raw_qam.cc:39-52,120-147: bits per axis and symbol power per constellation, float arithmetic as there
    const int numbits = qambits == 1 ? 1 : qambits / 2;
    const float power = qambits == 1 ? 1.0f : 0.5f;
    const int nn = 1 << (numbits - 1);
    const int sum2 = (4 * nn * nn * nn - nn) / 3;
    Q->scale = (float)sqrt(power * float(nn) / float(sum2));
    return OFDMTX_OK;
}

extern "C" {

int ofdmtx_raw_plan_frames(ofdmtx_ctx *ctx, uint32_t fft_length, uint32_t cp_length, const int32_t *carrier_map,
                           const float *preambles, uint32_t n_pre, uint64_t n_frames, uint32_t frame_size,
                           ofdmtx_raw_plan **out)
{
    if (frame_size == 0) return OFDMTX_EINVAL;
    const uint64_t n_sym = n_frames * frame_size, n_rows = n_frames * ((uint64_t)frame_size + n_pre);
    if (n_rows >= 0xFFFFFFF0ull) return OFDMTX_ETOOBIG;
    ofdmtx_raw_plan *pl;
    int rc = raw_plan_common(ctx, fft_length, cp_length, carrier_map, preambles, n_pre, &pl);
    if (rc) return rc;
    pl->rp.n_sym = (uint32_t)n_sym;
    pl->rp.n_frames = n_pre ? (uint32_t)n_frames : 0;
    pl->rp.frame_size = frame_size;
    raw_finish_sizes(pl, n_sym, n_frames, n_rows);
    *out = pl;
    return OFDMTX_OK;
}

int ofdmtx_raw_plan_flags(ofdmtx_ctx *ctx, uint32_t fft_length, uint32_t cp_length, const int32_t *carrier_map,
                          const float *preambles, uint32_t n_pre, const uint8_t *flags, uint64_t n_sym,
                          ofdmtx_raw_plan **out)
{
    if (n_sym && !flags) return OFDMTX_EINVAL;
    if (n_sym * ((uint64_t)n_pre + 1) >= 0xFFFFFFF0ull) return OFDMTX_ETOOBIG;
    ofdmtx_raw_plan *pl;
    int rc = raw_plan_common(ctx, fft_length, cp_length, carrier_map, preambles, n_pre, &pl);
    if (rc) return rc;
    // symbols in front of the first flagged one are dropped; a flagged symbol is preceded by
    // every preamble symbol
    std::vector<uint32_t> sym_row(n_sym ? n_sym : 1), frame_row;
    uint64_t row = 0;
    bool started = false;
    for (uint64_t j = 0; j < n_sym; j++) {
        if (flags[j] & 1u) {
            started = true;
            frame_row.push_back((uint32_t)row);
            row += n_pre;
        }
        sym_row[j] = started ? (uint32_t)row++ : ROW_DROPPED;
    }
    const size_t nf = frame_row.size();
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(&pl->d_sym_row, sym_row.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&pl->d_frame_row, (nf ? nf : 1) * sizeof(uint32_t));
    if (e == hipSuccess)
        e = hipMemcpy(pl->d_sym_row, sym_row.data(), sym_row.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && nf)
        e = hipMemcpy(pl->d_frame_row, frame_row.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ofdmtx_g_last_hip = (int)e;
        ofdmtx_raw_plan_destroy(pl);
        return OFDMTX_EHIP;
    }
    pl->rp.n_sym = (uint32_t)n_sym;
    pl->rp.n_frames = n_pre ? (uint32_t)nf : 0;
    pl->rp.frame_size = 0;
    pl->rp.sym_row = pl->d_sym_row;
    pl->rp.frame_row = pl->d_frame_row;
    raw_finish_sizes(pl, n_sym, nf, row);
    *out = pl;
    return OFDMTX_OK;
}

int ofdmtx_raw_plan_destroy(ofdmtx_raw_plan *pl)
{
    if (!pl) return OFDMTX_EINVAL;
    (void)hipSetDevice(pl->device);
    if (pl->d_src) (void)hipFree(pl->d_src);
    if (pl->d_tw) (void)hipFree(pl->d_tw);
    if (pl->d_pre_rows) (void)hipFree(pl->d_pre_rows);
    if (pl->d_pre_freq) (void)hipFree(pl->d_pre_freq);
    if (pl->d_sym_row) (void)hipFree(pl->d_sym_row);
    if (pl->d_frame_row) (void)hipFree(pl->d_frame_row);
    delete pl;
    return OFDMTX_OK;
}

int ofdmtx_raw_plan_get_sizes(const ofdmtx_raw_plan *pl, ofdmtx_raw_sizes *out)
{
    if (!pl || !out) return OFDMTX_EINVAL;
    *out = pl->sizes;
    return OFDMTX_OK;
}

int ofdmtx_raw_modulate(ofdmtx_ctx *ctx, const ofdmtx_raw_plan *pl, const float *d_data, float *d_mapped,
                        float *d_iq, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    if (pl->rp.n_sym == 0) return OFDMTX_OK;
    if (!d_iq || !d_data) return OFDMTX_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    return launch_raw_n(ctx, pl->rp, reinterpret_cast<const float2 *>(d_data), reinterpret_cast<float2 *>(d_mapped),
                        reinterpret_cast<float2 *>(d_iq), (hipStream_t)stream);
}

int ofdmtx_raw_taps(ofdmtx_ctx *ctx, const ofdmtx_raw_plan *pl, const float *d_mapped, const float *d_iq, float *d_pre,
                    float *d_ifft, float *d_cp, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx || !d_iq || (d_pre && !d_mapped && pl->rp.n_sym)) return OFDMTX_EINVAL;
    if (!d_pre && !d_ifft && !d_cp) return OFDMTX_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n_rows = pl->sizes.n_rows;
    const uint64_t items = n_rows + (d_pre ? pl->sizes.n_sym + (uint64_t)pl->rp.n_frames * pl->rp.n_pre : 0u);
    if (items == 0) return OFDMTX_OK;
    const uint64_t cap = (uint64_t)ctx->n_cu * 32u;
    // the block multiplies by the float32 constant 1/sqrt(N) (raw_ofdm_mod.py:48); the taps sit in front of it
    const float unscale = (float)sqrt((double)pl->rp.n);
    hipLaunchKernelGGL(k_raw_taps, dim3((unsigned)(items < cap ? items : cap)), dim3(256), 0, (hipStream_t)stream, pl->rp,
                       (uint32_t)n_rows, unscale, pl->d_pre_freq, reinterpret_cast<const float2 *>(d_mapped),
                       reinterpret_cast<const float2 *>(d_iq), reinterpret_cast<float2 *>(d_pre),
                       reinterpret_cast<float2 *>(d_ifft), reinterpret_cast<float2 *>(d_cp));
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_raw_framebytes(int qambits, int nc, int np, uint64_t size, uint32_t *out)
{
    // rawofdm/src/python/raw_qam.py:30-37 (np doubled for the rate-1/2 code), minus the padding byte
    if (!out || nc <= 0 || np <= 0 || np > nc) return OFDMTX_EINVAL;
    if (qambits != 1 && qambits != 2 && qambits != 4 && qambits != 6) return OFDMTX_ESHAPE;
    const uint64_t num = size * (uint64_t)qambits * (uint64_t)nc, den = (uint64_t)np * 2u * 8u;
    if (num % den != 0 || num / den < 1 || num / den > 0xFFFFFFFFull) return OFDMTX_ESHAPE;
    *out = (uint32_t)(num / den) - 1u;
    return OFDMTX_OK;
}

int ofdmtx_raw_qam_tx(ofdmtx_ctx *ctx, int qambits, int nc, int np, uint32_t ncarriers, uint32_t nsymbols,
                      const uint8_t *d_bytes, uint64_t n_frames, float *d_out, void *stream)
{
    if (!ctx || (n_frames && (!d_bytes || !d_out))) return OFDMTX_EINVAL;
    QamTxParams Q;
    int rc = qam_params(qambits, nc, np, ncarriers, nsymbols, n_frames, &Q);
    if (rc) return rc;
    if (Q.n_values == 0) return OFDMTX_OK;
    const uint32_t s = qambits / 2 > 1 ? (uint32_t)qambits / 2 : 1u;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n_rows = n_frames * (uint64_t)nsymbols;
    const uint64_t cap = (uint64_t)ctx->n_cu * 8u;
    const uint32_t cod_words = ((Q.cbps / (uint32_t)np * (uint32_t)nc + 63u) / 64u) * 2u;
    const uint32_t rows_per_iter = ncarriers >= 1024u ? 1u : 1024u / ncarriers;
    const size_t lds = (size_t)cod_words * rows_per_iter * sizeof(uint32_t) + (size_t)Q.cbps * sizeof(uint16_t);
    if (lds > 60000) return OFDMTX_ESHAPE;
    const unsigned threads = 256u;
    const uint64_t n_iters = (n_rows + rows_per_iter - 1) / rows_per_iter;
    const unsigned grid = (unsigned)(n_iters < cap ? n_iters : cap);
    hipStream_t st = (hipStream_t)stream;
    float2 *o2 = reinterpret_cast<float2 *>(d_out);
    // table entries of raw_qam.py:19-28.  Whole groups of 16 carriers per lane (at most 64 * FUSED_IT
    // groups per symbol, 16-byte stores): the wave-local form, k_raw_qam_wave; other carrier counts:
    // the workgroup form, k_raw_qam_tx_fast
    const uint32_t cod_bits = Q.cbps / (uint32_t)np * (uint32_t)nc;
    const bool wave_ok = ncarriers % 16u == 0 && ((uintptr_t)d_out & 15u) == 0 && ncarriers / 16u <= 64u * FUSED_IT &&
                         n_rows < 0xFFFFFFF0ull;
    // interleaved coded words (the puncturing pass reads whole units) and punctured words
    const uint32_t cod_stride = cod_words + 4u;
    const uint32_t pun_stride = nc == np ? 0u : nc == 4 ? 3u * ((cod_bits + 127u) / 128u) : 2u * ((cod_bits + 95u) / 96u);
    // staged input per row: 4 bytes ahead of the row's first input bit, the row's input bits, slack
    // for a bit offset and the word read past the end
    const uint32_t in_stride = (cod_bits / 2u + 31u) / 32u + 4u;
    const uint32_t row_words = in_stride + cod_stride + pun_stride + 4u;
    uint32_t rows_fast = rows_per_iter;
    // behind the rows: the group table (cbps / s entries)
    const size_t tail_fixed = (size_t)(Q.cbps / s) * sizeof(uint16_t);
    const size_t row_bytes = row_words * sizeof(uint32_t);
    const uint32_t rows_fit = tail_fixed < 40000u ? (uint32_t)((48000u - tail_fixed) / row_bytes) : 0u;
    if (rows_fast > rows_fit) rows_fast = rows_fit;
    const size_t lds_fast = row_bytes * rows_fast + tail_fixed + 16u;
    const bool fast_ok = rows_fast >= 1u && Q.total_bytes >= 4u && Q.total_bytes < 0x7FFFFFF0ull && (Q.cbps / s) * sizeof(uint16_t) < 40000u && Q.cbps / 16u < 4096u;
    const uint64_t iters_fast = fast_ok ? (n_rows + rows_fast - 1) / rows_fast : 1;
    const unsigned grid_fast = (unsigned)(iters_fast < cap ? iters_fast : cap);
#define LAUNCH_QAM(QB_, NC_, NP_)                                                                         \
    do {                                                                                                  \
        if (wave_ok && launch_qam_wave<QB_, NC_, NP_>(ctx, Q, d_bytes, o2, st))                           \
            break;                                                                                        \
        if (fast_ok)                                                                                      \
            hipLaunchKernelGGL((k_raw_qam_tx_fast<QB_, NC_, NP_>), dim3(grid_fast), dim3(threads),        \
                               lds_fast, st, Q, d_bytes, o2, in_stride, cod_stride, pun_stride,           \
                               rows_fast);                                                                \
        else                                                                                              \
            hipLaunchKernelGGL((k_raw_qam_tx<0, 0, 0>), dim3(grid), dim3(threads), lds, st, Q, d_bytes,   \
                               o2, cod_words, rows_per_iter);                                             \
    } while (0)
    const int key = qambits * 100 + nc * 10 + np;
    switch (key) {
    case 144: LAUNCH_QAM(1, 4, 4); break;
    case 164: LAUNCH_QAM(1, 6, 4); break;
    case 244: LAUNCH_QAM(2, 4, 4); break;
    case 264: LAUNCH_QAM(2, 6, 4); break;
    case 444: LAUNCH_QAM(4, 4, 4); break;
    case 464: LAUNCH_QAM(4, 6, 4); break;
    case 643: LAUNCH_QAM(6, 4, 3); break;
    case 664: LAUNCH_QAM(6, 6, 4); break;
    default:
        hipLaunchKernelGGL((k_raw_qam_tx<0, 0, 0>), dim3(grid), dim3(threads), lds, st, Q, d_bytes, o2, cod_words,
                           rows_per_iter);
        break;
    }
#undef LAUNCH_QAM
    HIPCHK(hipGetLastError());
    return OFDMTX_OK;
}

int ofdmtx_raw_tx(ofdmtx_ctx *ctx, const ofdmtx_raw_plan *pl, int qambits, int nc, int np, const uint8_t *d_bytes,
                  float *d_iq, void *stream)
{
    if (!ctx || !pl || pl->ctx != ctx) return OFDMTX_EINVAL;
    const RawParams &rp = pl->rp;
    if (rp.n_sym == 0) return OFDMTX_OK;
    if (!d_bytes || !d_iq) return OFDMTX_EINVAL;
    // whole frames of equal size, every data carrier fed by the bit chain, 16 carriers per thread
    if (rp.frame_size == 0 || rp.n_sym % rp.frame_size != 0 || rp.data_tones % 16u != 0) return OFDMTX_ESHAPE;
    if (((uintptr_t)d_iq & 15u) != 0) return OFDMTX_EINVAL;
    QamTxParams Q;
    int rc = qam_params(qambits, nc, np, rp.data_tones, rp.frame_size, rp.n_sym / rp.frame_size, &Q);
    if (rc) return rc;
    if (Q.total_bytes < 4u || Q.total_bytes >= 0x7FFFFFF0ull || Q.cbps / 16u >= 4096u) return OFDMTX_ESHAPE;
    HIPCHK(hipSetDevice(ctx->device));
    const int key = qambits * 100 + nc * 10 + np;
    float2 *iq = reinterpret_cast<float2 *>(d_iq);
    hipStream_t st = (hipStream_t)stream;
    // Waves per workgroup as measured per size (64-QAM r3/4, 1.05e6 / 2.6e5 symbols, one box, ms):
    //   N = 64: 4 -> 0.183 (2: 0.185, 8: 0.181, 12: 0.219)     N = 128: 4 -> 0.372 (2: 0.380, 8: 0.378, 12: 0.429)
    //   N = 256: 12 -> 0.684 (4: 0.715, 8: 0.770, 2: 0.845)    N = 512: 8 -> 0.484 (4: 0.558, 12: 0.559)
    //   N = 1024: 6 -> 1.065 (2: 1.075, 4: 1.206; the two kernels: 1.34)
    // Every wave takes as many rounds per pass as FUSED_IT groups per lane cover.
    switch (rp.n) {
    case 64: return launch_raw_tx<8, 8, 4>(ctx, rp, Q, key, d_bytes, iq, 0, st);
    case 128: return launch_raw_tx<8, 16, 4>(ctx, rp, Q, key, d_bytes, iq, 0, st);
    case 256: return launch_raw_tx<16, 16, 12>(ctx, rp, Q, key, d_bytes, iq, 0, st);
    case 512: return launch_raw_tx<64, 8, 16>(ctx, rp, Q, key, d_bytes, iq, 0, st);
    case 1024: return launch_raw_tx<64, 16, 12>(ctx, rp, Q, key, d_bytes, iq, 0, st);
    default: return OFDMTX_ESHAPE;
    }
}

}  // extern "C"
