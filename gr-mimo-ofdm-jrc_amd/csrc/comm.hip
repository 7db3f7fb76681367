// comm.hip — comm-side rows of the hot path: C1 mimo_ofdm_equalizer, C2 mimo_precoder, C3 steering
//
//   C1 replaces mimo_ofdm_equalizer_impl::general_work + helpers (reference lib/mimo_ofdm_equalizer_impl.cc:191-922)
//   C2 replaces mimo_precoder_impl::work + generate_signal_field   (reference lib/mimo_precoder_impl.cc:275-741, :985-1060)
//   C3 replaces the Eigen JacobiSVD / phased-steering bodies       (reference lib/mimo_precoder_impl.cc:846-861, :880-893, :961-974)
//
// All three are small per-subcarrier problems (K = N_tx <= 8): no GEMM shape, no MFMA.  The equalizer is a
// per-frame sequential state machine over OFDM symbols, so the parallel axes are subcarriers (lanes) and
// independent RX streams / frames (workgroups); its state lives in LDS for the duration of a launch.
#include "jrc_internal.h"
#include "fft_device.h"

#include <algorithm>
#include <type_traits>
#include <cmath>

// ------------------------------------------------------------------------------------------------
// scalar helpers restating libstdc++ / libgcc complex arithmetic (the reference's std::complex<float> ops)
__device__ __forceinline__ float2 c_mul(float2 a, float2 b)
{
#pragma clang fp contract(off)
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 c_conj(float2 a) { return make_float2(a.x, -a.y); }
// std::complex<float> operator/ = __divsc3.  A g++ build of the reference links it from libgcc_s.so.1 (ahead of the static libgcc), and that
// library (GCC >= 12; the image's is 12.3) forms the float quotient in double and rounds once — not Smith's method in float, which is what
// gcc <= 11 and its static libgcc.a have.  tests/test_second_source.py pins the formula against the box's libgcc_s bit for bit.
__host__ __device__ __forceinline__ float2 c_div(float2 n, float2 dn)
{
#pragma clang fp contract(off)
    const double a = n.x, b = n.y, c = dn.x, d = dn.y;
    const double denom = (c * c) + (d * d);
    return make_float2((float)(((a * c) + (b * d)) / denom), (float)(((b * c) - (a * d)) / denom));
}
__device__ __forceinline__ float2 c_expj(double x)   // std::exp(gr_complex(0, x)): the double is narrowed first
{
    const float xf = (float)x;
    float sn, cs;
    if (fabsf(xf) <= 0.78539816f) {
        // |x| <= pi/4 — every sampling-offset and residual-phase rotation of a sane link: no argument reduction, the two
        // single-precision minimax polynomials (Cephes sinf / cosf kernels, < 1 ulp here — the same error class as any libm's
        // sincosf, which is all the reference's std::exp guarantees)
        const float z = xf * xf;
        sn = fmaf(xf * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), xf);
        cs = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(z, -0.5f, 1.0f));
    } else {
        sincosf(xf, &sn, &cs);                         // one argument reduction for both
    }
    return make_float2(cs, sn);
}

__device__ __forceinline__ float2 c_expj_small(double x)   // c_expj for |(float)x| <= pi/4, which the caller has established: the same two polynomials, no test
{
    const float xf = (float)x;
    const float z = xf * xf;
    const float sn = fmaf(xf * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), xf);
    const float cs = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(z, -0.5f, 1.0f));
    return make_float2(cs, sn);
}

__host__ __device__ inline int popc8(int n) { int s = 0; for (int i = 0; i < 8; i++) s += (n >> i) & 1; return s; }

// ------------------------------------------------------------------------------------------------
// host-side SIG helpers (lib/utils.cc:26-111)
static int mcs_params(int mcs, int n_data, int* n_dbps, int* rate_field)
{
    int bpsc, num, den, rf;
    switch (mcs) {
        case 0: bpsc = 1; num = 1; den = 2; rf = 0x0D; break;
        case 1: bpsc = 1; num = 3; den = 4; rf = 0x0F; break;
        case 2: bpsc = 2; num = 1; den = 2; rf = 0x05; break;
        case 3: bpsc = 2; num = 3; den = 4; rf = 0x07; break;
        case 4: bpsc = 4; num = 1; den = 2; rf = 0x09; break;
        case 5: bpsc = 4; num = 3; den = 4; rf = 0x0B; break;
        default: return -1;
    }
    if (n_dbps) *n_dbps = n_data * bpsc * num / den;
    if (rate_field) *rate_field = rf;
    return 0;
}

__host__ __device__ inline int n_ofdm_sym_dev(int mcs, int n_data, int nbytes)
{
    int bpsc = mcs <= 1 ? 1 : (mcs <= 3 ? 2 : 4);
    int cbps = n_data * bpsc;
    int dbps = (mcs & 1) ? cbps * 3 / 4 : cbps / 2;
    return (int)ceil((16 + 8 * nbytes + 6) / (double)dbps);     // lib/utils.cc:31
}

extern "C" int jrc_n_ofdm_sym(int mcs, int n_data_carriers, int data_size_byte)
{
    if (mcs < 0 || mcs > 5 || n_data_carriers <= 0) return JRC_ERR_INVALID_ARG;
    return n_ofdm_sym_dev(mcs, n_data_carriers, data_size_byte);
}

extern "C" int jrc_sig_encode(int n_data, int mcs, int packet_type, int length, float* out_re)
{
    int rate_field;
    if (!out_re || n_data < 48 || mcs_params(mcs, n_data, nullptr, &rate_field) < 0) return JRC_ERR_INVALID_ARG;
    char hdr[24] = {0};                                                  // lib/mimo_precoder_impl.cc:1003-1038
    hdr[0] = (rate_field >> 3) & 1; hdr[1] = (rate_field >> 2) & 1; hdr[2] = (rate_field >> 1) & 1; hdr[3] = rate_field & 1;
    hdr[4] = (packet_type == 2) ? 1 : 0;
    for (int i = 0; i < 12; i++) hdr[5 + i] = (length >> i) & 1;
    int sum = 0;
    for (int i = 0; i < 17; i++) sum += hdr[i];
    hdr[17] = sum % 2;
    int state = 0;
    for (int i = 0; 2 * i + 1 < n_data; i++) {                           // convolutional_encoding, lib/utils.cc:207-217
        state = ((state << 1) & 0x7e) | (i < 24 ? hdr[i] : 0);
        out_re[2 * i] = (popc8(state & 0155) & 1) ? 1.0f : -1.0f;        // BPSK map: 0 -> -1, 1 -> +1
        out_re[2 * i + 1] = (popc8(state & 0117) & 1) ? 1.0f : -1.0f;
    }
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// C1 equalizer
#ifdef JRC_TIMING_EXPERIMENTS
#define EQ_EXP(d) ((d).exp)
#else
#define EQ_EXP(d) 0
#endif
struct EqDev {
    int N, cp, ND, NP, NAct, NL, T, mapped_cols, n_pilot_rows, estimator, lds_tables;
    int exp;            // timing experiments of docs/history.md §6 — WRONG RESULTS, so only in builds with -DJRC_TIMING_EXPERIMENTS (tools/ra_variants.py), where JRC_EQ_EXP selects:
                        // 1 = the pilot phase reads the batch's first symbol for every symbol; 2 = the MIMO-LTF symbols are not stored to / read from HBM;
                        // 3 = the equalisation phase reads the batch's first row for every symbol; 4 = no pilot phase; 5 = the equalisation phase stores its input cells; 6 = it stores nothing; 7 = it stores through the caches
    int sig_full;       // JRC_EQ_SIG_FULL: always run the windowed Viterbi on the SIG field (no codeword shortcut)
    int pre_lds;        // the MIMO-LTF symbols of a frame are kept in LDS (on the arrays that are dead between the SIG field and the data symbols) instead of HBM
    double freq, bw;
    const int* data_c; const int* pilot_c; const int* active_c;
    const float2* pilot_sym; const float2* ltf; const float2* mapped;
};

struct EqState {
    int symbol_ind, total_out, n_ofdm_symbols_SIG, sig_ok, equalize_done;
    int mcs, packet_type, data_length, snr_est_count, n_chan_mean;
    double freq_offset, er, epsilon0, snr_est, precoded_snr_est, signal_power_sum, noise_power_sum;
    float2 chan_mean[16];
};

struct EqIo {
    const float2* in; long in_stride; int ninput;
    const long long* tag_offsets; const double* tag_values; int n_tags;   // tag_offsets == nullptr: one tag at item 0 per stream
    float2* out; long out_stride; int noutput;
    int* n_out; int* n_consumed; jrc_eq_event* events; int max_events; int* n_events;
    float2* chan_est; int* chan_est_written;
    int stream0;
};

__device__ __forceinline__ float2 demod_point(int bps, float2 z)
{
    if (bps == 1) return make_float2(z.x > 0 ? 1.0f : -1.0f, 0.f);       // constellation_bpsk
    if (bps == 4) {
        // constellation_16qam (gr-digital 3.8, not in the reference tree: recollection, parity unpinned — the table of codec.hip):
        // decide (re > 0, |re| < 2 level, im > 0, |im| < 2 level) and map back: (+-1|3, +-1|3) * level, not scaled (:511, :567)
        const float level = 0.316227766016837933f;                         // sqrt(float(0.1)) rounded to float
        const float a = (fabsf(z.x) < 2 * level) ? 1.0f : 3.0f, b = (fabsf(z.y) < 2 * level) ? 1.0f : 3.0f;
        return make_float2((z.x > 0 ? a : -a) * level, (z.y > 0 ? b : -b) * level);
    }
    const float a = 0.707107f;                                             // constellation_qpsk, then /2 (:511-514)
    return make_float2((z.x > 0 ? a : -a) / 2.0f, (z.y > 0 ? a : -a) / 2.0f);
}

// ---- SIG-field Viterbi on one wavefront ------------------------------------------------------------------------------------
// decode_signal_field (lib/mimo_ofdm_equalizer_impl.cc:650-667) hands the hard BPSK decisions of the ND data cells to the reference's
// windowed SSE2 decoder (lib/viterbi_decoder.cc:99-331; ofdm_mcs(BPSK_1_2, ND) -> d_ntraceback 5, no depuncturing).  Its byte lanes map
// one to one onto the wavefront, lane = new trellis state, as in stream_decode_kernel (codec.hip): 8-bit metrics with wrap-around adds,
// the signed-byte compare of the difference (:118-120), 8-bit path chunks, viterbi_get_output_sse2 (:183-225) after steps 6, 14, 22, ...
// with the first maximum as best state and a traceback over 4 stored chunks.  Output byte j leaves at call 5 + j, so the 24 header
// bits need calls 0..7 = 62 trellis steps = coded bits 0..123: the decoder runs past the ND coded bits of a 48-carrier SIG symbol, where
// the reference reads whatever lies behind calloc(ND) (:175) — 0 here, as in the oracle.  Bit errors therefore resolve exactly as in the
// reference's decoder (ties, window truncation), not as a maximum-likelihood decoder would.
__host__ __device__ inline int eq_surv_words(int ND) { return ((ND / 2 + 59) / 60) * 64; }     // LDS scratch: >= 512 bytes, the ring takes 320
__host__ __device__ inline int eq_pair_bytes(int ND) { return ((ND / 2 + 5) / 6 + 1) * 8; }

__device__ __forceinline__ int sig_wave_max(int v) { for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off)); return v; }
__device__ __forceinline__ int sig_wave_min(int v) { for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off)); return v; }

// wave 0 only (lane = tid < 64).  Z: the ND equalised SIG cells; ring: 5 x 64 bytes of scratch.  Returns decoded bits 0..23 (bit i of the word = header bit i).
//
// Codeword shortcut (round 4).  When the 124 hard decisions the decoder would read ARE the encoder's output for some input sequence, that
// sequence is what the decoder returns: its path scores 2 on every step, any path ending in another state has lost 2 at the step where the two
// inputs first differed (both generators tap the newest bit, 0155 and 0117 are odd), so the true state is the strict maximum at every
// viterbi_get_output call and every traceback follows the true path, whatever the window.  The input sequence comes straight from the coded
// bits — g0 = 1+D^2+D^3+D^5+D^6 and g1 = 1+D+D^2+D^3+D^6 are coprime, (D^2+D^4) g0 + (1+D+D^2+D^3+D^4) g1 = 1 over GF(2), hence
// u_t = c0_{t-2} ^ c0_{t-4} ^ c1_t ^ c1_{t-1} ^ c1_{t-2} ^ c1_{t-3} ^ c1_{t-4} — a lane per step, and is accepted only if re-encoding it gives back
// every one of the 124 bits; one flipped decision anywhere and the trellis below runs as before.  tests/test_oracle_comm.py checks the claim
// against the oracle's windowed decoder, tests/test_gpu_comm.py that both ways agree on clean and corrupted fields (JRC_EQ_SIG_FULL).
__device__ __forceinline__ unsigned sig_viterbi_wave(const float2* Z, int ND, unsigned char* ring, int lane, bool full_only)
{
    const unsigned long long w0 = __ballot(lane < ND && Z[min(lane, ND - 1)].x > 0);                 // constellation_bpsk::decision_maker: re > 0
    const unsigned long long w1 = __ballot(lane + 64 < ND && Z[min(lane + 64, ND - 1)].x > 0);
    if (!full_only) {
        auto cb = [&](int j) -> unsigned { return j < 0 ? 0u : (j < 64 ? (unsigned)((w0 >> j) & 1ull) : (unsigned)((w1 >> (j - 64)) & 1ull)); };
        const int t = lane;
        const unsigned u = t < 62 ? (cb(2 * t - 4) ^ cb(2 * t - 8) ^ cb(2 * t + 1) ^ cb(2 * t - 1) ^ cb(2 * t - 3) ^ cb(2 * t - 5) ^ cb(2 * t - 7)) : 0u;
        const unsigned long long U = __ballot(u != 0);
        auto ub = [&](int i) -> unsigned { return i < 0 ? 0u : (unsigned)((U >> i) & 1ull); };
        const unsigned e0 = ub(t) ^ ub(t - 2) ^ ub(t - 3) ^ ub(t - 5) ^ ub(t - 6);                   // state & 0155
        const unsigned e1 = ub(t) ^ ub(t - 1) ^ ub(t - 2) ^ ub(t - 3) ^ ub(t - 6);                   // state & 0117
        const bool same = t >= 62 || (e0 == cb(2 * t) && e1 == cb(2 * t + 1));
        if (__all(same)) return (unsigned)(U & 0xffffffull);
    }
    for (int i = lane; i < 5 * 64; i += 64) ring[i] = 0;                                            // d_ppresult zeroed (:333-337)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int k = lane >> 1, odd = lane & 1;
    const int bt0 = __popc((2 * k) & 0x6d) & 1, bt1 = __popc((2 * k) & 0x4f) & 1;                    // d_branchtab27_sse2 (:323-326)
    const int nt = 5;
    unsigned st = 0, sig = 0;                                                                        // metric | path << 8
    int store_pos = 0, out_count = 0;
    for (int t = 0; t < 62; t++) {
        const unsigned long long w = t < 32 ? w0 : w1;
        const int sh = (2 * t) & 63;
        const int s0 = (int)((w >> sh) & 1ull), s1 = (int)((w >> (sh + 1)) & 1ull);
        const int metsvm = (bt0 ^ s0) + (bt1 ^ s1), metsv = 2 - metsvm;                             // :110-112
        const unsigned a = __shfl(st, k), b = __shfl(st, k + 32);
        const int ma = a & 0xff, mb = b & 0xff;
        const int x = (ma + (odd ? metsvm : metsv)) & 0xff, y = (mb + (odd ? metsv : metsvm)) & 0xff;
        const int dec = (signed char)((x - y) & 0xff) > 0;                                           // _mm_cmpgt_epi8(_mm_sub_epi8(m0, m1), 0)
        const unsigned pa = ((a >> 8) << 1) & 0xff, pb = ((((b >> 8) << 1) & 0xff) + 1) & 0xff;
        st = (unsigned)(dec ? x : y) | ((dec ? pa : pb) << 8);
        if (t >= 5 && ((t - 5) & 7) == 0) {                                                          // viterbi_get_output_sse2
            store_pos = (store_pos + 1 == nt) ? 0 : store_pos + 1;
            const int metric = st & 0xff;
            ring[store_pos * 64 + lane] = (unsigned char)(st >> 8);
            const int best = sig_wave_max(metric), mn = sig_wave_min(metric);
            int beststate = __ffsll((unsigned long long)__ballot(metric == best)) - 1;               // first maximum (strict > in the scan)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int pos = store_pos;
            for (int i = 0; i < nt - 1; i++) {
                beststate = ring[pos * 64 + beststate] >> 2;
                pos = (pos == 0) ? nt - 1 : pos - 1;
            }
            const unsigned c = ring[pos * 64 + beststate];
            st = (unsigned)((metric - mn) & 0xff);                                                   // paths zeroed, metrics renormalised
            if (out_count >= nt) {                                                                   // decoded bits, MSB first (:277-281)
                const int j = out_count - nt;
#pragma unroll
                for (int bb = 0; bb < 8; bb++) sig |= ((c >> (7 - bb)) & 1u) << (8 * j + bb);
            }
            out_count++;
            __builtin_amdgcn_wave_barrier();
        }
    }
    return sig;
}

#ifndef EQ_BATCH
#define EQ_BATCH 64   // data symbols per three-phase pass of the equalizer
#endif
#ifndef EQ_PD
#define EQ_PD 2       // input symbols in flight per lane in the equalisation phase (measured at config C once the loop had no branch around its loads: 1: 0.602-0.613 ms, 2: 0.593-0.599, 3: 0.596, 4: 0.602-0.65 with 31 spilled registers)
#endif

// NTMAX = workgroup size the variant is compiled for, WPE = waves per SIMD it must allow (register budget 512 / WPE)
template <int NTMAX, int WPE, int EPT /* subcarriers per lane: fft_len <= EPT * blockDim */>
__global__ __launch_bounds__(NTMAX, WPE) void equalizer_kernel(EqDev d, EqState* states, float2* H_all, float2* Hm_all,
                                                         float2* pre_all, EqIo io)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char eq_smem[];
    const int N = d.N, ND = d.ND, NP = d.NP, NL = d.NL, T = d.T;
    float2* Y = reinterpret_cast<float2*>(eq_smem);
    float2* H = Y + N;
    float2* Hm = H + N;
    float2* Z = Hm + N;                       // [ND]
    float2* est = Z + ND;                     // [NP]
    unsigned long long* surv = reinterpret_cast<unsigned long long*>(est + NP);   // [eq_surv_words(ND)]: scratch of the SIG Viterbi (its 5 x 64-byte path ring)
    // constant tables staged in LDS: the serial (one-lane) sections below walk them with dependent loads, which from
    // global memory cost microseconds each once every CU is busy
    int* dc = reinterpret_cast<int*>(eq_smem + d.lds_tables);                     // [ND] data carriers
    int* pc = dc + ND;                                                            // [NP] pilot carriers
    int* ac = pc + NP;                                                            // [NAct] sorted active carriers
    float2* s_ltf = reinterpret_cast<float2*>(ac + d.NAct + ((ND + NP + d.NAct) & 1));   // [N]
    float2* s_ref = s_ltf + N;                                                    // [NP] pilot row of the current symbol
    __shared__ EqState S;
    __shared__ float2 s_rot;
    __shared__ int s_flag;
    __shared__ double s_dred[32];
    __shared__ float2 s_brot[EQ_BATCH];
    __shared__ double s_bsig[EQ_BATCH], s_bnoi[EQ_BATCH];

    const int tid = threadIdx.x, NT = blockDim.x;
    const int b = blockIdx.x;
    const int stream = io.stream0 + b;
    EqState* gst = states + stream;
    float2* gH = H_all + (size_t)stream * N;
    float2* gHm = Hm_all + (size_t)stream * N;
    // MIMO-LTF store [N][NL] (:213: a stack array in the reference).  round 5: in LDS where it fits, on Z / est / the Viterbi scratch — all dead
    // between the SIG field and the first data symbol — grown to N x NL cells; it only goes to HBM (pre_g) when a call ends between two MIMO-LTF
    // symbols of a frame and comes back when the next call starts there (a frame may span calls).  (A flat pointer: LDS or global.)
    float2* const pre_g = pre_all + (size_t)stream * N * NL;
    float2* const pre = d.pre_lds ? Z : pre_g;
    const float2* in = io.in + (size_t)b * io.in_stride;
    float2* out = io.out + (size_t)b * io.out_stride;
    float2* chan_est = io.chan_est ? io.chan_est + (size_t)b * N * T : nullptr;
    jrc_eq_event* events = io.events + (size_t)b * io.max_events;

    if (tid == 0) S = *gst;
    for (int i = tid; i < N; i += NT) { H[i] = gH[i]; Hm[i] = gHm[i]; s_ltf[i] = d.ltf[i]; }
    for (int i = tid; i < ND; i += NT) dc[i] = d.data_c[i];
    for (int i = tid; i < NP; i += NT) pc[i] = d.pilot_c[i];
    for (int i = tid; i < d.NAct; i += NT) ac[i] = d.active_c[i];
    __syncthreads();
    if (d.pre_lds && S.sig_ok && S.symbol_ind >= 4 && S.symbol_ind <= 2 + NL)        // the call before this one ended between two MIMO-LTF symbols
        for (int i = tid; i < N * NL; i += NT) pre[i] = pre_g[i];

    int n_in = 0, n_out = 0, nev = 0, ce_written = 0, advance = 0;
    // input symbols are prefetched one symbol ahead into registers: a symbol is only a few microseconds of work, so
    // an HBM round trip per symbol would otherwise dominate the per-frame latency
    float2 xin[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) { const int i = tid + e * NT; if (i < N && io.ninput > 0) xin[e] = in[i]; }
    while (n_in < io.ninput && n_out < io.noutput) {                                   // :219
        float2 cur[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            cur[e] = xin[e];
            const int i = tid + e * NT;
            if (i < N && n_in + 1 < io.ninput) xin[e] = in[(size_t)(n_in + 1) * N + i];
        }
        if (tid == 0) {
            if (advance) S.symbol_ind++;                                                // d_symbol_ind++ of the previous turn (:607)
            int hit = -1;
            if (io.tag_offsets == nullptr) { if (n_in == 0) hit = stream; }
            else for (int t = 0; t < io.n_tags; t++) if (io.tag_offsets[t] == n_in) { hit = t; break; }
            if (hit >= 0) {                                                             // frame_start :221-245
                const double v = io.tag_values[io.tag_offsets == nullptr ? b : hit];
                S.symbol_ind = 0; S.total_out = 0; S.n_ofdm_symbols_SIG = 0;
                S.freq_offset = v * d.bw / (2 * M_PI);
                S.epsilon0 = v * d.bw / (2 * M_PI * d.freq);
                S.er = 0; S.sig_ok = 1; S.equalize_done = 0;
                S.signal_power_sum = 0; S.noise_power_sum = 0; S.snr_est_count = 0;
            }
        }
        __syncthreads();
        const int sym = S.symbol_ind;
        if (sym > S.n_ofdm_symbols_SIG + 2 + NL || !S.sig_ok) { n_in++; advance = 0; __syncthreads(); continue; }   // :250-255

        // ---- data symbols without decision feedback (LS): all that this call holds, in three phases -----------
        // A data symbol depends on the ones before it only through the running noise / signal sums (:484-493, :545), so: (A) each
        // wavefront takes symbols and does their pilot work — residual CFO (:908-922) and the two sums; (B) one lane accumulates the
        // sums in symbol order; (C) every lane equalises its subcarriers symbol after symbol with no barrier in between.  Same
        // expressions as the symbol-at-a-time path below, which keeps the STA estimator and single symbols.
        {
            const bool sta_b = d.estimator == 1;
            int nb = 0;
            if (sym > 2 + NL && !sta_b && (S.packet_type == 1 || S.packet_type == 2)) {
                nb = io.ninput - n_in;
                nb = min(nb, io.noutput - n_out);
                nb = min(nb, S.n_ofdm_symbols_SIG + 2 + NL - sym + 1);
                if (io.tag_offsets)
                    for (int t = 0; t < io.n_tags; t++) {
                        const long long off = io.tag_offsets[t];
                        if (off > n_in && off - n_in < nb) nb = (int)(off - n_in);
                    }
                nb = min(nb, EQ_BATCH);
            }
            if (nb >= 2) {
                const int ln = tid & 63, wv = tid >> 6, NW = NT >> 6;
                const int ptype = S.packet_type;
                const float2* Hsel = ptype == 1 ? H : Hm;
                const double eps = S.epsilon0 + S.er;
                if (EQ_EXP(d) == 4) {                                                   // timing experiment: no pilot phase at all
                    for (int j = tid; j < nb; j += NT) { s_brot[j] = make_float2(1.f, 0.f); s_bsig[j] = 1.0; s_bnoi[j] = 1e-3; }
                } else
                if (NP <= 64) {                                                         // (A) 64 / G symbols per wavefront pass, G = NP rounded up
                    const int G = NP <= 1 ? 1 : (1 << (32 - __clz(NP - 1)));            //     to a power of two lanes per symbol
                    const int spw = 64 / G, kk = ln & (G - 1), sl = ln / G;
                    for (int j0 = wv * spw; j0 < nb; j0 += NW * spw) {
                        const int j = j0 + sl;
                        const bool on = j < nb && kk < NP;
                        const int jc = j < nb ? j : nb - 1, kc = kk < NP ? kk : 0;
                        const int sy = sym + jc;
                        const float2* prow = d.pilot_sym + (size_t)((sy - 3 - NL) % d.n_pilot_rows) * NP;
                        const double k0 = 2 * M_PI * sy * ((N + d.cp) * 1.0 / N) * eps;
                        const int c = pc[kc];
                        const float2 yk = c_mul(in[(size_t)(n_in + (EQ_EXP(d) == 1 ? 0 : jc)) * N + c], c_expj(k0 * (c - N / 2)));
                        const float2 e = c_mul(Hsel[c], prow[kc]);
                        float2 sum = make_float2(0.f, 0.f);
                        if (on) { const float2 pp = c_mul(yk, c_conj(e)); sum.x = sum.x + pp.x; sum.y = sum.y + pp.y; }
                        for (int off = G >> 1; off > 0; off >>= 1) { sum.x += __shfl_xor(sum.x, off); sum.y += __shfl_xor(sum.y, off); }
                        const float2 r0 = c_expj(-(double)atan2f(sum.y, sum.x));
                        double sig = 0, noi = 0;
                        if (on) {
                            sig += (double)c_mul(e, c_conj(e)).x;
                            const float2 yr = c_mul(yk, r0);
                            const float2 er = make_float2(e.x - yr.x, e.y - yr.y);
                            noi += (double)c_mul(er, c_conj(er)).x;
                        }
                        for (int off = G >> 1; off > 0; off >>= 1) { sig += __shfl_xor(sig, off); noi += __shfl_xor(noi, off); }
                        if (kk == 0 && j < nb) { s_brot[j] = r0; s_bsig[j] = sig; s_bnoi[j] = noi; }
                    }
                } else
                for (int j = wv; j < nb; j += NW) {                                     // (A) more than 64 pilots: one symbol per pass, lanes stride the pilots
                    const int sy = sym + j;
                    const float2* prow = d.pilot_sym + (size_t)((sy - 3 - NL) % d.n_pilot_rows) * NP;
                    const double k0 = 2 * M_PI * sy * ((N + d.cp) * 1.0 / N) * eps;
                    const float2* x = in + (size_t)(n_in + j) * N;
                    float2 sum = make_float2(0.f, 0.f);
                    for (int k = ln; k < NP; k += 64) {
                        const int c = pc[k];
                        const float2 yk = c_mul(x[c], c_expj(k0 * (c - N / 2)));
                        const float2 e = c_mul(Hsel[c], prow[k]);
                        const float2 pp = c_mul(yk, c_conj(e));
                        sum.x = sum.x + pp.x; sum.y = sum.y + pp.y;
                    }
                    for (int off = 32; off > 0; off >>= 1) { sum.x += __shfl_xor(sum.x, off); sum.y += __shfl_xor(sum.y, off); }
                    const float2 r0 = c_expj(-(double)atan2f(sum.y, sum.x));
                    double sig = 0, noi = 0;
                    for (int k = ln; k < NP; k += 64) {
                        const int c = pc[k];
                        const float2 yk = c_mul(x[c], c_expj(k0 * (c - N / 2)));
                        const float2 e = c_mul(Hsel[c], prow[k]);
                        sig += (double)c_mul(e, c_conj(e)).x;
                        const float2 yr = c_mul(yk, r0);
                        const float2 er = make_float2(e.x - yr.x, e.y - yr.y);
                        noi += (double)c_mul(er, c_conj(er)).x;
                    }
                    for (int off = 32; off > 0; off >>= 1) { sig += __shfl_xor(sig, off); noi += __shfl_xor(noi, off); }
                    if (ln == 0) { s_brot[j] = r0; s_bsig[j] = sig; s_bnoi[j] = noi; }
                }
                __syncthreads();
                if (tid == 0) {                                                         // (B)
                    for (int j = 0; j < nb; j++) {
                        S.signal_power_sum += s_bsig[j]; S.noise_power_sum += s_bnoi[j]; S.snr_est_count += NP;
                        s_bnoi[j] = S.noise_power_sum / S.snr_est_count;                // nvar of symbol j (:545)
                    }
                    S.symbol_ind = sym + nb - 1;
                }
                __syncthreads();
                // (C) no branch around a load or a store: the memory counter is in order, and with loads or stores that may or may not be
                // issued the compiler cannot count what lies between a load and its use — it drains the counter (vmcnt(0)) at every symbol,
                // which turns the EQ_PD symbols of prefetch into none.  So: lanes past the last data carrier repeat the last one (they
                // compute the same value and store it to the same address), the prefetch index is clamped to the last symbol (read again
                // from cache), and the packet type selects one of two copies of the loop instead of a branch per cell.
                int scv[EPT], oi[EPT];
                float2 hv[EPT];
                double hm2[EPT];
                // EQ_PD symbols of input in flight per lane: one symbol is ~0.3 us of arithmetic against a ~2 us HBM round trip
                float2 xq[EQ_PD][EPT];
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    oi[e] = min(tid + e * NT, ND - 1);
                    scv[e] = dc[oi[e]];
                    hv[e] = Hsel[scv[e]];
                    hm2[e] = (double)c_mul(hv[e], c_conj(hv[e])).x;
#pragma unroll
                    for (int q = 0; q < EQ_PD; q++) xq[q][e] = in[(size_t)(n_in + (EQ_EXP(d) == 3 ? 0 : min(q, nb - 1))) * N + scv[e]];
                }
                // The rotation angle of a cell is k0 (sc - N/2) with |sc - N/2| <= N/2 and |k0| growing with the symbol index: when the last
                // symbol's k0 * N/2 rounds to at most pi/4 every angle of the batch does (rounding is monotonic), and the loop runs on
                // the polynomial alone — no test, no exec-masked slow path splitting the cell's code into blocks.
                const double k0_last = 2 * M_PI * (sym + nb - 1) * ((N + d.cp) * 1.0 / N) * eps;
                const bool all_small = fabsf((float)(k0_last * (N / 2))) <= 0.78539816f;
                auto equalise = [&](auto PT, auto SMALL) {
                    constexpr int pt = decltype(PT)::value;
                    constexpr bool small = decltype(SMALL)::value;
                    for (int j0 = 0; j0 < nb; j0 += EQ_PD) {
#pragma unroll
                        for (int q = 0; q < EQ_PD; q++) {
                            const int j = j0 + q;
                            if (j >= nb) break;
                            float2 xc[EPT];
                            const size_t nxt = (size_t)(n_in + (EQ_EXP(d) == 3 ? 0 : min(j + EQ_PD, nb - 1))) * N;   // (3: timing experiment, every symbol reads the batch's first row)
#pragma unroll
                            for (int e = 0; e < EPT; e++) {
                                xc[e] = xq[q][e];
                                xq[q][e] = in[nxt + scv[e]];
                            }
                            const double k0 = 2 * M_PI * (sym + j) * ((N + d.cp) * 1.0 / N) * eps;
                            const float2 rot = s_brot[j];
                            const double nvar = s_bnoi[j];
                            float2* o = out + (size_t)(n_out + j) * ND;
#pragma unroll
                            for (int e = 0; e < EPT; e++) {
                                const double ang = k0 * (scv[e] - N / 2);
                                const float2 yr = c_mul(c_mul(xc[e], small ? c_expj_small(ang) : c_expj(ang)), rot);
                                float2 z;
                                if (EQ_EXP(d) == 5) z = xc[e];                          // timing experiment: no arithmetic on the cell
                                else
                                if constexpr (pt == 1) z = c_div(yr, hv[e]);            // symbol_equalize :900-906
                                else {                                                  // :540-550
                                    const float csi = (float)(hm2[e] + nvar);
                                    const float2 num = c_mul(yr, c_conj(hv[e]));
                                    z = make_float2(num.x / csi, num.y / csi);
                                }
                                typedef float v2f __attribute__((ext_vector_type(2)));     // :602; write-once output: around the caches
                                const v2f zz = {z.x, z.y};
                                if (EQ_EXP(d) == 6) { if (z.x == 1.2345e30f) o[oi[e]] = z; }   // timing experiment: (practically) no stores
                                else if (EQ_EXP(d) == 7) o[oi[e]] = z;                         // timing experiment: cached stores
                                else
                                __builtin_nontemporal_store(zz, reinterpret_cast<v2f*>(o + oi[e]));
                            }
                        }
                    }
                };
                if (all_small) {
                    if (ptype == 1) equalise(std::integral_constant<int, 1>{}, std::true_type{}); else equalise(std::integral_constant<int, 2>{}, std::true_type{});
                } else {
                    if (ptype == 1) equalise(std::integral_constant<int, 1>{}, std::false_type{}); else equalise(std::integral_constant<int, 2>{}, std::false_type{});
                }
                n_in += nb; n_out += nb; advance = 1;
#pragma unroll
                for (int e = 0; e < EPT; e++) { const int i = tid + e * NT; if (i < N && n_in < io.ninput) xin[e] = in[(size_t)n_in * N + i]; }
                __syncthreads();
                continue;
            }
        }

        if (sym >= 2) {               // pilot row of this symbol (SIG: row 0; data symbol m: row m mod n_rows), read by lane 0 below
            const int row = sym == 2 ? 0 : (sym > 2 + NL ? (sym - 3 - NL) % d.n_pilot_rows : 0);
            for (int k = tid; k < NP; k += NT) s_ref[k] = d.pilot_sym[(size_t)row * NP + k];
        }
        {   // sampling-offset de-rotation :261-264
            const double k0 = 2 * M_PI * sym * ((N + d.cp) * 1.0 / N) * (S.epsilon0 + S.er);
#pragma unroll
            for (int e = 0; e < EPT; e++) { const int i = tid + e * NT; if (i < N) Y[i] = c_mul(cur[e], c_expj(k0 * (i - N / 2))); }
        }
        __syncthreads();

        if (sym == 0) {                                                                 // :272-275
            for (int i = tid; i < N; i += NT) H[i] = Y[i];
        } else if (sym == 1) {                                                          // :277-306
            {   // SNR from the two L-LTF periods (:279-305); double sums reduced across the workgroup
                double signal = 0, noise = 0;
                for (int k = tid; k < d.NAct; k += NT) {
                    const int c = ac[k];
                    const double hn = (double)ref_hypotf(make_float2(H[c].x - Y[c].x, H[c].y - Y[c].y));
                    const double hs = (double)ref_hypotf(make_float2(H[c].x + Y[c].x, H[c].y + Y[c].y));
                    noise += hn * hn; signal += hs * hs;
                }
                for (int off = 32; off > 0; off >>= 1) { noise += __shfl_down(noise, off); signal += __shfl_down(signal, off); }
                if ((tid & 63) == 0) { s_dred[tid >> 6] = noise; s_dred[16 + (tid >> 6)] = signal; }
                __syncthreads();
                if (tid == 0) {
                    double n2 = 0, s2 = 0;
                    for (int w = 0; w < (NT + 63) / 64; w++) { n2 += s_dred[w]; s2 += s_dred[16 + w]; }
                    S.snr_est = 10 * log10(s2 / n2 / 2);
                }
            }
            for (int k = tid; k < d.NAct; k += NT) {
                const int c = ac[k];
                const float2 l = s_ltf[c];
                H[c] = c_div(make_float2(H[c].x + Y[c].x, H[c].y + Y[c].y), c_mul(l, make_float2(2.f, 0.f)));
            }
        } else if (sym == 2) {                                                          // SIG :308-344
            if (tid == 0) {
                float2 sum = make_float2(0.f, 0.f);
                for (int k = 0; k < NP; k++) {                                          // estimate_residual_cfo :908-922
                    est[k] = c_mul(H[pc[k]], s_ref[k]);
                    const float2 p = c_mul(Y[pc[k]], c_conj(est[k]));
                    sum.x = sum.x + p.x; sum.y = sum.y + p.y;
                }
                s_rot = c_expj(-(double)atan2f(sum.y, sum.x));
            }
            __syncthreads();
            for (int i = tid; i < N; i += NT) Y[i] = c_mul(Y[i], s_rot);
            __syncthreads();
            for (int i = tid; i < ND; i += NT) {
                Z[i] = c_div(Y[dc[i]], H[dc[i]]);                           // symbol_equalize :900-906 (BPSK decision: sig_viterbi_wave)
            }
            __syncthreads();
            if (tid < 64) {   // the reference's windowed K=7 decoder over the ND hard decisions: one wavefront (above)
                const unsigned sig_word = sig_viterbi_wave(Z, ND, reinterpret_cast<unsigned char*>(surv), tid, d.sig_full != 0);
                if (tid == 0) {
                    // parse :669-781
                    const int rate = (int)(sig_word & 0xfu), pt = (int)((sig_word >> 4) & 1u), len = (int)((sig_word >> 5) & 0xfffu);
                    const int parity = __popc(sig_word & 0x1ffffu) & 1, dec17 = (int)((sig_word >> 17) & 1u);
                    const int trailing_ok = ((sig_word >> 17) & 0x3fu) == 0;                // bits 17..22, as the reference tests them (:702)
                    int ok = 1, mcs = 0;
                    S.data_length = len;
                    if (parity != dec17 && trailing_ok) { ok = 0; S.data_length = 0; S.n_ofdm_symbols_SIG = 0; }
                    else {
                        switch (rate) {
                            case 11: mcs = 0; break; case 15: mcs = 1; break; case 10: mcs = 2; break;
                            case 14: mcs = 3; break; case 9: mcs = 4; break; case 13: mcs = 5; break;
                            default: ok = 0;
                        }
                        S.packet_type = pt == 0 ? 1 : 2;
                        if (ok) { S.mcs = mcs; S.n_ofdm_symbols_SIG = n_ofdm_sym_dev(mcs, ND, len); }
                    }
                    S.sig_ok = ok;
                    if (ok && nev < io.max_events) {                                    // stream_start tag :331-337
                        jrc_eq_event e;
                        e.kind = 1; e.n_chan_mean = 0; e.offset = n_out; e.data_bytes = (unsigned long long)len;
                        e.mcs = (unsigned long long)mcs; e.packet_type = (unsigned long long)S.packet_type;
                        e.snr = S.snr_est; e.freq_offset = S.freq_offset; e.snr_data = 0;
                        for (int t = 0; t < 16; t++) { e.chan_mean[t].re = 0; e.chan_mean[t].im = 0; }
                        events[nev] = e;
                    }
                    s_flag = ok;
                }
            }
            __syncthreads();
            if (s_flag && nev < io.max_events) nev++;
        } else if (sym <= 2 + NL) {                                                     // MIMO-LTFs :346-463
            const int l = sym - 3;
            if (EQ_EXP(d) != 2) for (int i = tid; i < N; i += NT) pre[(size_t)i * NL + l] = Y[i];
            __syncthreads();
            if (l == NL - 1) {
                if (S.packet_type == 1) {                                               // NDP :375-422
                    for (int sc = tid; sc < N; sc += NT)
                        for (int t = 0; t < T; t++) {
                            float2 h = make_float2(0.f, 0.f);
                            for (int q = 0; q < NL; q++) {
                                const float2 p = c_mul(c_conj(d.mapped[(size_t)sc * d.mapped_cols + t * NL + q]), pre[(size_t)sc * NL + q]);
                                h.x = h.x + p.x; h.y = h.y + p.y;
                            }
                            if (chan_est) chan_est[(size_t)sc * T + t] = h;     // content of chan_est_file (:392-416)
                        }
                    __syncthreads();
                    if (tid == 0) {
                        for (int t = 0; t < T; t++) {
                            float2 m = make_float2(0.f, 0.f);
                            for (int k = 0; k < d.NAct; k++) {
                                const int sc = ac[k];
                                float2 h;
                                if (chan_est) h = chan_est[(size_t)sc * T + t];
                                else {
                                    h = make_float2(0.f, 0.f);
                                    for (int q = 0; q < NL; q++) {
                                        const float2 p = c_mul(c_conj(d.mapped[(size_t)sc * d.mapped_cols + t * NL + q]), pre[(size_t)sc * NL + q]);
                                        h.x = h.x + p.x; h.y = h.y + p.y;
                                    }
                                }
                                m.x = m.x + h.x; m.y = m.y + h.y;
                            }
                            S.chan_mean[t] = make_float2(m.x / (float)d.NAct, m.y / (float)d.NAct);
                        }
                        S.n_chan_mean = T;
                    }
                    ce_written = 1;
                } else if (S.packet_type == 2) {                                        // DATA :423-456
                    for (int k = tid; k < d.NAct; k += NT) {
                        const int sc = k < ND ? dc[k] : pc[k - ND];
                        float2 acc = make_float2(0.f, 0.f);
                        for (int q = 0; q < NL; q++) {                                  // row(0).dot(y): conjugates the row
                            const float2 p = c_mul(c_conj(d.mapped[(size_t)sc * d.mapped_cols + q]), EQ_EXP(d) == 2 ? Y[sc] : pre[(size_t)sc * NL + q]);
                            acc.x = acc.x + p.x; acc.y = acc.y + p.y;
                        }
                        Hm[sc] = make_float2(acc.x / (float)NL, acc.y / (float)NL);
                        Y[k] = Hm[sc];                                                  // in summation order (Y is free: the symbol sits in `pre`)
                    }
                    __syncthreads();
                    if (tid == 0) {                                                     // the mean is a sequential sum: contiguous operands, no index chasing
                        float2 m = make_float2(0.f, 0.f);
                        for (int k = 0; k < d.NAct; k++) { m.x = m.x + Y[k].x; m.y = m.y + Y[k].y; }
                        S.chan_mean[0] = make_float2(m.x / (float)d.NAct, m.y / (float)d.NAct);
                        S.n_chan_mean = 1;
                    }
                }
            }
        } else {                                                                        // data symbols :465-605
            const float2* ref = s_ref;          // d_pilot_symbols[(sym - 3 - N_ltf) % size], staged above (:467)
            float2* Hsel = S.packet_type == 1 ? H : Hm;
            if (tid < 64) {   // one wavefront: lane k owns pilot k (strided when there are more than 64 pilots)
                float2 sum = make_float2(0.f, 0.f);                                     // estimate_residual_cfo :908-922
                for (int k = tid; k < NP; k += 64) {
                    est[k] = c_mul(Hsel[pc[k]], ref[k]);
                    const float2 p = c_mul(Y[pc[k]], c_conj(est[k]));
                    sum.x = sum.x + p.x; sum.y = sum.y + p.y;
                }
                for (int off = 32; off > 0; off >>= 1) { sum.x += __shfl_xor(sum.x, off); sum.y += __shfl_xor(sum.y, off); }
                const float2 r0 = c_expj(-(double)atan2f(sum.y, sum.x));
                double sig = 0, noi = 0;
                for (int k = tid; k < NP; k += 64) {                                    // :484-493, on the de-rotated pilots
                    sig += (double)c_mul(est[k], c_conj(est[k])).x;
                    const float2 yr = c_mul(Y[pc[k]], r0);
                    const float2 e = make_float2(est[k].x - yr.x, est[k].y - yr.y);
                    noi += (double)c_mul(e, c_conj(e)).x;
                }
                for (int off = 32; off > 0; off >>= 1) { sig += __shfl_xor(sig, off); noi += __shfl_xor(noi, off); }
                if (tid == 0) {
                    s_rot = r0;
                    S.signal_power_sum += sig; S.noise_power_sum += noi; S.snr_est_count += NP;
                }
            }
            __syncthreads();
            const float2 rot = s_rot;           // Y is de-rotated on the fly (:480-482): it is not needed after this symbol
            const int bps = (S.mcs <= 1) ? 1 : (S.mcs <= 3 ? 2 : 4);
            const bool sta = d.estimator == 1;
            float2* o = out + (size_t)n_out * ND;
            if (S.packet_type == 1) {
                const float alpha = 0.5f;                                               // STA :498-535
                for (int i = tid; i < ND; i += NT) {
                    const int sc = dc[i];
                    const float2 yr = c_mul(Y[sc], rot);
                    const float2 z = c_div(yr, H[sc]);                                  // symbol_equalize :900-906
                    o[i] = z;                                                           // :602
                    if (sta) {
                        const float2 upd = c_div(yr, demod_point(bps, z));
                        const float2 a = c_mul(make_float2(1 - alpha, 0.f), H[sc]), bb = c_mul(make_float2(alpha, 0.f), upd);
                        H[sc] = make_float2(a.x + bb.x, a.y + bb.y);
                    }
                }
                if (sta)
                    for (int k = tid; k < NP; k += NT) {
                        const int sc = pc[k];
                        const float2 a = c_mul(make_float2(1 - alpha, 0.f), H[sc]);
                        const float2 bb = c_div(c_mul(make_float2(alpha, 0.f), c_mul(Y[sc], rot)), ref[k]);
                        H[sc] = make_float2(a.x + bb.x, a.y + bb.y);
                    }
            } else if (S.packet_type == 2) {
                const double nvar = S.noise_power_sum / S.snr_est_count;
                const float alpha = 0.4f;                                               // STA :552-592
                for (int i = tid; i < ND; i += NT) {                                    // :540-550
                    const int sc = dc[i];
                    const float2 yr = c_mul(Y[sc], rot);
                    const float csi = (float)((double)c_mul(Hm[sc], c_conj(Hm[sc])).x + nvar);
                    const float2 num = c_mul(yr, c_conj(Hm[sc]));
                    const float2 z = make_float2(num.x / csi, num.y / csi);
                    o[i] = z;
                    if (sta) {
                        const float2 a = c_mul(make_float2(1 - alpha, 0.f), Hm[sc]);
                        const float2 bb = c_div(c_mul(make_float2(alpha, 0.f), yr), demod_point(bps, z));
                        Hm[sc] = make_float2(a.x + bb.x, a.y + bb.y);
                    }
                }
                if (sta)
                    for (int k = tid; k < NP; k += NT) {
                        const int sc = pc[k];
                        const float2 a = c_mul(make_float2(1 - alpha, 0.f), Hm[sc]);
                        const float2 bb = c_div(c_mul(make_float2(alpha, 0.f), c_mul(Y[sc], rot)), ref[k]);
                        Hm[sc] = make_float2(a.x + bb.x, a.y + bb.y);
                    }
            } else {
                // unknown packet type: the reference emits the stale equalised SIG cells Z.  UNREACHABLE here — data symbols are only processed
                // with sig_ok set, and the SIG parse sets sig_ok only for packet_type 1 (NDP) or 2 (DATA) — which matters because with `pre_lds`
                // the MIMO-LTF store aliases Z: were this branch ever taken in that mode it would emit MIMO-LTF cells, not the reference's
                // stale SIG cells.  Kept so that the control flow reads like the reference's.
                for (int i = tid; i < ND; i += NT) o[i] = Z[i];
            }
            n_out++;
        }
        advance = 1;
        n_in++;
        __syncthreads();
    }
    if (tid == 0 && advance) S.symbol_ind++;


    if (tid == 0) {
        S.total_out += n_out;
        if (S.total_out == S.n_ofdm_symbols_SIG && S.sig_ok && !S.equalize_done) {      // stream_end tag :611-632
            if (S.snr_est_count != 0)
                S.precoded_snr_est = 10 * log10((S.signal_power_sum / S.snr_est_count) / (S.noise_power_sum / S.snr_est_count));
            if (nev < io.max_events) {
                jrc_eq_event e;
                e.kind = 2; e.n_chan_mean = S.n_chan_mean; e.offset = (long long)n_out - 1;
                e.data_bytes = 0; e.mcs = 0; e.packet_type = 0; e.snr = 0; e.freq_offset = 0;
                e.snr_data = S.precoded_snr_est;
                for (int t = 0; t < 16; t++) { e.chan_mean[t].re = t < S.n_chan_mean ? S.chan_mean[t].x : 0.f; e.chan_mean[t].im = t < S.n_chan_mean ? S.chan_mean[t].y : 0.f; }
                events[nev++] = e;
            }
            S.equalize_done = 1;
        }
        for (int k = nev; k < io.max_events; k++) events[k].kind = 0;   // every slot defined: callers need not clear the array
        *gst = S;
        io.n_out[b] = n_out;
        io.n_consumed[b] = n_in;
        io.n_events[b] = nev;
        if (io.chan_est_written) io.chan_est_written[b] = ce_written;
    }
    for (int i = tid; i < N; i += NT) { gH[i] = H[i]; gHm[i] = Hm[i]; }
    if (d.pre_lds) {                                                                   // this call ends between two MIMO-LTF symbols: their store goes to HBM
        __syncthreads();
        if (S.sig_ok && S.symbol_ind >= 4 && S.symbol_ind <= 2 + NL)
            for (int i = tid; i < N * NL; i += NT) pre_g[i] = pre[i];
    }
}

struct jrc_equalizer {
    jrc_ctx* ctx;
    EqDev d;
    int n_streams;
    void* tables = nullptr;      // one allocation for all constant tables
    EqState* states = nullptr;
    float2 *H = nullptr, *Hm = nullptr, *pre = nullptr;
    int* counters = nullptr;     // [n_streams][4]: n_out, n_consumed, n_events, chan_est_written
    size_t lds_bytes;
    int threads;
};

static void launch_equalizer(jrc_equalizer* eq, int grid, hipStream_t s, const EqIo& io);

extern "C" int jrc_equalizer_create(jrc_ctx* ctx, const jrc_eq_cfg* c, int n_streams, jrc_equalizer** out)
{
    if (!ctx || !c || !out || n_streams <= 0) return JRC_ERR_INVALID_ARG;
    if (c->fft_len <= 0 || c->n_data < 48 || c->n_pilot <= 0 || c->n_mimo_ltf <= 0 || c->mapped_cols % c->n_mimo_ltf ||
        c->n_pilot_rows <= 0 || !c->data_carriers || !c->pilot_carriers || !c->pilot_symbols || !c->ltf_seq || !c->mapped_ltf)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "mimo_ofdm_equalizer: bad configuration");
    if (c->estimator != 0 && c->estimator != 1)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "[OFDM Equalizer] Estimator not implemented");      // :941-943
    const int N = c->fft_len, ND = c->n_data, NP = c->n_pilot, T = c->mapped_cols / c->n_mimo_ltf;
    if (T > 16 || N > 4096) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "mimo_ofdm_equalizer: N_tx <= 16, fft_len <= 4096");
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int> dc(ND), pc(NP), ac;
    for (int i = 0; i < NP; i++) pc[i] = c->pilot_carriers[i] + N / 2;                 // :134-137
    for (int i = 0; i < ND; i++) dc[i] = c->data_carriers[i] + N / 2;                  // :139-142
    for (int v : dc) if (v < 0 || v >= N) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "data carrier index out of bounds");
    for (int v : pc) if (v < 0 || v >= N) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "pilot carrier index out of bounds");
    ac = dc; ac.insert(ac.end(), pc.begin(), pc.end());
    std::sort(ac.begin(), ac.end());                                                    // :159
    jrc_equalizer* eq = new jrc_equalizer();
    eq->ctx = ctx; eq->n_streams = n_streams;
    const size_t b_int = sizeof(int) * (size_t)(ND + NP + ac.size());
    const size_t b_ps = sizeof(float2) * (size_t)c->n_pilot_rows * NP, b_ltf = sizeof(float2) * N;
    const size_t b_map = sizeof(float2) * (size_t)N * c->mapped_cols;
    const size_t off_ps = (b_int + 15) & ~size_t(15);
    const size_t total = off_ps + b_ps + b_ltf + b_map;
    std::vector<unsigned char> host(total);
    memcpy(host.data(), dc.data(), sizeof(int) * ND);
    memcpy(host.data() + sizeof(int) * ND, pc.data(), sizeof(int) * NP);
    memcpy(host.data() + sizeof(int) * (ND + NP), ac.data(), sizeof(int) * ac.size());
    memcpy(host.data() + off_ps, c->pilot_symbols, b_ps);
    memcpy(host.data() + off_ps + b_ps, c->ltf_seq, b_ltf);
    memcpy(host.data() + off_ps + b_ps + b_ltf, c->mapped_ltf, b_map);
    hipError_t e = hipMalloc(&eq->tables, total);
    if (e == hipSuccess) e = hipMemcpy(eq->tables, host.data(), total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&eq->states, sizeof(EqState) * n_streams);
    if (e == hipSuccess) e = hipMemset(eq->states, 0, sizeof(EqState) * n_streams);    // sig_ok = 0: skip until a frame_start
    if (e == hipSuccess) e = hipMalloc((void**)&eq->H, sizeof(float2) * (size_t)N * n_streams);
    if (e == hipSuccess) e = hipMalloc((void**)&eq->Hm, sizeof(float2) * (size_t)N * n_streams);
    if (e == hipSuccess) e = hipMalloc((void**)&eq->pre, sizeof(float2) * (size_t)N * c->n_mimo_ltf * n_streams);
    if (e == hipSuccess) e = hipMalloc((void**)&eq->counters, sizeof(int) * 4 * (size_t)n_streams);
    if (e == hipSuccess) e = hipMemset(eq->H, 0, sizeof(float2) * (size_t)N * n_streams);
    if (e == hipSuccess) e = hipMemset(eq->Hm, 0, sizeof(float2) * (size_t)N * n_streams);
    if (e == hipSuccess) e = hipMemset(eq->pre, 0, sizeof(float2) * (size_t)N * c->n_mimo_ltf * n_streams);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);      // the memsets are on the null stream, the equalizer's launches on non-blocking streams: no order between them otherwise
    if (e != hipSuccess) { jrc_equalizer_destroy(eq); return jrc_fail(ctx, JRC_ERR_HIP, "jrc_equalizer_create: %s", hipGetErrorString(e)); }
    unsigned char* tb = (unsigned char*)eq->tables;
    EqDev& d = eq->d;
    d.N = N; d.cp = c->cp_len; d.ND = ND; d.NP = NP; d.NAct = (int)ac.size(); d.NL = c->n_mimo_ltf; d.T = T;
    d.mapped_cols = c->mapped_cols; d.n_pilot_rows = c->n_pilot_rows; d.estimator = c->estimator;
    d.freq = c->freq; d.bw = c->bw;
    d.sig_full = ctx->tune.eq_sig_full ? 1 : 0;
#ifdef JRC_TIMING_EXPERIMENTS
    d.exp = getenv("JRC_EQ_EXP") ? atoi(getenv("JRC_EQ_EXP")) : 0;
#else
    d.exp = 0;
#endif
    d.data_c = (const int*)tb; d.pilot_c = d.data_c + ND; d.active_c = d.pilot_c + NP;
    d.pilot_sym = (const float2*)(tb + off_ps); d.ltf = (const float2*)(tb + off_ps + b_ps);
    d.mapped = (const float2*)(tb + off_ps + b_ps + b_ltf);
    eq->threads = N >= 1024 ? 1024 : ((N + 63) / 64) * 64;
    {
        size_t off = sizeof(float2) * (size_t)(3 * N + ND + NP) + sizeof(unsigned long long) * eq_surv_words(ND) + eq_pair_bytes(ND);
        off = (off + 15) & ~size_t(15);
        // MIMO-LTF store in LDS, on Z / est / the Viterbi scratch grown to N x NL cells — when the dynamic LDS then still lets eight narrow
        // workgroups share a CU (18 KiB + ~2 KiB of static arrays each); JRC_EQ_PRE_LDS=0 keeps it in HBM
        d.pre_lds = 0;
        {
            const size_t tables = sizeof(int) * (size_t)(ND + NP + ac.size() + 1) + sizeof(float2) * (size_t)(N + NP) + 16;
            const size_t grown = std::max(off, (sizeof(float2) * (size_t)(3 * N + N * c->n_mimo_ltf) + 15) & ~size_t(15));
            const char* e = getenv("JRC_EQ_PRE_LDS");
            if (!(e && atoi(e) == 0) && grown + tables <= 18432) { d.pre_lds = 1; off = grown; }
        }
        d.lds_tables = (int)off;
        eq->lds_bytes = off + sizeof(int) * (size_t)(ND + NP + ac.size() + 1) + sizeof(float2) * (size_t)(N + NP) + 16;
    }
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<1024, 4, 4>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<1024, 4, 1>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 4, 1>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 2, 1>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 6, 1>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 8, 1>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 2, 2>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 2, 4>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 4, 2>, eq->lds_bytes));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)equalizer_kernel<256, 4, 4>, eq->lds_bytes));
    *out = eq;
    return JRC_OK;
}

// Launch geometry.  "Wide": one lane per subcarrier (fft_len threads, 128 VGPRs) — the shortest time for a single stream, used by the
// per-block entry point.  "Narrow": a quarter of the lanes with up to four subcarriers each and 256 VGPRs (no spills) — the sections of
// a frame that only one wavefront or one lane can work on (SIG Viterbi, running sums, channel means) then hold up a quarter of the
// waves, and twice as many streams are in flight per CU; used when a launch has at least a workgroup per CU.  Config C (fft_len 256,
// 8192 streams): 64 threads x 4 subcarriers 2.89 M frames/s, 128 x 2 2.53 M, wide 2.01 M (narrow at 168 VGPRs, with spills: 2.50 M).
// JRC_EQ_THREADS forces a workgroup size (-1: always wide), JRC_EQ_WPE the register budget (waves per SIMD: 2, 4; wide also 6, 8).
static void launch_equalizer(jrc_equalizer* eq, int grid, hipStream_t s, const EqIo& io)
{
    const int N = eq->d.N, forced = eq->ctx->tune.eq_threads;
    int threads = eq->threads, wpe = eq->ctx->tune.eq_wpe;
    if (forced > 0 && forced % 64 == 0 && forced <= 256 && forced < threads && N <= 4 * forced) threads = forced;
    else if (forced == 0 && N > 64 && N <= 1024 && grid >= eq->ctx->n_cus) {
        threads = N <= 128 ? 64 : ((N / 4 + 63) / 64) * 64;
        if (!wpe) wpe = 2;
    }
    if (!wpe) wpe = threads <= 64 ? 2 : 4;      // a single wavefront (fft_len <= 64): 171 VGPRs without spills beat 128 with 29 (comm receive chain, 1024 frames: 0.56 -> 0.53 ms)
#define EQ_LAUNCH(NTM, W, E) hipLaunchKernelGGL((equalizer_kernel<NTM, W, E>), dim3(grid), dim3(threads), eq->lds_bytes, s, eq->d, eq->states, eq->H, eq->Hm, eq->pre, io)
    if (N > threads && threads <= 256) {        // narrow
        if (N > 2 * threads) { if (wpe == 2) EQ_LAUNCH(256, 2, 4); else EQ_LAUNCH(256, 4, 4); }
        else { if (wpe == 2) EQ_LAUNCH(256, 2, 2); else EQ_LAUNCH(256, 4, 2); }
    }
    else if (N > threads) EQ_LAUNCH(1024, 4, 4);               // fft_len 2048 / 4096: several subcarriers per lane
    else if (threads > 256) EQ_LAUNCH(1024, 4, 1);
    else if (wpe == 2) EQ_LAUNCH(256, 2, 1);
    else if (wpe == 6) EQ_LAUNCH(256, 6, 1);
    else if (wpe == 8) EQ_LAUNCH(256, 8, 1);
    else EQ_LAUNCH(256, 4, 1);                                  // wide, measured on config C (frames/s): 8 waves/SIMD 1.27 M, 6: 1.35 M, 4: 1.44 M, 2: 1.32 M
#undef EQ_LAUNCH
}

extern "C" void jrc_equalizer_destroy(jrc_equalizer* eq)
{
    if (!eq) return;
    (void)hipSetDevice(eq->ctx->device);
    (void)hipStreamSynchronize(eq->ctx->stream);
    if (eq->tables) (void)hipFree(eq->tables);
    if (eq->states) (void)hipFree(eq->states);
    if (eq->H) (void)hipFree(eq->H);
    if (eq->Hm) (void)hipFree(eq->Hm);
    if (eq->pre) (void)hipFree(eq->pre);
    if (eq->counters) (void)hipFree(eq->counters);
    delete eq;
}

extern "C" int jrc_equalizer_set_estimator(jrc_equalizer* eq, int algo)
{
    if (!eq) return JRC_ERR_INVALID_ARG;
    if (algo != 0 && algo != 1) return jrc_fail(eq->ctx, JRC_ERR_INVALID_ARG, "[OFDM Equalizer] Estimator not implemented");
    eq->d.estimator = algo;
    return JRC_OK;
}
extern "C" int jrc_equalizer_set_bandwidth(jrc_equalizer* eq, double bw) { if (!eq) return JRC_ERR_INVALID_ARG; eq->d.bw = bw; return JRC_OK; }
extern "C" int jrc_equalizer_set_frequency(jrc_equalizer* eq, double f) { if (!eq) return JRC_ERR_INVALID_ARG; eq->d.freq = f; return JRC_OK; }

extern "C" int jrc_equalizer_work(jrc_equalizer* eq, int stream, int noutput_items, int ninput_items, const jrc_cf32* in,
                                  const int64_t* tag_offsets, const double* tag_values, int n_tags, jrc_cf32* out,
                                  int* n_consumed, jrc_eq_event* events, int max_events, int* n_events,
                                  jrc_cf32* chan_est, int* chan_est_written)
{
    if (!eq || !in || !out || !n_consumed || !n_events || (max_events > 0 && !events)) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = eq->ctx;
    if (stream < 0 || stream >= eq->n_streams || noutput_items < 0 || ninput_items < 0 || n_tags < 0 || max_events < 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_equalizer_work: bad stream/sizes");
    *n_consumed = 0; *n_events = 0;
    if (chan_est_written) *chan_est_written = 0;
    if (ninput_items == 0 || noutput_items == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const EqDev& d = eq->d;
    const size_t b_in = sizeof(float2) * (size_t)ninput_items * d.N, b_out = sizeof(float2) * (size_t)noutput_items * d.ND;
    const size_t b_tag = (sizeof(long long) + sizeof(double)) * (size_t)(n_tags ? n_tags : 1);
    const size_t b_ev = sizeof(jrc_eq_event) * (size_t)(max_events ? max_events : 1), b_ce = sizeof(float2) * (size_t)d.N * d.T;
    JRC_TRY(jrc_ensure_pinned(ctx, b_in + b_tag + b_out + b_ev + b_ce + 64));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, b_in + b_tag));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, b_out));
    JRC_TRY(jrc_ensure_scratch(ctx, 2, b_ev + b_ce));
    unsigned char* hp = (unsigned char*)ctx->pinned;
    memcpy(hp, in, b_in);
    long long* h_off = (long long*)(hp + b_in);
    double* h_val = (double*)(h_off + (n_tags ? n_tags : 1));
    for (int t = 0; t < n_tags; t++) { h_off[t] = tag_offsets[t]; h_val[t] = tag_values[t]; }
    if (!n_tags) { h_off[0] = -1; h_val[0] = 0; }
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], hp, b_in + b_tag, hipMemcpyHostToDevice, ctx->stream));
    EqIo io;
    io.in = (const float2*)ctx->scratch[0]; io.in_stride = 0; io.ninput = ninput_items;
    io.tag_offsets = (const long long*)((unsigned char*)ctx->scratch[0] + b_in);
    io.tag_values = (const double*)(io.tag_offsets + (n_tags ? n_tags : 1)); io.n_tags = n_tags ? n_tags : 1;
    io.out = (float2*)ctx->scratch[1]; io.out_stride = 0; io.noutput = noutput_items;
    int* cnt = eq->counters + 4 * (size_t)stream;
    io.n_out = cnt; io.n_consumed = cnt + 1; io.n_events = cnt + 2; io.chan_est_written = cnt + 3;
    io.events = (jrc_eq_event*)ctx->scratch[2]; io.max_events = max_events;
    io.chan_est = (float2*)((unsigned char*)ctx->scratch[2] + b_ev);
    io.stream0 = stream;
    launch_equalizer(eq, 1, ctx->stream, io);
    JRC_HIP(ctx, hipGetLastError());
    unsigned char* h_out = hp + b_in + b_tag;
    int* h_cnt = (int*)(h_out + b_out + b_ev + b_ce);
    JRC_HIP(ctx, hipMemcpyAsync(h_cnt, cnt, sizeof(int) * 4, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync(h_out, ctx->scratch[1], b_out, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync(h_out + b_out, ctx->scratch[2], b_ev + b_ce, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n_out = h_cnt[0];
    memcpy(out, h_out, sizeof(float2) * (size_t)n_out * d.ND);
    *n_consumed = h_cnt[1];
    *n_events = h_cnt[2];
    for (int i = 0; i < h_cnt[2]; i++) events[i] = ((jrc_eq_event*)(h_out + b_out))[i];
    if (h_cnt[3]) {
        if (chan_est) memcpy(chan_est, h_out + b_out + b_ev, b_ce);
        if (chan_est_written) *chan_est_written = 1;
    }
    return n_out;
}

extern "C" int jrc_equalizer_frames_dev(jrc_equalizer* eq, int n_streams, int n_symbols, const jrc_cf32* d_in,
                                        const double* d_phase, int max_out, jrc_cf32* d_out, int32_t* d_n_out,
                                        jrc_eq_event* d_events, void* stream)
{
    JRC_TRACE("jrc_equalizer_frames_dev");
    if (!eq || !d_in || !d_phase || !d_out || !d_n_out || !d_events) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = eq->ctx;
    if (n_streams <= 0 || n_streams > eq->n_streams || n_symbols <= 0 || max_out <= 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_equalizer_frames_dev: bad sizes");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    EqIo io;
    io.in = (const float2*)d_in; io.in_stride = (long)n_symbols * eq->d.N; io.ninput = n_symbols;
    io.tag_offsets = nullptr; io.tag_values = d_phase; io.n_tags = 1;
    io.out = (float2*)d_out; io.out_stride = (long)max_out * eq->d.ND; io.noutput = max_out;
    io.n_out = d_n_out; io.n_consumed = eq->counters; io.n_events = eq->counters + eq->n_streams;
    io.chan_est_written = nullptr; io.events = d_events; io.max_events = 2; io.chan_est = nullptr; io.stream0 = 0;
    launch_equalizer(eq, n_streams, s, io);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// C3 steering
__global__ void steering_kernel(const float2* __restrict__ h, float2* __restrict__ Q, int T, int n, int phased)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2* hv = h + (size_t)i * T;
    float2* q = Q + (size_t)i * T * T;
    const float sqT = sqrtf((float)T);
    if (phased) {                                                                        // :848-853
        float nrm = 0.f;
        for (int t = 0; t < T; t++) nrm = nrm + (hv[t].x * hv[t].x + hv[t].y * hv[t].y);
        nrm = sqrtf(nrm);
        // Q = Q * sqrt(T) / Q.norm() runs over the whole matrix (:851): the zero columns stay zero unless the row is all zero, where 0 / 0 makes every entry NaN
        for (int k = T; k < T * T; k++) q[k] = make_float2(0.f / nrm, 0.f / nrm);
        for (int t = 0; t < T; t++) q[t] = make_float2(hv[t].x * sqT / nrm, -hv[t].y * sqT / nrm);
        return;
    }
    float2 x[8], v[8];
    for (int t = 0; t < T; t++) x[t] = c_conj(hv[t]);
    float tail = 0.f;
    for (int t = 1; t < T; t++) tail = tail + (x[t].x * x[t].x + x[t].y * x[t].y);
    const float2 c0 = x[0];
    float2 tau;
    v[0] = make_float2(1.f, 0.f);
    if (tail <= 1.17549435e-38f && c0.y * c0.y <= 1.17549435e-38f) {                      // Eigen makeHouseholder degenerate case
        tau = make_float2(0.f, 0.f);
        for (int t = 1; t < T; t++) v[t] = make_float2(0.f, 0.f);
    } else {
        float beta = sqrtf((c0.x * c0.x + c0.y * c0.y) + tail);
        if (c0.x >= 0) beta = -beta;
        const float2 den = make_float2(c0.x - beta, c0.y);
        for (int t = 1; t < T; t++) v[t] = c_div(x[t], den);
        const float2 tt = make_float2((beta - c0.x) / beta, (-c0.y) / beta);
        tau = c_conj(tt);
    }
    float fro = 0.f;
    const float2 ctau = c_conj(tau);
    for (int col = 0; col < T; col++)
        for (int rw = 0; rw < T; rw++) {                                                  // V = H^H = I - conj(tau) v v^H
            const float2 vv = c_mul(c_mul(ctau, v[rw]), c_conj(v[col]));
            const float2 e = make_float2((rw == col ? 1.f : 0.f) - vv.x, -vv.y);
            q[(size_t)col * T + rw] = e;
            fro = fro + (e.x * e.x + e.y * e.y);
        }
    fro = sqrtf(fro);
    for (int k = 0; k < T * T; k++) q[k] = make_float2(q[k].x * sqT / fro, q[k].y * sqT / fro);   // :858
}

extern "C" int jrc_steering_from_channel(jrc_ctx* ctx, int T, int n, const jrc_cf32* h, int phased, jrc_cf32* Q)
{
    if (!ctx || !h || !Q || n < 0) return JRC_ERR_INVALID_ARG;
    if (T < 1 || T > 8) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "steering: N_tx must be in [1, 8]");
    if (n == 0) return JRC_OK;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t b_h = sizeof(float2) * (size_t)n * T, b_q = sizeof(float2) * (size_t)n * T * T;
    JRC_TRY(jrc_ensure_pinned(ctx, b_h + b_q));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, b_h));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, b_q));
    memcpy(ctx->pinned, h, b_h);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, b_h, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(steering_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, (const float2*)ctx->scratch[0],
                       (float2*)ctx->scratch[1], T, n, phased);
    JRC_HIP(ctx, hipGetLastError());
    JRC_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + b_h, ctx->scratch[1], b_q, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(Q, (char*)ctx->pinned + b_h, b_q);
    return JRC_OK;
}

static void dft_matrix_host(int T, float2* F)   // get_dft_matrix_eigen :761-772, column-major
{
    for (int r = 0; r < T; r++)
        for (int c = 0; c < T; c++) {
            const float ang = (float)(-2 * M_PI * float(r * c) / float(T));
            F[(size_t)c * T + r] = c_div(make_float2(cosf(ang), sinf(ang)), make_float2((float)std::sqrt((double)T), 0.f));   // std::exp(.) / (gr_complex) std::sqrt(N)
        }
}

extern "C" int jrc_dft_matrix(jrc_ctx* ctx, int T, jrc_cf32* F)
{
    if (!ctx || !F || T < 1 || T > 16) return JRC_ERR_INVALID_ARG;
    dft_matrix_host(T, (float2*)F);
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// C2 precoder: one lane per (OFDM symbol k, subcarrier sc), writing all T output ports
struct PreDev {
    int N, T, ND, NP, NS, n_pilot_rows;
    const int* data_c; const int* pilot_c; const short* role;   // role[sc]: -1 unused, 0..ND-1 data index, 0x4000|k pilot index
    const float2* pilot_sym; const float2* sync; const float2* mapped;
};

__global__ __launch_bounds__(256) void precoder_kernel(PreDev d, const float2* __restrict__ in, const float* __restrict__ sig,
                                                       int n_sym, int packet_type, int steer_mode,
                                                       const float2* __restrict__ Qm, const float2* __restrict__ Qsc,
                                                       const float2* __restrict__ rs, float2* __restrict__ out /* [T][n_total][N] */,
                                                       size_t in_stride, size_t rs_stride, size_t out_stride /* per frame (blockIdx.y) */)
{
    const int N = d.N, T = d.T, NL = d.T;
    const int n_total = n_sym + d.NS + T + 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_total * N) return;
    in += (size_t)blockIdx.y * in_stride;
    if (rs) rs += (size_t)blockIdx.y * rs_stride;
    out += (size_t)blockIdx.y * out_stride;
    const int k = (int)(idx / N), sc = (int)(idx % N);
    const short role = d.role[sc];
    float2 o[8];
    for (int t = 0; t < T; t++) o[t] = make_float2(0.f, 0.f);                             // memset :337
    if (k < d.NS) {                                                                       // sync words, ports 0,1 only :340-347
        for (int t = 0; t < T && t < 2; t++) o[t] = d.sync[(size_t)k * N + sc];
    } else if (k == d.NS) {                                                               // SIG :353-371
        float2 v = make_float2(0.f, 0.f);
        if (role >= 0 && !(role & 0x4000)) v = make_float2(sig[role], 0.f);
        else if (role >= 0) v = d.pilot_sym[role & 0x3fff];
        for (int t = 0; t < T && t < 2; t++) o[t] = v;
    } else if (k < d.NS + 1 + T) {                                                        // MIMO-LTFs
        const int l = k - d.NS - 1;
        const float2* X = d.mapped + (size_t)sc * T * NL;                                 // row-major T x NL
        if (packet_type == 1) {                                                           // NDP :379-388
            for (int t = 0; t < T; t++) o[t] = X[t * NL + l];
        } else {                                                                          // DATA :536-581
            bool zero = true;
            for (int i = 0; i < T * NL; i++) if (X[i].x != 0.f || X[i].y != 0.f) zero = false;
            if (!zero) {
                const float2* Q = steer_mode == 2 ? Qsc + (size_t)sc * T * T : Qm;
                for (int t = 0; t < T; t++) {
                    float2 acc = make_float2(0.f, 0.f);
                    for (int j = 0; j < T; j++) acc = cadd(acc, cmul(Q[(size_t)j * T + t], X[j * NL + l]));
                    o[t] = acc;
                }
            }
        }
    } else if (role >= 0) {                                                               // data / pilot carriers
        const int m = k - d.NS - 1 - T;
        float2 s0;
        if (role & 0x4000) s0 = d.pilot_sym[(size_t)(m % d.n_pilot_rows) * d.NP + (role & 0x3fff)];
        else s0 = in[(size_t)m * d.ND + role];
        if (packet_type == 1) {                                                           // NDP :394-428
            for (int t = 0; t < T && t < 2; t++) o[t] = s0;
        } else {                                                                          // DATA :589-712
            const float2* Q = steer_mode == 2 ? Qsc + (size_t)sc * T * T : Qm;
            for (int t = 0; t < T; t++) {
                float2 acc = cmul(Q[t], s0);                                              // column 0 = the data stream
                if (rs)
                    for (int j = 1; j < T; j++) acc = cadd(acc, cmul(Q[(size_t)j * T + t], rs[((size_t)(j - 1) * n_sym + m) * N + sc]));
                o[t] = acc;
            }
        }
    }
    for (int t = 0; t < T; t++) out[((size_t)t * n_total + k) * N + sc] = o[t];
}

// The same cell arithmetic with the steering matrix of the lane's subcarrier held in registers: one lane per subcarrier, the four
// waves of a workgroup (times gridDim.y workgroups) walk the OFDM symbols, so Q[sc] — T*T values at a 8*T*T-byte stride between
// lanes, the one uncoalesced read of the block — is fetched once per lane instead of once per cell, and every store and every
// radar-stream load is a full 512-byte wave access.  N_tx in {1, 2, 4, 8}; other sizes run precoder_kernel.
// four waves per SIMD (128 VGPRs): left to itself the compiler takes 178 and two waves fit (measured, config C: 0.304 ms at two, 0.271 at
// three, 0.260 at four; five and six spill: 0.55 / 0.78); eight TX antennas (64 matrix entries per lane) keep two
template <int T>
__global__ __launch_bounds__(256, T <= 4 ? 4 : 2) void precoder_frames_kernel(PreDev d, const float2* __restrict__ in, const float* __restrict__ sig,
                                                              int n_sym, int packet_type, int steer_mode,
                                                              const float2* __restrict__ Qm, const float2* __restrict__ Qsc,
                                                              const float2* __restrict__ rs, float2* __restrict__ out,
                                                              size_t in_stride, size_t rs_stride, size_t out_stride)
{
    const int N = d.N, NL = T;
    const int n_total = n_sym + d.NS + T + 1;
    const int sc = blockIdx.x * 64 + (threadIdx.x & 63);
    if (sc >= N) return;
    in += (size_t)blockIdx.z * in_stride;
    if (rs) rs += (size_t)blockIdx.z * rs_stride;
    out += (size_t)blockIdx.z * out_stride;
    const short role = d.role[sc];
    float2 q[T * T];
    {
        const float2* Q = steer_mode == 2 ? Qsc + (size_t)sc * T * T : Qm;
#pragma unroll
        for (int i = 0; i < T * T; i++) q[i] = Q[i];
    }
    const int kstep = 4 * gridDim.y;
    // the data symbol (and the radar streams) of the NEXT trip are requested before this trip's stores go out: the wait counter is in
    // order, so a load issued behind the stores would only come back after they have all completed
    const int k_data0 = d.NS + 1 + T;
    const bool want_rs = rs != nullptr && packet_type != 1;
    float2 s0n = make_float2(0.f, 0.f), rn[T];
#pragma unroll
    for (int j = 0; j < T; j++) rn[j] = make_float2(0.f, 0.f);
    auto fetch = [&](int kk) {
        if (kk >= k_data0 && kk < n_total && role >= 0) {
            const int m = kk - k_data0;
            if (role & 0x4000) s0n = d.pilot_sym[(size_t)(m % d.n_pilot_rows) * d.NP + (role & 0x3fff)];
            else s0n = in[(size_t)m * d.ND + role];
            if (want_rs) {
#pragma unroll
                for (int j = 1; j < T; j++) {          // read once: around the caches
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(rs + ((size_t)(j - 1) * n_sym + m) * N + sc));
                    rn[j] = make_float2(v.x, v.y);
                }
            }
        }
    };
    fetch(blockIdx.y * 4 + (threadIdx.x >> 6));
    for (int k = blockIdx.y * 4 + (threadIdx.x >> 6); k < n_total; k += kstep) {
        const float2 s0 = s0n;
        float2 r[T];
#pragma unroll
        for (int j = 0; j < T; j++) r[j] = rn[j];
        fetch(k + kstep);
        float2 o[T];
#pragma unroll
        for (int t = 0; t < T; t++) o[t] = make_float2(0.f, 0.f);                         // memset :337
        if (k < d.NS) {                                                                   // sync words, ports 0,1 only :340-347
            const float2 v = d.sync[(size_t)k * N + sc];
#pragma unroll
            for (int t = 0; t < T && t < 2; t++) o[t] = v;
        } else if (k == d.NS) {                                                           // SIG :353-371
            float2 v = make_float2(0.f, 0.f);
            if (role >= 0 && !(role & 0x4000)) v = make_float2(sig[role], 0.f);
            else if (role >= 0) v = d.pilot_sym[role & 0x3fff];
#pragma unroll
            for (int t = 0; t < T && t < 2; t++) o[t] = v;
        } else if (k < d.NS + 1 + T) {                                                    // MIMO-LTFs
            const int l = k - d.NS - 1;
            const float2* X = d.mapped + (size_t)sc * T * NL;                             // row-major T x NL
            if (packet_type == 1) {                                                       // NDP :379-388
#pragma unroll
                for (int t = 0; t < T; t++) o[t] = X[t * NL + l];
            } else {                                                                      // DATA :536-581
                bool zero = true;
                for (int i = 0; i < T * NL; i++) if (X[i].x != 0.f || X[i].y != 0.f) zero = false;
                if (!zero) {
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                        for (int j = 0; j < T; j++) acc = cadd(acc, cmul(q[j * T + t], X[j * NL + l]));
                        o[t] = acc;
                    }
                }
            }
        } else if (role >= 0) {                                                           // data / pilot carriers
            if (packet_type == 1) {                                                       // NDP :394-428
#pragma unroll
                for (int t = 0; t < T && t < 2; t++) o[t] = s0;
            } else {                                                                      // DATA :589-712
#pragma unroll
                for (int t = 0; t < T; t++) {
                    float2 acc = cmul(q[t], s0);                                          // column 0 = the data stream
                    if (rs) {
#pragma unroll
                        for (int j = 1; j < T; j++) acc = cadd(acc, cmul(q[j * T + t], r[j]));
                    }
                    o[t] = acc;
                }
            }
        }
        typedef float v2f __attribute__((ext_vector_type(2)));
        asm volatile("" ::: "memory");           // keeps the next trip's loads (above) ahead of these stores in the instruction stream
#pragma unroll
        for (int t = 0; t < T; t++) {            // write-once output, far larger than the caches for a batch: stored around them
            v2f v = {o[t].x, o[t].y};
            __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(out + ((size_t)t * n_total + k) * N + sc));
        }
    }
}

// one launch for n_frames packets (gridDim.z / gridDim.y = frame); k_blocks workgroups share a packet's symbols
static int launch_precoder(jrc_ctx* ctx, const PreDev& d, int n_frames, int k_blocks, const float2* in, const float* sig, int n_sym, int packet_type,
                           int steer_mode, const float2* Qm, const float2* Qsc, const float2* rs, float2* out, hipStream_t s)
{
    const int N = d.N, T = d.T;
    const int n_total = n_sym + d.NS + T + 1;
    const size_t total = (size_t)n_total * N;
    const size_t in_stride = (size_t)n_sym * d.ND, rs_stride = (size_t)(T - 1) * n_sym * N, out_stride = (size_t)T * n_total * N;
    for (int f0 = 0; f0 < n_frames; f0 += 65535) {
        const int nf = n_frames - f0 < 65535 ? n_frames - f0 : 65535;
        const float2* in_f = in + (size_t)f0 * in_stride;
        const float2* rs_f = rs ? rs + (size_t)f0 * rs_stride : nullptr;
        float2* out_f = out + (size_t)f0 * out_stride;
        const dim3 g((unsigned)((N + 63) / 64), (unsigned)k_blocks, (unsigned)nf), b(256);
        switch (T) {
            case 1: hipLaunchKernelGGL(precoder_frames_kernel<1>, g, b, 0, s, d, in_f, sig, n_sym, packet_type, steer_mode, Qm, Qsc, rs_f, out_f, in_stride, rs_stride, out_stride); break;
            case 2: hipLaunchKernelGGL(precoder_frames_kernel<2>, g, b, 0, s, d, in_f, sig, n_sym, packet_type, steer_mode, Qm, Qsc, rs_f, out_f, in_stride, rs_stride, out_stride); break;
            case 4: hipLaunchKernelGGL(precoder_frames_kernel<4>, g, b, 0, s, d, in_f, sig, n_sym, packet_type, steer_mode, Qm, Qsc, rs_f, out_f, in_stride, rs_stride, out_stride); break;
            case 8: hipLaunchKernelGGL(precoder_frames_kernel<8>, g, b, 0, s, d, in_f, sig, n_sym, packet_type, steer_mode, Qm, Qsc, rs_f, out_f, in_stride, rs_stride, out_stride); break;
            default:
                hipLaunchKernelGGL(precoder_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)nf), dim3(256), 0, s, d, in_f, sig, n_sym, packet_type,
                                   steer_mode, Qm, Qsc, rs_f, out_f, in_stride, rs_stride, out_stride);
        }
    }
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

struct jrc_precoder {
    jrc_ctx* ctx;
    PreDev d;
    void* tables = nullptr;
    float2 dft[64];
    std::vector<int> dc, pc;
    // batched entry point: SIG field of the last (mcs, packet_type, pdu_len) and the DFT matrix, on the device
    float* d_sig = nullptr;
    float2* d_dft = nullptr;
    int sig_key[3] = {-1, -1, -1};
};

extern "C" int jrc_precoder_create(jrc_ctx* ctx, const jrc_pre_cfg* c, jrc_precoder** out)
{
    if (!ctx || !c || !out) return JRC_ERR_INVALID_ARG;
    const int N = c->fft_len, T = c->N_tx, ND = c->n_data, NP = c->n_pilot;
    if (N <= 0 || T <= 0 || !c->data_carriers || ND <= 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "Data carriers must be of type vector of vector i.e. ().");           // :119-122
    if (!c->pilot_carriers || NP <= 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "Pilot carriers must be of type vector of vector i.e. ((),).");        // :139-141
    if (!c->pilot_symbols || c->n_pilot_rows <= 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "Pilot symbols must be of type vector of vector i.e. ((),).");         // :155-157
    if (!c->sync_words || !c->mapped_ltf || c->n_sync < 0) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "mimo_precoder: missing tables");
    if (T > 8 || ND < 48 || ND >= 0x4000) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "mimo_precoder: N_tx <= 8, 48 <= data carriers < 16384");
    jrc_precoder* p = new jrc_precoder();
    p->ctx = ctx;
    p->dc.resize(ND); p->pc.resize(NP);
    std::vector<short> role(N, -1);
    for (int i = 0; i < ND; i++) {                                                        // :124-137
        int v = c->data_carriers[i];
        if (v < 0) v += N;
        if (v > N || v < 0) { delete p; return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "data carrier index out of bounds"); }
        p->dc[i] = (v + N / 2) % N;
        role[p->dc[i]] = (short)i;
    }
    for (int i = 0; i < NP; i++) {                                                        // :143-153
        int v = c->pilot_carriers[i];
        if (v < 0) v += N;
        if (v > N || v < 0) { delete p; return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "pilot carrier index out of bounds"); }
        p->pc[i] = (v + N / 2) % N;
        role[p->pc[i]] = (short)(0x4000 | i);
    }
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t b_role = (sizeof(short) * N + 15) & ~size_t(15);
    const size_t b_ps = sizeof(float2) * (size_t)c->n_pilot_rows * NP, b_sync = sizeof(float2) * (size_t)c->n_sync * N;
    const size_t b_map = sizeof(float2) * (size_t)N * T * T;
    std::vector<unsigned char> host(b_role + b_ps + b_sync + b_map);
    memcpy(host.data(), role.data(), sizeof(short) * N);
    memcpy(host.data() + b_role, c->pilot_symbols, b_ps);
    memcpy(host.data() + b_role + b_ps, c->sync_words, b_sync);
    memcpy(host.data() + b_role + b_ps + b_sync, c->mapped_ltf, b_map);
    hipError_t e = hipMalloc(&p->tables, host.size());
    if (e == hipSuccess) e = hipMemcpy(p->tables, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { jrc_precoder_destroy(p); return jrc_fail(ctx, JRC_ERR_HIP, "jrc_precoder_create: %s", hipGetErrorString(e)); }
    unsigned char* tb = (unsigned char*)p->tables;
    PreDev& d = p->d;
    d.N = N; d.T = T; d.ND = ND; d.NP = NP; d.NS = c->n_sync; d.n_pilot_rows = c->n_pilot_rows;
    d.data_c = nullptr; d.pilot_c = nullptr; d.role = (const short*)tb;
    d.pilot_sym = (const float2*)(tb + b_role); d.sync = (const float2*)(tb + b_role + b_ps);
    d.mapped = (const float2*)(tb + b_role + b_ps + b_sync);
    dft_matrix_host(T, p->dft);
    *out = p;
    return JRC_OK;
}

extern "C" void jrc_precoder_destroy(jrc_precoder* p)
{
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->tables) (void)hipFree(p->tables);
    if (p->d_sig) (void)hipFree(p->d_sig);
    if (p->d_dft) (void)hipFree(p->d_dft);
    delete p;
}

extern "C" int jrc_precoder_output_length(const jrc_precoder* p, int ninput_items)
{
    if (!p) return JRC_ERR_INVALID_ARG;
    return p->d.NS + 1 + p->d.T + ninput_items / p->d.ND;                                 // :265-272
}

extern "C" int jrc_precoder_work(jrc_precoder* p, int ninput_items, const jrc_cf32* in, int mcs, int packet_type,
                                 int pdu_len, int steer_mode, const jrc_cf32* Q_mean, const jrc_cf32* Q_sc,
                                 const jrc_cf32* radar_streams, jrc_cf32* const* out)
{
    if (!p || !in || !out) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = p->ctx;
    const PreDev& d = p->d;
    const int N = d.N, T = d.T;
    if (packet_type != 1 && packet_type != 2)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "[MIMO PRECODER] packet type is not defined!");              // :716-719
    if (mcs < 0 || mcs > 5) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "[MIMO PRECODER] unknown mcs");
    const int n_sym = ninput_items / d.ND;
    if (n_ofdm_sym_dev(mcs, d.ND, pdu_len) != n_sym)
        return jrc_fail(ctx, JRC_ERR_SIG_FIELD, "%s", jrc_strerror(JRC_ERR_SIG_FIELD));                         // :327-333
    if (steer_mode < 0 || steer_mode > 2 || (steer_mode == 1 && !Q_mean) || (steer_mode == 2 && !Q_sc))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "mimo_precoder: steering matrices missing for steer_mode %d", steer_mode);
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const int n_total = n_sym + d.NS + T + 1;
    const size_t b_in = sizeof(float2) * (size_t)n_sym * d.ND, b_sig = sizeof(float) * (size_t)d.ND;
    const size_t b_qm = sizeof(float2) * (size_t)T * T, b_qsc = steer_mode == 2 ? sizeof(float2) * (size_t)N * T * T : 0;
    const size_t b_rs = radar_streams ? sizeof(float2) * (size_t)(T - 1) * n_sym * N : 0;
    const size_t b_out = sizeof(float2) * (size_t)T * n_total * N;
    const size_t b_up = b_in + b_sig + b_qm + b_qsc + b_rs;
    JRC_TRY(jrc_ensure_pinned(ctx, b_up > b_out ? b_up : b_out));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, b_up));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, b_out));
    unsigned char* hp = (unsigned char*)ctx->pinned;
    memcpy(hp, in, b_in);
    JRC_TRY(jrc_sig_encode(d.ND, mcs, packet_type, pdu_len, (float*)(hp + b_in)));                              // generate_signal_field
    memcpy(hp + b_in + b_sig, steer_mode == 1 ? (const void*)Q_mean : (const void*)p->dft, b_qm);
    if (b_qsc) memcpy(hp + b_in + b_sig + b_qm, Q_sc, b_qsc);
    if (b_rs) memcpy(hp + b_in + b_sig + b_qm + b_qsc, radar_streams, b_rs);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], hp, b_up, hipMemcpyHostToDevice, ctx->stream));
    unsigned char* dp = (unsigned char*)ctx->scratch[0];
    // one packet: spread its symbols over up to 16 workgroups per subcarrier tile
    int kb = (n_total + 3) / 4; if (kb > 16) kb = 16;
    JRC_TRY(launch_precoder(ctx, p->d, 1, kb, (const float2*)dp, (const float*)(dp + b_in), n_sym, packet_type, steer_mode,
                            (const float2*)(dp + b_in + b_sig), b_qsc ? (const float2*)(dp + b_in + b_sig + b_qm) : nullptr,
                            b_rs ? (const float2*)(dp + b_in + b_sig + b_qm + b_qsc) : nullptr, (float2*)ctx->scratch[1], ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync(hp, ctx->scratch[1], b_out, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < T; t++) memcpy(out[t], hp + sizeof(float2) * (size_t)t * n_total * N, sizeof(float2) * (size_t)n_total * N);
    return n_total;
}

// Batched, device-resident C2: n_frames packets of one (mcs, packet_type, pdu_len) per launch, symbols in and port buffers out in
// HBM; grid.y = frame.  Same kernel, same arithmetic as jrc_precoder_work.
extern "C" int jrc_precoder_frames_dev(jrc_precoder* p, int n_frames, int ninput_items, const jrc_cf32* d_in, int mcs, int packet_type,
                                       int pdu_len, int steer_mode, const jrc_cf32* d_Q_mean, const jrc_cf32* d_Q_sc,
                                       const jrc_cf32* d_radar_streams, jrc_cf32* d_out, void* stream)
{
    JRC_TRACE("jrc_precoder_frames_dev");
    if (!p || !d_in || !d_out || n_frames < 0) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = p->ctx;
    const PreDev& d = p->d;
    const int T = d.T;
    if (packet_type != 1 && packet_type != 2)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "[MIMO PRECODER] packet type is not defined!");              // :716-719
    if (mcs < 0 || mcs > 5) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "[MIMO PRECODER] unknown mcs");
    const int n_sym = ninput_items / d.ND;
    if (n_ofdm_sym_dev(mcs, d.ND, pdu_len) != n_sym)
        return jrc_fail(ctx, JRC_ERR_SIG_FIELD, "%s", jrc_strerror(JRC_ERR_SIG_FIELD));                         // :327-333
    if (steer_mode < 0 || steer_mode > 2 || (steer_mode == 1 && !d_Q_mean) || (steer_mode == 2 && !d_Q_sc))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "mimo_precoder: steering matrices missing for steer_mode %d", steer_mode);
    if (n_frames == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (!p->d_sig) {
        JRC_HIP(ctx, hipMalloc((void**)&p->d_sig, sizeof(float) * (size_t)d.ND));
        JRC_HIP(ctx, hipMalloc((void**)&p->d_dft, sizeof(float2) * (size_t)T * T));
        JRC_HIP(ctx, hipMemcpy(p->d_dft, p->dft, sizeof(float2) * (size_t)T * T, hipMemcpyHostToDevice));
    }
    if (p->sig_key[0] != mcs || p->sig_key[1] != packet_type || p->sig_key[2] != pdu_len) {                    // generate_signal_field, once per format
        std::vector<float> sig((size_t)d.ND);
        JRC_TRY(jrc_sig_encode(d.ND, mcs, packet_type, pdu_len, sig.data()));
        JRC_HIP(ctx, hipStreamSynchronize(s));                                                                  // earlier launches may still read the old field
        JRC_HIP(ctx, hipMemcpy(p->d_sig, sig.data(), sizeof(float) * (size_t)d.ND, hipMemcpyHostToDevice));
        p->sig_key[0] = mcs; p->sig_key[1] = packet_type; p->sig_key[2] = pdu_len;
    }
    const int n_total = n_sym + d.NS + T + 1;
    JRC_TRY(launch_precoder(ctx, p->d, n_frames, n_frames >= 64 ? 1 : 4, (const float2*)d_in, (const float*)p->d_sig, n_sym, packet_type, steer_mode,
                            steer_mode == 1 ? (const float2*)d_Q_mean : (const float2*)p->d_dft, steer_mode == 2 ? (const float2*)d_Q_sc : nullptr,
                            (const float2*)d_radar_streams, (float2*)d_out, s));
    JRC_HIP(ctx, hipGetLastError());
    return n_total;
}


// =====================================================================================================================
// ofdm_frame_generator (lib/ofdm_frame_generator_impl.cc:155-216): the SISO carrier allocator.  One workgroup per packet:
// zero fill, sync words, one lane per input symbol scatters it to (OFDM symbol, carrier) through the cyclic carrier sets,
// one lane per (OFDM symbol, pilot) writes the pilots.
// =====================================================================================================================
__global__ __launch_bounds__(256) void frame_generator_kernel(int N, int n_occ_sets, const int* __restrict__ occ_off, const int* __restrict__ occ_flat,
                                                              int n_pil_sets, const int* __restrict__ pil_off, const int* __restrict__ pil_flat,
                                                              int n_psym_sets, const int* __restrict__ psym_off, const float2* __restrict__ psym_flat,
                                                              int n_sync, const float2* __restrict__ sync_words, int n_in, long in_stride,
                                                              const float2* __restrict__ in, int n_out_sym, long out_stride, float2* __restrict__ out)
{
    const size_t b = blockIdx.x;
    const float2* x = in + b * (size_t)in_stride;
    float2* o = out + b * (size_t)out_stride;
    const int tid = threadIdx.x;
    const int sps = occ_off[n_occ_sets];
    const long total = (long)n_out_sym * N;
    for (long i = tid; i < total; i += 256) o[i] = (i < (long)n_sync * N) ? sync_words[i] : make_float2(0.f, 0.f);   // :165-170
    __syncthreads();
    float2* d = o + (size_t)n_sync * N;
    for (int i = tid; i < n_in; i += 256) {                                              // :175-200
        const int cycle = i / sps, rem = i % sps;
        int k = 0;
        while (rem >= occ_off[k + 1]) k++;
        d[(size_t)(cycle * n_occ_sets + k) * N + occ_flat[rem]] = x[i];
    }
    __syncthreads();
    const int n_ofdm = n_out_sym - n_sync;
    for (int sy = 0; sy < n_ofdm; sy++) {                                                // :202-208 (after the data: pilots win on a shared carrier)
        const int pk = sy % n_pil_sets, sk = sy % n_psym_sets;
        const int np = pil_off[pk + 1] - pil_off[pk];
        for (int k = tid; k < np; k += 256) d[(size_t)sy * N + pil_flat[pil_off[pk] + k]] = psym_flat[psym_off[sk] + k];
    }
}

struct jrc_frame_generator {
    jrc_ctx* ctx;
    int N, n_occ_sets, n_pil_sets, n_psym_sets, n_sync, sps;
    std::vector<int> occ_sizes;
    int *d_occ_off = nullptr, *d_occ_flat = nullptr, *d_pil_off = nullptr, *d_pil_flat = nullptr, *d_psym_off = nullptr;
    float2 *d_psym = nullptr, *d_sync = nullptr;
};

extern "C" void jrc_frame_generator_destroy(jrc_frame_generator* g)
{
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    (void)hipFree(g->d_occ_off); (void)hipFree(g->d_occ_flat); (void)hipFree(g->d_pil_off); (void)hipFree(g->d_pil_flat);
    (void)hipFree(g->d_psym_off); (void)hipFree(g->d_psym); (void)hipFree(g->d_sync);
    delete g;
}

template <class T> static bool fg_upload(T** dst, const std::vector<T>& v)
{
    const size_t n = v.size() ? v.size() : 1;
    if (hipMalloc((void**)dst, sizeof(T) * n) != hipSuccess) return false;
    return v.empty() || hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) == hipSuccess;
}

// carrier sets are passed flattened with per-set sizes, exactly as given to make(): negative indices wrap (+fft_len), and with
// output_is_shifted every index is rotated by fft_len/2 (:83-113).  Returns NULL on the constructor's invalid_argument cases.
extern "C" jrc_frame_generator* jrc_frame_generator_create(jrc_ctx* ctx, int fft_len, int n_occ_sets, const int* occ_sizes, const int* occ_flat,
                                                           int n_pil_sets, const int* pil_sizes, const int* pil_flat, int n_psym_sets,
                                                           const int* psym_sizes, const jrc_cf32* psym_flat, int n_sync, const jrc_cf32* sync_words,
                                                           int output_is_shifted)
{
    if (!ctx) return nullptr;
    auto fail = [&](const char* m) -> jrc_frame_generator* { jrc_fail(ctx, JRC_ERR_INVALID_ARG, "%s", m); return nullptr; };
    if (fft_len < 1 || n_occ_sets < 1 || !occ_sizes) return fail("Occupied carriers must be of type vector of vector i.e. ((),).");
    if (n_pil_sets < 1 || !pil_sizes) return fail("Pilot carriers must be of type vector of vector i.e. ((),).");
    if (n_psym_sets < 1 || !psym_sizes) return fail("Pilot symbols must be of type vector of vector i.e. ((),).");
    if (n_sync < 0 || (n_sync > 0 && !sync_words)) return fail("sync words must be fft length");
    std::vector<int> occ_off(1, 0), occ, pil_off(1, 0), pil, ps_off(1, 0);
    for (int k = 0, p = 0; k < n_occ_sets; k++) {
        for (int j = 0; j < occ_sizes[k]; j++, p++) {
            int c = occ_flat[p];
            if (c < 0) c += fft_len;
            if (c > fft_len || c < 0) return fail("data carrier index out of bounds");
            if (output_is_shifted) c = (c + fft_len / 2) % fft_len;
            occ.push_back(c);
        }
        occ_off.push_back((int)occ.size());
    }
    for (int k = 0, p = 0; k < n_pil_sets; k++) {
        for (int j = 0; j < pil_sizes[k]; j++, p++) {
            int c = pil_flat[p];
            if (c < 0) c += fft_len;
            if (c > fft_len || c < 0) return fail("pilot carrier index out of bounds");
            if (output_is_shifted) c = (c + fft_len / 2) % fft_len;
            pil.push_back(c);
        }
        pil_off.push_back((int)pil.size());
    }
    for (int k = 0; k < n_psym_sets; k++) ps_off.push_back(ps_off.back() + psym_sizes[k]);
    for (int i = 0; i < std::max(n_pil_sets, n_psym_sets); i++)
        if (pil_sizes[i % n_pil_sets] != psym_sizes[i % n_psym_sets]) return fail("pilot_carriers do not match pilot_symbols");
    for (size_t i = 0; i < occ.size(); i++) if (occ[i] >= fft_len) return fail("data carrier index out of bounds");     // index == fft_len passes :88 but is no carrier
    for (size_t i = 0; i < pil.size(); i++) if (pil[i] >= fft_len) return fail("pilot carrier index out of bounds");
    if (occ.empty()) return fail("Occupied carriers must be of type vector of vector i.e. ((),).");
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    jrc_frame_generator* g = new jrc_frame_generator();
    g->ctx = ctx; g->N = fft_len; g->n_occ_sets = n_occ_sets; g->n_pil_sets = n_pil_sets; g->n_psym_sets = n_psym_sets; g->n_sync = n_sync;
    g->sps = (int)occ.size();
    g->occ_sizes.assign(occ_sizes, occ_sizes + n_occ_sets);
    std::vector<float2> ps((const float2*)psym_flat, (const float2*)psym_flat + ps_off.back());
    std::vector<float2> sw((const float2*)sync_words, (const float2*)sync_words + (size_t)n_sync * fft_len);
    if (!fg_upload(&g->d_occ_off, occ_off) || !fg_upload(&g->d_occ_flat, occ) || !fg_upload(&g->d_pil_off, pil_off) || !fg_upload(&g->d_pil_flat, pil) ||
        !fg_upload(&g->d_psym_off, ps_off) || !fg_upload(&g->d_psym, ps) || !fg_upload(&g->d_sync, sw)) {
        jrc_fail(ctx, JRC_ERR_NOMEM, "ofdm_frame_generator: allocation failed");
        jrc_frame_generator_destroy(g);
        return nullptr;
    }
    return g;
}

extern "C" int jrc_frame_generator_output_length(const jrc_frame_generator* g, int ninput_items)        // :143-153
{
    if (!g || ninput_items < 0) return JRC_ERR_INVALID_ARG;
    int nout = (ninput_items / g->sps) * g->n_occ_sets;
    int k = 0;
    for (int i = 0; i < ninput_items % g->sps; k++) { nout++; i += g->occ_sizes[k % g->n_occ_sets]; }
    return nout + g->n_sync;
}

extern "C" int jrc_frame_generator_dev(jrc_frame_generator* g, int n_packets, int ninput_items, const jrc_cf32* d_in, jrc_cf32* d_out, void* stream)
{
    if (!g || n_packets < 0 || ninput_items < 0) return JRC_ERR_INVALID_ARG;
    const int nout = jrc_frame_generator_output_length(g, ninput_items);
    if (n_packets == 0) return nout;
    if ((ninput_items > 0 && !d_in) || !d_out) return jrc_fail(g->ctx, JRC_ERR_INVALID_ARG, "ofdm_frame_generator: null buffers");
    JRC_BIND(g->ctx);
    hipStream_t s = stream ? (hipStream_t)stream : g->ctx->stream;
    hipLaunchKernelGGL(frame_generator_kernel, dim3(n_packets), dim3(256), 0, s, g->N, g->n_occ_sets, (const int*)g->d_occ_off, (const int*)g->d_occ_flat,
                       g->n_pil_sets, (const int*)g->d_pil_off, (const int*)g->d_pil_flat, g->n_psym_sets, (const int*)g->d_psym_off,
                       (const float2*)g->d_psym, g->n_sync, (const float2*)g->d_sync, ninput_items, (long)ninput_items, (const float2*)d_in, nout,
                       (long)nout * g->N, (float2*)d_out);
    JRC_HIP(g->ctx, hipGetLastError());
    return nout;
}

extern "C" int jrc_frame_generator_work(jrc_frame_generator* g, int ninput_items, const jrc_cf32* in, jrc_cf32* out)
{
    if (!g || ninput_items < 0 || (ninput_items > 0 && !in) || !out) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = g->ctx;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const int nout = jrc_frame_generator_output_length(g, ninput_items);
    const size_t ib = sizeof(float2) * (size_t)(ninput_items > 0 ? ninput_items : 1), ob = sizeof(float2) * (size_t)nout * g->N;
    if (ob == 0) return 0;
    JRC_TRY(jrc_ensure_pinned(ctx, ib + ob));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, ib));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, ob));
    if (ninput_items) memcpy(ctx->pinned, in, sizeof(float2) * (size_t)ninput_items);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, ib, hipMemcpyHostToDevice, ctx->stream));
    JRC_TRY(jrc_frame_generator_dev(g, 1, ninput_items, (const jrc_cf32*)ctx->scratch[0], (jrc_cf32*)ctx->scratch[1], ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + ib, ctx->scratch[1], ob, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, (char*)ctx->pinned + ib, ob);
    return nout;
}
