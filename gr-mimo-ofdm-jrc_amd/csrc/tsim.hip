// tsim.hip — target_simulator on the device (reference lib/target_simulator_impl.cc:132-385).
//
// Per burst of n samples (n is arbitrary: (preamble + symbols + 3 pad symbols) x (fft_len + cp), e.g. 22080) and per
// target k the reference computes, for every RX antenna l,
//      out_l = IFFT_n( FFT_n( in . doppler_k ) . timeshift_{l,k} ) [. phase_k]   (+ self coupling)
// with two FFTW3f transforms of length n per (l, k).  Here:
//   * FFT_n( in . doppler_k ) does not depend on l and is computed once per target;
//   * lengths n = n1 x 2^a (2^a >= 16, n1 <= 512: every burst of the flowgraphs) are transformed directly as four-step DFTs — the td_* kernels
//     further down (round 5); what follows here is the route for all other lengths:
//   * the length-n DFTs are chirp-z (Bluestein) transforms over a power-of-two M >= 2n-1:
//         X[k] = c[k] . sum_i (x[i] c[i]) conj(c)[k-i],   c[i] = exp(-j pi i^2 / n)
//     and because the inverse DFT uses conj(c), the chirp factors between the two transforms cancel
//     (X[i] ts[i] conj(c)[i] = conv[i] ts[i]):  the whole block is two circular convolutions of length M;
//   * every length-M FFT is a four-step transform M = 256 x n2: a column pass (256-point FFT over a stride-n2 column,
//     16 columns per workgroup so each row access is a 128-byte segment, FFT in registers as 16 x 16 with one LDS
//     exchange) and a row pass (n2-point Stockham FFT in LDS).  The forward transform leaves its output in
//     [k1][k2] (digit-swapped) order; the pointwise product with the chirp spectrum and the inverse row FFT happen
//     on the same row in the same workgroup, and the inverse column pass restores natural order — no transposes;
//   * the inverse column pass of the first convolution, the product with timeshift_{l,k} and the forward column pass
//     of the second convolution for all R antennas are one kernel (the data is already distributed as the next
//     transform needs it).
// The channel filters themselves (doppler_k, timeshift_{l,k}) are the reference's float/double recurrences,
// evaluated on the host once per burst length like the reference does (:249-300) and kept in HBM.
#include "fft_device.h"
#include "tsim_device.h"

#include <cmath>
#include <complex>
#include <vector>

#define TS_N1 256      // column-pass FFT length
#define TS_CW 16       // columns per workgroup (16 x 8 B = 128-byte row segments)
#define TS_XPAD 272    // LDS stride between q planes of the 16x16 exchange (256 + 16: see fft256_cols)

struct jrc_tsim {
    jrc_ctx* ctx = nullptr;
    int K = 0, R = 0;
    std::vector<float> range, velocity, rcs, azimuth, position_rx;
    int samp_rate = 0;
    float center_freq = 0.f, self_coupling_db = 0.f;
    int rndm_phaseshift = 0, self_coupling = 0, sum_targets = 0;
    int max_bursts = 1;
    std::vector<float> doppler, scale_ampl, timeshift;     // [K], [K], [R][K]
    // per burst length
    int n = 0, M = 0, n2 = 0;
    float2* d_dop = nullptr;      // [K][n]
    float2* d_ts = nullptr;       // [R][K][n]
    float2* d_chirp = nullptr;    // [n]   c[i] = exp(-j pi i^2 / n)
    float2* d_bhat = nullptr;     // [256][n2] FFT_M(conj(c) wrapped) / M in [k1][k2] order
    float2* d_u = nullptr;        // [max_bursts][Kz][M], Kz = K when sum_targets else 1
    float2* d_g = nullptr;        // [max_bursts][R][M]
    float2* d_phase = nullptr;    // [K]
    size_t u_cap = 0, g_cap = 0;
    // direct four-step plan (td_*): n = n1 x n2, n2 a power of two — no chirp, no padding; d_ts then holds timeshift permuted to [R][K][n1][n2]
    bool direct = false;
    int n1 = 0;                   // column transform length (rows of the [n1][n2] view); n2 above is the row length
    int nrad = 0, rad[24] = {0};  // radices of the n1-point column transform
    float2* d_w1 = nullptr;       // [n1]      exp(-j 2 pi q / n1)
    float2* d_two = nullptr;      // [n1][n2]  exp(-j 2 pi i2 k1 / n)
};

__device__ __forceinline__ float2 conjf2(float2 a) { return make_float2(a.x, -a.y); }

// ---- 256-point forward FFT over 16 columns, in registers ---------------------------------------
// thread (c = tid & 15, s = tid >> 4) enters with x[j] = element (s + 16 j) of column c and leaves with
// x[r] = element (q + 16 r), q = s:  X[q + 16 r] = sum_s w16^{s r} [ w256^{s q} sum_j x[s + 16 j] w16^{j q} ].
// xch: 16 * TS_XPAD float2 of LDS; plane q at q*TS_XPAD, then [s][c] with c fastest, so the 64 lanes of a wave write
// 512 contiguous bytes and read four 128-byte runs whose banks pair up without conflict.
__device__ __forceinline__ void fft256_cols(float2 (&x)[16], float2* xch, const float2* tw256 /* LDS, exp(-j2pi k/256) */)
{
    const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
    fft_fwd_small<16>(x);
#pragma unroll
    for (int q = 1; q < 16; q++) x[q] = cmul(x[q], tw256[s * q]);
    __syncthreads();                                   // previous user of xch is done
#pragma unroll
    for (int q = 0; q < 16; q++) xch[q * TS_XPAD + s * 16 + c] = x[q];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = xch[s * TS_XPAD + j * 16 + c];   // my q = s: A[s' = j][q]
    fft_fwd_small<16>(x);
}
__device__ __forceinline__ void swap_reim(float2 (&x)[16])
{
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = make_float2(x[j].y, x[j].x);
}
// inverse (unnormalised) via IFFT(x) = swap(FFT(swap(x)))
__device__ __forceinline__ void ifft256_cols(float2 (&x)[16], float2* xch, const float2* tw256)
{
    swap_reim(x); fft256_cols(x, xch, tw256); swap_reim(x);
}

// ---- the same 256-point FFT over 16 contiguous rows of 256 points (row pass when n2 == 256) --------------------
// thread (rl = tid >> 4, s = tid & 15) holds x[j] = element (s + 16 j) of row rl on entry and element (s + 16 r) of the
// transform on exit, so a forward transform, a pointwise product and an inverse transform chain without any
// redistribution.  Exchange layout: row rl at rl*TS_XPAD, plane q at q*17, then s — writes are 16-lane contiguous
// runs, reads have stride 17 (odd) inside a row and the four rows of a wave pair up on disjoint banks.
__device__ __forceinline__ void fft256_rows(float2 (&x)[16], float2* xch, const float2* tw256)
{
    const int s = threadIdx.x & 15, rl = threadIdx.x >> 4;
    fft_fwd_small<16>(x);
#pragma unroll
    for (int q = 1; q < 16; q++) x[q] = cmul(x[q], tw256[s * q]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; q++) xch[rl * TS_XPAD + q * 17 + s] = x[q];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = xch[rl * TS_XPAD + s * 17 + j];
    fft_fwd_small<16>(x);
}

// row pass for n2 == 256: 16 rows per workgroup, forward FFT -> . Bhat (or conj) -> inverse FFT, all in registers
__global__ __launch_bounds__(256) void tsim_rowconv256_kernel(float2* __restrict__ X, const float2* __restrict__ bhat, int conj_b,
                                                              const float2* __restrict__ tw256_g, size_t rows)
{
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    const int s = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const size_t row = (size_t)blockIdx.x * 16 + rl;            // rows is a multiple of 256
    tw256[threadIdx.x] = tw256_g[threadIdx.x];
    float2* g = X + row * 256;
    const float2* brow = bhat + (size_t)(row & (TS_N1 - 1)) * 256;
    float2 x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = g[s + 16 * j];
    __syncthreads();
    fft256_rows(x, xch, tw256);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float2 bv = brow[s + 16 * r];
        if (conj_b) bv.y = -bv.y;
        x[r] = cmul(x[r], bv);
    }
    swap_reim(x); fft256_rows(x, xch, tw256); swap_reim(x);
#pragma unroll
    for (int r = 0; r < 16; r++) g[s + 16 * r] = x[r];
    (void)rows;
}

// row pass for n2 = m * 256, m in {2, 4, 8, 16}: the row is itself a two-step transform  i = i_a*256 + i_b  ->
// k = k_a + m*k_b  (m-point FFTs over i_a in registers, twiddle w_n2^{i_b k_a}, then 256-point FFTs over i_b as above).
// The transform is left in [k_a][k_b] order: the chirp spectrum is produced by the same kernel (FWD_ONLY), and the
// inverse runs the steps backwards, so no reordering pass exists.  16*m threads per row, 16/m rows per workgroup.
template <int M_, bool FWD_ONLY>
__global__ __launch_bounds__(256) void tsim_rowconv_m_kernel(float2* __restrict__ X, const float2* __restrict__ bhat, int conj_b,
                                                             const float2* __restrict__ tw256_g, const float2* __restrict__ twn2_g)
{
    constexpr int TPR = 16 * M_;             // threads per row
    constexpr int U = 16 / M_;               // i_b values per thread in the m-point step
    constexpr int N2 = 256 * M_;
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    const int rowl = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int s = threadIdx.x & 15, plane = threadIdx.x >> 4;          // plane = rowl * M_ + k_a
    const size_t row = (size_t)blockIdx.x * (256 / TPR) + rowl;
    tw256[threadIdx.x] = tw256_g[threadIdx.x];
    float2* g = X + row * N2;
    float2 y[16];                             // [u][i_a]
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int a = 0; a < M_; a++) y[u * M_ + a] = g[a * 256 + t + TPR * u];
#pragma unroll
    for (int u = 0; u < U; u++) {
        fft_fwd_small<M_>(*reinterpret_cast<float2(*)[M_]>(&y[u * M_]));
        const int ib = t + TPR * u;
#pragma unroll
        for (int a = 1; a < M_; a++) y[u * M_ + a] = cmul(y[u * M_ + a], twn2_g[ib * a]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int a = 0; a < M_; a++) xch[(rowl * M_ + a) * TS_XPAD + t + TPR * u] = y[u * M_ + a];
    __syncthreads();
    float2 x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = xch[plane * TS_XPAD + s + 16 * j];
    fft256_rows(x, xch, tw256);               // x[r] = transform element k_b = s + 16 r of plane k_a
    const int ka = plane % M_;
    if (FWD_ONLY) {
#pragma unroll
        for (int r = 0; r < 16; r++) g[ka * 256 + s + 16 * r] = x[r];
        return;
    }
    const float2* brow = bhat + (row & (TS_N1 - 1)) * (size_t)N2 + ka * 256;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float2 bv = brow[s + 16 * r];
        if (conj_b) bv.y = -bv.y;
        x[r] = cmul(x[r], bv);
    }
    swap_reim(x); fft256_rows(x, xch, tw256); swap_reim(x);            // x[r] = element i_b = s + 16 r of plane k_a
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) xch[plane * TS_XPAD + s + 16 * r] = x[r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int ib = t + TPR * u;
#pragma unroll
        for (int a = 0; a < M_; a++) {
            float2 v = xch[(rowl * M_ + a) * TS_XPAD + ib];
            if (a) v = cmul(v, conjf2(twn2_g[ib * a]));
            y[u * M_ + a] = make_float2(v.y, v.x);                      // swapped: inverse m-point FFT via the forward one
        }
        fft_fwd_small<M_>(*reinterpret_cast<float2(*)[M_]>(&y[u * M_]));
#pragma unroll
        for (int a = 0; a < M_; a++) g[a * 256 + ib] = make_float2(y[u * M_ + a].y, y[u * M_ + a].x);
    }
}

// outer four-step twiddles w_M^{i2 * k1}, k1 = s + 16 j:  w^{i2 s} . (w^{16 i2})^j, the second factor shared by the
// 16 threads of a column through LDS (one sincospi per thread for each factor; arguments are exact dyadic fractions)
__device__ __forceinline__ float2 unit_pow(long num, int M /* pow2 */)
{
    const int m = (int)(num & (long)(M - 1));
    float sn, cs;
    sincospif(-2.0f * (float)m / (float)M, &sn, &cs);
    return make_float2(cs, sn);                         // exp(-j 2 pi m / M)
}
__device__ __forceinline__ void outer_twiddle_setup(float2* twc /* LDS [16][16] */, float2& t1, int col0, int M)
{
    const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
    const long i2 = col0 + c;
    twc[s * 16 + c] = unit_pow(16L * i2 * s, M);        // (w^{16 i2})^s, [power][column]
    t1 = unit_pow(i2 * s, M);
}

__device__ __forceinline__ void load_tw256(float2* tw256, const float2* __restrict__ tw256_g)
{
    tw256[threadIdx.x] = tw256_g[threadIdx.x];
}

// ---- kernel A: x = in . doppler_k . c  (zero padded to M)  ->  column FFT  ->  outer twiddle  ->  U[k1][i2] ---------
// PLAIN (table setup): x = in (already padded, length M)
template <bool PLAIN>
__global__ __launch_bounds__(256) void tsim_col_first_kernel(const float2* __restrict__ in, long in_stride,
                                                             const float2* __restrict__ dop, const float2* __restrict__ chirp,
                                                             float2* __restrict__ U, const float2* __restrict__ tw256_g,
                                                             int n, int n2, int M)
{
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    __shared__ float2 twc[256];
    const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int col0 = blockIdx.x * TS_CW;
    const size_t b = blockIdx.y, z = blockIdx.z;          // z = target within this launch (dop advances n per target)
    const float2* src = in + b * (size_t)in_stride;
    if (!PLAIN) dop += z * (size_t)n;
    float2 t1;
    load_tw256(tw256, tw256_g);
    outer_twiddle_setup(twc, t1, col0, M);
    float2 x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const long i = (long)(s + 16 * j) * n2 + col0 + c;
        if (PLAIN) {
            x[j] = src[i];
        } else if (i < n) {
            const float2 v = cmul(src[i], dop[i]);       // volk_32fc_x2_multiply_32fc (:345)
            x[j] = cmul(v, chirp[i]);
        } else {
            x[j] = make_float2(0.f, 0.f);
        }
    }
    __syncthreads();                                      // tw256 / twc visible
    fft256_cols(x, xch, tw256);
    float2* dst = U + (b * gridDim.z + z) * (size_t)M;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float2 w = cmul(t1, twc[r * 16 + c]);       // w_M^{i2 (s + 16 r)}
        dst[(size_t)(s + 16 * r) * n2 + col0 + c] = cmul(x[r], w);
    }
}

// ---- row kernel: row FFT (n2) -> . Bhat[k1][.] (or conj) -> inverse row FFT, in place -------------------------------
// FWD_ONLY (table setup): row FFT only.
template <bool FWD_ONLY>
__global__ __launch_bounds__(256) void tsim_rowconv_kernel(float2* __restrict__ X, const float2* __restrict__ bhat, int conj_b,
                                                           const float2* __restrict__ tw_f, const float2* __restrict__ tw_i,
                                                           int n2, int logn2, size_t rows, int tp)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int per_block = blockDim.x / tp;
    const int lt = threadIdx.x % tp, lb = threadIdx.x / tp;
    const size_t row = (size_t)blockIdx.x * per_block + lb;
    const bool live = row < rows;
    float2* buf0 = lds + (size_t)lb * 2 * n2;
    float2* buf1 = buf0 + n2;
    float2* g = X + row * (size_t)n2;
    const int k1 = (int)(row & (TS_N1 - 1));
    const float2* brow = bhat ? bhat + (size_t)k1 * n2 : nullptr;

    const float2* cur = nullptr;
    float2* nxt = buf0;
    for (int dir = 0; dir < (FWD_ONLY ? 1 : 2); dir++) {
        const float2* tw = dir ? tw_i : tw_f;
        const int sign = dir ? 1 : -1;
        int Ns = 1;
        bool first = true;
        while (Ns < n2) {
            const int Rx = ((logn2 & 1) && first) ? 2 : 4;
            const bool last = Ns * Rx == n2;
            const bool from_g = (dir == 0 && first);
            const bool to_g = last && (FWD_ONLY || dir == 1);
            if (live) {
                if (Rx == 2) stockham_pass<2>(from_g ? g : nullptr, 0, nullptr, cur, to_g ? nullptr : nxt, to_g ? g : nullptr, 0, tw, n2, Ns, sign, lt, tp);
                else stockham_pass<4>(from_g ? g : nullptr, 0, nullptr, cur, to_g ? nullptr : nxt, to_g ? g : nullptr, 0, tw, n2, Ns, sign, lt, tp);
            }
            __syncthreads();
            cur = nxt; nxt = (nxt == buf0) ? buf1 : buf0;
            Ns *= Rx; first = false;
        }
        if (!FWD_ONLY && dir == 0) {
            if (live) {
                float2* w = const_cast<float2*>(cur);
                for (int k2 = lt; k2 < n2; k2 += tp) {
                    float2 bv = brow[k2];
                    if (conj_b) bv.y = -bv.y;
                    w[k2] = cmul(w[k2], bv);
                }
            }
            __syncthreads();
        }
    }
}

// ---- kernel MID: for every target z of the launch: Z_z -> conj outer twiddle -> inverse column FFT = conv_z (natural
//      order); acc_l += conv_z . timeshift_{l,z} [. phase_z]  (zero beyond n);  then per antenna l: column FFT -> outer
//      twiddle -> G[l][k1][i2].  The sum over targets is taken here, in the middle of the two convolutions (both are
//      linear), so K targets cost K + R convolutions instead of K (1 + R).  RC antennas per launch (accumulators in
//      registers). -----------------------------------------------------------------------------------------------------
template <int RC>
__global__ __launch_bounds__(256) void tsim_col_mid_kernel(const float2* __restrict__ Z, float2* __restrict__ G,
                                                           const float2* __restrict__ ts /* [R][K][n] + (l0*K + k0)*n */, long ts_l_stride,
                                                           const float2* __restrict__ phase /* + k0, or null */,
                                                           const float2* __restrict__ tw256_g, int Kz, int R, int l0, int n, int n2, int M)
{
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    __shared__ float2 twc[256];
    const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int col0 = blockIdx.x * TS_CW;
    const size_t b = blockIdx.y;
    float2 t1;
    load_tw256(tw256, tw256_g);
    outer_twiddle_setup(twc, t1, col0, M);
    float2 acc[RC][16];
#pragma unroll
    for (int l = 0; l < RC; l++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[l][r] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int z = 0; z < Kz; z++) {
        const float2* src = Z + (b * Kz + z) * (size_t)M;
        float2 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = src[(size_t)(s + 16 * j) * n2 + col0 + c];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = cmul(x[j], conjf2(cmul(t1, twc[j * 16 + c])));
        ifft256_cols(x, xch, tw256);                      // x[r] = conv_z[(s + 16 r) n2 + i2]
        if (phase) {
            const float2 ph = phase[z];                   // :358-362, folded in before the (linear) second transform
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = cmul(x[r], ph);
        }
#pragma unroll
        for (int l = 0; l < RC; l++) {
            const float2* tsl = ts + (size_t)l * ts_l_stride + (size_t)z * n;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long i = (long)(s + 16 * r) * n2 + col0 + c;
                if (i < n) acc[l][r] = cadd(acc[l][r], cmul(x[r], tsl[i]));      // :352
            }
        }
    }
#pragma unroll
    for (int l = 0; l < RC; l++) {
        fft256_cols(acc[l], xch, tw256);
        float2* dst = G + (b * R + l0 + l) * (size_t)M;
#pragma unroll
        for (int r = 0; r < 16; r++)
            dst[(size_t)(s + 16 * r) * n2 + col0 + c] = cmul(acc[l][r], cmul(t1, twc[r * 16 + c]));
    }
}

// single target per launch (the reference's own behaviour, and K = 1): no accumulators — one inverse column FFT, then the
// antennas one after the other out of the same registers
__global__ __launch_bounds__(256) void tsim_col_mid1_kernel(const float2* __restrict__ Z, float2* __restrict__ G,
                                                            const float2* __restrict__ ts /* [R][K][n] + k*n */, long ts_l_stride,
                                                            const float2* __restrict__ phase /* + k, or null */,
                                                            const float2* __restrict__ tw256_g, int R, int n, int n2, int M)
{
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    __shared__ float2 twc[256];
    const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int col0 = blockIdx.x * TS_CW;
    const size_t b = blockIdx.y;
    const float2* src = Z + b * (size_t)M;
    float2 t1;
    load_tw256(tw256, tw256_g);
    outer_twiddle_setup(twc, t1, col0, M);
    float2 x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = src[(size_t)(s + 16 * j) * n2 + col0 + c];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = cmul(x[j], conjf2(cmul(t1, twc[j * 16 + c])));
    ifft256_cols(x, xch, tw256);                          // x[r] = conv[(s + 16 r) n2 + i2]
    if (phase) {
        const float2 ph = phase[0];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = cmul(x[r], ph);
    }
    for (int l = 0; l < R; l++) {
        const float2* tsl = ts + (size_t)l * ts_l_stride;
        float2 y[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long i = (long)(s + 16 * r) * n2 + col0 + c;
            y[r] = (i < n) ? cmul(x[r], tsl[i]) : make_float2(0.f, 0.f);     // :352
        }
        fft256_cols(y, xch, tw256);
        float2* dst = G + (b * R + l) * (size_t)M;
#pragma unroll
        for (int r = 0; r < 16; r++)
            dst[(size_t)(s + 16 * r) * n2 + col0 + c] = cmul(y[r], cmul(t1, twc[r * 16 + c]));
    }
}

// ---- kernel OUT: Y_l -> conj outer twiddle -> inverse column FFT -> . conj(c)[m] -> out_l[m]
//      (= or +=), plus the self-coupling term sc . in[m] (:372-378) when asked ---------------------------------------
__global__ __launch_bounds__(256) void tsim_col_out_kernel(const float2* __restrict__ Y, float2* __restrict__ out,
                                                           long out_burst_stride, long out_rx_stride,
                                                           const float2* __restrict__ chirp,
                                                           const float2* __restrict__ in, long in_stride, float self_coupling,
                                                           int add_self, int accumulate,
                                                           const float2* __restrict__ tw256_g, int R, int n, int n2, int M)
{
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    __shared__ float2 twc[256];
    const int c = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int col0 = blockIdx.x * TS_CW;
    const size_t bl = blockIdx.y;                         // burst * R + l
    const size_t b = bl / R, l = bl % R;
    const float2* src = Y + bl * (size_t)M;
    float2 t1;
    load_tw256(tw256, tw256_g);
    outer_twiddle_setup(twc, t1, col0, M);
    float2 x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = src[(size_t)(s + 16 * j) * n2 + col0 + c];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = cmul(x[j], conjf2(cmul(t1, twc[j * 16 + c])));
    ifft256_cols(x, xch, tw256);
    float2* o = out + b * (size_t)out_burst_stride + l * (size_t)out_rx_stride;
    const float2* inb = in + b * (size_t)in_stride;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const long m = (long)(s + 16 * r) * n2 + col0 + c;
        if (m < n) {
            float2 v = cmul(x[r], conjf2(chirp[m]));
            if (accumulate) v = cadd(o[m], v);
            if (add_self) {                               // out += (gr_complex)pow(10, db/20) * in  (:376)
                const float2 xi = inb[m];
                v = cadd(v, make_float2(self_coupling * xi.x - 0.0f * xi.y, self_coupling * xi.y + 0.0f * xi.x));
            }
            o[m] = v;
        }
    }
}

// no targets at all: out = [out +] sc . in
__global__ void tsim_passthrough_kernel(float2* __restrict__ out, long out_burst_stride, long out_rx_stride,
                                        const float2* __restrict__ in, long in_stride, float self_coupling, int add_self,
                                        int accumulate, int R, int n)
{
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const size_t b = blockIdx.y / R, l = blockIdx.y % R;
    float2* o = out + b * (size_t)out_burst_stride + l * (size_t)out_rx_stride;
    float2 v = accumulate ? o[m] : make_float2(0.f, 0.f);
    if (add_self) { const float2 xi = in[b * (size_t)in_stride + m]; v = cadd(v, make_float2(self_coupling * xi.x, self_coupling * xi.y)); }
    o[m] = v;
}

// =====================================================================================================================
// Direct four-step path (round 5).  Every burst the flowgraphs produce has n = n_symbols x (fft_len + cp) samples with
// fft_len + cp = 5 x 2^k (lib/target_simulator_impl.cc:243-245 takes n from the packet_len tag; 23,040 = 45 x 512 at config B),
// and the chirp-z route above pads such a burst to M = 65,536 = 2.84 n and walks it five times through HBM.  Here the length-n
// DFT itself is split  n = n1 x n2  with n2 the largest power of two dividing n (<= 4096) and n1 = n / n2 (any integer <= 512):
//      i = i1 n2 + i2,  k = k1 + n1 k2:   X[k1 + n1 k2] = sum_i2 w_n2^{i2 k2} [ w_n^{i2 k1} sum_i1 x[i1 n2 + i2] w_n1^{i1 k1} ]
//   * td_col_fwd_kernel   (in . doppler) viewed as [n1][n2]: n1-point DFTs down 16 columns per workgroup (mixed-radix Stockham in LDS:
//                         radix 4 / 2 / 3 / 5 butterflies, any other prime factor p by its p-term sums), . w_n^{i2 k1} -> U[k1][i2]
//   * td_rows*_kernel     per row k1: n2-point FFT -> . timeshift_l[k1 + n1 k2] (summed over the targets of the launch) -> inverse
//                         n2-point FFT, for every RX antenna out of the one forward transform -> G[l][k1][i2]
//   * td_col_inv_kernel   . conj w_n^{i2 k1}, inverse n1-point DFTs down the columns -> out_l[i1 n2 + i2] (natural order)
// "doppler -> FFT once per target -> x timeshift per antenna -> IFFT" is the algebra of the route above; the spectrum stays in
// [k1][k2] order between the passes (the timeshift table is stored in that order once per burst length), so there is no transpose and
// no reordering pass.  HBM traffic per burst: (3 + 3R) n x 8 B against (1 + R) n x 8 B algorithmic = 3x at R = 4 (chirp-z: 13.6x).
// Lengths that do not split this way (n2 < 16 or n1 > 512) keep the chirp-z route; JRC_TSIM_BLUESTEIN=1 forces it for all.
// =====================================================================================================================
// ---- column pass, forward:  x = in . doppler_z  as [n1][n2]  ->  n1-point DFT per column  ->  . w_n^{i2 k1}  ->  U[b][z][k1][i2] ------------
template <int CW>
__global__ __launch_bounds__(256) void td_col_fwd_kernel(td_srcs srcs, long in_stride,
                                                         float2* __restrict__ U, const float2* __restrict__ w1_g,
                                                         const float2* __restrict__ two, td_plan pl, int n)
{
    extern __shared__ __attribute__((aligned(16))) float2 td_lds[];
    const int n1 = pl.n1, n2 = pl.n2;
    float2* buf0 = td_lds;
    float2* buf1 = buf0 + (size_t)n1 * CW;
    float2* w1 = buf1 + (size_t)n1 * CW;
    const int c = threadIdx.x & (CW - 1), w = threadIdx.x / CW, nw = 256 / CW;
    const int col = blockIdx.x * CW + c;
    const size_t b = blockIdx.y, z = blockIdx.z;
    const float2* __restrict__ src = srcs.in[z] + b * (size_t)in_stride;
    const float2* __restrict__ dz = srcs.dop[z];
    for (int i = threadIdx.x; i < n1; i += 256) w1[i] = w1_g[i];
    // global accesses two columns (16 bytes) per lane; the passes below work a column per lane on the same tile
    const int c2 = (threadIdx.x % (CW / 2)) * 2, wp = threadIdx.x / (CW / 2), nwp = 256 / (CW / 2);
    const int colp = blockIdx.x * CW + c2;
    for (int i1 = wp; i1 < n1; i1 += nwp) {
        const size_t i = (size_t)i1 * n2 + colp;
        const float4 a = *reinterpret_cast<const float4*>(src + i), dd = *reinterpret_cast<const float4*>(dz + i);
        const float2 p0 = cmul(make_float2(a.x, a.y), make_float2(dd.x, dd.y));      // volk_32fc_x2_multiply_32fc (:345)
        const float2 p1 = cmul(make_float2(a.z, a.w), make_float2(dd.z, dd.w));
        *reinterpret_cast<float4*>(buf0 + i1 * CW + c2) = make_float4(p0.x, p0.y, p1.x, p1.y);
    }
    __syncthreads();
    const float2* cur = td_col_transform<CW>(buf0, buf1, w1, pl, c, w, nw);
    float2* dst = U + (b * gridDim.z + z) * (size_t)n;
    for (int k1 = wp; k1 < n1; k1 += nwp) {
        const size_t o = (size_t)k1 * n2 + colp;
        const float4 v = *reinterpret_cast<const float4*>(cur + k1 * CW + c2), t = *reinterpret_cast<const float4*>(two + o);
        const float2 p0 = cmul(make_float2(v.x, v.y), make_float2(t.x, t.y)), p1 = cmul(make_float2(v.z, v.w), make_float2(t.z, t.w));
        *reinterpret_cast<float4*>(dst + o) = make_float4(p0.x, p0.y, p1.x, p1.y);
    }
    (void)col;
}

// ---- column pass, inverse:  G[bl][k1][i2] . conj w_n^{i2 k1}  ->  inverse n1-point DFT per column (conj, forward, conj)  ->  out_l[i1 n2 + i2]
//      (= or +=), plus the self-coupling term sc . in (:372-378) when asked ---------------------------------------------------------------------
template <int CW>
__global__ __launch_bounds__(256) void td_col_inv_kernel(const float2* __restrict__ G, float2* __restrict__ out, long out_burst_stride,
                                                         long out_rx_stride, td_self self, long in_stride,
                                                         float self_coupling, int accumulate,
                                                         const float2* __restrict__ w1_g, const float2* __restrict__ two, td_plan pl, int R, int n)
{
    extern __shared__ __attribute__((aligned(16))) float2 td_lds[];
    const int n1 = pl.n1, n2 = pl.n2;
    float2* buf0 = td_lds;
    float2* buf1 = buf0 + (size_t)n1 * CW;
    float2* w1 = buf1 + (size_t)n1 * CW;
    const int c = threadIdx.x & (CW - 1), w = threadIdx.x / CW, nw = 256 / CW;
    const int col = blockIdx.x * CW + c;
    const size_t bl = blockIdx.y, b = bl / R, l = bl % R;
    const float2* src = G + bl * (size_t)n;
    for (int i = threadIdx.x; i < n1; i += 256) w1[i] = w1_g[i];
    const int c2 = (threadIdx.x % (CW / 2)) * 2, wp = threadIdx.x / (CW / 2), nwp = 256 / (CW / 2);      // two columns (16 bytes) per lane on the global side
    const int colp = blockIdx.x * CW + c2;
    for (int k1 = wp; k1 < n1; k1 += nwp) {
        const size_t o = (size_t)k1 * n2 + colp;
        const float4 g = *reinterpret_cast<const float4*>(src + o), t = *reinterpret_cast<const float4*>(two + o);
        const float2 v0 = cmul(make_float2(g.x, g.y), make_float2(t.x, -t.y)), v1 = cmul(make_float2(g.z, g.w), make_float2(t.z, -t.w));
        *reinterpret_cast<float4*>(buf0 + k1 * CW + c2) = make_float4(v0.x, -v0.y, v1.x, -v1.y);          // conjugated: the inverse runs on the forward passes
    }
    __syncthreads();
    const float2* cur = td_col_transform<CW>(buf0, buf1, w1, pl, c, w, nw);
    float2* o = out + b * (size_t)out_burst_stride + l * (size_t)out_rx_stride;
    for (int i1 = wp; i1 < n1; i1 += nwp) {
        const size_t m = (size_t)i1 * n2 + colp;
        const float4 r = *reinterpret_cast<const float4*>(cur + i1 * CW + c2);
        float2 v0 = make_float2(r.x, -r.y), v1 = make_float2(r.z, -r.w);
        if (accumulate) { const float4 e = *reinterpret_cast<const float4*>(o + m); v0 = cadd(make_float2(e.x, e.y), v0); v1 = cadd(make_float2(e.z, e.w), v1); }
        for (int q = 0; q < self.n; q++) {                                   // out += (gr_complex)pow(10, db/20) * in  (:376), per simulator
            const float4 xi = *reinterpret_cast<const float4*>(self.in[q] + b * (size_t)in_stride + m);
            v0 = cadd(v0, make_float2(self_coupling * xi.x - 0.0f * xi.y, self_coupling * xi.y + 0.0f * xi.x));
            v1 = cadd(v1, make_float2(self_coupling * xi.z - 0.0f * xi.w, self_coupling * xi.w + 0.0f * xi.z));
        }
        *reinterpret_cast<float4*>(o + m) = make_float4(v0.x, v0.y, v1.x, v1.y);
    }
    (void)col;
}

// ---- row pass, n2 == 256: 16 rows per workgroup, everything in registers.  RC antennas of the launch out of one forward transform per target;
//      tsp = timeshift as [R][K][n1][n2] (+ (l0 K + k0) n), the sum over the Kz targets of the launch is taken on the spectrum ----------------
// RC == 0: one target per launch (the reference's own behaviour) — the forward spectrum stays in 16 registers and the R antennas are taken one
// after the other out of it, no accumulators.  Two workgroups per CU (<= 256 VGPRs, no spills) measured faster at config B than three with
// spills or than one antenna per workgroup with the forward transform repeated (0.221 / 0.236 / 0.236 ms per 256 bursts, round 5).
template <int RC>
__global__ __launch_bounds__(256, RC == 1 ? 3 : 2) void td_rows256_kernel(const float2* __restrict__ U, float2* __restrict__ G, td_ts ts,
                                                            long ts_l_stride, const float2* __restrict__ tw256_g,
                                                            int Kz, int R, int l0, int n1, long rows)
{
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    const int s = threadIdx.x & 15, rl = threadIdx.x >> 4;
    long row = (long)blockIdx.x * 16 + rl;                                    // over bursts x n1
    const bool live = row < rows;
    if (!live) row = rows - 1;
    const size_t b = (size_t)(row / n1), k1 = (size_t)(row % n1);
    tw256[threadIdx.x] = tw256_g[threadIdx.x];
    __syncthreads();
    if (RC == 0) {
        const float2* g = U + (b * n1 + k1) * 256;
        float2 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = g[s + 16 * j];
        fft256_rows(x, xch, tw256);
        if (ts.use_phase) {
            const float2 ph = ts.phase[0];                                    // :358-362, on the spectrum (the inverse transform is linear)
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = cmul(x[r], ph);
        }
#pragma unroll 1
        for (int l = 0; l < R; l++) {
            const float2* __restrict__ tr = ts.tsp[0] + (size_t)l * ts_l_stride + k1 * 256;
            float2 y[16];
#pragma unroll
            for (int r = 0; r < 16; r++) y[r] = cmul(x[r], tr[s + 16 * r]);   // :352
            swap_reim(y); fft256_rows(y, xch, tw256); swap_reim(y);
            float2* d = G + ((b * R + l) * n1 + k1) * 256;
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; r++) d[s + 16 * r] = y[r];
            }
        }
        return;
    }
    constexpr int RCA = RC > 0 ? RC : 1;
    float2 acc[RCA][16];
    for (int z = 0; z < Kz; z++) {
        const float2* g = U + ((b * Kz + z) * n1 + k1) * 256;
        float2 x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = g[s + 16 * j];
        fft256_rows(x, xch, tw256);
        if (ts.use_phase) {
            const float2 ph = ts.phase[z];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = cmul(x[r], ph);
        }
#pragma unroll
        for (int l = 0; l < RCA; l++) {
            const float2* __restrict__ tr = ts.tsp[z] + (size_t)(l0 + l) * ts_l_stride + k1 * 256;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float2 p = cmul(x[r], tr[s + 16 * r]);
                acc[l][r] = z ? cadd(acc[l][r], p) : p;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < RCA; l++) {
        swap_reim(acc[l]); fft256_rows(acc[l], xch, tw256); swap_reim(acc[l]);
        float2* d = G + ((b * R + l0 + l) * n1 + k1) * 256;
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; r++) d[s + 16 * r] = acc[l][r];
        }
    }
}

// ---- row pass, n2 = m x 256 (m = 2, 4, 8, 16): the row transform of tsim_rowconv_m_kernel (m-point step in registers, 256-point step through
//      LDS, spectrum in [k_a][k_b] order — the timeshift table is stored in that order), RC antennas out of one forward transform per target ----
template <int M_>
__device__ __forceinline__ void td_row_m_fwd(const float2* __restrict__ g, float2 (&x)[16], float2* xch, const float2* tw256,
                                             const float2* __restrict__ twn2_g, int rowl, int t, int s, int plane)
{
    constexpr int TPR = 16 * M_;
    constexpr int UU = 16 / M_;
    float2 y[16];                                                             // [u][i_a]
#pragma unroll
    for (int u = 0; u < UU; u++)
#pragma unroll
        for (int a = 0; a < M_; a++) y[u * M_ + a] = g[a * 256 + t + TPR * u];
#pragma unroll
    for (int u = 0; u < UU; u++) {
        fft_fwd_small<M_>(*reinterpret_cast<float2(*)[M_]>(&y[u * M_]));
        const int ib = t + TPR * u;
#pragma unroll
        for (int a = 1; a < M_; a++) y[u * M_ + a] = cmul(y[u * M_ + a], twn2_g[ib * a]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UU; u++)
#pragma unroll
        for (int a = 0; a < M_; a++) xch[(rowl * M_ + a) * TS_XPAD + t + TPR * u] = y[u * M_ + a];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = xch[plane * TS_XPAD + s + 16 * j];
    fft256_rows(x, xch, tw256);                                               // x[r] = spectrum element k_b = s + 16 r of plane k_a
}
// v[r] = spectrum element k_b = s + 16 r of plane k_a  ->  the row in time order, stored to d (natural order)
template <int M_>
__device__ __forceinline__ void td_row_m_inv(float2 (&v)[16], float2* __restrict__ d, bool live, float2* xch, const float2* tw256,
                                             const float2* __restrict__ twn2_g, int rowl, int t, int s, int plane)
{
    constexpr int TPR = 16 * M_;
    constexpr int UU = 16 / M_;
    swap_reim(v); fft256_rows(v, xch, tw256); swap_reim(v);                   // v[r] = element i_b = s + 16 r of plane k_a
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) xch[plane * TS_XPAD + s + 16 * r] = v[r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UU; u++) {
        const int ib = t + TPR * u;
        float2 y[M_];
#pragma unroll
        for (int a = 0; a < M_; a++) {
            float2 q = xch[(rowl * M_ + a) * TS_XPAD + ib];
            if (a) q = cmul(q, conjf2(twn2_g[ib * a]));
            y[a] = make_float2(q.y, q.x);                                     // swapped: inverse m-point FFT via the forward one
        }
        fft_fwd_small<M_>(y);
        if (live) {
#pragma unroll
            for (int a = 0; a < M_; a++) d[a * 256 + ib] = make_float2(y[a].y, y[a].x);
        }
    }
}

template <int M_, int RC>
__global__ __launch_bounds__(256, RC == 1 ? 3 : 2) void td_rows_m_kernel(const float2* __restrict__ U, float2* __restrict__ G, td_ts ts,
                                                           long ts_l_stride, const float2* __restrict__ tw256_g,
                                                           const float2* __restrict__ twn2_g, int Kz, int R, int l0, int n1, long rows)
{
    constexpr int TPR = 16 * M_;
    constexpr int N2 = 256 * M_;
    __shared__ float2 xch[16 * TS_XPAD];
    __shared__ float2 tw256[256];
    const int rowl = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int s = threadIdx.x & 15, plane = threadIdx.x >> 4;                 // plane = rowl * M_ + k_a
    long row = (long)blockIdx.x * (256 / TPR) + rowl;
    const bool live = row < rows;
    if (!live) row = rows - 1;
    const size_t b = (size_t)(row / n1), k1 = (size_t)(row % n1);
    const int ka = plane % M_;
    tw256[threadIdx.x] = tw256_g[threadIdx.x];
    if (RC == 0) {                                                            // one target: the antennas one after the other out of x
        float2 x[16];
        td_row_m_fwd<M_>(U + (b * n1 + k1) * N2, x, xch, tw256, twn2_g, rowl, t, s, plane);
        if (ts.use_phase) {
            const float2 ph = ts.phase[0];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = cmul(x[r], ph);
        }
#pragma unroll 1
        for (int l = 0; l < R; l++) {
            const float2* __restrict__ tr = ts.tsp[0] + (size_t)l * ts_l_stride + k1 * N2 + ka * 256;
            float2 y[16];
#pragma unroll
            for (int r = 0; r < 16; r++) y[r] = cmul(x[r], tr[s + 16 * r]);
            td_row_m_inv<M_>(y, G + ((b * R + l) * n1 + k1) * N2, live, xch, tw256, twn2_g, rowl, t, s, plane);
        }
        return;
    }
    constexpr int RCA = (RC == 1 || RC == 2) ? RC : 1;
    float2 acc[RCA][16];
    for (int z = 0; z < Kz; z++) {
        float2 x[16];
        td_row_m_fwd<M_>(U + ((b * Kz + z) * n1 + k1) * N2, x, xch, tw256, twn2_g, rowl, t, s, plane);
        if (ts.use_phase) {
            const float2 ph = ts.phase[z];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = cmul(x[r], ph);
        }
#pragma unroll
        for (int l = 0; l < RCA; l++) {
            const float2* __restrict__ tr = ts.tsp[z] + (size_t)(l0 + l) * ts_l_stride + k1 * N2 + ka * 256;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float2 p = cmul(x[r], tr[s + 16 * r]);
                acc[l][r] = z ? cadd(acc[l][r], p) : p;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < RCA; l++)
        td_row_m_inv<M_>(acc[l], G + ((b * R + l0 + l) * n1 + k1) * N2, live, xch, tw256, twn2_g, rowl, t, s, plane);
}

// ---- row pass, any power of two 16 <= n2 <= 128 (the 64-carrier flowgraphs: n2 = 16 x 2^v2(n_symbols)): Stockham passes in LDS, tp threads per
//      row, one antenna per blockIdx.y; the spectrum sum over the targets is kept in a third LDS row -----------------------------------------
__global__ __launch_bounds__(256) void td_rows_small_kernel(const float2* __restrict__ U, float2* __restrict__ G, td_ts ts,
                                                            long ts_l_stride, const float2* __restrict__ tw_f,
                                                            const float2* __restrict__ tw_i, int Kz, int R, int n1, int n2, int logn2, long rows, int tp)
{
    extern __shared__ __attribute__((aligned(16))) float2 td_lds[];
    const int per_block = blockDim.x / tp;
    const int lt = threadIdx.x % tp, lb = threadIdx.x / tp;
    long row = (long)blockIdx.x * per_block + lb;
    const bool live = row < rows;
    if (!live) row = rows - 1;
    const size_t b = (size_t)(row / n1), k1 = (size_t)(row % n1);
    const int l = blockIdx.y;
    float2* buf0 = td_lds + (size_t)lb * 3 * n2;
    float2* buf1 = buf0 + n2;
    float2* accb = buf1 + n2;
    for (int z = 0; z < Kz; z++) {
        const float2* g = U + ((b * Kz + z) * n1 + k1) * n2;
        const float2* cur = nullptr;
        float2* nxt = buf0;
        int Ns = 1;
        bool first = true;
        while (Ns < n2) {
            const int Rx = ((logn2 & 1) && first) ? 2 : 4;
            if (Rx == 2) stockham_pass<2>(first ? g : nullptr, 0, nullptr, cur, nxt, nullptr, 0, tw_f, n2, Ns, -1, lt, tp);
            else stockham_pass<4>(first ? g : nullptr, 0, nullptr, cur, nxt, nullptr, 0, tw_f, n2, Ns, -1, lt, tp);
            __syncthreads();
            cur = nxt; nxt = (nxt == buf0) ? buf1 : buf0;
            Ns *= Rx; first = false;
        }
        const float2* __restrict__ tr = ts.tsp[z] + (size_t)l * ts_l_stride + k1 * n2;
        for (int k2 = lt; k2 < n2; k2 += tp) {
            float2 v = cur[k2];
            if (ts.use_phase) v = cmul(v, ts.phase[z]);
            v = cmul(v, tr[k2]);
            accb[k2] = z ? cadd(accb[k2], v) : v;
        }
        __syncthreads();
    }
    {
        const float2* cur = accb;
        float2* nxt = buf0;
        float2* d = G + ((b * R + l) * n1 + k1) * n2;
        int Ns = 1;
        bool first = true;
        while (Ns < n2) {
            const int Rx = ((logn2 & 1) && first) ? 2 : 4;
            const bool last = Ns * Rx == n2;
            float2* dg = (last && live) ? d : nullptr;
            float2* dl = last ? (live ? nullptr : nxt) : nxt;
            if (Rx == 2) stockham_pass<2>(nullptr, 0, nullptr, cur, dl, dg, 0, tw_i, n2, Ns, +1, lt, tp);
            else stockham_pass<4>(nullptr, 0, nullptr, cur, dl, dg, 0, tw_i, n2, Ns, +1, lt, tp);
            __syncthreads();
            cur = nxt; nxt = (nxt == buf0) ? buf1 : buf0;
            Ns *= Rx; first = false;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------
static const double TS_FOUR_PI_CUBED_SQRT = 44.54662397465366;   // :33
static const float TS_C_LIGHT = 3e8f;                           // target_simulator_impl.h c_light

static void tsim_setup_targets(jrc_tsim* h)                     // :132-198
{
    const int K = h->K, R = h->R;
    h->doppler.resize(K); h->scale_ampl.resize(K); h->timeshift.resize((size_t)K * R);
    for (int k = 0; k < K; k++) h->doppler[k] = 2 * h->velocity[k] * h->center_freq / TS_C_LIGHT;
    for (int l = 0; l < R; l++)
        for (int k = 0; k < K; k++)
            h->timeshift[(size_t)l * K + k] = (2.0 * h->range[k] - h->position_rx[l] * std::sin(h->azimuth[k] * M_PI / 180.0)) / TS_C_LIGHT;
    for (int k = 0; k < K; k++)
        h->scale_ampl[k] = TS_C_LIGHT * std::sqrt(h->rcs[k]) / TS_FOUR_PI_CUBED_SQRT / (h->range[k] * h->range[k]) / h->center_freq;
}

static int tsim_rows_launch(jrc_tsim* h, bool fwd_only, float2* X, int conj_b, size_t rows, hipStream_t stream)
{
    jrc_ctx* ctx = h->ctx;
    const int n2 = h->n2;
    const float2 *twf = nullptr, *twi = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, n2, -1, &twf));
    JRC_TRY(jrc_get_twiddles(ctx, n2, +1, &twi));
    if (n2 >= 512 && n2 <= 4096) {
        const float2* tw256 = nullptr;
        JRC_TRY(jrc_get_twiddles(ctx, TS_N1, -1, &tw256));
        const int m = n2 / 256;
        const dim3 grid((unsigned)(rows / (16 / m)));
        const float2* bh = fwd_only ? nullptr : (const float2*)h->d_bhat;
#define TS_LAUNCH_M(MM)                                                                                                   \
        do {                                                                                                              \
            if (fwd_only) hipLaunchKernelGGL((tsim_rowconv_m_kernel<MM, true>), grid, dim3(256), 0, stream, X, bh, conj_b, tw256, twf); \
            else hipLaunchKernelGGL((tsim_rowconv_m_kernel<MM, false>), grid, dim3(256), 0, stream, X, bh, conj_b, tw256, twf);        \
        } while (0)
        if (m == 2) TS_LAUNCH_M(2); else if (m == 4) TS_LAUNCH_M(4); else if (m == 8) TS_LAUNCH_M(8); else TS_LAUNCH_M(16);
#undef TS_LAUNCH_M
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    if (!fwd_only && n2 == 256) {
        const float2* tw256 = nullptr;
        JRC_TRY(jrc_get_twiddles(ctx, TS_N1, -1, &tw256));
        hipLaunchKernelGGL(tsim_rowconv256_kernel, dim3((unsigned)(rows / 16)), dim3(256), 0, stream, X, (const float2*)h->d_bhat,
                           conj_b, tw256, rows);
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    int tp = n2 / 4; if (tp > 256) tp = 256;
    const int per_block = 256 / tp;
    const size_t blocks = (rows + per_block - 1) / per_block;
    const size_t lds_bytes = sizeof(float2) * 2 * (size_t)n2 * per_block;
    if (fwd_only) {
        JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)tsim_rowconv_kernel<true>, lds_bytes));
        hipLaunchKernelGGL(tsim_rowconv_kernel<true>, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, X,
                           (const float2*)nullptr, 0, twf, twi, n2, jrc_ilog2(n2), rows, tp);
    } else {
        JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)tsim_rowconv_kernel<false>, lds_bytes));
        hipLaunchKernelGGL(tsim_rowconv_kernel<false>, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, X,
                           (const float2*)h->d_bhat, conj_b, twf, twi, n2, jrc_ilog2(n2), rows, tp);
    }
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

static void tsim_free_tables(jrc_tsim* h)
{
    (void)hipFree(h->d_dop); (void)hipFree(h->d_ts); (void)hipFree(h->d_chirp); (void)hipFree(h->d_bhat);
    (void)hipFree(h->d_w1); (void)hipFree(h->d_two);
    h->d_dop = h->d_ts = h->d_chirp = h->d_bhat = h->d_w1 = h->d_two = nullptr;
    h->n = 0; h->direct = false;
}

// n = n1 x n2 with n2 the largest power of two dividing n (<= 4096) and n1 <= TD_MAX_N1; the radices of the n1-point column transform
static size_t td_col_lds_bytes(int n1, int cw) { return sizeof(float2) * ((size_t)2 * n1 * cw + (size_t)n1); }
// `pairs` = (simulator, target) pairs one launch of this simulator carries (K with sum_targets, else 1): the launches hold at most TD_MAXV
// of them, a simulator with more keeps the chirp-z route, which takes any K.  `max_lds` = the device's LDS per workgroup: the column passes
// stage 2 x n1 x 16 cells (+ n1 twiddles), a length whose tile does not fit keeps the chirp-z route too.
static bool tsim_plan_direct(int n, int pairs, size_t max_lds, td_plan* pl)
{
    if (const char* e = getenv("JRC_TSIM_BLUESTEIN")) { if (atoi(e) != 0) return false; }
    if (pairs > TD_MAXV) return false;
    int n2 = 1;
    while (n2 < 4096 && n % (n2 * 2) == 0) n2 *= 2;
    if (n2 < TD_CW) return false;
    const int n1 = n / n2;
    if (n1 > TD_MAX_N1) return false;
    if (td_col_lds_bytes(n1, TD_CW) > max_lds) return false;
    pl->n1 = n1; pl->n2 = n2; pl->nrad = 0;
    int m = n1;
    while (m % 4 == 0) { pl->rad[pl->nrad++] = 4; m /= 4; }
    if (m % 2 == 0) { pl->rad[pl->nrad++] = 2; m /= 2; }
    for (int p = 3; m > 1; p += 2)
        while (m % p == 0) { pl->rad[pl->nrad++] = p; m /= p; }
    return true;
}
static td_plan tsim_plan_of(const jrc_tsim* h)
{
    td_plan pl;
    pl.n1 = h->n1; pl.n2 = h->n2; pl.nrad = h->nrad;
    for (int i = 0; i < 24; i++) pl.rad[i] = h->rad[i];
    return pl;
}
// columns per workgroup of the column passes: 32 (256-byte row segments) while the tile fits 64 KB of LDS, else 16; JRC_TSIM_CW overrides —
// 32 only where its tile fits the 64 KB the <32> kernels may use without a dynamic-LDS opt-in (they never get one)
static int td_pick_cw(int n1, int n2)
{
    const bool fits32 = n2 >= 32 && td_col_lds_bytes(n1, 32) <= 64 * 1024;
    int cw = fits32 ? 32 : 16;
    if (const char* e = getenv("JRC_TSIM_CW")) { const int v = atoi(e); if (v == 16 || (v == 32 && fits32)) cw = v; }
    return cw;
}

// channel filters and chirp tables for bursts of n samples (:249-300 + the Bluestein tables)
static int tsim_prepare(jrc_tsim* h, int n, hipStream_t stream)
{
    jrc_ctx* ctx = h->ctx;
    if (h->n == n) return JRC_OK;
    if (n > (1 << 20))
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "target_simulator: burst of %d samples exceeds 2^20", n);
    JRC_HIP(ctx, hipStreamSynchronize(stream));
    tsim_free_tables(h);
    const int K = h->K, R = h->R;
    td_plan pl;
    const bool direct = tsim_plan_direct(n, (h->sum_targets && K > 1) ? K : 1, ctx->max_lds_per_block, &pl);
    long M = 32768;
    while (M < 2L * n - 1) M <<= 1;
    if (direct) M = n;                                 // the work buffers hold bursts of n samples, nothing is padded
    const int n2 = direct ? pl.n2 : (int)(M / TS_N1);
    // work buffers
    const size_t per = sizeof(float2) * (size_t)h->max_bursts * M;
    const size_t need_u = per * (size_t)(h->sum_targets && K > 1 ? K : 1), need_g = per * (size_t)(R > 0 ? R : 1);
    if (need_u > h->u_cap) { (void)hipFree(h->d_u); h->d_u = nullptr; h->u_cap = 0; JRC_HIP(ctx, hipMalloc((void**)&h->d_u, need_u)); h->u_cap = need_u; }
    if (need_g > h->g_cap) { (void)hipFree(h->d_g); h->d_g = nullptr; h->g_cap = 0; JRC_HIP(ctx, hipMalloc((void**)&h->d_g, need_g)); h->g_cap = need_g; }

    std::vector<float> freq((size_t)n);
    for (int i = 0; i < n; i++) {                      // :262-268, float arithmetic
        if (i < n / 2) freq[i] = i * (float)h->samp_rate / (float)n;
        else freq[i] = i * (float)h->samp_rate / (float)n - (float)h->samp_rate;
    }
    std::vector<std::complex<float>> dop((size_t)K * n), ts((size_t)R * K * n);
    for (int k = 0; k < K; k++) {
        std::complex<float> phase_doppler = 0;         // :281
        for (int i = 0; i < n; i++) {                  // :282-287
            dop[(size_t)k * n + i] = std::exp(phase_doppler) * h->scale_ampl[k];
            phase_doppler = std::complex<float>(0, std::fmod(std::imag(phase_doppler) + 2 * M_PI * h->doppler[k] / (float)h->samp_rate, 2 * M_PI));
        }
        for (int l = 0; l < R; l++) {                  // :291-305
            std::complex<float>* t = &ts[((size_t)l * K + k) * n];
            for (int i = 0; i < n; i++) {
                std::complex<float> phase_time(0, std::fmod(2 * M_PI * (h->timeshift[(size_t)l * K + k]) * (freq[i] + h->center_freq), 2 * M_PI));
                t[i] = std::exp(-phase_time) / (float)n;
            }
        }
    }
    if (direct) {
        // timeshift in the order the row pass leaves the spectrum in: row k1, then position p <-> k2 (natural for n2 <= 256, [k_a][k_b] with
        // k2 = k_a + m k_b for n2 = m x 256), k = k1 + n1 k2;  w_n1 and the outer twiddles w_n^{i2 k1} in double
        const int n1 = pl.n1, m = n2 > 256 ? n2 / 256 : 1;
        std::vector<std::complex<float>> tsp(ts.size());
        for (size_t lk = 0; lk < (size_t)R * K; lk++)
            for (int k1 = 0; k1 < n1; k1++)
                for (int p = 0; p < n2; p++) {
                    const int k2 = m > 1 ? (p / 256) + m * (p % 256) : p;
                    tsp[lk * n + (size_t)k1 * n2 + p] = ts[lk * n + (size_t)k1 + (size_t)n1 * k2];
                }
        std::vector<float2> w1((size_t)n1), two((size_t)n);
        for (int q = 0; q < n1; q++) {
            const double a = -2.0 * M_PI * (double)q / (double)n1;
            w1[q] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        for (int k1 = 0; k1 < n1; k1++)
            for (int i2 = 0; i2 < n2; i2++) {
                const double a = -2.0 * M_PI * (double)(((long)i2 * k1) % n) / (double)n;
                two[(size_t)k1 * n2 + i2] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        if (K > 0) {
            JRC_HIP(ctx, hipMalloc((void**)&h->d_dop, sizeof(float2) * (size_t)K * n));
            JRC_HIP(ctx, hipMalloc((void**)&h->d_ts, sizeof(float2) * (size_t)R * K * n));
            JRC_HIP(ctx, hipMemcpy(h->d_dop, dop.data(), sizeof(float2) * (size_t)K * n, hipMemcpyHostToDevice));
            JRC_HIP(ctx, hipMemcpy(h->d_ts, tsp.data(), sizeof(float2) * (size_t)R * K * n, hipMemcpyHostToDevice));
        }
        JRC_HIP(ctx, hipMalloc((void**)&h->d_w1, sizeof(float2) * (size_t)n1));
        JRC_HIP(ctx, hipMalloc((void**)&h->d_two, sizeof(float2) * (size_t)n));
        JRC_HIP(ctx, hipMemcpy(h->d_w1, w1.data(), sizeof(float2) * (size_t)n1, hipMemcpyHostToDevice));
        JRC_HIP(ctx, hipMemcpy(h->d_two, two.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
        JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)td_col_fwd_kernel<16>, td_col_lds_bytes(n1, 16)));
        JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)td_col_inv_kernel<16>, td_col_lds_bytes(n1, 16)));
        h->n = n; h->M = n; h->n2 = n2; h->n1 = n1; h->nrad = pl.nrad; h->direct = true;
        for (int i = 0; i < 24; i++) h->rad[i] = i < pl.nrad ? pl.rad[i] : 0;
        return JRC_OK;
    }
    // chirp c[i] = exp(-j pi i^2 / n) with i^2 reduced mod 2n in integers; b = conj(c) wrapped to length M, / M
    std::vector<float2> chirp((size_t)n), bpad((size_t)M, make_float2(0.f, 0.f));
    for (long i = 0; i < n; i++) {
        const long q = (i * i) % (2L * n);
        const double a = M_PI * (double)q / (double)n;
        chirp[i] = make_float2((float)std::cos(a), (float)(-std::sin(a)));
        const float2 bv = make_float2((float)(std::cos(a) / (double)M), (float)(std::sin(a) / (double)M));
        bpad[i] = bv;
        if (i) bpad[M - i] = bv;
    }
    if (K > 0) {
        JRC_HIP(ctx, hipMalloc((void**)&h->d_dop, sizeof(float2) * (size_t)K * n));
        JRC_HIP(ctx, hipMalloc((void**)&h->d_ts, sizeof(float2) * (size_t)R * K * n));
        JRC_HIP(ctx, hipMemcpy(h->d_dop, dop.data(), sizeof(float2) * (size_t)K * n, hipMemcpyHostToDevice));
        JRC_HIP(ctx, hipMemcpy(h->d_ts, ts.data(), sizeof(float2) * (size_t)R * K * n, hipMemcpyHostToDevice));
    }
    JRC_HIP(ctx, hipMalloc((void**)&h->d_chirp, sizeof(float2) * (size_t)n));
    JRC_HIP(ctx, hipMemcpy(h->d_chirp, chirp.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    JRC_HIP(ctx, hipMalloc((void**)&h->d_bhat, sizeof(float2) * (size_t)M));
    float2* d_tmp = nullptr;
    JRC_HIP(ctx, hipMalloc((void**)&d_tmp, sizeof(float2) * (size_t)M));
    JRC_HIP(ctx, hipMemcpy(d_tmp, bpad.data(), sizeof(float2) * (size_t)M, hipMemcpyHostToDevice));
    h->n = n; h->M = (int)M; h->n2 = n2;
    const float2* tw256 = nullptr;
    int st = jrc_get_twiddles(ctx, TS_N1, -1, &tw256);
    if (st == JRC_OK) {
        hipLaunchKernelGGL(tsim_col_first_kernel<true>, dim3(n2 / TS_CW, 1), dim3(256), 0, stream, (const float2*)d_tmp, (long)M,
                           (const float2*)nullptr, (const float2*)nullptr, h->d_bhat, tw256, n, n2, (int)M);
        st = tsim_rows_launch(h, true, h->d_bhat, 0, TS_N1, stream);
    }
    hipError_t e = hipStreamSynchronize(stream);
    (void)hipFree(d_tmp);
    if (st != JRC_OK) { h->n = 0; return st; }
    if (e != hipSuccess) { h->n = 0; return jrc_fail(ctx, JRC_ERR_HIP, "target_simulator table setup: %s", hipGetErrorString(e)); }
    return JRC_OK;
}

extern "C" jrc_tsim* jrc_tsim_create(jrc_ctx* ctx, const jrc_tsim_cfg* cfg)
{
    if (!ctx) return nullptr;
    if (!cfg || cfg->n_targets < 0 || cfg->n_rx < 1 || cfg->samp_rate <= 0 || cfg->max_bursts < 1 ||
        (cfg->n_targets > 0 && (!cfg->range || !cfg->velocity || !cfg->rcs || !cfg->azimuth)) || !cfg->position_rx) {
        jrc_fail(ctx, JRC_ERR_INVALID_ARG, "target_simulator: invalid configuration");
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { jrc_fail(ctx, JRC_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    jrc_tsim* h = new jrc_tsim();
    h->ctx = ctx;
    h->K = cfg->n_targets; h->R = cfg->n_rx;
    h->range.assign(cfg->range, cfg->range + h->K); h->velocity.assign(cfg->velocity, cfg->velocity + h->K);
    h->rcs.assign(cfg->rcs, cfg->rcs + h->K); h->azimuth.assign(cfg->azimuth, cfg->azimuth + h->K);
    h->position_rx.assign(cfg->position_rx, cfg->position_rx + h->R);
    h->samp_rate = cfg->samp_rate; h->center_freq = cfg->center_freq; h->self_coupling_db = cfg->self_coupling_db;
    h->rndm_phaseshift = cfg->rndm_phaseshift; h->self_coupling = cfg->self_coupling; h->sum_targets = cfg->sum_targets;
    h->max_bursts = cfg->max_bursts;
    tsim_setup_targets(h);
    if (hipMalloc((void**)&h->d_phase, sizeof(float2) * (size_t)(h->K > 0 ? h->K : 1)) != hipSuccess) {
        jrc_fail(ctx, JRC_ERR_NOMEM, "target_simulator: hipMalloc failed");
        delete h;
        return nullptr;
    }
    return h;
}

extern "C" void jrc_tsim_destroy(jrc_tsim* h)
{
    if (!h) return;
    (void)hipSetDevice(h->ctx->device);
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    tsim_free_tables(h);
    (void)hipFree(h->d_u); (void)hipFree(h->d_g); (void)hipFree(h->d_phase);
    delete h;
}

// The direct route for one simulator or for several whose outputs are summed (all prepared for bursts of n samples, same plan, same
// antenna count): the (simulator, target) pairs are the "virtual targets" of the three launches.  Work buffers: the first simulator's.
static int tsim_run_direct(jrc_tsim* const* sims, int n_sims, int n_bursts, int n, const jrc_cf32* const* d_in, jrc_cf32* d_out,
                           const jrc_cf32* const* target_phase /* host, per simulator, or null */, int accumulate_out, hipStream_t s)
{
    jrc_tsim* h = sims[0];
    jrc_ctx* ctx = h->ctx;
    const int R = h->R, n1 = h->n1, n2 = h->n2;
    td_srcs srcs; td_ts ts; td_self self;
    self.n = 0; ts.use_phase = 0;
    int V = 0;
    for (int q = 0; q < n_sims; q++) {
        const jrc_tsim* g = sims[q];
        const int K = g->K, k0 = g->sum_targets ? 0 : K - 1;
        const bool ph = g->rndm_phaseshift && target_phase && target_phase[q];
        for (int k = k0; k < K; k++, V++) {
            if (V >= TD_MAXV) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "target_simulator: more than %d (simulator, target) pairs in one launch", TD_MAXV);
            srcs.in[V] = (const float2*)d_in[q];
            srcs.dop[V] = (const float2*)g->d_dop + (size_t)k * n;
            ts.tsp[V] = (const float2*)g->d_ts + (size_t)k * n;             // [R][K][n]: antenna l at + l K n
            ts.phase[V] = ph ? make_float2(target_phase[q][k].re, target_phase[q][k].im) : make_float2(1.f, 0.f);
            if (ph) ts.use_phase = 1;
        }
        if (g->self_coupling) self.in[self.n++] = (const float2*)d_in[q];
    }
    for (int v = V; v < TD_MAXV; v++) { srcs.in[v] = srcs.dop[v] = ts.tsp[v] = nullptr; ts.phase[v] = make_float2(1.f, 0.f); }
    for (int q = self.n; q < TD_MAXSIMS; q++) self.in[q] = nullptr;
    const long ts_l_stride = (long)h->K * n;
    const float sc = (float)std::pow(10, h->self_coupling_db / 20.0);         // :376
    const size_t need_u = sizeof(float2) * (size_t)n_bursts * V * n;
    if (need_u > h->u_cap) {                                                // more pairs than this simulator's own targets: grow its spectrum buffer
        JRC_HIP(ctx, hipStreamSynchronize(s));
        (void)hipFree(h->d_u); h->d_u = nullptr; h->u_cap = 0;
        JRC_HIP(ctx, hipMalloc((void**)&h->d_u, need_u));
        h->u_cap = need_u;
    }
    if (const char* e_on = getenv("JRC_TSIM_ONCHIP"); e_on && *e_on && *e_on != '0') {     // read per call, like JRC_TSIM_BLUESTEIN
        // the burst-on-chip kernel where (2 + R) x n cells fit a workgroup's LDS (opt-in, JRC_TSIM_ONCHIP=1: written without a device, never timed)
        const size_t lds_on = sizeof(float2) * (size_t)(2 + R) * n;
        if (lds_on <= ctx->max_lds_per_block && n >= 2)
            return td_onchip_launch(ctx, s, srcs, ts, ts_l_stride, self, V, R, n, n1, n2, n_bursts, (float2*)d_out, sc, accumulate_out ? 1 : 0);
    }
    const float2* tw256 = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, TS_N1, -1, &tw256));
    const td_plan pl = tsim_plan_of(h);
    const int cw = td_pick_cw(n1, n2);
    const size_t lds = td_col_lds_bytes(n1, cw);
    if (cw == 32)
        hipLaunchKernelGGL(td_col_fwd_kernel<32>, dim3(n2 / 32, n_bursts, V), dim3(256), lds, s, srcs, (long)n, h->d_u, (const float2*)h->d_w1, (const float2*)h->d_two, pl, n);
    else
        hipLaunchKernelGGL(td_col_fwd_kernel<16>, dim3(n2 / 16, n_bursts, V), dim3(256), lds, s, srcs, (long)n, h->d_u, (const float2*)h->d_w1, (const float2*)h->d_two, pl, n);
    JRC_HIP(ctx, hipGetLastError());
    const long rows = (long)n_bursts * n1;
    if (n2 >= 256) {
        const float2* twn2 = nullptr;
        if (n2 > 256) JRC_TRY(jrc_get_twiddles(ctx, n2, -1, &twn2));
        const int m = n2 / 256, rpb = 16 / m;                        // rows per workgroup
        const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
        for (int l0 = 0; l0 < R;) {
            // one spectrum per launch: all R antennas out of one forward transform, one after the other (RC 0); a sum of spectra keeps the
            // antennas' sums in accumulators, two antennas per launch
            const int rc = (V == 1) ? 0 : ((R - l0 >= 2) ? 2 : 1);
#define TD_ROWS(MM, RC_)                                                                                                                  \
            hipLaunchKernelGGL((td_rows_m_kernel<MM, RC_>), grid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g, ts, ts_l_stride, tw256, twn2, V, R, l0, n1, rows)
#define TD_ROWS_RC(MM) do { if (rc == 0) TD_ROWS(MM, 0); else if (rc == 2) TD_ROWS(MM, 2); else TD_ROWS(MM, 1); } while (0)
            if (m == 1) {
                if (rc == 0) hipLaunchKernelGGL(td_rows256_kernel<0>, grid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g, ts, ts_l_stride, tw256, V, R, l0, n1, rows);
                else if (rc == 2) hipLaunchKernelGGL(td_rows256_kernel<2>, grid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g, ts, ts_l_stride, tw256, V, R, l0, n1, rows);
                else hipLaunchKernelGGL(td_rows256_kernel<1>, grid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g, ts, ts_l_stride, tw256, V, R, l0, n1, rows);
            } else if (m == 2) TD_ROWS_RC(2);
            else if (m == 4) TD_ROWS_RC(4);
            else if (m == 8) TD_ROWS_RC(8);
            else TD_ROWS_RC(16);
#undef TD_ROWS_RC
#undef TD_ROWS
            JRC_HIP(ctx, hipGetLastError());
            l0 += rc ? rc : R;
        }
    } else {
        const float2 *twf = nullptr, *twi = nullptr;
        JRC_TRY(jrc_get_twiddles(ctx, n2, -1, &twf));
        JRC_TRY(jrc_get_twiddles(ctx, n2, +1, &twi));
        const int tp = n2 / 4, per_block = 256 / tp;
        const size_t lds_rows = sizeof(float2) * 3 * (size_t)n2 * per_block;
        hipLaunchKernelGGL(td_rows_small_kernel, dim3((unsigned)((rows + per_block - 1) / per_block), R), dim3(256), lds_rows, s, (const float2*)h->d_u,
                           h->d_g, ts, ts_l_stride, twf, twi, V, R, n1, n2, jrc_ilog2(n2), rows, tp);
        JRC_HIP(ctx, hipGetLastError());
    }
    float2* out = (float2*)d_out;
    if (cw == 32)
        hipLaunchKernelGGL(td_col_inv_kernel<32>, dim3(n2 / 32, n_bursts * R), dim3(256), lds, s, (const float2*)h->d_g, out, (long)R * n, (long)n, self, (long)n,
                           sc, accumulate_out ? 1 : 0, (const float2*)h->d_w1, (const float2*)h->d_two, pl, R, n);
    else
        hipLaunchKernelGGL(td_col_inv_kernel<16>, dim3(n2 / 16, n_bursts * R), dim3(256), lds, s, (const float2*)h->d_g, out, (long)R * n, (long)n, self, (long)n,
                           sc, accumulate_out ? 1 : 0, (const float2*)h->d_w1, (const float2*)h->d_two, pl, R, n);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

extern "C" int jrc_tsim_run_dev(jrc_tsim* h, int n_bursts, int n_input, const jrc_cf32* d_in, jrc_cf32* d_out,
                                const jrc_cf32* target_phase, int accumulate_out, void* stream)
{
    JRC_TRACE("jrc_tsim_run_dev");
    if (!h) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = h->ctx;
    if (n_bursts < 0 || n_input < 0 || (n_bursts > 0 && n_input > 0 && (!d_in || !d_out)))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "target_simulator: invalid arguments");
    if (n_bursts > h->max_bursts)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "target_simulator: %d bursts exceed max_bursts %d", n_bursts, h->max_bursts);
    if (n_bursts == 0 || n_input == 0) return JRC_OK;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int n = n_input, K = h->K, R = h->R;
    const float2* in = (const float2*)d_in;
    float2* out = (float2*)d_out;
    const float sc = (float)std::pow(10, h->self_coupling_db / 20.0);         // :376
    if (K == 0) {
        hipLaunchKernelGGL(tsim_passthrough_kernel, dim3((n + 255) / 256, n_bursts * R), dim3(256), 0, s, out, (long)R * n, (long)n,
                           in, (long)n, sc, h->self_coupling, accumulate_out, R, n);
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    JRC_TRY(tsim_prepare(h, n, s));
    const bool use_phase = h->rndm_phaseshift && target_phase;
    if (h->direct) {
        jrc_tsim* one[1] = {h};
        const jrc_cf32* ins[1] = {d_in};
        const jrc_cf32* phs[1] = {use_phase ? target_phase : nullptr};
        return tsim_run_direct(one, 1, n_bursts, n, ins, d_out, phs, accumulate_out, s);
    }
    if (use_phase) JRC_HIP(ctx, hipMemcpyAsync(h->d_phase, target_phase, sizeof(float2) * (size_t)K, hipMemcpyHostToDevice, s));
    const float2* tw256 = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, TS_N1, -1, &tw256));
    const int M = h->M, n2 = h->n2;
    // as written in the reference every target overwrites the output buffer (:354-366), so only the last one is
    // observable; sum_targets accumulates them instead (in the middle kernel, between the two convolutions)
    const int k0 = h->sum_targets ? 0 : K - 1, Kz = K - k0;
    const dim3 grid_first(n2 / TS_CW, n_bursts, Kz), grid_mid(n2 / TS_CW, n_bursts), grid_out(n2 / TS_CW, n_bursts * R);
    hipLaunchKernelGGL(tsim_col_first_kernel<false>, grid_first, dim3(256), 0, s, in, (long)n, (const float2*)h->d_dop + (size_t)k0 * n,
                       (const float2*)h->d_chirp, h->d_u, tw256, n, n2, M);
    JRC_HIP(ctx, hipGetLastError());
    JRC_TRY(tsim_rows_launch(h, false, h->d_u, 0, (size_t)n_bursts * Kz * TS_N1, s));
    const float2* ph = use_phase ? (const float2*)h->d_phase + k0 : (const float2*)nullptr;
    if (Kz == 1) {
        hipLaunchKernelGGL(tsim_col_mid1_kernel, grid_mid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g,
                           (const float2*)h->d_ts + (size_t)k0 * n, (long)K * n, ph, tw256, R, n, n2, M);
        JRC_HIP(ctx, hipGetLastError());
    }
    for (int l0 = (Kz == 1) ? R : 0; l0 < R;) {
        const int rc = (R - l0 >= 4) ? 4 : ((R - l0 >= 2) ? 2 : 1);
        const float2* tsp = (const float2*)h->d_ts + ((size_t)l0 * K + k0) * n;
        if (rc == 4) hipLaunchKernelGGL(tsim_col_mid_kernel<4>, grid_mid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g, tsp, (long)K * n, ph, tw256, Kz, R, l0, n, n2, M);
        else if (rc == 2) hipLaunchKernelGGL(tsim_col_mid_kernel<2>, grid_mid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g, tsp, (long)K * n, ph, tw256, Kz, R, l0, n, n2, M);
        else hipLaunchKernelGGL(tsim_col_mid_kernel<1>, grid_mid, dim3(256), 0, s, (const float2*)h->d_u, h->d_g, tsp, (long)K * n, ph, tw256, Kz, R, l0, n, n2, M);
        JRC_HIP(ctx, hipGetLastError());
        l0 += rc;
    }
    JRC_TRY(tsim_rows_launch(h, false, h->d_g, 1, (size_t)n_bursts * R * TS_N1, s));
    hipLaunchKernelGGL(tsim_col_out_kernel, grid_out, dim3(256), 0, s, (const float2*)h->d_g, out, (long)R * n, (long)n,
                       (const float2*)h->d_chirp, in, (long)n, sc, h->self_coupling ? 1 : 0, accumulate_out ? 1 : 0, tw256, R, n, n2, M);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// Several simulators whose RX outputs a flowgraph adds (the target_simulator per TX port feeding one blocks_add_xx per RX antenna,
// examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc): out_l (+)= sum_t sim_t(in_t)_l with the sum taken on the spectrum — one inverse
// transform per RX antenna instead of one per (TX, RX) pair, the output written once.  Needs the direct route for this burst length and
// simulators of one context with the same antenna count; anything else: JRC_ERR_UNSUPPORTED, and the caller runs them one by one with
// accumulate_out (which is what this call equals, to the rounding of a float sum taken in another order).
extern "C" int jrc_tsim_run_sum_dev(jrc_tsim* const* sims, int n_sims, int n_bursts, int n_input, const jrc_cf32* const* d_in, jrc_cf32* d_out,
                                    const jrc_cf32* const* target_phase, int accumulate_out, void* stream)
{
    JRC_TRACE("jrc_tsim_run_sum_dev");
    if (!sims || n_sims < 1 || !sims[0]) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = sims[0]->ctx;
    if (n_sims > TD_MAXSIMS) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "jrc_tsim_run_sum_dev: at most %d simulators per call", TD_MAXSIMS);
    if (n_bursts < 0 || n_input < 0 || !d_in || (n_bursts > 0 && n_input > 0 && !d_out))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_tsim_run_sum_dev: invalid arguments");
    for (int q = 0; q < n_sims; q++) {
        if (!sims[q] || (n_bursts > 0 && n_input > 0 && !d_in[q])) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_tsim_run_sum_dev: null simulator or input");
        if (sims[q]->ctx != ctx || sims[q]->R != sims[0]->R)
            return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "jrc_tsim_run_sum_dev: simulators of one context and one antenna count only");
        if (sims[q]->K == 0) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "jrc_tsim_run_sum_dev: a simulator without targets");
        // one antenna stride (K n) and one self-coupling gain serve the whole launch: the simulators of a flowgraph's TX ports are built alike
        if (sims[q]->K != sims[0]->K || sims[q]->sum_targets != sims[0]->sum_targets)
            return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "jrc_tsim_run_sum_dev: simulators with different target counts or sum_targets settings");
        if (sims[q]->self_coupling != sims[0]->self_coupling || (sims[q]->self_coupling && sims[q]->self_coupling_db != sims[0]->self_coupling_db))
            return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "jrc_tsim_run_sum_dev: simulators with different self-coupling settings");
        if (n_bursts > sims[q]->max_bursts)
            return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "target_simulator: %d bursts exceed max_bursts %d", n_bursts, sims[q]->max_bursts);
    }
    if (n_bursts == 0 || n_input == 0) return JRC_OK;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    for (int q = 0; q < n_sims; q++) {
        JRC_TRY(tsim_prepare(sims[q], n_input, s));
        if (!sims[q]->direct)
            return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "jrc_tsim_run_sum_dev: bursts of %d samples do not take the direct route (n = n1 x 2^a, 2^a >= 16, n1 <= %d)",
                            n_input, TD_MAX_N1);
    }
    return tsim_run_direct(sims, n_sims, n_bursts, n_input, d_in, d_out, target_phase, accumulate_out, s);
}

extern "C" int jrc_tsim_work(jrc_tsim* h, const jrc_cf32* in, int n_input, jrc_cf32* const* out, const jrc_cf32* target_phase)
{
    if (!h) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = h->ctx;
    if (n_input < 0 || (n_input > 0 && (!in || !out))) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "target_simulator: null buffers");
    if (n_input == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = sizeof(float2) * (size_t)n_input, obytes = bytes * (size_t)h->R;
    JRC_TRY(jrc_ensure_pinned(ctx, bytes + obytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, obytes));
    memcpy(ctx->pinned, in, bytes);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    JRC_TRY(jrc_tsim_run_dev(h, 1, n_input, (const jrc_cf32*)ctx->scratch[0], (jrc_cf32*)ctx->scratch[1], target_phase, 0, ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + bytes, ctx->scratch[1], obytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int l = 0; l < h->R; l++) memcpy(out[l], (char*)ctx->pinned + bytes + (size_t)l * bytes, bytes);
    return n_input;
}

extern "C" int jrc_tsim_set_targets(jrc_tsim* h, int n_targets, const float* range, const float* velocity, const float* rcs,
                                    const float* azimuth)
{
    if (!h) return JRC_ERR_INVALID_ARG;
    if (n_targets < 0 || (n_targets > 0 && (!range || !velocity || !rcs || !azimuth)))
        return jrc_fail(h->ctx, JRC_ERR_INVALID_ARG, "target_simulator: invalid targets");
    JRC_HIP(h->ctx, hipSetDevice(h->ctx->device));
    JRC_HIP(h->ctx, hipStreamSynchronize(h->ctx->stream));
    h->K = n_targets;
    h->range.assign(range, range + n_targets); h->velocity.assign(velocity, velocity + n_targets);
    h->rcs.assign(rcs, rcs + n_targets); h->azimuth.assign(azimuth, azimuth + n_targets);
    tsim_setup_targets(h);
    tsim_free_tables(h);                                  // d_new_channel = true (:150)
    (void)hipFree(h->d_phase); h->d_phase = nullptr;
    JRC_HIP(h->ctx, hipMalloc((void**)&h->d_phase, sizeof(float2) * (size_t)(n_targets > 0 ? n_targets : 1)));
    return JRC_OK;
}

extern "C" int jrc_tsim_burst_capacity(const jrc_tsim* h) { return h ? h->max_bursts : 0; }

// ---- zero_pad (lib/zero_pad_impl.cc:62-94): out = [pad_front noise | in | pad_tail noise], noise ~ N(0, 1e-2) per component.
// The reference draws from std::random_device every call; here a counter-based generator (splitmix64 -> Box-Muller) keyed by
// (seed, burst, sample) makes the padding reproducible. ------------------------------------------------------------------
__global__ void zero_pad_kernel(const float2* __restrict__ in, float2* __restrict__ out, int n_in, int pad_front, int pad_tail,
                                unsigned long long seed, float sigma, long in_stride, long out_stride)
{
    const int n_out = n_in + pad_front + pad_tail;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i >= n_out) return;
    float2 v;
    if (i >= pad_front && i < pad_front + n_in) {
        v = in[b * (size_t)in_stride + (i - pad_front)];
    } else {
        const unsigned long long r = splitmix64(seed ^ splitmix64((b << 32) | (unsigned)i));
        const float u1 = ((float)(unsigned)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
        const float u2 = (float)(unsigned)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f);     // [0, 1)
        const float rad = sigma * sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);
        v = make_float2(rad * cs, rad * sn);
    }
    out[b * (size_t)out_stride + i] = v;
}

extern "C" int jrc_zero_pad_strided_dev(jrc_ctx* ctx, int n_bursts, int n_input, unsigned pad_front, unsigned pad_tail, uint64_t seed,
                                        const jrc_cf32* d_in, long in_stride, jrc_cf32* d_out, long out_stride, void* stream)
{
    if (!ctx || n_bursts < 0 || n_input < 0) return JRC_ERR_INVALID_ARG;
    const long n_out = (long)n_input + pad_front + pad_tail;
    if (n_bursts == 0 || n_out == 0) return (int)n_out;
    if ((n_input > 0 && !d_in) || !d_out) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "zero_pad: null buffers");
    if (in_stride < n_input || out_stride < n_out) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "zero_pad: a row stride shorter than the row");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    hipLaunchKernelGGL(zero_pad_kernel, dim3((unsigned)((n_out + 255) / 256), n_bursts), dim3(256), 0, s, (const float2*)d_in, (float2*)d_out,
                       n_input, (int)pad_front, (int)pad_tail, (unsigned long long)seed, 1e-2f, in_stride, out_stride);
    JRC_HIP(ctx, hipGetLastError());
    return (int)n_out;
}

extern "C" int jrc_zero_pad_dev(jrc_ctx* ctx, int n_bursts, int n_input, unsigned pad_front, unsigned pad_tail, uint64_t seed,
                                const jrc_cf32* d_in, jrc_cf32* d_out, void* stream)
{
    return jrc_zero_pad_strided_dev(ctx, n_bursts, n_input, pad_front, pad_tail, seed, d_in, (long)n_input, d_out,
                                    (long)n_input + pad_front + pad_tail, stream);
}

extern "C" int jrc_zero_pad(jrc_ctx* ctx, int n_input, unsigned pad_front, unsigned pad_tail, uint64_t seed, const jrc_cf32* in, jrc_cf32* out)
{
    if (!ctx || n_input < 0 || (n_input > 0 && !in) || !out) return JRC_ERR_INVALID_ARG;
    const size_t n_out = (size_t)n_input + pad_front + pad_tail;
    if (n_out == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ib = sizeof(float2) * (size_t)(n_input > 0 ? n_input : 1), ob = sizeof(float2) * n_out;
    JRC_TRY(jrc_ensure_pinned(ctx, ib + ob));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, ib));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, ob));
    if (n_input) memcpy(ctx->pinned, in, sizeof(float2) * (size_t)n_input);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, ib, hipMemcpyHostToDevice, ctx->stream));
    JRC_TRY(jrc_zero_pad_dev(ctx, 1, n_input, pad_front, pad_tail, seed, (const jrc_cf32*)ctx->scratch[0], (jrc_cf32*)ctx->scratch[1], ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + ib, ctx->scratch[1], ob, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, (char*)ctx->pinned + ib, ob);
    return (int)n_out;
}
