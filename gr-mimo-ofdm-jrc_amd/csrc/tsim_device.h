// tsim_device.h — device helpers of the target simulator's direct route shared by tsim.hip (the three-pass kernels, round 5) and onchip.hip (the
// kernels written in round 6 without a device): the launch structures, the small DFT butterflies, the mixed-radix Stockham column passes over an
// LDS tile, and the counter-based generator of zero_pad's padding.  Moved here verbatim from tsim.hip — tools/device_code_diff.py shows tsim.hip's
// device code byte-identical to round 5's with it.
#pragma once

#include "fft_device.h"

#define TD_CW 16
#define TD_MAX_N1 512

struct td_plan { int n1, n2, nrad; int rad[24]; };
// the spectra a launch sums: one entry per (simulator, target) pair — the targets of one simulator (sum_targets), and the simulators of a
// flowgraph's TX ports whose outputs a blocks_add_xx adds (jrc_tsim_run_sum_dev)
#define TD_MAXV 32
#define TD_MAXSIMS 8
struct td_srcs { const float2* in[TD_MAXV]; const float2* dop[TD_MAXV]; };            // burst 0 of the pair's input; its doppler filter [n]
struct td_ts { const float2* tsp[TD_MAXV]; float2 phase[TD_MAXV]; int use_phase; };   // timeshift of (antenna 0, target) in row-pass order; per-target phase
struct td_self { const float2* in[TD_MAXSIMS]; int n; };                              // inputs whose self-coupling term is added (:372-378)

template <int R>
__device__ __forceinline__ void td_dft_small(float2 (&v)[R]);
template <> __device__ __forceinline__ void td_dft_small<2>(float2 (&v)[2]) { fft_fwd_small<2>(v); }
template <> __device__ __forceinline__ void td_dft_small<4>(float2 (&v)[4]) { fft_fwd_small<4>(v); }
template <> __device__ __forceinline__ void td_dft_small<3>(float2 (&v)[3])
{
    const float s3 = 0.86602540378443864676f;
    const float2 t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const float2 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    const float2 jd = make_float2(s3 * d.y, -s3 * d.x);                      // -j s3 d
    v[0] = cadd(v[0], t); v[1] = cadd(m, jd); v[2] = csub(m, jd);
}
template <> __device__ __forceinline__ void td_dft_small<5>(float2 (&v)[5])
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]), b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    const float2 r1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 r2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    const float2 i1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 i2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = make_float2(r1.x + i1.y, r1.y - i1.x); v[4] = make_float2(r1.x - i1.y, r1.y + i1.x);      // r -/+ j i
    v[2] = make_float2(r2.x + i2.y, r2.y - i2.x); v[3] = make_float2(r2.x - i2.y, r2.y + i2.x);
}

// one forward Stockham pass of radix R down the columns of an LDS tile [n1][CW]; thread (c, w) of (CW, nw); Ns = product of the
// radices already done.  w1[q] = exp(-j 2 pi q / n1).
template <int R, int CW>
__device__ __forceinline__ void td_col_pass(const float2* x, float2* y, const float2* w1, int n1, int Ns, int c, int w, int nw)
{
    const int m = n1 / R, tws = n1 / (Ns * R);
    for (int j = w; j < m; j += nw) {
        const int k = j % Ns;
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; t++) {
            v[t] = x[(j + t * m) * CW + c];
            if (t && k) v[t] = cmul(v[t], w1[k * t * tws]);                  // k t tws < n1
        }
        td_dft_small<R>(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int u = 0; u < R; u++) y[(j0 + u * Ns) * CW + c] = v[u];
    }
}
// any radix r (the prime factors of n1 beyond 2, 3, 5): every thread forms outputs, each as its r-term sum
template <int CW>
__device__ __forceinline__ void td_col_pass_any(const float2* x, float2* y, const float2* w1, int n1, int r, int Ns, int c, int w, int nw)
{
    const int m = n1 / r, tws = n1 / (Ns * r);
    for (int e = w; e < n1; e += nw) {
        const int j = e % m, u = e / m, k = j % Ns;
        int step = k * tws + u * m;                                          // exponent per input t: twiddle w_{Ns r}^{k t} and w_r^{u t}
        if (step >= n1) step -= n1;
        int q = 0;
        float2 acc = x[j * CW + c];
        for (int t = 1; t < r; t++) {
            q += step; if (q >= n1) q -= n1;
            acc = cadd(acc, cmul(x[(j + t * m) * CW + c], w1[q]));
        }
        y[((j - k) * r + k + u * Ns) * CW + c] = acc;
    }
}
// the whole n1-point forward transform of the tile in buf0; returns the buffer that holds the result
template <int CW>
__device__ __forceinline__ float2* td_col_transform(float2* buf0, float2* buf1, const float2* w1, const td_plan& pl, int c, int w, int nw)
{
    float2 *cur = buf0, *nxt = buf1;
    int Ns = 1;
    for (int p = 0; p < pl.nrad; p++) {
        const int r = pl.rad[p];
        if (r == 4) td_col_pass<4, CW>(cur, nxt, w1, pl.n1, Ns, c, w, nw);
        else if (r == 2) td_col_pass<2, CW>(cur, nxt, w1, pl.n1, Ns, c, w, nw);
        else if (r == 3) td_col_pass<3, CW>(cur, nxt, w1, pl.n1, Ns, c, w, nw);
        else if (r == 5) td_col_pass<5, CW>(cur, nxt, w1, pl.n1, Ns, c, w, nw);
        else td_col_pass_any<CW>(cur, nxt, w1, pl.n1, r, Ns, c, w, nw);
        __syncthreads();
        float2* t = cur; cur = nxt; nxt = t;
        Ns *= r;
    }
    return cur;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// radices of an m-point column transform: 4s, a 2, then the odd prime factors in rising order (3 and 5 have butterflies of their own)
static inline void td_factor(int m, td_plan* pl)
{
    pl->n1 = m; pl->n2 = 1; pl->nrad = 0;
    while (m % 4 == 0) { pl->rad[pl->nrad++] = 4; m /= 4; }
    if (m % 2 == 0) { pl->rad[pl->nrad++] = 2; m /= 2; }
    for (int p = 3; m > 1; p += 2)
        while (m % p == 0) { pl->rad[pl->nrad++] = p; m /= p; }
}

// onchip.hip: the whole burst through one kernel (one workgroup per burst; the caller has checked that (2 + R) x n cells fit a workgroup's LDS)
int td_onchip_launch(jrc_ctx* ctx, hipStream_t s, const td_srcs& srcs, const td_ts& ts, long ts_l_stride, const td_self& self, int V, int R, int n,
                     int d_n1, int d_n2, int n_bursts, float2* d_out, float self_coupling, int accumulate);
