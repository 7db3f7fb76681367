/*
 * jrc_oracle_tsim.c — CPU restatement (TEST INFRASTRUCTURE, not product) of
 * target_simulator_impl (lib/target_simulator_impl.cc:132-385).  PARITY UNPINNED
 * (see jrc_oracle.h): the reference has no tests or vectors for this block; this file
 * follows its float/double evaluation order line by line, and the two FFTW3f calls
 * (gr::fft::fft_complex forward / reverse, unnormalised, any length) are restated as a
 * mixed-radix DFT evaluated in double and rounded once to float.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "jrc_oracle.h"

#define TS_PI 3.14159265358979323846 /* GR_M_PI */
static const double FOUR_PI_CUBED_SQRT = 44.54662397465366; /* :33 */
static const float C_LIGHT = 3e8f;                          /* target_simulator_impl.h: c_light */

struct orc_tsim_state {
    int K, R;
    float *range, *velocity, *rcs, *azimuth, *position_rx;
    int samp_rate;
    float center_freq, self_coupling_db;
    int rndm_phaseshift, self_coupling;
    float *doppler, *scale_ampl; /* [K] */
    float* timeshift;            /* [R][K] */
    int buff_size;               /* :89 d_buff_size = 2 */
    float* freq;                 /* [n] */
    float* filt_doppler;         /* [K][n] complex */
    float* filt_time;            /* [R][K][n] complex */
};

static float* dupf(const float* s, int n)
{
    float* d = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(d, s, sizeof(float) * (size_t)n);
    return d;
}

/* setup_targets (:132-198) */
orc_tsim_state* orc_tsim_create(int K, const float* range, const float* velocity, const float* rcs,
                                const float* azimuth, int R, const float* position_rx, int samp_rate,
                                float center_freq, float self_coupling_db, int rndm_phaseshift, int self_coupling)
{
    orc_tsim_state* st = (orc_tsim_state*)calloc(1, sizeof(*st));
    st->K = K; st->R = R;
    st->range = dupf(range, K); st->velocity = dupf(velocity, K); st->rcs = dupf(rcs, K);
    st->azimuth = dupf(azimuth, K); st->position_rx = dupf(position_rx, R);
    st->samp_rate = samp_rate; st->center_freq = center_freq; st->self_coupling_db = self_coupling_db;
    st->rndm_phaseshift = rndm_phaseshift; st->self_coupling = self_coupling;
    st->doppler = (float*)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    st->scale_ampl = (float*)malloc(sizeof(float) * (size_t)(K > 0 ? K : 1));
    st->timeshift = (float*)malloc(sizeof(float) * (size_t)(K * R > 0 ? K * R : 1));
    for (int k = 0; k < K; k++)            /* :163 — all-float arithmetic */
        st->doppler[k] = 2 * st->velocity[k] * st->center_freq / C_LIGHT;
    for (int l = 0; l < R; l++)            /* :175 — double, stored to float */
        for (int k = 0; k < K; k++)
            st->timeshift[l * K + k] = (float)((2.0 * st->range[k] - st->position_rx[l] * sin(st->azimuth[k] * TS_PI / 180.0)) / C_LIGHT);
    for (int k = 0; k < K; k++) {          /* :187 — c_light*sqrtf(rcs) in float, then double */
        float num = C_LIGHT * sqrtf(st->rcs[k]);
        st->scale_ampl[k] = (float)(num / FOUR_PI_CUBED_SQRT / (st->range[k] * st->range[k]) / st->center_freq);
    }
    st->buff_size = 2;
    return st;
}

void orc_tsim_destroy(orc_tsim_state* st)
{
    if (!st) return;
    free(st->range); free(st->velocity); free(st->rcs); free(st->azimuth); free(st->position_rx);
    free(st->doppler); free(st->scale_ampl); free(st->timeshift);
    free(st->freq); free(st->filt_doppler); free(st->filt_time);
    free(st);
}

/* channel filters for a burst of n samples (:249-300) */
static void tsim_filters(orc_tsim_state* st, int n)
{
    const int K = st->K, R = st->R;
    free(st->freq); free(st->filt_doppler); free(st->filt_time);
    st->freq = (float*)malloc(sizeof(float) * (size_t)n);
    st->filt_doppler = (float*)malloc(sizeof(float) * 2 * (size_t)n * (size_t)(K > 0 ? K : 1));
    st->filt_time = (float*)malloc(sizeof(float) * 2 * (size_t)n * (size_t)(K * R > 0 ? K * R : 1));
    for (int i = 0; i < n; i++) {          /* :262-268 — float arithmetic */
        if (i < n / 2) st->freq[i] = i * (float)st->samp_rate / (float)n;
        else st->freq[i] = i * (float)st->samp_rate / (float)n - (float)st->samp_rate;
    }
    for (int k = 0; k < K; k++) {
        float complex phase_doppler = 0;   /* :281 gr_complex */
        float* fd = st->filt_doppler + 2 * (size_t)k * (size_t)n;
        for (int i = 0; i < n; i++) {      /* :282-287 */
            float complex e = cexpf(phase_doppler) * st->scale_ampl[k];
            fd[2 * i] = crealf(e); fd[2 * i + 1] = cimagf(e);
            double next = fmod(cimagf(phase_doppler) + 2 * TS_PI * st->doppler[k] / (float)st->samp_rate, 2 * TS_PI);
            phase_doppler = CMPLXF(0.0f, (float)next);
        }
        for (int l = 0; l < R; l++) {      /* :291-305 */
            float* ft = st->filt_time + 2 * ((size_t)l * K + k) * (size_t)n;
            for (int i = 0; i < n; i++) {
                double ph = fmod(2 * TS_PI * (st->timeshift[l * K + k]) * (st->freq[i] + st->center_freq), 2 * TS_PI);
                float complex phase_time = CMPLXF(0.0f, (float)ph);
                float complex e = cexpf(-phase_time);
                ft[2 * i] = crealf(e) / (float)n; ft[2 * i + 1] = cimagf(e) / (float)n;
            }
        }
    }
    st->buff_size = n;
}

const float* orc_tsim_filt_doppler(orc_tsim_state* st, int n, int k)
{
    if (st->buff_size != n || !st->filt_doppler) tsim_filters(st, n);
    return st->filt_doppler + 2 * (size_t)k * (size_t)n;
}
const float* orc_tsim_filt_time(orc_tsim_state* st, int n, int l, int k)
{
    if (st->buff_size != n || !st->filt_time) tsim_filters(st, n);
    return st->filt_time + 2 * ((size_t)l * st->K + k) * (size_t)n;
}

/* ---- unnormalised DFT of any length (FFTW3f restated): recursive mixed radix, double ---- */
static void dft_rec(int n, int stride, const double complex* in, double complex* out, const double complex* w, int wstep,
                    double complex* tmp)
{
    if (n == 1) { out[0] = in[0]; return; }
    int p = 2;
    while (n % p) p += (p == 2) ? 1 : 2;
    const int m = n / p;
    /* p sub-transforms of length m over the decimated inputs */
    for (int r = 0; r < p; r++) dft_rec(m, stride * p, in + (size_t)r * stride, out + (size_t)r * m, w, wstep * p, tmp);
    /* combine: X[q + m*s] = sum_r w^(r*(q+m*s)) * Y_r[q] */
    for (int q = 0; q < m; q++) {
        for (int s = 0; s < p; s++) {
            const long kk = q + (long)m * s;
            double complex acc = 0;
            for (int r = 0; r < p; r++) acc += out[(size_t)r * m + q] * w[((long)r * kk % n) * wstep];
            tmp[s] = acc;
        }
        for (int s = 0; s < p; s++) out[q + (size_t)m * s] = tmp[s];
    }
}
/* note: the in-place combine above is safe because column q only reads and writes the p entries {r*m+q} */

/* in-place double-precision form for orc_fft_vcc's non-power-of-two sizes */
void orc_dft_any_f64(int n, int sign, double* re, double* im)
{
    if (n <= 0) return;
    double complex* a = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    double complex* b = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    double complex* w = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    double complex* tmp = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    for (int i = 0; i < n; i++) {
        a[i] = re[i] + I * im[i];
        w[i] = cos(2.0 * TS_PI * i / n) + I * (double)sign * sin(2.0 * TS_PI * i / n);
    }
    dft_rec(n, 1, a, b, w, 1, tmp);
    for (int i = 0; i < n; i++) { re[i] = creal(b[i]); im[i] = cimag(b[i]); }
    free(a); free(b); free(w); free(tmp);
}

void orc_dft_any(int n, int forward, const float* in, float* out)
{
    if (n <= 0) return;
    double complex* a = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    double complex* b = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    double complex* w = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    double complex* tmp = (double complex*)malloc(sizeof(double complex) * (size_t)n);
    const double sgn = forward ? -1.0 : 1.0;
    for (int i = 0; i < n; i++) {
        a[i] = in[2 * i] + I * (double)in[2 * i + 1];
        w[i] = cos(2.0 * TS_PI * i / n) + I * sgn * sin(2.0 * TS_PI * i / n);
    }
    dft_rec(n, 1, a, b, w, 1, tmp);
    for (int i = 0; i < n; i++) { out[2 * i] = (float)creal(b[i]); out[2 * i + 1] = (float)cimag(b[i]); }
    free(a); free(b); free(w); free(tmp);
}

static void cmul_vec(float* dst, const float* a, const float* b, int n) /* volk_32fc_x2_multiply_32fc, generic kernel */
{
    for (int i = 0; i < n; i++) {
        const float ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
        dst[2 * i] = ar * br - ai * bi;
        dst[2 * i + 1] = ar * bi + ai * br;
    }
}

/* work (:202-385).  target_phase: K complex multipliers (the exp(j*2*pi*rand) of :316-321, drawn by the caller) used when
 * rndm_phaseshift; sum_targets = 0 restates the reference as written (every target overwrites `out`, :354-362, so the
 * last target is what leaves the block), 1 accumulates the targets instead. */
int orc_tsim_work(orc_tsim_state* st, const float* in, int n_input, float* const* out, const float* target_phase,
                  int sum_targets)
{
    const int K = st->K, R = st->R, n = n_input;
    if (n <= 0) return 0;
    if (st->buff_size != n || !st->freq) tsim_filters(st, n);
    float* bt = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    float* bf = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    float* fo = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    for (int l = 0; l < R; l++) {
        float* o = out[l];
        memset(o, 0, sizeof(float) * 2 * (size_t)n);            /* :338 */
        for (int k = 0; k < K; k++) {
            cmul_vec(bt, in, st->filt_doppler + 2 * (size_t)k * n, n);              /* :345 */
            orc_dft_any(n, 1, bt, fo);                                               /* :348-349 */
            cmul_vec(bf, fo, st->filt_time + 2 * ((size_t)l * K + k) * n, n);       /* :352 */
            orc_dft_any(n, 0, bf, fo);                                               /* :355-356 */
            if (st->rndm_phaseshift && target_phase) {                                /* :358-362 */
                for (int i = 0; i < n; i++) {
                    const float ar = fo[2 * i], ai = fo[2 * i + 1], br = target_phase[2 * k], bi = target_phase[2 * k + 1];
                    bt[2 * i] = ar * br - ai * bi; bt[2 * i + 1] = ar * bi + ai * br;
                }
            } else {
                memcpy(bt, fo, sizeof(float) * 2 * (size_t)n);                         /* :366 */
            }
            if (sum_targets && k > 0)
                for (int i = 0; i < 2 * n; i++) o[i] += bt[i];
            else
                memcpy(o, bt, sizeof(float) * 2 * (size_t)n);
        }
        if (st->self_coupling) {                                                        /* :372-378 */
            const float sc = (float)pow(10, st->self_coupling_db / 20.0);              /* (gr_complex)pow(...) */
            for (int i = 0; i < n; i++) {
                /* complex * complex with imag(sc) = 0 */
                const float pr = sc * in[2 * i] - 0.0f * in[2 * i + 1];
                const float pi_ = sc * in[2 * i + 1] + 0.0f * in[2 * i];
                o[2 * i] += pr; o[2 * i + 1] += pi_;
            }
        }
    }
    free(bt); free(bf); free(fo);
    return n;
}
