/*
 * jrc_oracle.c — see jrc_oracle.h.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED
 * (no reference tests/fixtures exist and the reference cannot be built here).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: the reference is built for baseline x86-64 (no FMA),
 * so every float product and sum below is individually rounded, exactly like
 * libstdc++'s std::complex<float> operators in the reference build.
 */
#include "jrc_oracle.h"

void orc_dft_any_f64(int n, int sign, double* re, double* im);   /* jrc_oracle_tsim.c */

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * A1  mimo_ofdm_radar
 * ---------------------------------------------------------------------------------------- */
struct orc_radar_state {
    int N, T, R, S, Npre, Ir, interleave;
    int bg_removal, bg_recording, record_len;
    /* boost::circular_buffer<std::vector<gr_complex>> radar_chan_est_buffer
     * (lib/mimo_ofdm_radar_impl.h:52): index 0 = oldest */
    float* ring;     /* record_len x (P*N) complex */
    int    ring_size;
    int    ring_head; /* slot of the oldest element */
    float* temp;     /* radar_chan_est_temp */
    float* est;      /* radar_chan_est */
};

orc_radar_state* orc_radar_create(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre,
                                  int background_removal, int background_recording,
                                  int record_len, int interp_factor, int enable_tx_interleave)
{
    orc_radar_state* st = (orc_radar_state*)calloc(1, sizeof(*st));
    st->N = fft_len; st->T = N_tx; st->R = N_rx; st->S = N_sym; st->Npre = N_pre;
    st->Ir = interp_factor; st->interleave = enable_tx_interleave;
    st->bg_removal = background_removal; st->bg_recording = background_recording;
    st->record_len = record_len;
    size_t pn = (size_t)N_tx * N_rx * fft_len;
    st->ring = (float*)calloc((size_t)(record_len > 0 ? record_len : 1) * pn * 2, sizeof(float));
    st->temp = (float*)calloc(pn * 2, sizeof(float)); /* std::vector::resize value-initialises (:115) */
    st->est  = (float*)calloc(pn * 2, sizeof(float));
    return st;
}

void orc_radar_destroy(orc_radar_state* st)
{
    if (!st) return;
    free(st->ring); free(st->temp); free(st->est); free(st);
}

void orc_radar_set_background_record(orc_radar_state* st, int on) { st->bg_recording = on; }
int  orc_radar_ring_size(const orc_radar_state* st) { return st->ring_size; }

void orc_radar_work(orc_radar_state* st, const float* const* tx, const float* const* rx,
                    long tx_discard, float* out)
{
    const int N = st->N, T = st->T, R = st->R, S = st->S, Ir = st->Ir;
    const size_t pn = (size_t)T * R * N;
    float* est = st->est;

    memset(out, 0, pn * Ir * 2 * sizeof(float));          /* :243 */
    memset(est, 0, pn * 2 * sizeof(float));               /* :244 */

    for (int sc = 0; sc < N; sc++) {                      /* :250 */
        for (int r = 0; r < R; r++) {
            const float* in_rx = rx[r] + (size_t)2 * N * st->Npre;               /* :254-255 */
            for (int t = 0; t < T; t++) {
                const float* in_tx = tx[t] + (size_t)2 * N * st->Npre
                                           + (size_t)2 * N * tx_discard;        /* :258-260 */
                size_t idx = st->interleave ? (size_t)sc + (size_t)N * (t * R + r)   /* :264 */
                                            : (size_t)sc + (size_t)N * (r * T + t);  /* :268 */
                float ar = est[2 * idx], ai = est[2 * idx + 1];
                for (int sym = 0; sym < S; sym++) {       /* :271-274 */
                    float a = in_rx[2 * (sc + (size_t)sym * N)], b = in_rx[2 * (sc + (size_t)sym * N) + 1];
                    float c = in_tx[2 * (sc + (size_t)sym * N)], d = in_tx[2 * (sc + (size_t)sym * N) + 1];
                    /* (a+jb)*(c-jd): libstdc++ complex multiply = (ac - b(-d)) + j(a(-d) + bc) */
                    float pr = a * c + b * d;
                    float pi = b * c - a * d;
                    ar = ar + pr;
                    ai = ai + pi;
                }
                est[2 * idx] = ar; est[2 * idx + 1] = ai;

                if (st->bg_recording) {                   /* :276-279 */
                    st->temp[2 * idx] = ar; st->temp[2 * idx + 1] = ai;
                }
                if (st->bg_removal) {                     /* :281-293 */
                    int n = st->ring_size;
                    float mr = 0.0f, mi = 0.0f;
                    for (int i = 0; i < n; i++) {
                        int slot = (st->ring_head + i) % st->record_len;
                        const float* e = st->ring + ((size_t)slot * pn + idx) * 2;
                        /* complex / float divides each component (:289) */
                        mr = mr + e[0] / (float)n;
                        mi = mi + e[1] / (float)n;
                    }
                    est[2 * idx]     = ar - mr;
                    est[2 * idx + 1] = ai - mi;
                }
            }
        }
    }

    if (st->bg_removal && st->record_len > 0) {           /* :297-300 push_back (drops oldest when full) */
        int slot;
        if (st->ring_size < st->record_len) {
            slot = (st->ring_head + st->ring_size) % st->record_len;
            st->ring_size++;
        } else {
            slot = st->ring_head;
            st->ring_head = (st->ring_head + 1) % st->record_len;
        }
        memcpy(st->ring + (size_t)slot * pn * 2, st->temp, pn * 2 * sizeof(float));
    }

    for (int p = 0; p < T * R; p++)                       /* :312-315 */
        memcpy(out + (size_t)p * N * Ir * 2, est + (size_t)p * N * 2, (size_t)N * 2 * sizeof(float));
}

/* ------------------------------------------------------------------------------------------
 * stock fft_vcc (double-precision definition oracle)
 * ---------------------------------------------------------------------------------------- */
static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static void fft_core_f64(int n, int sign, double* re, double* im)
{
    /* sign = -1 forward, +1 backward; unnormalised (FFTW convention) */
    if (is_pow2(n)) {
        for (int i = 1, j = 0; i < n; i++) {              /* bit reversal */
            int bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
        }
        for (int len = 2; len <= n; len <<= 1) {
            int half = len >> 1;
            for (int k = 0; k < half; k++) {
                double ang = sign * 2.0 * M_PI * (double)k / (double)len;
                double wr = cos(ang), wi = sin(ang);
                for (int i = k; i < n; i += len) {
                    int j = i + half;
                    double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
                    re[j] = re[i] - xr; im[j] = im[i] - xi;
                    re[i] += xr; im[i] += xi;
                }
            }
        }
    } else {
        double* tr = (double*)malloc(sizeof(double) * 2 * (size_t)n);
        double* ti = tr + n;
        for (int k = 0; k < n; k++) {
            double sr = 0, si = 0;
            for (int m = 0; m < n; m++) {
                long long mk = ((long long)m * k) % n;
                double ang = sign * 2.0 * M_PI * (double)mk / (double)n;
                double wr = cos(ang), wi = sin(ang);
                sr += re[m] * wr - im[m] * wi; si += re[m] * wi + im[m] * wr;
            }
            tr[k] = sr; ti[k] = si;
        }
        memcpy(re, tr, sizeof(double) * n); memcpy(im, ti, sizeof(double) * n);
        free(tr);
    }
}

void orc_fft_vcc(int n, int forward, int shift, const float* window, long batch,
                 const float* in, float* out)
{
    double* re = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* im = re + n;
    const int half_in = n / 2;                 /* reverse+shift: offset = fft_size/2 */
    const int half_out = (n + 1) / 2;          /* forward+shift: len = ceil(fft_size/2) */
    for (long b = 0; b < batch; b++) {
        const float* x = in + (size_t)b * n * 2;
        float* y = out + (size_t)b * n * 2;
        for (int i = 0; i < n; i++) {
            int src = i;
            if (!forward && shift) src = (i + half_in) % n;   /* first half of fft input = second half of in */
            /* GR multiplies gr_complex by the float window in float (volk_32fc_32f_multiply_32fc) */
            float xr = window ? x[2 * src] * window[src] : x[2 * src];
            float xi = window ? x[2 * src + 1] * window[src] : x[2 * src + 1];
            re[i] = xr; im[i] = xi;
        }
        if ((n & (n - 1)) == 0) fft_core_f64(n, forward ? -1 : +1, re, im);
        else orc_dft_any_f64(n, forward ? -1 : +1, re, im);        /* FFTW handles any size: mixed radix in double */
        for (int i = 0; i < n; i++) {
            int src = i;
            if (forward && shift) src = (i + half_out) % n;
            y[2 * i] = (float)re[src]; y[2 * i + 1] = (float)im[src];
        }
    }
    free(re);
}

/* float radix-2 FFT, for CPU-baseline timing only (FFTW3f stand-in arithmetic type).  Twiddle and bit-reversal
 * tables are cached per (size, direction) so the timed loop does only butterflies, like a planned FFT. */
typedef struct { int n, sign; float* tw; int* rev; } f32_plan;
static _Thread_local f32_plan g_plans[8];

static const f32_plan* get_plan(int n, int sign)
{
    for (int i = 0; i < 8; i++) if (g_plans[i].n == n && g_plans[i].sign == sign) return &g_plans[i];
    int slot = 0;
    for (int i = 0; i < 8; i++) if (g_plans[i].n == 0) { slot = i; break; }
    f32_plan* p = &g_plans[slot];
    free(p->tw); free(p->rev);
    p->n = n; p->sign = sign;
    p->tw = (float*)malloc(sizeof(float) * (size_t)n);
    p->rev = (int*)malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n / 2; k++) {
        double ang = sign * 2.0 * M_PI * (double)k / (double)n;
        p->tw[2 * k] = (float)cos(ang); p->tw[2 * k + 1] = (float)sin(ang);
    }
    int bits = 0; while ((1 << bits) < n) bits++;
    for (int i = 0; i < n; i++) {
        unsigned j = 0;
        for (int q = 0; q < bits; q++) if (i & (1 << q)) j |= 1u << (bits - 1 - q);
        p->rev[i] = (int)j;
    }
    return p;
}

void orc_fft_vcc_f32(int n, int forward, int shift, long batch, const float* in, float* out)
{
    const f32_plan* pl = get_plan(n, forward ? -1 : +1);
    const float* tw = pl->tw; const int* rev = pl->rev;
    float* buf = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    const int half = n / 2;
    for (long b = 0; b < batch; b++) {
        const float* x = in + (size_t)b * n * 2;
        float* y = out + (size_t)b * n * 2;
        if (!forward && shift)
            for (int i = 0; i < n; i++) { int src = (i + half) % n, j = rev[i]; buf[2 * j] = x[2 * src]; buf[2 * j + 1] = x[2 * src + 1]; }
        else
            for (int i = 0; i < n; i++) { int j = rev[i]; buf[2 * j] = x[2 * i]; buf[2 * j + 1] = x[2 * i + 1]; }
        for (int len = 2; len <= n; len <<= 1) {
            int h = len >> 1, step = n / len;
            for (int i0 = 0; i0 < n; i0 += len)
                for (int k = 0; k < h; k++) {
                    float wr = tw[2 * k * step], wi = tw[2 * k * step + 1];
                    int i = i0 + k, j = i + h;
                    float xr = buf[2 * j] * wr - buf[2 * j + 1] * wi;
                    float xi = buf[2 * j] * wi + buf[2 * j + 1] * wr;
                    buf[2 * j] = buf[2 * i] - xr; buf[2 * j + 1] = buf[2 * i + 1] - xi;
                    buf[2 * i] += xr; buf[2 * i + 1] += xi;
                }
        }
        if (forward && shift) {
            memcpy(y, buf + 2 * half, sizeof(float) * 2 * (size_t)(n - half));
            memcpy(y + 2 * (n - half), buf, sizeof(float) * 2 * (size_t)half);
        } else {
            memcpy(y, buf, sizeof(float) * 2 * (size_t)n);
        }
    }
    free(buf);
}

/* ------------------------------------------------------------------------------------------
 * A3  matrix_transpose
 * ---------------------------------------------------------------------------------------- */
int orc_matrix_transpose(int input_len, int output_len, int interp_factor, int ninput_items,
                         const float* in, float* out)
{
    /* :82-83  float vs integer division mismatch -> runtime_error */
    if (ninput_items * (float)input_len / (float)output_len
            - ninput_items * input_len / output_len != 0)
        return -1;
    memset(out, 0, sizeof(float) * 2 * (size_t)interp_factor * output_len * input_len);   /* :97 */
    for (int l = 0; l < input_len; l++)                                                    /* :100-104 */
        for (int k = 0; k < ninput_items; k++) {
            size_t o = (size_t)l * output_len * interp_factor + k, i = (size_t)k * input_len + l;
            out[2 * o] = in[2 * i]; out[2 * o + 1] = in[2 * i + 1];
        }
    return input_len;
}

/* ------------------------------------------------------------------------------------------
 * A5  range_angle_estimator
 * ---------------------------------------------------------------------------------------- */
static int lower_bound_f(const float* v, int n, float x)
{
    int lo = 0, hi = n;                    /* first index with v[i] >= x  (std::lower_bound) */
    while (lo < hi) { int mid = lo + (hi - lo) / 2; if (v[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

void orc_ra_estimate(int vlen, int n_inputs, const float* in,
                     const float* range_bins, int n_range_bins,
                     const float* angle_bins, int n_angle_bins,
                     float noise_discard_range_m, float noise_discard_angle_deg,
                     float snr_threshold, float power_threshold, orc_ra_result* res)
{
    float peak_power = -1;
    float curr_power;
    int peak_range_idx = -1, peak_angle_idx = -1;

    for (int i_range = 0; i_range < n_inputs; i_range++)               /* :137-151 */
        for (int i_angle = 0; i_angle < vlen; i_angle++) {
            size_t k = (size_t)i_angle + (size_t)vlen * i_range;
            /* std::pow(std::abs(z), 2): abs -> hypotf (float); pow(float,int) promotes to double */
            curr_power = (float)pow((double)hypotf(in[2 * k], in[2 * k + 1]), 2.0);
            if (curr_power > peak_power) {
                peak_power = curr_power; peak_range_idx = i_range; peak_angle_idx = i_angle;
            }
        }
    float angle_val = angle_bins[peak_angle_idx];
    float range_val = range_bins[peak_range_idx];

    float angle_null = angle_val + 90;                                  /* :155-160 */
    if (angle_null >= 90) angle_null = angle_null - 180;

    int angle_null_idx;
    int it = lower_bound_f(angle_bins, n_angle_bins, angle_null);       /* :163-167 */
    /* :169-182.  The reference dereferences *(iter-1) and *iter before the bounds checks
     * (undefined behaviour at both ends).  Defined behaviour adopted here:
     * iter==begin -> 0; iter==end -> size-1 (then clamped to size-2 below). */
    if (it == 0) {
        angle_null_idx = 0;
    } else if (it == n_angle_bins) {
        angle_null_idx = n_angle_bins - 1;
    } else {
        double a = angle_bins[it - 1], b = angle_bins[it];
        if (fabs(angle_null - a) < fabs(angle_null - b)) angle_null_idx = it - 1;
        else angle_null_idx = it;
    }
    if (angle_null_idx == n_angle_bins - 1) angle_null_idx = n_angle_bins - 2;   /* :184-187 */

    int discard_range_idx = (int)(noise_discard_range_m / (range_bins[1] - range_bins[0]));   /* :189 */
    int discard_angle_idx = (int)(noise_discard_angle_deg /
            (angle_bins[(angle_null_idx + 1) % n_angle_bins] - angle_bins[angle_null_idx]));  /* :190 */
    if (discard_angle_idx <= 0) discard_angle_idx = 1;                                          /* :192-195 */

    int start_range_idx = peak_range_idx + n_range_bins / 2 - discard_range_idx;   /* :197-198 */
    int end_range_idx   = peak_range_idx + n_range_bins / 2 + discard_range_idx;
    int start_angle_idx = angle_null_idx - discard_angle_idx;                       /* :200-201 */
    int end_angle_idx   = angle_null_idx + discard_angle_idx;

    float noise_power = 0;
    int n_noise_samples = 0;
    for (int i_range = start_range_idx; i_range < end_range_idx; i_range++) {      /* :209-226 */
        int r_idx = ((i_range % n_inputs) + n_inputs) % n_inputs;
        for (int i_angle = start_angle_idx; i_angle < end_angle_idx; i_angle++) {
            int a_idx = ((i_angle % vlen) + vlen) % vlen;
            size_t k = (size_t)a_idx + (size_t)vlen * r_idx;
            /* float += double: the sum is formed in double and rounded to float each step */
            noise_power = (float)((double)noise_power + pow((double)hypotf(in[2 * k], in[2 * k + 1]), 2.0));
            n_noise_samples++;
        }
    }
    noise_power = noise_power / n_noise_samples;                                    /* :231 */
    float snr_est = 10 * log10f(peak_power / noise_power);                          /* :232 */

    res->peak_range_idx = peak_range_idx; res->peak_angle_idx = peak_angle_idx;
    res->angle_null_idx = angle_null_idx;
    res->discard_range_idx = discard_range_idx; res->discard_angle_idx = discard_angle_idx;
    res->n_noise_samples = n_noise_samples;
    res->peak_power = peak_power; res->noise_power = noise_power; res->snr_est = snr_est;
    res->range_val = range_val; res->angle_val = angle_val;
    res->published = (snr_est >= snr_threshold && peak_power >= power_threshold);   /* :234 */
}

/* ------------------------------------------------------------------------------------------
 * A6  ofdm_cyclic_prefix_remover
 * ---------------------------------------------------------------------------------------- */
int orc_cp_remove(int fft_len, int cp_len, long ninput_items, const float* in, float* out)
{
    int noutput_items = (int)(ninput_items / (fft_len + cp_len));                   /* :86 */
    for (int k = 0; k < noutput_items; k++)                                         /* :92-95 */
        memcpy(out + (size_t)2 * fft_len * k,
               in + (size_t)2 * (cp_len + (size_t)k * (fft_len + cp_len)),
               (size_t)fft_len * 2 * sizeof(float));
    return noutput_items;
}

/* ------------------------------------------------------------------------------------------
 * B1  fft_peak_detect
 * ---------------------------------------------------------------------------------------- */
int orc_fft_peak_detect(int samp_rate, float interp_factor, float threshold, int samp_protect,
                        long ninput_items, const float* in,
                        float* out_freq, float* out_phase, float* out_mag)
{
    int k = -1;
    float hold = -1;
    /* std::pow(10, d_threshold / 10.0): int base, double exponent -> double */
    double thr = pow(10.0, threshold / 10.0);
    for (long p = samp_protect; p < ninput_items - samp_protect; p++) {             /* :90-95 */
        float mag = hypotf(in[2 * p], in[2 * p + 1]);
        if (mag > hold && pow((double)mag, 2.0) > thr) { hold = mag; k = (int)p; }
    }
    if (k != -1) {                                                                  /* :98-107 */
        int n = (int)ninput_items;
        if (k <= n / 2)
            out_freq[0] = k / (float)n * (samp_rate * interp_factor);
        else
            out_freq[0] = -((float)samp_rate * interp_factor) + k * (samp_rate * interp_factor / (float)n);
        out_phase[0] = atan2f(in[2 * k + 1], in[2 * k]);
        out_mag[0] = hypotf(in[2 * k], in[2 * k + 1]);
    }
    return k;
}

/* ------------------------------------------------------------------------------------------
 * composed radar chain for one frame (CPU baseline leg + chain-level parity)
 *   A1 (mimo_ofdm_radar) -> A2 fft_vxx reverse/no-shift N*Ir -> A3 matrix_transpose
 *   -> A4 fft_vxx forward/shift P*Ia   (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2189-2197)
 * ---------------------------------------------------------------------------------------- */
void orc_radar_chain(orc_radar_state* st, const float* const* tx, const float* const* rx,
                     int interp_angle, float* Hpad, float* range, float* tr, float* map)
{
    const int P = st->T * st->R, NR = st->N * st->Ir, NA = P * interp_angle;
    orc_radar_work(st, tx, rx, 0, Hpad);
    orc_fft_vcc_f32(NR, 0, 0, P, Hpad, range);
    orc_matrix_transpose(NR, P, interp_angle, P, range, tr);
    orc_fft_vcc_f32(NA, 1, 1, NR, tr, map);
}
