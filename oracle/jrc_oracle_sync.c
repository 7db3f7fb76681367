/*
 * jrc_oracle_sync.c — CPU restatement (TEST INFRASTRUCTURE, not product) of the sample-serial sync front-end
 * (SURVEY §8(f) rank 4): moving_avg_impl::work (lib/moving_avg_impl.cc:62-98), frame_detector_impl::general_work
 * (lib/frame_detector_impl.cc:70-205), frame_sync_impl::general_work / search_frame_start (lib/frame_sync_impl.cc:89-289),
 * and the stock blocks wired in front of them in examples/simulation/communication/mimo_ofdm_jrc_comm_sim.grc
 * (blocks_delay, conjugate, multiply, complex_to_mag[_squared], moving_average_ff, abs, divide).  PARITY UNPINNED (see
 * jrc_oracle.h).  gr::filter::kernel::fir_filter_ccc is restated from its published definition
 * (y[n] = sum_k taps[k] x[n + ntaps - 1 - k]); its VOLK dot product's summation order is not reproduced (float, in order).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "jrc_oracle.h"

typedef float complex cf;

/* moving_avg_impl::work: `in` carries length-1 items of history in front (set_history), returns items produced */
int orc_moving_avg_work(int length, float scale, int max_iter, int noutput_items, const float* in_, float* out_)
{
    const cf* in = (const cf*)in_;
    cf* out = (cf*)out_;
    const unsigned num_iter = (unsigned)((noutput_items > max_iter) ? max_iter : noutput_items);   /* :78 */
    cf sum = in[0];
    for (int i = 1; i < length - 1; i++) sum += in[i];                                             /* :80-83 */
    for (unsigned i = 0; i < num_iter; i++) {                                                      /* :85-90 */
        sum += in[i + length - 1];
        out[i] = sum * scale;
        sum -= in[i];
    }
    return (int)num_iter;
}

/* the stock blocks in front of the detector for a whole capture x[n]: in_abs[i] (complex moving average of
 * x[i] conj(x[i - delay]) over `window` samples), in_cor[i] = |in_abs[i]| / |scale_p * sum_{pw} |x|^2|, and the delayed
 * samples xd[i] = x[i - delay].  Moving averages run as one work() call each (max_iter >= n). */
void orc_sync_metrics(const float* x_, int n, int delay, int window, int pwindow, float pscale, float* xd_, float* in_abs_,
                      float* in_cor)
{
    const cf* x = (const cf*)x_;
    cf* xd = (cf*)xd_;
    cf* ia = (cf*)in_abs_;
    cf* prod = (cf*)calloc((size_t)n + window, sizeof(cf));       /* window-1 zeros of history, then conj(xd) * x */
    float* pw = (float*)calloc((size_t)n + pwindow, sizeof(float));
    for (int i = 0; i < n; i++) {
        xd[i] = i >= delay ? x[i - delay] : 0;
        prod[window - 1 + i] = conjf(xd[i]) * x[i];                /* blocks_conjugate_cc -> blocks_multiply_xx */
        pw[pwindow - 1 + i] = crealf(x[i]) * crealf(x[i]) + cimagf(x[i]) * cimagf(x[i]);   /* complex_to_mag_squared */
    }
    orc_moving_avg_work(window, 1.0f, n, n, (const float*)prod, in_abs_);
    {   /* blocks_moving_average_ff(length pwindow, scale): same running-sum loop on floats */
        float sum = pw[0];
        for (int i = 1; i < pwindow - 1; i++) sum += pw[i];
        for (int i = 0; i < n; i++) {
            sum += pw[i + pwindow - 1];
            const float p = sum * pscale;
            sum -= pw[i];
            in_cor[i] = cabsf(ia[i]) / fabsf(p);                   /* complex_to_mag / abs -> divide */
        }
    }
    free(prod); free(pw);
}

/* ---- frame_detector ---- */
struct orc_fd_state {
    int fft_len, cp_len, min_n_peaks, ignore_gap;
    double threshold, MAX_PEAK_VALUE;
    int MAX_PEAK_DISTANCE, MAX_SAMPLES;
    int state;               /* 0 SEARCH, 1 COPY */
    unsigned n_peaks;
    uint64_t first_peak_ind;
    int copied_samples;
    float coarse_cfo_est;
    uint64_t nread, nwritten;
};

orc_fd_state* orc_fd_create(int fft_len, int cp_len, double threshold, int min_n_peaks, int ignore_gap)
{
    orc_fd_state* s = (orc_fd_state*)calloc(1, sizeof(*s));
    s->fft_len = fft_len; s->cp_len = cp_len; s->threshold = threshold; s->min_n_peaks = min_n_peaks; s->ignore_gap = ignore_gap;
    s->MAX_PEAK_VALUE = 2.0; s->MAX_PEAK_DISTANCE = 2 * (fft_len + cp_len); s->MAX_SAMPLES = 540 * (fft_len + cp_len);   /* :56-58 */
    return s;
}
void orc_fd_destroy(orc_fd_state* s) { free(s); }

/* one general_work call.  tag_off / tag_cfo receive the frame_start tags added in this call (absolute output offsets).
 * Returns items produced; *consumed = items consumed on every input. */
int orc_fd_work(orc_fd_state* s, int noutput, int ninput, const float* in_, const float* in_abs_, const float* in_cor, float* out_,
                int* consumed, uint64_t* tag_off, double* tag_cfo, int max_tags, int* n_tags)
{
    const cf* in = (const cf*)in_;
    const cf* in_abs = (const cf*)in_abs_;
    cf* out = (cf*)out_;
    *n_tags = 0; *consumed = 0;
    if (s->state == 0) {                                                                   /* SEARCH (:89-134) */
        int n_in;
        for (n_in = 0; n_in < ninput; n_in++) {
            if (in_cor[n_in] > s->threshold && in_cor[n_in] < s->MAX_PEAK_VALUE) {
                if (s->n_peaks < (unsigned)s->min_n_peaks) {
                    s->n_peaks++;
                    if (s->n_peaks == 1) s->first_peak_ind = s->nread + n_in;
                } else if ((s->nread + n_in - s->first_peak_ind) < (uint64_t)s->MAX_PEAK_DISTANCE) {
                    s->state = 1;
                    s->copied_samples = 0;
                    s->coarse_cfo_est = (float)(cargf(in_abs[n_in]) / (s->fft_len / 4.0));  /* :112, stored in a float member */
                    s->n_peaks = 0;
                    s->first_peak_ind = 0;
                    if (*n_tags < max_tags) { tag_off[*n_tags] = s->nwritten; tag_cfo[*n_tags] = s->coarse_cfo_est; (*n_tags)++; }
                    break;
                } else {
                    s->n_peaks = 0;
                    s->first_peak_ind = 0;
                }
            } else if ((s->nread + n_in - s->first_peak_ind) > (uint64_t)s->MAX_PEAK_DISTANCE) {
                s->n_peaks = 0;
                s->first_peak_ind = 0;
            }
        }
        *consumed = n_in;
        s->nread += (uint64_t)n_in;
        return 0;
    }
    int n_out = 0;                                                                         /* COPY (:136-191) */
    while (n_out < ninput && n_out < noutput && s->copied_samples < s->MAX_SAMPLES) {
        if (in_cor[n_out] > s->threshold && in_cor[n_out] < s->MAX_PEAK_VALUE) {
            if (s->n_peaks < (unsigned)s->min_n_peaks) {
                s->n_peaks++;
                if (s->n_peaks == 1) s->first_peak_ind = s->nread + n_out;
            } else if ((s->nread + n_out - s->first_peak_ind) < (uint64_t)s->MAX_PEAK_DISTANCE) {
                if (s->copied_samples > s->ignore_gap) {
                    s->copied_samples = 0;
                    s->n_peaks = 0;
                    s->first_peak_ind = 0;
                    s->coarse_cfo_est = (float)(cargf(in_abs[n_out]) / (s->fft_len / 4.0));
                    if (*n_tags < max_tags) { tag_off[*n_tags] = s->nwritten + n_out; tag_cfo[*n_tags] = s->coarse_cfo_est; (*n_tags)++; }
                    break;
                }
            } else {
                s->n_peaks = 0;
                s->first_peak_ind = 0;
            }
        } else if ((s->nread + n_out - s->first_peak_ind) > (uint64_t)s->MAX_PEAK_DISTANCE) {
            s->n_peaks = 0;
            s->first_peak_ind = 0;
        }
        out[n_out] = in[n_out] * cexpf(CMPLXF(0.0f, -s->coarse_cfo_est * s->copied_samples));   /* :178 */
        n_out++;
        s->copied_samples++;
    }
    if (s->copied_samples == s->MAX_SAMPLES) s->state = 0;
    *consumed = n_out;
    s->nread += (uint64_t)n_out;
    s->nwritten += (uint64_t)n_out;
    return n_out;
}

/* ---- frame_sync ---- */
struct orc_fs_state {
    int fft_len, cp_len, SYNC_LENGTH, ntaps;
    cf* taps;
    int state;              /* 0 SYNC, 1 COPY, 2 RESET */
    int sample_offset, frame_start;
    float freq_offset;      /* frame_sync_impl.h:46 */
    double cfo_coarse_est;
    int total_out_count;
    cf* cor_val; int* cor_idx; int n_cor;
    uint64_t nread, nwritten;
};

orc_fs_state* orc_fs_create(int fft_len, int cp_len, int sync_length, const float* ltf_seq_time, int ntaps)
{
    orc_fs_state* s = (orc_fs_state*)calloc(1, sizeof(*s));
    s->fft_len = fft_len; s->cp_len = cp_len; s->SYNC_LENGTH = sync_length; s->ntaps = ntaps;
    s->taps = (cf*)malloc(sizeof(cf) * (size_t)ntaps);
    memcpy(s->taps, ltf_seq_time, sizeof(cf) * (size_t)ntaps);
    s->cor_val = (cf*)malloc(sizeof(cf) * (size_t)(sync_length + 8));
    s->cor_idx = (int*)malloc(sizeof(int) * (size_t)(sync_length + 8));
    return s;
}
void orc_fs_destroy(orc_fs_state* s) { if (s) { free(s->taps); free(s->cor_val); free(s->cor_idx); free(s); } }
int orc_fs_frame_start(const orc_fs_state* s) { return s->frame_start; }
double orc_fs_freq_offset(const orc_fs_state* s) { return s->freq_offset; }

static void fs_search_frame_start(orc_fs_state* s)                                       /* :232-287 */
{
    /* d_cor.sort(compare_abs2): stable, descending |value| -> only the first four entries are used */
    cf v[4]; int ix[4];
    char* used = (char*)calloc((size_t)s->n_cor, 1);
    for (int k = 0; k < 4; k++) {
        int best = -1; float bm = -1.0f;
        for (int i = 0; i < s->n_cor; i++) {
            if (used[i]) continue;
            const float m = cabsf(s->cor_val[i]);
            if (best < 0 || m > bm) { best = i; bm = m; }
        }
        used[best] = 1; v[k] = s->cor_val[best]; ix[k] = s->cor_idx[best];
    }
    free(used);
    s->n_cor = 0;
    s->frame_start = s->SYNC_LENGTH;                                                      /* :242 */
    for (int i = 0; i < 3; i++)
        for (int k = i + 1; k < 4; k++) {
            cf first, second;
            if (ix[i] > ix[k]) { first = v[k]; second = v[i]; } else { first = v[i]; second = v[k]; }
            const int diff = abs(ix[i] - ix[k]);
            const int mn = ix[i] < ix[k] ? ix[i] : ix[k];
            if (diff == s->fft_len) {
                s->frame_start = mn;
                s->freq_offset = cargf(first * conjf(second)) / s->fft_len;
                return;
            } else if (diff == s->fft_len - 1) {
                s->frame_start = mn;
                s->freq_offset = cargf(first * conjf(second)) / (s->fft_len - 1);
            } else if (diff == s->fft_len + 1) {
                s->frame_start = mn;
                s->freq_offset = cargf(first * conjf(second)) / (s->fft_len + 1);
            }
        }
}

/* one general_work call.  tags_in: frame_start tags on input 0 as (absolute offset, value) sorted by offset, within
 * [nread, nread + ninput).  tag_out_*: the frame_start tag this call may add.  Returns items produced. */
int orc_fs_work(orc_fs_state* s, int noutput, int ninput0, int ninput1, const float* in_, const float* in_delayed_,
                const uint64_t* tin_off, const double* tin_val, int n_tin, float* out_, int* consumed, uint64_t* tag_out_off,
                double* tag_out_val, int* n_tag_out)
{
    const cf* in = (const cf*)in_;
    const cf* in_delayed = (const cf*)in_delayed_;
    cf* out = (cf*)out_;
    int ninput = ninput0 < ninput1 ? ninput0 : ninput1;
    if (ninput > 8192) ninput = 8192;                                                     /* :111 */
    *n_tag_out = 0;
    int have = 0; uint64_t first_off = 0; double first_val = 0;
    for (int i = 0; i < n_tin; i++)
        if (tin_off[i] >= s->nread && tin_off[i] < s->nread + (uint64_t)ninput) { have = 1; first_off = tin_off[i]; first_val = tin_val[i]; break; }
    if (have) {                                                                           /* :120-146 */
        if (first_off > s->nread) {
            ninput = (int)(first_off - s->nread);
        } else {
            if (s->sample_offset && s->state == 0) return -1;                            /* runtime_error("[FRAME SYNC] Something is wrong!") */
            if (s->state == 1) s->state = 2;
            s->cfo_coarse_est = first_val;
        }
    }
    int n_in = 0, n_out = 0;
    if (s->state == 0) {                                                                  /* SYNC (:153-173) */
        int ncor = ninput - s->fft_len - 1; if (ncor < 0) ncor = 0; if (ncor > s->SYNC_LENGTH) ncor = s->SYNC_LENGTH;
        cf* corr = (cf*)calloc((size_t)s->SYNC_LENGTH + 8192, sizeof(cf));
        for (int i = 0; i < ncor; i++) {                                                  /* d_ltf_fir.filterN */
            cf acc = 0;
            for (int k = 0; k < s->ntaps; k++) acc += s->taps[k] * in[i + s->ntaps - 1 - k];
            corr[i] = acc;
        }
        while (n_in + s->fft_len - 1 < ninput) {
            s->cor_val[s->n_cor] = corr[n_in]; s->cor_idx[s->n_cor] = s->sample_offset; s->n_cor++;
            n_in++;
            s->sample_offset++;
            if (s->sample_offset == s->SYNC_LENGTH) {
                fs_search_frame_start(s);
                s->sample_offset = 0;
                s->total_out_count = 0;
                s->state = 1;
                break;
            }
        }
        free(corr);
    } else if (s->state == 1) {                                                           /* COPY (:175-202) */
        while (n_in < ninput && n_out < noutput) {
            const int rel = s->sample_offset - s->frame_start;
            if (!rel) { tag_out_off[0] = s->nwritten; tag_out_val[0] = s->cfo_coarse_est - s->freq_offset; *n_tag_out = 1; }
            if (rel >= 0 && (rel < s->fft_len * 2 || ((rel - s->fft_len * 2) % (s->fft_len + s->cp_len)) > s->cp_len - 1)) {
                out[n_out] = in_delayed[n_in] * cexpf(CMPLXF(0.0f, s->sample_offset * s->freq_offset));   /* :193, int * float */
                n_out++;
            }
            n_in++;
            s->sample_offset++;
        }
    } else {                                                                              /* RESET (:204-223) */
        while (n_out < noutput) {
            if (((s->total_out_count + n_out) % s->fft_len) == 0) {
                s->sample_offset = 0;
                s->state = 0;
                break;
            } else {
                out[n_out] = 0;
                n_out++;
            }
        }
    }
    s->total_out_count += n_out;
    *consumed = n_in;
    s->nread += (uint64_t)n_in;
    s->nwritten += (uint64_t)n_out;
    return n_out;
}
