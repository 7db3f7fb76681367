/*
 * jrc_oracle_comm.c — CPU restatement (TEST INFRASTRUCTURE) of the comm-side rows of the hot path:
 *   C1 mimo_ofdm_equalizer, C2 mimo_precoder, C3 steering, and the SIG-field codec they share.
 * PARITY UNPINNED (see jrc_oracle.h): no reference test/fixture exists and the reference cannot be built
 * here.  Third-party pieces restated from their published behaviour:
 *   - gr::digital::constellation_bpsk / _qpsk (GNU Radio 3.8.5): points and decision rules
 *   - Eigen3 JacobiSVD of a 1 x T row (unversioned find_package(Eigen3)): Householder construction, C3 below
 *   - the SIG field goes through the windowed SSE2 Viterbi of lib/viterbi_decoder.cc as restated lane by lane in
 *     jrc_oracle_codec.c (orc_viterbi_windowed), symbols read beyond the coded bits = 0; orc_viterbi_k7 (full traceback,
 *     maximum likelihood) stays as a cross-check for the tests.
 */
#include "jrc_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef float complex cf;

/* ------------------------------------------------------------------------------------------
 * ofdm_mcs / packet_param  (lib/utils.cc:26-111)
 * ---------------------------------------------------------------------------------------- */
int orc_mcs_params(int mcs, int n_data_carriers, int* n_bpsc, int* n_cbps, int* n_dbps, int* rate_field)
{
    int bpsc, num, den, rf;
    switch (mcs) {
        case 0: bpsc = 1; num = 1; den = 2; rf = 0x0D; break;   /* BPSK_1_2  */
        case 1: bpsc = 1; num = 3; den = 4; rf = 0x0F; break;   /* BPSK_3_4  */
        case 2: bpsc = 2; num = 1; den = 2; rf = 0x05; break;   /* QPSK_1_2  */
        case 3: bpsc = 2; num = 3; den = 4; rf = 0x07; break;   /* QPSK_3_4  */
        case 4: bpsc = 4; num = 1; den = 2; rf = 0x09; break;   /* QAM16_1_2 */
        case 5: bpsc = 4; num = 3; den = 4; rf = 0x0B; break;   /* QAM16_3_4 */
        default: return -1;
    }
    int cbps = n_data_carriers * bpsc;
    if (n_bpsc) *n_bpsc = bpsc;
    if (n_cbps) *n_cbps = cbps;
    if (n_dbps) *n_dbps = cbps * num / den;        /* n_cbps/2 or n_cbps*3/4, integer arithmetic (:59-95) */
    if (rate_field) *rate_field = rf;
    return 0;
}

int orc_n_ofdm_sym(int mcs, int n_data_carriers, int data_size_byte)
{
    int n_dbps;
    if (orc_mcs_params(mcs, n_data_carriers, 0, 0, &n_dbps, 0) < 0) return -1;
    return (int)ceil((16 + 8 * data_size_byte + 6) / (double)n_dbps);          /* lib/utils.cc:31 */
}

/* ------------------------------------------------------------------------------------------
 * SIG field  (generate: lib/mimo_precoder_impl.cc:985-1060; parse: lib/mimo_ofdm_equalizer_impl.cc:650-781)
 * ---------------------------------------------------------------------------------------- */
static int ones8(int n) { int s = 0; for (int i = 0; i < 8; i++) if (n & (1 << i)) s++; return s; }

/* out: n_data_carriers BPSK symbols (re only; +-1), via convolutional_encoding (lib/utils.cc:207-217) and
 * split_symbols with n_bpsc = 1; NOT interleaved. packet_type: 1 = NDP, 2 = DATA (enum), field bit 0 / 1 */
int orc_sig_encode(int n_data_carriers, int mcs, int packet_type, int length, float* out_re)
{
    int rate_field;
    if (orc_mcs_params(mcs, n_data_carriers, 0, 0, 0, &rate_field) < 0) return -1;
    int n_bits = n_data_carriers / 2;              /* signal_ofdm BPSK_1_2: n_dbps = n_cbps/2, one symbol */
    /* packet_param(signal_ofdm, 0, type): n_ofdm_sym = ceil(22/n_dbps); n_data_bits = n_ofdm_sym*n_dbps */
    int n_sym = (int)ceil(22 / (double)n_bits);
    int n_data_bits = n_sym * n_bits;
    char* hdr = (char*)calloc((size_t)(n_data_bits > 24 ? n_data_bits : 24), 1);
    hdr[0] = (rate_field >> 3) & 1; hdr[1] = (rate_field >> 2) & 1;
    hdr[2] = (rate_field >> 1) & 1; hdr[3] = rate_field & 1;
    hdr[4] = (packet_type == 2) ? 1 : 0;                                        /* packet_type_field bit 0 */
    for (int i = 0; i < 12; i++) hdr[5 + i] = (length >> i) & 1;
    int sum = 0;
    for (int i = 0; i < 17; i++) sum += hdr[i];
    hdr[17] = sum % 2;
    /* 18..23 zero */
    int state = 0;
    for (int i = 0; i < n_data_bits && 2 * i + 1 < n_data_carriers * n_sym; i++) {
        state = ((state << 1) & 0x7e) | hdr[i];
        int b0 = ones8(state & 0155) % 2, b1 = ones8(state & 0117) % 2;
        /* only the first n_data_carriers coded bits are mapped (one OFDM symbol) */
        if (2 * i < n_data_carriers) out_re[2 * i] = b0 ? 1.0f : -1.0f;           /* BPSK: 0 -> -1, 1 -> +1 */
        if (2 * i + 1 < n_data_carriers) out_re[2 * i + 1] = b1 ? 1.0f : -1.0f;
    }
    free(hdr);
    return 0;
}

/* hard-decision K=7 (0155, 0117) Viterbi, full traceback from the best end state */
void orc_viterbi_k7(const uint8_t* coded, int n_decoded, uint8_t* decoded)
{
    const int INF = 1 << 28;
    int metric[64], next[64];
    uint8_t* surv = (uint8_t*)malloc((size_t)n_decoded * 64);
    for (int s = 0; s < 64; s++) metric[s] = s ? INF : 0;
    for (int i = 0; i < n_decoded; i++) {
        int r0 = coded[2 * i], r1 = coded[2 * i + 1];
        for (int s = 0; s < 64; s++) {
            int in = s & 1, best = INF, bp = 0;
            for (int h = 0; h < 2; h++) {
                int prev = (s >> 1) | (h << 5);
                int reg = (prev << 1) | in;                   /* 7-bit encoder register */
                int e0 = ones8(reg & 0155) & 1, e1 = ones8(reg & 0117) & 1;
                int m = metric[prev] + (e0 != r0) + (e1 != r1);
                if (m < best) { best = m; bp = h; }
            }
            next[s] = best; surv[(size_t)i * 64 + s] = (uint8_t)bp;
        }
        memcpy(metric, next, sizeof(metric));
    }
    int s = 0;
    for (int k = 1; k < 64; k++) if (metric[k] < metric[s]) s = k;
    for (int i = n_decoded - 1; i >= 0; i--) {
        decoded[i] = (uint8_t)(s & 1);
        s = (s >> 1) | (surv[(size_t)i * 64 + s] << 5);
    }
    free(surv);
}

/* parse the 24 SIG bits (lib/mimo_ofdm_equalizer_impl.cc:669-781). returns 1 on success */
int orc_sig_parse(const uint8_t* bits, int n_data_carriers, int* mcs, int* packet_type, int* length, int* n_ofdm_sym)
{
    int rate_bitmap = 0, pt_bitmap = 0, len = 0, parity = 0;
    for (int i = 0; i < 17; i++) {
        parity ^= bits[i];
        if (i < 4 && bits[i]) rate_bitmap |= 1 << i;
        if (i == 4 && bits[i]) pt_bitmap |= 1;
        if (bits[i] && i > 4 && i < 17) len |= 1 << (i - 5);
    }
    int trailing_ok = 1;
    for (int i = 17; i < 23; i++) if (bits[i] != 0) trailing_ok = 0;            /* sic: 17..22 (:695-700) */
    *length = len;
    if (parity != bits[17] && trailing_ok) { *length = 0; *n_ofdm_sym = 0; return 0; }   /* :702-710 */
    *packet_type = pt_bitmap == 0 ? 1 : 2;                                       /* NDP = 1, DATA = 2 */
    switch (rate_bitmap) {
        case 11: *mcs = 0; break; case 15: *mcs = 1; break; case 10: *mcs = 2; break;
        case 14: *mcs = 3; break; case 9: *mcs = 4; break; case 13: *mcs = 5; break;
        default: return 0;
    }
    *n_ofdm_sym = orc_n_ofdm_sym(*mcs, n_data_carriers, len);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * C1  mimo_ofdm_equalizer  (lib/mimo_ofdm_equalizer_impl.cc:191-648)
 * ---------------------------------------------------------------------------------------- */
struct orc_eq_state {
    orc_eq_cfg c;
    int* data_c; int* pilot_c; int* active_c; int n_active;
    cf* pilot_sym; cf* ltf; cf* mapped; int n_tx;
    /* frame state */
    int symbol_ind, total_out, n_ofdm_symbols_SIG, sig_ok, equalize_done;
    int mcs, packet_type, data_length;
    double freq_offset, er, epsilon0, snr_est, precoded_snr_est;
    double signal_power_sum, noise_power_sum; int snr_est_count;
    cf* H; cf* H_mimo; cf* pre;            /* pre[sc][ltf] kept across calls (reference bug: stack VLA) */
    cf chan_mean[16]; int n_chan_mean;
};

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

orc_eq_state* orc_eq_create(const orc_eq_cfg* c)
{
    orc_eq_state* s = (orc_eq_state*)calloc(1, sizeof(*s));
    s->c = *c;
    const int N = c->fft_len;
    s->data_c = (int*)malloc(sizeof(int) * c->n_data);
    s->pilot_c = (int*)malloc(sizeof(int) * c->n_pilot);
    for (int i = 0; i < c->n_pilot; i++) s->pilot_c[i] = c->pilot_carriers[i] + N / 2;    /* :134-137 */
    for (int i = 0; i < c->n_data; i++) s->data_c[i] = c->data_carriers[i] + N / 2;       /* :139-142 */
    s->n_active = c->n_data + c->n_pilot;
    s->active_c = (int*)malloc(sizeof(int) * s->n_active);
    memcpy(s->active_c, s->data_c, sizeof(int) * c->n_data);
    memcpy(s->active_c + c->n_data, s->pilot_c, sizeof(int) * c->n_pilot);
    qsort(s->active_c, s->n_active, sizeof(int), cmp_int);                               /* :159 */
    size_t np = (size_t)c->n_pilot_rows * c->n_pilot;
    s->pilot_sym = (cf*)malloc(sizeof(cf) * np); memcpy(s->pilot_sym, c->pilot_symbols, sizeof(cf) * np);
    s->ltf = (cf*)malloc(sizeof(cf) * N); memcpy(s->ltf, c->ltf_seq, sizeof(cf) * N);
    size_t nm = (size_t)N * c->mapped_cols;
    s->mapped = (cf*)malloc(sizeof(cf) * nm); memcpy(s->mapped, c->mapped_ltf, sizeof(cf) * nm);
    s->n_tx = c->mapped_cols / c->n_mimo_ltf;                                            /* :168 */
    s->H = (cf*)calloc(N, sizeof(cf)); s->H_mimo = (cf*)calloc(N, sizeof(cf));
    s->pre = (cf*)calloc((size_t)N * c->n_mimo_ltf, sizeof(cf));
    s->sig_ok = 0; s->n_ofdm_symbols_SIG = 0; s->symbol_ind = 0;
    /* before the first frame_start the reference's members are uninitialised; a fresh block that sees no
     * tag skips everything, which is what sig_ok = 0 gives */
    return s;
}

void orc_eq_destroy(orc_eq_state* s)
{
    if (!s) return;
    free(s->data_c); free(s->pilot_c); free(s->active_c); free(s->pilot_sym); free(s->ltf); free(s->mapped);
    free(s->H); free(s->H_mimo); free(s->pre); free(s);
}

void orc_eq_set_estimator(orc_eq_state* s, int algo) { s->c.estimator = algo; }

static cf cexp_f(double x) { float xf = (float)x; return cosf(xf) + I * sinf(xf); }   /* std::exp(gr_complex(0, x)) */

/* std::complex<float> operator/ is libgcc's __divsc3.  A g++ build of the reference takes it from libgcc_s.so.1 (g++ links -lgcc_s
 * ahead of -lgcc; this image: libgcc-s1 12.3), whose float version evaluates the textbook quotient in double and rounds once
 * (libgcc2.c since GCC 12: "float is handled with double precision").  gcc links C programs against the static libgcc.a of the
 * compiler (11.4 here), which still has Smith's method in float: `a / b` in this file would differ from the reference in the last
 * bits.  Written out so the oracle does not depend on the linker's choice; tests/test_second_source.py pins it, bit for bit,
 * against the __divsc3 exported by the box's libgcc_s.so.1. */
static cf c_div(cf x, cf y)
{
    const double aa = crealf(x), bb = cimagf(x), cc = crealf(y), dd = cimagf(y);
    const double denom = (cc * cc) + (dd * dd);
    return CMPLXF((float)(((aa * cc) + (bb * dd)) / denom), (float)(((bb * cc) - (aa * dd)) / denom));
}
void orc_cdiv(const float* a, const float* b, float* q) { cf r = c_div(CMPLXF(a[0], a[1]), CMPLXF(b[0], b[1])); q[0] = crealf(r); q[1] = cimagf(r); }

/* gr::digital constellation decisions (GNU Radio 3.8.5) */
static cf demod_point(int bps, cf z)
{
    if (bps == 1) return crealf(z) > 0 ? 1.0f : -1.0f;
    if (bps == 4) {
        /* constellation_16qam (gr-digital 3.8, NOT in the tree: recollection, PARITY UNPINNED — same table as oracle/jrc_oracle_codec.c):
         * decision bits re > 0, |re| < 2 level, im > 0, |im| < 2 level; the point of that value is (+-1|3, +-1|3) * level, level = sqrt(0.1).
         * The reference does not scale it (bits_per_symbol() != qpsk's, :511, :567). */
        const float level = sqrtf(0.1f);
        const float re = crealf(z), im = cimagf(z);
        const float a = (fabsf(re) < 2 * level) ? 1.0f : 3.0f, b = (fabsf(im) < 2 * level) ? 1.0f : 3.0f;
        return ((re > 0 ? a : -a) * level) + I * ((im > 0 ? b : -b) * level);
    }
    /* QPSK: index 2*(im>0)+(re>0) -> (+-0.707107, +-0.707107); the reference then divides by 2 (:511-514) */
    const float a = 0.707107f;
    cf p = (crealf(z) > 0 ? a : -a) + I * (cimagf(z) > 0 ? a : -a);
    return p / 2.0f;
}

static double residual_cfo(const orc_eq_state* s, const cf* Y, const cf* chan, const cf* ref, cf* est)
{
    cf sum = 0;                                                                     /* :908-922 */
    for (int k = 0; k < s->c.n_pilot; k++) {
        est[k] = chan[s->pilot_c[k]] * ref[k];
        sum += Y[s->pilot_c[k]] * conjf(est[k]);
    }
    return cargf(sum);
}

int orc_eq_work(orc_eq_state* s, int noutput_items, int ninput_items, const float* in_f,
                const long* tag_offsets, const double* tag_values, int n_tags,
                float* out_f, int* n_consumed, orc_eq_event* events, int max_events, int* n_events,
                float* chan_est /* [N][n_tx] complex, written at an NDP channel estimate */, int* chan_est_written)
{
    const int N = s->c.fft_len, ND = s->c.n_data, NP = s->c.n_pilot, NL = s->c.n_mimo_ltf;
    const cf* in = (const cf*)in_f; cf* out = (cf*)out_f;
    int n_in = 0, n_out = 0, nev = 0;
    cf* Y = (cf*)malloc(sizeof(cf) * N);
    cf* Z = (cf*)malloc(sizeof(cf) * ND);
    cf* est = (cf*)malloc(sizeof(cf) * NP);
    if (chan_est_written) *chan_est_written = 0;

    while (n_in < ninput_items && n_out < noutput_items) {                                   /* :219 */
        for (int t = 0; t < n_tags; t++)
            if (tag_offsets[t] == n_in) {                                                    /* :221-245 */
                s->symbol_ind = 0; s->total_out = 0; s->n_ofdm_symbols_SIG = 0;
                s->freq_offset = tag_values[t] * s->c.bw / (2 * M_PI);
                s->epsilon0 = tag_values[t] * s->c.bw / (2 * M_PI * s->c.freq);
                s->er = 0; s->sig_ok = 1; s->equalize_done = 0;
                s->signal_power_sum = 0; s->noise_power_sum = 0; s->snr_est_count = 0;
                break;
            }
        if (s->symbol_ind > s->n_ofdm_symbols_SIG + 2 + NL || !s->sig_ok) { n_in++; continue; }   /* :250-255 */

        for (int i = 0; i < N; i++)                                                          /* :261-264 */
            Y[i] = in[(size_t)n_in * N + i] *
                   cexp_f(2 * M_PI * s->symbol_ind * ((N + s->c.cp_len) * 1.0 / N) * (s->epsilon0 + s->er) * (i - N / 2));

        if (s->symbol_ind == 0) {                                                            /* :272-275 */
            memcpy(s->H, Y, sizeof(cf) * N);
        } else if (s->symbol_ind == 1) {                                                     /* :277-306 */
            double signal = 0, noise = 0;
            for (int k = 0; k < s->n_active; k++) {
                int c = s->active_c[k];
                noise += pow((double)cabsf(s->H[c] - Y[c]), 2.0);
                signal += pow((double)cabsf(s->H[c] + Y[c]), 2.0);
                s->H[c] += Y[c];
                s->H[c] = c_div(s->H[c], s->ltf[c] * (cf)(2.0f));
            }
            /* the CPE computed and applied to Y here (:288-303) is discarded with Y */
            s->snr_est = 10 * log10(signal / noise / 2);
        } else if (s->symbol_ind == 2) {                                                     /* :308-344 */
            double cfo = residual_cfo(s, Y, s->H, s->pilot_sym, est);
            cf rot = cexp_f(-cfo);
            for (int i = 0; i < N; i++) Y[i] *= rot;
            for (int i = 0; i < ND; i++) Z[i] = c_div(Y[s->data_c[i]], s->H[s->data_c[i]]);   /* symbol_equalize */
            /* decode_signal_field (:650-667): BPSK decisions -> d_decoder.decode(ofdm_mcs(BPSK_1_2, ND), packet_param(., 0, NDP), rx_bits):
             * the windowed decoder of lib/viterbi_decoder.cc (oracle/jrc_oracle_codec.c), which runs until n_ofdm_sym * ND/2 bits are out
             * and so reads 5 traceback chunks past the ND coded bits; rx_bits is calloc(ND) (:175), what lies behind it is undefined
             * in the reference and 0 here */
            const int sig_dbps = ND / 2, sig_nsym = (int)ceil(22 / (double)sig_dbps);
            uint8_t* bits = (uint8_t*)calloc((size_t)sig_nsym * ND + 8, 1); uint8_t* dec = (uint8_t*)calloc((size_t)sig_nsym * sig_dbps + 64, 1);
            for (int i = 0; i < ND; i++) bits[i] = crealf(Z[i]) > 0;
            orc_viterbi_windowed(0, sig_nsym, ND, sig_nsym * sig_dbps, bits, dec);
            s->sig_ok = orc_sig_parse(dec, ND, &s->mcs, &s->packet_type, &s->data_length, &s->n_ofdm_symbols_SIG);
            free(bits); free(dec);
            if (s->sig_ok && nev < max_events) {
                orc_eq_event* e = &events[nev++];
                memset(e, 0, sizeof(*e));
                e->kind = 1; e->offset = n_out; e->data_bytes = s->data_length; e->mcs = s->mcs;
                e->packet_type = s->packet_type; e->snr = s->snr_est; e->freq_offset = s->freq_offset;
            }
        } else if (s->symbol_ind >= 3 && s->symbol_ind <= 2 + NL) {                          /* :346-463 */
            int l = s->symbol_ind - 3;
            for (int i = 0; i < N; i++) s->pre[(size_t)i * NL + l] = Y[i];
            if (l == NL - 1) {
                const int T = s->n_tx;
                if (s->packet_type == 1) {                                                   /* NDP :375-422 */
                    cf mean[16]; for (int t = 0; t < T; t++) mean[t] = 0;
                    for (int sc = 0; sc < N; sc++) {
                        for (int t = 0; t < T; t++) {                                        /* H = conj(X_ltf) * y */
                            cf h = 0;
                            for (int q = 0; q < NL; q++)
                                h += conjf(s->mapped[(size_t)sc * s->c.mapped_cols + t * NL + q]) * s->pre[(size_t)sc * NL + q];
                            if (chan_est) ((cf*)chan_est)[(size_t)sc * T + t] = h;
                            int act = 0;
                            for (int k = 0; k < s->n_active; k++) if (s->active_c[k] == sc) act = 1;
                            if (act) mean[t] += h;
                        }
                    }
                    for (int t = 0; t < T; t++) s->chan_mean[t] = mean[t] / (float)s->n_active;
                    s->n_chan_mean = T;
                    if (chan_est_written) *chan_est_written = 1;
                } else if (s->packet_type == 2) {                                            /* DATA :423-456 */
                    cf mean = 0;
                    for (int pass = 0; pass < 2; pass++) {
                        int cnt = pass ? NP : ND; const int* list = pass ? s->pilot_c : s->data_c;
                        for (int k = 0; k < cnt; k++) {
                            int sc = list[k]; cf d = 0;                                      /* row(0).dot(y): conj left */
                            for (int q = 0; q < NL; q++)
                                d += conjf(s->mapped[(size_t)sc * s->c.mapped_cols + q]) * s->pre[(size_t)sc * NL + q];
                            s->H_mimo[sc] = d / (float)NL;
                            mean += s->H_mimo[sc];
                        }
                    }
                    s->chan_mean[0] = mean / (float)s->n_active; s->n_chan_mean = 1;
                }
            }
        } else {                                                                             /* data :465-605 */
            int row = (s->symbol_ind - 3 - NL) % s->c.n_pilot_rows;
            const cf* ref = s->pilot_sym + (size_t)row * NP;
            cf* Hsel = s->packet_type == 1 ? s->H : s->H_mimo;
            double cfo = residual_cfo(s, Y, Hsel, ref, est);
            cf rot = cexp_f(-cfo);
            for (int i = 0; i < N; i++) Y[i] *= rot;
            for (int k = 0; k < NP; k++) {                                                   /* :484-493 */
                s->signal_power_sum += crealf(est[k] * conjf(est[k]));
                cf err = est[k] - Y[s->pilot_c[k]];
                s->noise_power_sum += crealf(err * conjf(err));
                s->snr_est_count++;
            }
            int bps = (s->mcs <= 1) ? 1 : (s->mcs <= 3 ? 2 : 4);
            if (s->packet_type == 1) {
                for (int i = 0; i < ND; i++) Z[i] = c_div(Y[s->data_c[i]], s->H[s->data_c[i]]);
                if (s->c.estimator == 1) {                                                   /* STA :498-535 */
                    const float alpha = 0.5f;
                    for (int i = 0; i < ND; i++) {
                        int sc = s->data_c[i];
                        cf X = demod_point(bps, Z[i]);
                        cf upd = c_div(Y[sc], X);
                        s->H[sc] = (cf)(1 - alpha) * s->H[sc] + (cf)(alpha) * upd;
                    }
                    for (int k = 0; k < NP; k++) {
                        int sc = s->pilot_c[k];
                        s->H[sc] = (cf)(1 - alpha) * s->H[sc] + c_div((cf)(alpha) * Y[sc], ref[k]);
                    }
                }
            } else if (s->packet_type == 2) {
                for (int i = 0; i < ND; i++) {                                               /* :540-550 */
                    int sc = s->data_c[i];
                    float csi = (float)((double)crealf(s->H_mimo[sc] * conjf(s->H_mimo[sc])) + s->noise_power_sum / s->snr_est_count);   /* float + double, rounded once */
                    Z[i] = Y[sc] * conjf(s->H_mimo[sc]) / csi;
                }
                if (s->c.estimator == 1) {                                                   /* STA :552-592 */
                    const float alpha = 0.4f;
                    for (int i = 0; i < ND; i++) {
                        int sc = s->data_c[i];
                        cf X = demod_point(bps, Z[i]);
                        s->H_mimo[sc] = (cf)(1 - alpha) * s->H_mimo[sc] + c_div((cf)(alpha) * Y[sc], X);
                    }
                    for (int k = 0; k < NP; k++) {
                        int sc = s->pilot_c[k];
                        s->H_mimo[sc] = (cf)(1 - alpha) * s->H_mimo[sc] + c_div((cf)(alpha) * Y[sc], ref[k]);
                    }
                }
            }
            memcpy(out + (size_t)n_out * ND, Z, sizeof(cf) * ND);                            /* :602 */
            n_out++;
        }
        n_in++;
        s->symbol_ind++;
    }
    s->total_out += n_out;
    if (s->total_out == s->n_ofdm_symbols_SIG && s->sig_ok && !s->equalize_done) {           /* :611-632 */
        if (s->snr_est_count != 0)
            s->precoded_snr_est = 10 * log10((s->signal_power_sum / s->snr_est_count) / (s->noise_power_sum / s->snr_est_count));
        if (nev < max_events) {
            orc_eq_event* e = &events[nev++];
            memset(e, 0, sizeof(*e));
            e->kind = 2; e->offset = n_out - 1; e->snr_data = s->precoded_snr_est; e->n_chan_mean = s->n_chan_mean;
            for (int t = 0; t < s->n_chan_mean; t++) { e->chan_mean[2 * t] = crealf(s->chan_mean[t]); e->chan_mean[2 * t + 1] = cimagf(s->chan_mean[t]); }
        }
        s->equalize_done = 1;
    }
    *n_consumed = n_in; *n_events = nev;
    free(Y); free(Z); free(est);
    return n_out;
}

/* ------------------------------------------------------------------------------------------
 * C3  steering matrices  (lib/mimo_precoder_impl.cc:846-861, :880-893, :961-974)
 *   phased: Q[:,0] = conj(h) * sqrt(T)/||conj(h)||, other columns 0
 *   SVD   : Q = V * sqrt(T)/||V||_F, V = full right-singular basis of the 1 x T row h^T.
 * Eigen's JacobiSVD of a 1 x T input preconditions with ColPivHouseholderQR of the T x 1 adjoint
 * x = conj(h): one Householder reflector H = I - tau v v^H with beta = -sign(Re x0) ||x||,
 * v = [1; x1/(x0-beta) ...], tau = conj((beta - x0)/beta); matrixV = householderQ = I - conj(tau)... applied as
 * H^H; first column = x/beta up to the reflector's sign convention.  (version unpinned.)
 * out: column-major T x T (Eigen default), as steering_matrix[sc] stores Q.data().
 * ---------------------------------------------------------------------------------------- */
void orc_steering_from_channel(int T, const float* h_f, int phased, float* Q_f)
{
    const cf* h = (const cf*)h_f; cf* Q = (cf*)Q_f;
    memset(Q, 0, sizeof(cf) * T * T);
    if (phased) {
        float nrm = 0;
        for (int t = 0; t < T; t++) nrm += crealf(h[t] * conjf(h[t]));
        nrm = sqrtf(nrm);
        for (int t = 0; t < T; t++) Q[t] = conjf(h[t]) * sqrtf((float)T) / nrm;     /* column 0 */
        /* Q = Q * sqrt(T) / Q.norm() runs over the whole matrix (:851): the zero columns stay zero unless the row is all zero, where 0 / 0 makes every entry NaN */
        for (int k = T; k < T * T; k++) Q[k] = (0.0f / nrm) + (0.0f / nrm) * I;
        return;
    }
    cf x[16];
    for (int t = 0; t < T; t++) x[t] = conjf(h[t]);
    float tail = 0;
    for (int t = 1; t < T; t++) tail += crealf(x[t] * conjf(x[t]));
    cf c0 = x[0];
    cf tau; float beta; cf v[16];
    v[0] = 1;
    if (tail <= 1.17549435e-38f && cimagf(c0) * cimagf(c0) <= 1.17549435e-38f) {   /* makeHouseholder degenerate case */
        tau = 0; beta = crealf(c0);
        for (int t = 1; t < T; t++) v[t] = 0;
    } else {
        beta = sqrtf(crealf(c0 * conjf(c0)) + tail);
        if (crealf(c0) >= 0) beta = -beta;
        for (int t = 1; t < T; t++) v[t] = c_div(x[t], c0 - beta);
        tau = conjf((beta - c0) / beta);   /* complex / real */
    }
    /* V = H^H = I - conj(tau) v v^H */
    for (int col = 0; col < T; col++)
        for (int rw = 0; rw < T; rw++) {
            cf e = (rw == col) ? 1.0f : 0.0f;
            Q[(size_t)col * T + rw] = e - conjf(tau) * v[rw] * conjf(v[col]);
        }
    /* normalise: V*sqrt(T)/||V||_F; ||V||_F = sqrt(T) for a unitary V up to rounding (computed, not assumed) */
    float fro = 0;
    for (int i = 0; i < T * T; i++) fro += crealf(Q[i] * conjf(Q[i]));
    fro = sqrtf(fro);
    for (int i = 0; i < T * T; i++) Q[i] = Q[i] * sqrtf((float)T) / fro;
}

/* get_dft_matrix_eigen (lib/mimo_precoder_impl.cc:761-772), column-major T x T */
void orc_dft_matrix(int T, float* F_f)
{
    cf* F = (cf*)F_f;
    for (int r = 0; r < T; r++)
        for (int c = 0; c < T; c++) {
            float ang = (float)(-2 * M_PI * (float)(r * c) / (float)T);
            F[(size_t)c * T + r] = c_div(cosf(ang) + I * sinf(ang), (cf)sqrt((double)T));
        }
}

/* ------------------------------------------------------------------------------------------
 * C2  mimo_precoder::work  (lib/mimo_precoder_impl.cc:275-741), deterministic sub-paths:
 *   radar streams (std::random_device) are an INPUT here: radar_streams[(T-1)][n_sym][N] or NULL.
 *   steering: mode 0 = DFT (fourier) precoding, 1 = one matrix Q_mean for all subcarriers
 *   (smoothing / radar-aided), 2 = per-subcarrier Q[sc].  Q matrices column-major T x T.
 * out[t]: [4+1+T+n_sym][N].  returns produced items or -1 (n_ofdm_sym mismatch, :327-333)
 * ---------------------------------------------------------------------------------------- */
int orc_precoder_work(const orc_pre_cfg* c, int ninput_items, const float* in_f, int mcs, int packet_type,
                      int pdu_len, int steer_mode, const float* Q_mean_f, const float* Q_sc_f,
                      const float* radar_streams_f, float* const* out_f)
{
    const int N = c->fft_len, T = c->n_tx, ND = c->n_data, NP = c->n_pilot, NS = c->n_sync, NL = T;
    const cf* in = (const cf*)in_f;
    int n_sym = ninput_items / ND;
    if (orc_n_ofdm_sym(mcs, ND, pdu_len) != n_sym) return -1;
    int n_total = n_sym + NS + T + 1;
    int* dc = (int*)malloc(sizeof(int) * ND); int* pc = (int*)malloc(sizeof(int) * NP);
    for (int i = 0; i < ND; i++) { int v = c->data_carriers[i]; if (v < 0) v += N; dc[i] = (v + N / 2) % N; }   /* :126-137 */
    for (int i = 0; i < NP; i++) { int v = c->pilot_carriers[i]; if (v < 0) v += N; pc[i] = (v + N / 2) % N; } /* :144-153 */
    const cf* sync = (const cf*)c->sync_words; const cf* pil = (const cf*)c->pilot_symbols;
    const cf* mapped = (const cf*)c->mapped_ltf;
    float* sig = (float*)calloc(ND, sizeof(float));
    orc_sig_encode(ND, mcs, packet_type, pdu_len, sig);
    for (int t = 0; t < T; t++) {
        cf* o = (cf*)out_f[t];
        memset(o, 0, sizeof(cf) * (size_t)N * n_total);                                      /* :337 */
        if (t < 2) {
            for (int k = 0; k < NS; k++) memcpy(o + (size_t)k * N, sync + (size_t)k * N, sizeof(cf) * N);   /* :340-347 */
            cf* s = o + (size_t)NS * N;
            for (int i = 0; i < ND; i++) s[dc[i]] = sig[i];                                  /* :356-366 */
            for (int k = 0; k < NP; k++) s[pc[k]] = pil[k];
        }
    }
    const int base_ltf = NS + 1, base_data = NS + 1 + T;
    if (packet_type == 1) {                                                                  /* NDP :374-429 */
        for (int t = 0; t < T; t++) {
            cf* o = (cf*)out_f[t];
            for (int sc = 0; sc < N; sc++)
                for (int l = 0; l < NL; l++) o[(size_t)(base_ltf + l) * N + sc] = mapped[(size_t)sc * T * NL + l + t * NL];
            if (t < 2)
                for (int m = 0; m < n_sym; m++) {
                    for (int i = 0; i < ND; i++) o[(size_t)(base_data + m) * N + dc[i]] = in[(size_t)m * ND + i];
                    for (int k = 0; k < NP; k++) o[(size_t)(base_data + m) * N + pc[k]] = pil[(size_t)(m % c->n_pilot_rows) * NP + k];
                }
        }
    } else {                                                                                 /* DATA :431-712 */
        cf F[256];
        if (steer_mode == 0) orc_dft_matrix(T, (float*)F);
        const cf* Qm = steer_mode == 0 ? F : (const cf*)Q_mean_f;
        const cf* rs = (const cf*)radar_streams_f;
        const int n_streams = rs ? T : 1;
        for (int sc = 0; sc < N; sc++) {                                                     /* LTF precoding :536-581 */
            const cf* X = mapped + (size_t)sc * T * NL;          /* row-major T x NL */
            int zero = 1;
            for (int i = 0; i < T * NL; i++) if (X[i] != 0) zero = 0;
            const cf* Q = (steer_mode == 2) ? (const cf*)Q_sc_f + (size_t)sc * T * T : Qm;
            for (int t = 0; t < T; t++)
                for (int l = 0; l < NL; l++) {
                    cf acc = 0;
                    if (!zero) for (int j = 0; j < T; j++) acc += Q[(size_t)j * T + t] * X[(size_t)j * NL + l];
                    ((cf*)out_f[t])[(size_t)(base_ltf + l) * N + sc] = acc;
                }
        }
        for (int m = 0; m < n_sym; m++) {                                                    /* data + pilots :589-712 */
            for (int pass = 0; pass < 2; pass++) {
                int cnt = pass ? NP : ND;
                for (int k = 0; k < cnt; k++) {
                    int sc = pass ? pc[k] : dc[k];
                    cf s0 = pass ? pil[(size_t)(m % c->n_pilot_rows) * NP + k] : in[(size_t)m * ND + k];
                    const cf* Q = (steer_mode == 2) ? (const cf*)Q_sc_f + (size_t)sc * T * T : Qm;
                    for (int t = 0; t < T; t++) {
                        cf acc = Q[t] * s0;                                                   /* column 0 */
                        for (int j = 1; j < n_streams; j++)
                            acc += Q[(size_t)j * T + t] * rs[((size_t)(j - 1) * n_sym + m) * N + sc];
                        ((cf*)out_f[t])[(size_t)(base_data + m) * N + sc] = acc;
                    }
                }
            }
        }
    }
    free(sig); free(dc); free(pc);
    return n_total;
}


/* ------------------------------------------------------------------------------------------
 * ofdm_frame_generator_impl::work (lib/ofdm_frame_generator_impl.cc:155-216; the SISO carrier allocator): occupied / pilot
 * carrier sets are flattened with their sizes; indices already normalised (negative + fft_len, shifted) as the constructor
 * does (:83-113).  Returns the number of OFDM symbols written incl. sync words.
 * ---------------------------------------------------------------------------------------- */
int orc_frame_generator(int fft_len, int n_occ_sets, const int* occ_sizes, const int* occ_flat, int n_pil_sets, const int* pil_sizes,
                        const int* pil_flat, int n_psym_sets, const float* psym_flat, int n_sync, const float* sync_words, int n_in,
                        const float* in, float* out, int noutput_items)
{
    cf* o = (cf*)out;
    memset(o, 0, sizeof(cf) * (size_t)fft_len * noutput_items);                         /* :165 */
    for (int i = 0; i < n_sync; i++) memcpy(o + (size_t)i * fft_len, (const cf*)sync_words + (size_t)i * fft_len, sizeof(cf) * fft_len);
    o += (size_t)n_sync * fft_len;
    int* occ_off = (int*)malloc(sizeof(int) * (n_occ_sets + 1));
    occ_off[0] = 0;
    for (int k = 0; k < n_occ_sets; k++) occ_off[k + 1] = occ_off[k] + occ_sizes[k];
    long n_ofdm = 0;
    int curr_set = 0, to_alloc = occ_sizes[0], allocated = 0;
    for (int i = 0; i < n_in; i++) {                                                    /* :175-200 */
        if (allocated == 0) n_ofdm++;
        o[(n_ofdm - 1) * fft_len + occ_flat[occ_off[curr_set] + allocated]] = ((const cf*)in)[i];
        allocated++;
        if (allocated == to_alloc) { curr_set = (curr_set + 1) % n_occ_sets; to_alloc = occ_sizes[curr_set]; allocated = 0; }
    }
    int* pil_off = (int*)malloc(sizeof(int) * (n_pil_sets + 1));
    pil_off[0] = 0;
    for (int k = 0; k < n_pil_sets; k++) pil_off[k + 1] = pil_off[k] + pil_sizes[k];
    /* pilot symbol sets have the sizes of the pilot carrier sets they pair with (:119-125); offsets by cumulative size */
    int* ps_off = (int*)malloc(sizeof(int) * (n_psym_sets + 1));
    ps_off[0] = 0;
    for (int k = 0; k < n_psym_sets; k++) ps_off[k + 1] = ps_off[k] + pil_sizes[k % n_pil_sets];
    for (long i = 0; i < n_ofdm; i++) {                                                 /* :202-208 */
        const int pk = (int)(i % n_pil_sets), sk = (int)(i % n_psym_sets);
        for (int k = 0; k < pil_sizes[pk]; k++) o[i * fft_len + pil_flat[pil_off[pk] + k]] = ((const cf*)psym_flat)[ps_off[sk] + k];
    }
    free(occ_off); free(pil_off); free(ps_off);
    return (int)n_ofdm + n_sync;
}
