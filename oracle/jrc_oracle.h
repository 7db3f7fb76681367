/*
 * jrc_oracle.h — CPU restatement (TEST INFRASTRUCTURE, not product) of the
 * MIMO-OFDM radar / equalizer / precoder hot path of gr-mimo-ofdm-jrc.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures,
 * and its sources cannot be built in this image (they need GNU Radio 3.8,
 * Eigen3, Boost, FFTW3f and VOLK headers, none of which exist here; writing
 * stand-ins for them is not a reference build).  Every function below is a
 * line-by-line restatement of the cited reference code in plain C with the
 * same float/double evaluation order, checked against closed forms and the
 * constant tables minted from the reference's flowgraph-embedded Python
 * module (tests/golden/ofdm_config_64.npz).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The shipped path is gr-mimo-ofdm-jrc_amd/csrc.
 *
 * All complex buffers are interleaved float pairs (gr_complex layout).
 * Reference citations are relative to /root/reference.
 */
#ifndef JRC_ORACLE_H
#define JRC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A1: mimo_ofdm_radar_impl::general_work (lib/mimo_ofdm_radar_impl.cc:131-340) ---- */
typedef struct orc_radar_state orc_radar_state;
orc_radar_state* orc_radar_create(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre,
                                  int background_removal, int background_recording,
                                  int record_len, int interp_factor, int enable_tx_interleave);
void orc_radar_destroy(orc_radar_state* st);
void orc_radar_set_background_record(orc_radar_state* st, int on);
int  orc_radar_ring_size(const orc_radar_state* st);
/* tx[t], rx[r]: port buffers exactly as GNU Radio hands them (item 0 = first
 * OFDM symbol of the tagged packet); tx_discard = items skipped on the TX
 * ports (lib/mimo_ofdm_radar_impl.cc:191-198).  out: P x (N*Ir). */
void orc_radar_work(orc_radar_state* st, const float* const* tx, const float* const* rx,
                    long tx_discard, float* out);

/* ---- stock gr::fft::fft_vcc semantics (GNU Radio 3.8 gr-fft, not in tree) ----
 * forward: out = [fftshift](FFT(in * window)); reverse: out = IFFT_unnorm([ifftshift](in*window)).
 * window may be NULL.  Computed in double, rounded once to float. */
void orc_fft_vcc(int n, int forward, int shift, const float* window, long batch,
                 const float* in, float* out);

/* ---- A3: matrix_transpose_impl::work (lib/matrix_transpose_impl.cc:69-110) ----
 * returns input_len (items produced) or -1 for the runtime_error at :82-83 */
int orc_matrix_transpose(int input_len, int output_len, int interp_factor, int ninput_items,
                         const float* in, float* out);

/* ---- A5: range_angle_estimator_impl::work (lib/range_angle_estimator_impl.cc:121-284) ---- */
typedef struct {
    int   peak_range_idx, peak_angle_idx;
    int   angle_null_idx;
    int   discard_range_idx, discard_angle_idx;
    int   n_noise_samples;
    float peak_power, noise_power, snr_est;
    float range_val, angle_val;
    int   published; /* snr_est >= snr_threshold && peak_power >= power_threshold (:234) */
} orc_ra_result;
void orc_ra_estimate(int vlen, int n_inputs, const float* in,
                     const float* range_bins, int n_range_bins,
                     const float* angle_bins, int n_angle_bins,
                     float noise_discard_range_m, float noise_discard_angle_deg,
                     float snr_threshold, float power_threshold, orc_ra_result* res);

/* ---- A6: ofdm_cyclic_prefix_remover_impl::work (lib/ofdm_cyclic_prefix_remover_impl.cc:69-99) ----
 * returns noutput_items */
int orc_cp_remove(int fft_len, int cp_len, long ninput_items, const float* in, float* out);

/* ---- B1: fft_peak_detect_impl::work (lib/fft_peak_detect_impl.cc:77-111) ----
 * returns k (or -1: nothing above threshold, outputs untouched as in the reference) */
int orc_fft_peak_detect(int samp_rate, float interp_factor, float threshold, int samp_protect,
                        long ninput_items, const float* in,
                        float* out_freq, float* out_phase, float* out_mag);

/* ---- composed radar chain A1 -> A2 -> A3 -> A4 (-> A5) for one frame, used as the CPU baseline ----
 * Hpad/range/tr are caller-provided scratch (P*N*Ir, P*N*Ir, N*Ir*P*Ia floats*2); map = N*Ir x P*Ia */
void orc_radar_chain(orc_radar_state* st, const float* const* tx, const float* const* rx,
                     int interp_angle, float* Hpad, float* range, float* tr, float* map);

/* single-precision radix-2 FFT (same semantics as orc_fft_vcc, float arithmetic):
 * only used to TIME a CPU baseline with the arithmetic type the reference (FFTW3f) uses. */
void orc_fft_vcc_f32(int n, int forward, int shift, long batch, const float* in, float* out);

/* ---- SIG field codec shared by C1/C2 (lib/utils.cc:26-111, :207-217; lib/mimo_precoder_impl.cc:985-1060;
 *      lib/mimo_ofdm_equalizer_impl.cc:650-781) ---- */
int  orc_mcs_params(int mcs, int n_data_carriers, int* n_bpsc, int* n_cbps, int* n_dbps, int* rate_field);
int  orc_n_ofdm_sym(int mcs, int n_data_carriers, int data_size_byte);
int  orc_sig_encode(int n_data_carriers, int mcs, int packet_type, int length, float* out_re);
void orc_viterbi_k7(const uint8_t* coded, int n_decoded, uint8_t* decoded);
void orc_cdiv(const float* a, const float* b, float* q);   /* std::complex<float> operator/ as a g++ build on this image evaluates it (libgcc_s 12: double, rounded once) */
int  orc_sig_parse(const uint8_t* bits, int n_data_carriers, int* mcs, int* packet_type, int* length, int* n_ofdm_sym);

/* ---- C1: mimo_ofdm_equalizer_impl::general_work (lib/mimo_ofdm_equalizer_impl.cc:191-648) ---- */
typedef struct {
    int    estimator;              /* 0 = LS, 1 = STA */
    double freq, bw;
    int    fft_len, cp_len;
    int    n_data, n_pilot;
    const int* data_carriers;      /* signed, as passed to make() */
    const int* pilot_carriers;
    int    n_pilot_rows;
    const float* pilot_symbols;    /* complex [n_pilot_rows][n_pilot] */
    const float* ltf_seq;          /* complex [fft_len] */
    const float* mapped_ltf;       /* complex [fft_len][mapped_cols], mapped_cols = N_tx*n_mimo_ltf */
    int    mapped_cols;
    int    n_mimo_ltf;
} orc_eq_cfg;
typedef struct {
    int      kind;                 /* 1 = stream_start (:331-337), 2 = stream_end (:626-629) */
    long     offset;               /* output item index relative to this call (nitems_written + offset) */
    uint64_t data_bytes, mcs, packet_type;
    double   snr, freq_offset;     /* stream_start */
    double   snr_data;             /* stream_end */
    int      n_chan_mean;
    float    chan_mean[32];        /* complex, stream_end */
} orc_eq_event;
typedef struct orc_eq_state orc_eq_state;
orc_eq_state* orc_eq_create(const orc_eq_cfg* cfg);
void orc_eq_destroy(orc_eq_state* st);
void orc_eq_set_estimator(orc_eq_state* st, int algo);
int  orc_eq_work(orc_eq_state* st, int noutput_items, int ninput_items, const float* in,
                 const long* tag_offsets, const double* tag_values, int n_tags,
                 float* out, int* n_consumed, orc_eq_event* events, int max_events, int* n_events,
                 float* chan_est, int* chan_est_written);

/* ---- C3 / C2 (lib/mimo_precoder_impl.cc:275-983) ---- */
void orc_steering_from_channel(int T, const float* h, int phased, float* Q /* col-major T x T */);
void orc_dft_matrix(int T, float* F /* col-major T x T */);
typedef struct {
    int fft_len, n_tx;
    int n_data, n_pilot;
    const int* data_carriers; const int* pilot_carriers;
    int n_pilot_rows; const float* pilot_symbols;
    int n_sync; const float* sync_words;        /* complex [n_sync][fft_len] */
    const float* mapped_ltf;                    /* complex [fft_len][n_tx*n_tx] */
} orc_pre_cfg;
int orc_precoder_work(const orc_pre_cfg* c, int ninput_items, const float* in, int mcs, int packet_type,
                      int pdu_len, int steer_mode, const float* Q_mean, const float* Q_sc,
                      const float* radar_streams, float* const* out);

/* ---- §8(f) rank 2: target_simulator_impl (lib/target_simulator_impl.cc:132-385) ---- */
typedef struct orc_tsim_state orc_tsim_state;
orc_tsim_state* orc_tsim_create(int K, const float* range, const float* velocity, const float* rcs,
                                const float* azimuth, int R, const float* position_rx, int samp_rate,
                                float center_freq, float self_coupling_db, int rndm_phaseshift, int self_coupling);
void orc_tsim_destroy(orc_tsim_state* st);
/* channel filters of a burst of n samples (:249-300): n complex floats each */
const float* orc_tsim_filt_doppler(orc_tsim_state* st, int n, int k);
const float* orc_tsim_filt_time(orc_tsim_state* st, int n, int l, int k);
/* one tagged burst: in[n_input] -> out[l][n_input], l < R.  target_phase: K complex multipliers (caller-drawn random
 * phases, used when rndm_phaseshift), may be NULL.  sum_targets 0 = as written in the reference (last target wins),
 * 1 = targets accumulated.  Returns n_input. */
int orc_tsim_work(orc_tsim_state* st, const float* in, int n_input, float* const* out, const float* target_phase,
                  int sum_targets);
/* unnormalised DFT of any length (gr::fft::fft_complex forward/reverse restated), double inside */
void orc_dft_any(int n, int forward, const float* in, float* out);

/* ---- §8(f) rank 4: bit codec (lib/stream_encoder_impl.cc, lib/stream_decoder_impl.cc, lib/utils.cc, lib/viterbi_decoder.cc) ---- */
uint32_t orc_crc32(const uint8_t* p, size_t n);
void orc_constellation_point(int bpsc, int value, float* re, float* im);
int  orc_constellation_decide(int bpsc, float re, float im);
int  orc_stream_encode(int mcs, int n_dc, const uint8_t* psdu, int len, int scrambler_init, float* out_sym, int* n_ofdm_sym,
                       int* pdu_len_tag);
int  orc_viterbi_windowed(int mcs, int n_sym, int n_cbps, int n_data_bits, const uint8_t* in, uint8_t* decoded);
int  orc_stream_decode(int mcs, int n_dc, int data_size_byte, const float* sym, uint8_t* out_payload);

/* ---- §8(f) rank 4: sync front-end (lib/moving_avg_impl.cc, lib/frame_detector_impl.cc, lib/frame_sync_impl.cc) ---- */
int  orc_moving_avg_work(int length, float scale, int max_iter, int noutput_items, const float* in, float* out);
void orc_sync_metrics(const float* x, int n, int delay, int window, int pwindow, float pscale, float* xd, float* in_abs,
                      float* in_cor);
typedef struct orc_fd_state orc_fd_state;
orc_fd_state* orc_fd_create(int fft_len, int cp_len, double threshold, int min_n_peaks, int ignore_gap);
void orc_fd_destroy(orc_fd_state* s);
int  orc_fd_work(orc_fd_state* s, int noutput, int ninput, const float* in, const float* in_abs, const float* in_cor, float* out,
                 int* consumed, uint64_t* tag_off, double* tag_cfo, int max_tags, int* n_tags);
typedef struct orc_fs_state orc_fs_state;
orc_fs_state* orc_fs_create(int fft_len, int cp_len, int sync_length, const float* ltf_seq_time, int ntaps);
void orc_fs_destroy(orc_fs_state* s);
int  orc_fs_frame_start(const orc_fs_state* s);
double orc_fs_freq_offset(const orc_fs_state* s);
/* returns items produced, -1 for the runtime_error of :135 */
int  orc_fs_work(orc_fs_state* s, int noutput, int ninput0, int ninput1, const float* in, const float* in_delayed,
                 const uint64_t* tin_off, const double* tin_val, int n_tin, float* out, int* consumed, uint64_t* tag_out_off,
                 double* tag_out_val, int* n_tag_out);

/* ofdm_frame_generator_impl::work (lib/ofdm_frame_generator_impl.cc:155-216) */
int orc_frame_generator(int fft_len, int n_occ_sets, const int* occ_sizes, const int* occ_flat, int n_pil_sets, const int* pil_sizes,
                        const int* pil_flat, int n_psym_sets, const float* psym_flat, int n_sync, const float* sync_words, int n_in,
                        const float* in, float* out, int noutput_items);

#ifdef __cplusplus
}
#endif
#endif
