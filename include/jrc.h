/*
 * jrc.h — C ABI of the MI355X (gfx950) implementation of the gr-mimo-ofdm-jrc hot path.
 *
 * This is the drop-in boundary: the GNU Radio block wrappers call these
 * entry points from their work()/general_work() bodies.  Plain pointers and sizes only; no C++
 * or torch types cross this line.  Citations are to the reference tree (/root/reference).
 *
 * Conventions
 *   - jrc_cf32 is gr_complex (std::complex<float>): interleaved re, im.
 *   - Functions return JRC_OK (0) or a negative jrc_status; jrc_strerror() names it and
 *     jrc_last_error() gives the detailed message (HIP error string, offending size ...).
 *     The reference signals the same conditions with C++ exceptions; the wrappers re-throw.
 *   - "host" entry points take caller-owned host memory (GNU Radio ring buffers), stage through
 *     pinned memory, run the HIP kernels and are synchronous on return.
 *   - "_dev" entry points take device pointers (hipMalloc / torch.cuda tensors) and a stream
 *     (hipStream_t passed as void*; NULL = the context's own, non-blocking stream, which does NOT order against
 *     the legacy default stream other libraries may be using: pass hipStreamLegacy = (void*)1 to run on that one)
 *     and are asynchronous.
 *   - One jrc_ctx per host thread; a ctx is bound to one GPU.  There is no CPU fallback: without a
 *     usable HIP device jrc_create() fails with JRC_ERR_NO_DEVICE.
 */
#ifndef JRC_H
#define JRC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JRC_ABI_VERSION 1

typedef struct { float re, im; } jrc_cf32;

typedef enum {
    JRC_OK = 0,
    JRC_ERR_NO_DEVICE = -1,        /* no HIP device / HIP runtime failure at init */
    JRC_ERR_HIP = -2,              /* a HIP call failed; see jrc_last_error() */
    JRC_ERR_INVALID_ARG = -3,      /* std::invalid_argument in the reference */
    JRC_ERR_UNSUPPORTED = -4,      /* size/shape outside what the HIP kernels implement */
    JRC_ERR_LENGTH_MISMATCH = -5,  /* matrix_transpose runtime_error (lib/matrix_transpose_impl.cc:82-83) */
    JRC_ERR_SHORT_INPUT = -6,      /* fewer items than the block needs for one frame */
    JRC_ERR_NOMEM = -7,
    JRC_ERR_SIG_FIELD = -8,        /* precoder: frame_param.n_ofdm_sym mismatch (lib/mimo_precoder_impl.cc:327-333) */
    JRC_ERR_IO = -9                /* file could not be opened (estimator / equalizer CSV side files) */
} jrc_status;

typedef struct jrc_ctx jrc_ctx;

/* ---- context ---------------------------------------------------------------------------- */
int         jrc_abi_version(void);
int         jrc_device_count(void);                       /* never initialises a device */
int         jrc_create(int device, jrc_ctx** ctx);
void        jrc_destroy(jrc_ctx* ctx);
const char* jrc_strerror(int status);
const char* jrc_last_error(const jrc_ctx* ctx);
int         jrc_device_name(const jrc_ctx* ctx, char* buf, size_t len);
int         jrc_sync(jrc_ctx* ctx);                        /* wait for the ctx stream */
void*       jrc_stream(jrc_ctx* ctx);                      /* the ctx's hipStream_t */

/* device memory helpers for non-torch hosts (the GNU Radio wrappers, C tests) */
int jrc_dev_malloc(jrc_ctx* ctx, size_t bytes, void** dptr);
int jrc_dev_free(jrc_ctx* ctx, void* dptr);
int jrc_dev_memset(jrc_ctx* ctx, void* dptr, int value, size_t bytes);
int jrc_memcpy_h2d(jrc_ctx* ctx, void* dptr, const void* hptr, size_t bytes);   /* synchronous */
int jrc_memcpy_d2h(jrc_ctx* ctx, void* hptr, const void* dptr, size_t bytes);   /* synchronous */

/* ---- A1  mimo_ofdm_radar  (replaces mimo_ofdm_radar_impl::general_work,
 *          lib/mimo_ofdm_radar_impl.cc:131-340; ctor :66-120; setter :342-346) ------------------- */
typedef struct jrc_radar jrc_radar;
int  jrc_radar_create(jrc_ctx* ctx, int fft_len, int N_tx, int N_rx, int N_sym, int N_pre,
                      int background_removal, int background_recording, int record_len,
                      int interp_factor, int enable_tx_interleave, jrc_radar** radar);
void jrc_radar_destroy(jrc_radar* radar);
int  jrc_radar_set_background_record(jrc_radar* radar, int background_recording);
int  jrc_radar_ring_size(const jrc_radar* radar);
/* tx[t] / rx[r]: the T / R GNU Radio input port buffers (item = fft_len cf32) positioned at the
 * tagged packet start; n_items_* = ninput_items on those ports; tx_discard = items of stale TX
 * packets skipped (:191-198).  out = output port buffer, P x (fft_len*interp_factor) cf32
 * (zero-padded rows, :303-315).  Returns the number of items produced (P) or < 0. */
int  jrc_radar_work(jrc_radar* radar, const jrc_cf32* const* tx, const jrc_cf32* const* rx,
                    size_t n_items_tx, size_t n_items_rx, size_t tx_discard, jrc_cf32* out);

/* batched, device-resident A1: d_frames [n_frames][T+R][n_items][fft_len] -> d_chanest [n_frames][P][fft_len] (no padding) */
int  jrc_radar_chanest_dev(jrc_ctx* ctx, int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int n_items,
                           int enable_tx_interleave, int n_frames, const jrc_cf32* d_frames, jrc_cf32* d_chanest,
                           void* stream);
/* A6 + A7 + A1 fused: as jrc_radar_chanest_dev with TX rows d_tx [n_frames][T][n_items][fft_len] and time-domain RX streams
 * d_rx_td [n_frames][R][rx_stream_len] (see jrc_chain_run_td_dev) */
int  jrc_radar_chanest_td_dev(jrc_ctx* ctx, int fft_len, int cp_len, int N_tx, int N_rx, int N_sym, int N_pre, int n_items,
                              long rx_stream_len, int enable_tx_interleave, int n_frames, const jrc_cf32* d_tx,
                              const jrc_cf32* d_rx_td, jrc_cf32* d_chanest, void* stream);

/* ---- A2/A4/A7  stock gr::fft::fft_vcc (FFTW3f in GNU Radio 3.8; flowgraph wiring
 *          examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:877-1047) -------------------------
 * forward: out = [fftshift] FFT(in .* window); reverse: out = unnormalised IFFT([ifftshift](in .* window)).
 * fft_size must be a power of two in [2, 16384].  window may be NULL (rectangular). */
int jrc_fft_vcc(jrc_ctx* ctx, int fft_size, int forward, int shift, const float* window,
                size_t batch, const jrc_cf32* in, jrc_cf32* out);
int jrc_fft_vcc_dev(jrc_ctx* ctx, int fft_size, int forward, int shift, const float* d_window,
                    size_t batch, const jrc_cf32* d_in, jrc_cf32* d_out, void* stream);

/* ---- A3  matrix_transpose (replaces matrix_transpose_impl::work, lib/matrix_transpose_impl.cc:69-110)
 * in: ninput_items x input_len; out: input_len x (output_len*interp_factor), zero padded.
 * Returns items produced (= input_len), or JRC_ERR_LENGTH_MISMATCH (:82-83). */
int jrc_matrix_transpose(jrc_ctx* ctx, int input_len, int output_len, int interp_factor,
                         int ninput_items, const jrc_cf32* in, jrc_cf32* out);
int jrc_matrix_transpose_dev(jrc_ctx* ctx, int input_len, int output_len, int interp_factor,
                             int ninput_items, size_t batch, const jrc_cf32* d_in, jrc_cf32* d_out,
                             void* stream);

/* ---- A5  range_angle_estimator (replaces range_angle_estimator_impl::work,
 *          lib/range_angle_estimator_impl.cc:121-284) ------------------------------------------ */
typedef struct {
    int32_t peak_range_idx, peak_angle_idx;   /* first arg-max of |z|^2 in scan order (:137-151) */
    int32_t angle_null_idx;                   /* (:155-187) */
    int32_t discard_range_idx, discard_angle_idx;   /* (:189-195) */
    int32_t n_noise_samples;
    float   peak_power, noise_power, snr_est; /* (:227-232) */
    float   range_val, angle_val;             /* range_bins[peak_range_idx], angle_bins[peak_angle_idx] */
    int32_t published;                        /* snr_est >= snr_threshold && peak_power >= power_threshold (:234) */
} jrc_ra_result;
int jrc_ra_estimate(jrc_ctx* ctx, int vlen, int n_inputs, const jrc_cf32* in,
                    const float* range_bins, int n_range_bins,
                    const float* angle_bins, int n_angle_bins,
                    float noise_discard_range_m, float noise_discard_angle_deg,
                    float snr_threshold, float power_threshold, jrc_ra_result* result);

/* ---- A6  ofdm_cyclic_prefix_remover (replaces ofdm_cyclic_prefix_remover_impl::work,
 *          lib/ofdm_cyclic_prefix_remover_impl.cc:69-99).  Returns noutput_items. ---------------- */
int jrc_cp_remove(jrc_ctx* ctx, int fft_len, int cp_len, size_t ninput_items,
                  const jrc_cf32* in, jrc_cf32* out);
/* A6+A7 fused: CP removal followed by fft_vxx forward+shift of size fft_len (RX OFDM demod). */
int jrc_cp_remove_fft(jrc_ctx* ctx, int fft_len, int cp_len, size_t ninput_items,
                      const jrc_cf32* in, jrc_cf32* out);
int jrc_cp_remove_fft_dev(jrc_ctx* ctx, int fft_len, int cp_len, size_t n_symbols,
                          const jrc_cf32* d_in, jrc_cf32* d_out, void* stream);

/* TX OFDM modulator, SURVEY.md §8(f) rank 1: fft_vxx reverse+shift+window followed by digital_ofdm_cyclic_prefixer
 * (rolloff 0) as wired after mimo_precoder (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:801-897).
 * in: n_symbols x fft_len (DC at fft_len/2); out: n_symbols x (cp_len + fft_len) time samples.  Returns n_symbols. */
int jrc_ofdm_mod(jrc_ctx* ctx, int fft_len, int cp_len, const float* window, size_t n_symbols,
                 const jrc_cf32* in, jrc_cf32* out);
int jrc_ofdm_mod_dev(jrc_ctx* ctx, int fft_len, int cp_len, const float* d_window, size_t n_symbols,
                     const jrc_cf32* d_in, jrc_cf32* d_out, void* stream);

/* ---- B1  fft_peak_detect (replaces fft_peak_detect_impl::work, lib/fft_peak_detect_impl.cc:77-111)
 * Returns 1 (items produced, always, :110).  *k_out = winning bin or -1; when -1 the three
 * outputs are left untouched exactly like the reference. */
int jrc_fft_peak_detect(jrc_ctx* ctx, int samp_rate, float interp_factor, float threshold,
                        int samp_protect, size_t ninput_items, const jrc_cf32* in,
                        float* out_freq, float* out_phase, float* out_mag, int* k_out);

/* ---- fused, device-resident radar chain  A1 -> A2 -> A3 -> A4 -> A5 over a batch of frames ------
 * One launch sequence replaces mimo_ofdm_radar + fft_vxx(reverse) + matrix_transpose +
 * fft_vxx(forward,shift) + range_angle_estimator of the radar flowgraph
 * (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2189-2197).  Frames are independent.
 * Shapes: both transform sizes fft_len*interp_range and N_tx*N_rx*interp_angle must be powers of two (<= 16384).
 * Power-of-two fft_len in [64, 1024], N_tx*N_rx <= 16 and 2 <= interp_angle <= 64 take the fused kernel (A2-A4 and
 * the arg-max never leave the chip); other shapes run block by block on the device with the same results. */
typedef struct {
    int32_t fft_len, N_tx, N_rx, N_sym, N_pre;
    int32_t interp_range, interp_angle;       /* Ir (mimo_ofdm_radar), Ia (matrix_transpose) */
    int32_t enable_tx_interleave;
    int32_t n_items;                          /* items per port per frame in d_frames (>= N_pre+N_sym) */
    float   noise_discard_range_m, noise_discard_angle_deg;
    float   snr_threshold, power_threshold;
} jrc_chain_cfg;

typedef struct jrc_chain jrc_chain;
int  jrc_chain_create(jrc_ctx* ctx, const jrc_chain_cfg* cfg,
                      const float* range_bins /* fft_len*interp_range */,
                      const float* angle_bins /* N_tx*N_rx*interp_angle */,
                      int max_frames, jrc_chain** chain);
void jrc_chain_destroy(jrc_chain* chain);
/* bytes per frame of each device buffer the caller provides */
size_t jrc_chain_frame_bytes(const jrc_chain* chain);     /* (T+R) * n_items * fft_len * 8 */
size_t jrc_chain_chanest_bytes(const jrc_chain* chain);   /* P * fft_len * 8 */
size_t jrc_chain_map_bytes(const jrc_chain* chain);       /* (fft_len*Ir) * (P*Ia) * 8 */
/* d_frames : [n_frames][T+R][n_items][fft_len] cf32, ports ordered TX0..TX(T-1), RX0..RX(R-1)
 * d_chanest: [n_frames][P][fft_len] cf32            (A1 output before zero padding)
 * d_map    : [n_frames][fft_len*Ir][P*Ia] cf32      (A4 output; the estimator's input)
 * d_results: [n_frames] jrc_ra_result (snr_est/published are filled by jrc_chain_finish_results)
 * Asynchronous on `stream`. */
int  jrc_chain_run_dev(jrc_chain* chain, int n_frames, const jrc_cf32* d_frames,
                       jrc_cf32* d_chanest, jrc_cf32* d_map, jrc_ra_result* d_results, void* stream);
/* Same chain with the receive side handed over in the TIME domain: A6 (ofdm_cyclic_prefix_remover,
 * lib/ofdm_cyclic_prefix_remover_impl.cc:69-99) + A7 (stock fft_vxx forward/shift, flowgraph `fft_vxx_0_0`) + A1 run as one
 * kernel, so the frequency-domain RX symbols never go to HBM and the N_pre preamble symbols A1 skips are not transformed.
 * d_tx   : [n_frames][T][n_items][fft_len] cf32   frequency-domain TX reference symbols (the TX ports of d_frames)
 * d_rx_td: [n_frames][R][rx_stream_len] cf32      RX streams, symbol k of a stream at k*(fft_len+cp_len);
 *                                                 rx_stream_len >= n_items*(fft_len+cp_len)
 * Results agree with jrc_cp_remove_fft_dev + jrc_chain_run_dev to float rounding (1e-7 relative).  fft_len: power of two 16..1024, N_tx in {1,2,3,4,8};
 * other shapes fail with JRC_ERR_UNSUPPORTED (run the two calls instead). */
int  jrc_chain_run_td_dev(jrc_chain* chain, int n_frames, const jrc_cf32* d_tx, const jrc_cf32* d_rx_td, int cp_len,
                          long rx_stream_len, jrc_cf32* d_chanest, jrc_cf32* d_map, jrc_ra_result* d_results, void* stream);
/* D2H the results of the last run and complete snr_est / published on the host with libm's log10f
 * (bit-identical to the reference's std::log10, :227). Synchronises `stream`. */
int  jrc_chain_fetch_results(jrc_chain* chain, int n_frames, const jrc_ra_result* d_results,
                             jrc_ra_result* h_results, void* stream);
/* The same in two halves, for a caller that keeps batches in flight: _begin orders the copy of the records behind the work queued on
 * `stream` so far and runs it on the chain's own copy stream (the caller's stream is not blocked and goes on with the next batch);
 * _end waits for the oldest copy begun and completes snr_est / published as above.  At most two copies may be in flight (the third
 * _begin fails with JRC_ERR_INVALID_ARG); d_results must stay untouched until the matching _end, so a caller alternates two buffers. */
int  jrc_chain_fetch_results_begin(jrc_chain* chain, int n_frames, const jrc_ra_result* d_results, void* stream);
int  jrc_chain_fetch_results_end(jrc_chain* chain, jrc_ra_result* h_results, int* n_frames);
/* HIP-event timing of the dominant kernel (range-angle FFT + detect) for bench.py's roofline:
 * when enabled every jrc_chain_run_dev brackets each kernel with events on `stream`. */
int  jrc_chain_set_timing(jrc_chain* chain, int enabled);
/* mean milliseconds per launch since timing was enabled/reset, per kernel:
 * ms[0]=radar_chanest, ms[1]=range_angle_fused, ms[2]=ra_finalize; *launches = runs measured */
int  jrc_chain_get_timing(jrc_chain* chain, float ms[3], int* launches);
/* launches of the dominant kernel per jrc_chain_run_dev of n_frames (a batch beyond one resident wave of workgroups runs in chunks) */
int  jrc_chain_launches_per_run(const jrc_chain* chain, int n_frames);

/* Background recording / removal of mimo_ofdm_radar (lib/mimo_ofdm_radar_impl.cc:276-300; make() arguments background_removal,
 * background_recording, record_len, include/mimo_ofdm_jrc/mimo_ofdm_radar.h:52-56; setter set_background_record :61) for the batched
 * chain.  The frames of a batch are consecutive frames of ONE radar stream: frame f's estimate has the mean of the <= record_len
 * estimates recorded before it subtracted (oldest first, each term divided by the count, exactly :281-292), across batches; the
 * history lives on the device.  Without this call the chain behaves as a block built with both flags false.  May be called between
 * batches to switch recording / removal; record_len is fixed by the first call. */
int  jrc_chain_set_background(jrc_chain* chain, int background_removal, int background_recording, int record_len);
/* `chain` continues the stream of `owner` (same shape, same GPU): both advance one history; batches are ordered by an event, so the
 * chains may run on different streams (the slots of a jrc_chain_feed) */
int  jrc_chain_share_background(jrc_chain* chain, jrc_chain* owner);
/* entries in the history (the reference's radar_chan_est_buffer.size()) */
int  jrc_chain_background_size(const jrc_chain* chain);
/* channel estimate + history update only, outputs dropped.  A GPU that owns frames [lo, hi) of a sharded stream first primes with
 * frames [lo - min(record_len, lo), lo): its history then equals the one a single GPU would hold at frame lo. */
int  jrc_chain_prime_background_dev(jrc_chain* chain, int n_frames, const jrc_cf32* d_frames, void* stream);

/* Detect-only mode.  write_map = 0: jrc_chain_run_dev / _td_dev store no range-angle map (d_map is ignored and may be NULL); the
 * arg-max runs on the values in registers and the estimator's noise-window rows (lib/range_angle_estimator_impl.cc:197-226) are
 * re-computed by the same kernel code into a small buffer, so d_results is bit-identical to write_map = 1.  For consumers that only
 * take range_angle_estimator's `params` message (:234-253).  Fused-kernel shapes only (else JRC_ERR_UNSUPPORTED). */
int  jrc_chain_set_write_map(jrc_chain* chain, int write_map);
/* Map format.  JRC_MAP_COMPLEX (default): d_map is the complex map the reference's estimator reads.  JRC_MAP_POWER: d_map is
 * float [n_frames][fft_len*Ir][P*Ia] holding |z|^2 = re*re + im*im — the stream the flowgraph's display branch consumes
 * (blocks_complex_to_mag_squared -> gui_heatmap_plot, examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2192) — at half the bytes;
 * d_results is bit-identical to the complex format (the estimator's window rows are re-computed as in detect-only mode).
 * jrc_chain_map_bytes follows the format.  Fused-kernel shapes only. */
enum { JRC_MAP_COMPLEX = 0, JRC_MAP_POWER = 1 };
int  jrc_chain_set_map_format(jrc_chain* chain, int format);

/* ---- host-fed pipeline over the chain: what a GNU Radio work() hands over is HOST memory (the T+R input ring buffers of
 *      mimo_ofdm_radar, lib/mimo_ofdm_radar_impl.cc:207-238) and what leaves the radar branch is one small record per
 *      frame (the PDU of lib/range_angle_estimator_impl.cc:199-235).  `n_slots` batches of up to `frames_per_slot` frames
 *      are kept in flight, each on its own stream with its own pinned staging and device buffers, so copy-in, kernels and
 *      copy-out of consecutive batches overlap.  Results come back in submission order.  Not thread-safe (one feeder thread).
 *   JRC_FEED_GRAPH: a full slot is recorded once as a hipGraph (copy-in, A1, fused A2-A4, A5, copy-out) and replayed. */
enum { JRC_FEED_GRAPH = 1 };
typedef struct jrc_chain_feed jrc_chain_feed;
int    jrc_chain_feed_create(jrc_ctx* ctx, const jrc_chain_cfg* cfg, const float* range_bins, const float* angle_bins,
                             int n_slots, int frames_per_slot, int maps_per_slot /* maps copied back per batch, 0 = none */,
                             int flags, jrc_chain_feed** feed);
void   jrc_chain_feed_destroy(jrc_chain_feed* feed);
size_t jrc_chain_feed_frame_bytes(const jrc_chain_feed* feed);   /* = jrc_chain_frame_bytes */
size_t jrc_chain_feed_map_bytes(const jrc_chain_feed* feed);
/* pinned staging of the next slot ([frames_per_slot] frames, layout of jrc_chain_run_dev's d_frames) to fill in place;
 * fails when every slot is in flight */
int    jrc_chain_feed_acquire(jrc_chain_feed* feed, jrc_cf32** h_frames);
/* enqueue the next slot; h_frames = NULL: the acquired buffer was filled in place, else n_frames frames are staged from
 * h_frames (pageable memory is fine).  Asynchronous. */
int    jrc_chain_feed_submit(jrc_chain_feed* feed, const jrc_cf32* h_frames, int n_frames);
/* wait for the OLDEST batch in flight: results[frames_per_slot], maps (may be NULL) [maps_per_slot] maps of its first
 * frames.  Returns the number of frames (also in *n_frames), 0 when nothing is in flight, < 0 on error. */
int    jrc_chain_feed_collect(jrc_chain_feed* feed, jrc_ra_result* results, jrc_cf32* maps, int* n_frames);
int    jrc_chain_feed_pending(const jrc_chain_feed* feed);        /* batches in flight */
/* 1 when the oldest batch in flight has finished (jrc_chain_feed_collect will not block), else 0 */
int    jrc_chain_feed_poll(const jrc_chain_feed* feed);
/* TX-resident submission.  mimo_ofdm_radar correlates every packet with its T reference ports (lib/mimo_ofdm_radar_impl.cc:250-274); in the
 * reference's flowgraph those are the MIMO-LTF rows, identical from packet to packet (N_pre = 5, N_sym = N_tx,
 * examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:1292-1295).  _set_tx keeps a copy of them on the device (h_tx: [N_tx][n_items][fft_len],
 * the first N_tx ports of a frame; NULL switches it off; no batch may be in flight); _submit_rx is jrc_chain_feed_submit for frames whose TX
 * ports EQUAL that copy (the caller's promise): only the N_rx receive ports of each frame are read from h_frames / the acquired buffer and
 * cross PCIe.  Full and receive-only submissions may alternate; results are those of a full submission of the same frames. */
int    jrc_chain_feed_set_tx(jrc_chain_feed* feed, const jrc_cf32* h_tx);
int    jrc_chain_feed_submit_rx(jrc_chain_feed* feed, const jrc_cf32* h_frames, int n_frames);
int    jrc_chain_feed_stats(const jrc_chain_feed* feed, long* graph_replays, long* direct_submits);
/* Several GPUs fed from one host process.  `devices[n_devices]` (a device may be listed more than once) each get their own context,
 * `slots_per_device` slots and one host thread; batch k of the submission order runs on devices[k mod n_devices]; results come back in
 * submission order through the same jrc_chain_feed_acquire / _submit / _collect calls.  Frames are independent units:
 * there is no exchange between the devices. */
int    jrc_chain_feed_create_multi(const int* devices, int n_devices, const jrc_chain_cfg* cfg, const float* range_bins,
                                   const float* angle_bins, int slots_per_device, int frames_per_slot, int maps_per_slot, int flags,
                                   jrc_chain_feed** feed);
int    jrc_chain_feed_n_devices(const jrc_chain_feed* feed);
/* message of the last failing jrc_chain_feed_* call on this feed (a multi-device feed owns its contexts: this is the way to its errors) */
const char* jrc_chain_feed_last_error(const jrc_chain_feed* feed);
/* n_batches (<= free slots) batches at once: batch k is staged from h_frames[k] (pageable memory is fine) and enqueued by the host
 * thread of the device it lands on, all devices in parallel; returns when every batch has left the caller's buffers */
int    jrc_chain_feed_submit_many(jrc_chain_feed* feed, const jrc_cf32* const* h_frames, const int* n_frames, int n_batches);
/* background recording / removal for the stream the feed carries (see jrc_chain_set_background): the slots share one history and
 * consecutive batches are ordered across their streams; hipGraph replay is switched off (the history pointers alternate) */
int    jrc_chain_feed_set_background(jrc_chain_feed* feed, int background_removal, int background_recording, int record_len);
/* detect-only mode for every slot (see jrc_chain_set_write_map); maps_per_slot must be 0 */
int    jrc_chain_feed_set_write_map(jrc_chain_feed* feed, int write_map);

/* ---- D  range-Doppler map — NO reference counterpart (the reference sums over symbols,
 *          lib/mimo_ofdm_radar_impl.cc:271-274); defined by this build, parity unpinned by construction:
 *   D[p][sym][sc] = rx_r[sym][sc]*conj(tx_t[sym][sc]);  out[f][p][k][d] = fftshift_d FFT_{S*Id}(IFFT_{N*Ir}(D zero-padded))
 * d_work: n_frames*P*S*(N*Ir) cf32 scratch; d_out: [n_frames][P][N*Ir][S*Id] cf32.  cfg: the chain's (interp_angle unused). */
int jrc_range_doppler_dev(jrc_ctx* ctx, const jrc_chain_cfg* cfg, int interp_doppler, int n_frames,
                          const jrc_cf32* d_frames, jrc_cf32* d_work, jrc_cf32* d_out, void* stream);

/* ---- C1  mimo_ofdm_equalizer (replaces mimo_ofdm_equalizer_impl::general_work and helpers,
 *          lib/mimo_ofdm_equalizer_impl.cc:191-922; ctor :66-180; setters :924-960) -------------------------
 * One jrc_equalizer holds `n_streams` independent per-RX-stream states (the reference block has one input
 * port; "4 RX" = 4 instances).  The per-symbol state machine (L-LTF LS estimate, SIG decode incl. the K=7
 * Viterbi, MIMO-LTF estimate, pilot CPE tracking, ZF / MMSE-like equalisation, optional STA update) runs in
 * one kernel, one workgroup per stream, one lane per subcarrier. */
typedef struct {
    int32_t estimator;                 /* ChannelEstimator: 0 = LS, 1 = STA (include/mimo_ofdm_jrc/mimo_ofdm_equalizer.h:27-30) */
    double  freq, bw;
    int32_t fft_len, cp_len;
    int32_t n_data, n_pilot;
    const int32_t* data_carriers;      /* signed subcarrier indices exactly as passed to make() */
    const int32_t* pilot_carriers;
    int32_t n_pilot_rows;
    const jrc_cf32* pilot_symbols;     /* [n_pilot_rows][n_pilot] */
    const jrc_cf32* ltf_seq;           /* [fft_len] */
    const jrc_cf32* mapped_ltf;        /* [fft_len][mapped_cols], mapped_cols = N_tx * n_mimo_ltf */
    int32_t mapped_cols, n_mimo_ltf;
} jrc_eq_cfg;

typedef struct {
    int32_t  kind;                     /* 1 = "stream_start" tag (:331-337), 2 = "stream_end" tag (:626-629) */
    int32_t  n_chan_mean;
    int64_t  offset;                   /* output item the tag sits on, relative to this call (may be -1) */
    uint64_t data_bytes, mcs, packet_type;
    double   snr, freq_offset;         /* stream_start dict */
    double   snr_data;                 /* stream_end dict */
    jrc_cf32 chan_mean[16];            /* stream_end dict "chan_mean" */
} jrc_eq_event;

typedef struct jrc_equalizer jrc_equalizer;
int  jrc_equalizer_create(jrc_ctx* ctx, const jrc_eq_cfg* cfg, int n_streams, jrc_equalizer** eq);
void jrc_equalizer_destroy(jrc_equalizer* eq);
int  jrc_equalizer_set_estimator(jrc_equalizer* eq, int algo);
int  jrc_equalizer_set_bandwidth(jrc_equalizer* eq, double bw);
int  jrc_equalizer_set_frequency(jrc_equalizer* eq, double freq);
/* general_work() of stream `stream`: in = ninput_items vectors of fft_len; frame_start tags given as item
 * offsets relative to `in` plus their double value; out = up to noutput_items vectors of n_data.
 * chan_est (may be NULL): [fft_len][N_tx], written when an NDP frame's MIMO-LTFs complete (the content of the
 * reference's chan_est_file, :378-416); *chan_est_written says so.  Returns items produced or < 0. */
int  jrc_equalizer_work(jrc_equalizer* eq, int stream, int noutput_items, int ninput_items, const jrc_cf32* in,
                        const int64_t* tag_offsets, const double* tag_values, int n_tags, jrc_cf32* out,
                        int* n_consumed, jrc_eq_event* events, int max_events, int* n_events,
                        jrc_cf32* chan_est, int* chan_est_written);
/* Batched, device-resident: every stream s gets one frame of n_symbols vectors starting at
 * d_in + s*n_symbols*fft_len with a frame_start tag of value d_phase[s] on its first item.
 * d_out: [n_streams][max_out][n_data]; d_n_out: [n_streams] items produced; d_events: [n_streams][2], every slot
 * written (kind 0 = unused), so the array needs no clearing by the caller. */
int  jrc_equalizer_frames_dev(jrc_equalizer* eq, int n_streams, int n_symbols, const jrc_cf32* d_in,
                              const double* d_phase, int max_out, jrc_cf32* d_out, int32_t* d_n_out,
                              jrc_eq_event* d_events, void* stream);

/* ---- C3  steering matrices (replaces the per-subcarrier body of compute_steering_matrix /
 *          compute_radar_aided_steering, lib/mimo_precoder_impl.cc:846-861, :880-893, :961-974) -----------
 * h: [n][T] channel rows; Q: [n][T*T] column-major (Eigen layout of steering_matrix[sc]).
 * phased: Q[:,0] = sqrt(T) conj(h)/||h||, rest 0.  Otherwise Q = V sqrt(T)/||V||_F with V the full
 * right-singular basis Eigen's JacobiSVD returns for a 1 x T row (Householder construction). T <= 8. */
int jrc_steering_from_channel(jrc_ctx* ctx, int T, int n, const jrc_cf32* h, int phased, jrc_cf32* Q);
/* get_dft_matrix_eigen (lib/mimo_precoder_impl.cc:761-772), column-major T x T */
int jrc_dft_matrix(jrc_ctx* ctx, int T, jrc_cf32* F);

/* ---- C2  mimo_precoder (replaces mimo_precoder_impl::work, lib/mimo_precoder_impl.cc:275-741;
 *          generate_signal_field :985-1060; calculate_output_stream_length :265-272) -------------------- */
typedef struct {
    int32_t fft_len, N_tx;
    int32_t n_data, n_pilot;
    const int32_t* data_carriers;
    const int32_t* pilot_carriers;
    int32_t n_pilot_rows;
    const jrc_cf32* pilot_symbols;     /* [n_pilot_rows][n_pilot] */
    int32_t n_sync;
    const jrc_cf32* sync_words;        /* [n_sync][fft_len] */
    const jrc_cf32* mapped_ltf;        /* [fft_len][N_tx*N_tx] */
} jrc_pre_cfg;
typedef struct jrc_precoder jrc_precoder;
int  jrc_precoder_create(jrc_ctx* ctx, const jrc_pre_cfg* cfg, jrc_precoder** pre);
void jrc_precoder_destroy(jrc_precoder* pre);
int  jrc_precoder_output_length(const jrc_precoder* pre, int ninput_items);
/* steer_mode: 0 = DFT ("fourier") precoding, 1 = one steering matrix Q_mean for every subcarrier (channel
 * smoothing / radar-aided), 2 = per-subcarrier Q_sc[fft_len][T*T].  Matrices column-major.
 * radar_streams: NULL (use_radar_streams = false) or [(T-1)][n_sym][fft_len] symbols for streams 1..T-1 (the
 * reference draws them from std::random_device, :435-437; here the caller supplies them).
 * out[t]: [n_sync + 1 + N_tx + n_sym][fft_len].  Returns items produced per port, or JRC_ERR_SIG_FIELD. */
int  jrc_precoder_work(jrc_precoder* pre, int ninput_items, const jrc_cf32* in, int mcs, int packet_type,
                       int pdu_len, int steer_mode, const jrc_cf32* Q_mean, const jrc_cf32* Q_sc,
                       const jrc_cf32* radar_streams, jrc_cf32* const* out);

/* batched, device-resident: n_frames packets of one format per launch.  d_in [n_frames][ninput_items], d_radar_streams NULL or
 * [n_frames][T-1][n_sym][fft_len], d_out [n_frames][T][n_sync + 1 + N_tx + n_sym][fft_len]; Q matrices on the device (column-major,
 * shared by all frames).  Asynchronous on `stream`; returns items per port per frame. */
int  jrc_precoder_frames_dev(jrc_precoder* pre, int n_frames, int ninput_items, const jrc_cf32* d_in, int mcs, int packet_type,
                             int pdu_len, int steer_mode, const jrc_cf32* d_Q_mean, const jrc_cf32* d_Q_sc,
                             const jrc_cf32* d_radar_streams, jrc_cf32* d_out, void* stream);

/* SIG-field helpers shared by C1/C2 (host side; lib/utils.cc:26-111, lib/mimo_precoder_impl.cc:985-1060) */
int  jrc_n_ofdm_sym(int mcs, int n_data_carriers, int data_size_byte);
int  jrc_sig_encode(int n_data_carriers, int mcs, int packet_type, int length, float* out_re);

/* ---- target_simulator (lib/target_simulator_impl.cc:66-385; make() in
 * include/mimo_ofdm_jrc/target_simulator.h) — one input stream (a TX burst), n_rx output streams.  Per target k and
 * antenna l: out_l = IFFT_n(FFT_n(in . doppler_k) . timeshift_{l,k}) [. phase_k], + 10^(self_coupling_db/20) . in
 * when self_coupling.  n = burst length, any value up to 2^20: lengths n = n1 x 2^a with 2^a >= 16 and n1 <= 512 — every burst the
 * flowgraphs produce, n_symbols x (fft_len + cp) with fft_len + cp = 5 x 2^k — are transformed directly as n1-point x 2^a-point
 * four-step DFTs, all others as chirp-z transforms (JRC_TSIM_BLUESTEIN=1 forces those). ---- */
typedef struct jrc_tsim jrc_tsim;
typedef struct {
    int n_targets;                 /* range.size() (:160) */
    const float* range;            /* [n_targets] m */
    const float* velocity;         /* [n_targets] m/s */
    const float* rcs;              /* [n_targets] m^2 */
    const float* azimuth;          /* [n_targets] deg */
    int n_rx;                      /* position_rx.size() = number of output streams */
    const float* position_rx;      /* [n_rx] m */
    int samp_rate;                 /* Hz */
    float center_freq;             /* Hz */
    float self_coupling_db;
    int rndm_phaseshift;           /* when set, jrc_tsim_work/run_dev apply the caller-drawn target_phase[k] (:316-321) */
    int self_coupling;
    int sum_targets;               /* 0 = as written in the reference: every target overwrites the output (:354-366), the
                                      last one is what leaves the block; 1 = targets are accumulated */
    int max_bursts;                /* capacity of jrc_tsim_run_dev (device work buffers are sized for it) */
} jrc_tsim_cfg;
/* returns NULL on error (jrc_last_error(ctx)) — the reference constructor throws */
jrc_tsim* jrc_tsim_create(jrc_ctx* ctx, const jrc_tsim_cfg* cfg);
void jrc_tsim_destroy(jrc_tsim* h);
/* setup_targets() (:121-198): new target list, channel filters are rebuilt on the next burst */
int jrc_tsim_set_targets(jrc_tsim* h, int n_targets, const float* range, const float* velocity, const float* rcs,
                         const float* azimuth);
/* target_simulator_impl::work (:202-385) on host buffers: one tagged burst in[n_input] -> out[l][n_input], l < n_rx.
 * target_phase: n_targets complex multipliers exp(j 2 pi rand) drawn by the caller (used only when rndm_phaseshift;
 * may be NULL).  Returns n_input (items produced per output) or a negative status. */
int jrc_tsim_work(jrc_tsim* h, const jrc_cf32* in, int n_input, jrc_cf32* const* out, const jrc_cf32* target_phase);
/* batched device form: d_in [n_bursts][n_input], d_out [n_bursts][n_rx][n_input]; accumulate_out != 0 adds to d_out
 * instead of overwriting it (absorbs the blocks_add_xx that sums the per-TX simulators,
 * examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2213-2220).  Asynchronous on `stream` (NULL = ctx stream). */
int jrc_tsim_run_dev(jrc_tsim* h, int n_bursts, int n_input, const jrc_cf32* d_in, jrc_cf32* d_out,
                     const jrc_cf32* target_phase, int accumulate_out, void* stream);
/* n_sims simulators whose RX outputs the flowgraph adds (one target_simulator per TX port into one blocks_add_xx per RX antenna,
 * mimo_ofdm_jrc_radar_sim.grc:2213-2220) in ONE pass: d_out[b][l] (+)= sum_q sims[q](d_in[q][b])_l, the sum taken on the spectrum, so each
 * RX antenna costs one inverse transform and one write of d_out whatever n_sims is.  Equals n_sims calls of jrc_tsim_run_dev with
 * accumulate_out to the rounding of a float sum taken in another order.  d_in[q]: [n_bursts][n_input] of simulator q; target_phase[q]:
 * that simulator's host phases or NULL (the array itself may be NULL).  Simulators of one context built alike (same n_rx, target count,
 * sum_targets and self-coupling settings — the simulators of a flowgraph's TX ports differ in their antenna positions), at most 8 of them
 * and 32 (simulator, target) pairs; JRC_ERR_UNSUPPORTED otherwise and when the burst length does not take the direct route (then run them
 * one by one). */
int jrc_tsim_run_sum_dev(jrc_tsim* const* sims, int n_sims, int n_bursts, int n_input, const jrc_cf32* const* d_in, jrc_cf32* d_out,
                         const jrc_cf32* const* target_phase, int accumulate_out, void* stream);
int jrc_tsim_burst_capacity(const jrc_tsim* h);

/* ---- bit codec.  stream_encoder (lib/stream_encoder_impl.cc:76-270, make(mod_encode, data_len,
 * N_ss_radar, debug) in include/mimo_ofdm_jrc/stream_encoder.h:62) and stream_decoder (lib/stream_decoder_impl.cc:100-435,
 * make(n_data_carriers, comm_log_file, stats_record, debug)); mcs = the reference's MCS enum 0..5
 * (BPSK 1/2, 3/4, QPSK 1/2, 3/4, 16QAM 1/2, 3/4), payloads up to 3100 bytes incl. CRC (lib/utils.h MAX_PAYLOAD_SIZE). ---- */
/* packet_param::n_ofdm_sym for data_size_byte bytes (PDU + 4 CRC bytes) */
int jrc_stream_n_ofdm_sym(int mcs, int n_data_carriers, int data_size_byte);
/* one PDU -> n_ofdm_sym * n_data_carriers constellation points.  psdu[0] is the packet type byte exactly as the block
 * receives it; scrambler_init = the block's d_scrambler (1..127).  Returns the number of symbols written, 0 if the PDU is
 * too large (the reference prints and drops it, :139-143), negative on error. */
int jrc_stream_encode(jrc_ctx* ctx, int mcs, int n_data_carriers, const uint8_t* psdu, int len, int scrambler_init,
                      jrc_cf32* out_symbols, int out_capacity);
/* batched: d_psdu [n_frames][psdu_stride] bytes, d_len[n_frames], d_scrambler[n_frames] -> d_out [n_frames][sym_stride],
 * d_n_sym[n_frames] = symbols produced per PDU (0 = dropped).  Asynchronous on `stream`. */
int jrc_stream_encode_dev(jrc_ctx* ctx, int mcs, int n_data_carriers, int n_frames, const uint8_t* d_psdu, long psdu_stride,
                          const int* d_len, const uint8_t* d_scrambler, jrc_cf32* d_out, long sym_stride, int* d_n_sym,
                          void* stream);
/* one equalised frame (n_ofdm_sym * n_data_carriers symbols; mcs / data_size_byte from the stream_start tag, :118-131) ->
 * payload bytes (PSDU without CRC, data_size_byte - 4 of them).  *crc_ok = 1 when the CRC-32 residue matches (:246).
 * Returns the payload length, JRC_ERR_UNSUPPORTED when the reference would refuse the frame (:133-146). */
int jrc_stream_decode(jrc_ctx* ctx, int mcs, int n_data_carriers, int data_size_byte, const jrc_cf32* symbols, int n_symbols,
                      uint8_t* out_payload, int* crc_ok);
/* batched: d_sym [n_frames][sym_stride], d_mcs / d_data_bytes [n_frames] -> d_payload [n_frames][payload_stride],
 * d_status[n_frames] = 1 (CRC ok), 0 (CRC wrong), -1 (refused).  Asynchronous on `stream`. */
int jrc_stream_decode_dev(jrc_ctx* ctx, int n_data_carriers, int n_frames, const jrc_cf32* d_sym, long sym_stride,
                          const int* d_mcs, const int* d_data_bytes, uint8_t* d_payload, long payload_stride, int* d_status,
                          void* stream);

/* ---- sync front-end of the comm receive chain.  moving_avg (lib/moving_avg_impl.cc:62-98;
 * make(length, scale, max_iter, debug)), frame_detector (lib/frame_detector_impl.cc:70-205; make(fft_len, cp_len,
 * threshold, min_n_peaks, ignore_gap, debug)), frame_sync (lib/frame_sync_impl.cc:89-289; make(fft_len, cp_len,
 * sync_length, ltf_seq_time, debug)).  Each *_work call is one general_work()/work() call of the block. ---- */
/* moving_avg::work: `in` holds length-1 items of history followed by the new items; returns items produced
 * (min(noutput_items, max_iter)).  Window sums, scaled. */
int jrc_moving_avg(jrc_ctx* ctx, int length, float scale, int max_iter, int noutput_items, const jrc_cf32* in, jrc_cf32* out);
int jrc_moving_avg_dev(jrc_ctx* ctx, int length, float scale, int n_out, const jrc_cf32* d_in, jrc_cf32* d_out, void* stream);
/* the stock blocks wired in front of the detector (blocks_delay, conjugate, multiply, moving averages, complex_to_mag[_squared],
 * abs, divide; examples/simulation/communication/mimo_ofdm_jrc_comm_sim.grc) for a capture resident on the device:
 * d_xd[i] = x[i - delay], d_in_abs[i] = sum over `window` of x conj(x delayed), d_in_cor[i] = |in_abs| / |pscale * sum over
 * `pwindow` of |x|^2| */
int jrc_sync_metrics_dev(jrc_ctx* ctx, int n, int delay, int window, int pwindow, float pscale, const jrc_cf32* d_x,
                         jrc_cf32* d_xd, jrc_cf32* d_in_abs, float* d_in_cor, void* stream);
typedef struct jrc_frame_detector jrc_frame_detector;
jrc_frame_detector* jrc_frame_detector_create(jrc_ctx* ctx, int fft_len, int cp_len, double threshold, unsigned min_n_peaks,
                                              unsigned ignore_gap);
void jrc_frame_detector_destroy(jrc_frame_detector* d);
/* frame_detector::general_work: inputs in / in_abs / in_cor (ninput_items each), output out.  Returns items produced;
 * *n_consumed = consume_each(); frame_start tags added in this call come back as (absolute output offset, coarse CFO). */
int jrc_frame_detector_work(jrc_frame_detector* d, int noutput_items, int ninput_items, const jrc_cf32* in, const jrc_cf32* in_abs,
                            const float* in_cor, jrc_cf32* out, int* n_consumed, uint64_t* tag_offsets, double* tag_cfo,
                            int max_tags, int* n_tags);
typedef struct jrc_frame_sync jrc_frame_sync;
jrc_frame_sync* jrc_frame_sync_create(jrc_ctx* ctx, int fft_len, int cp_len, unsigned sync_length, const jrc_cf32* ltf_seq_time, int ntaps);
void jrc_frame_sync_destroy(jrc_frame_sync* f);
/* frame_sync::general_work: in (port 0) and in_delayed (port 1); tag_offsets/values = the frame_start tags on port 0
 * (absolute offsets).  Returns items produced (JRC_ERR_LENGTH_MISMATCH for the runtime_error of :135); *n_consumed items
 * are consumed on both ports; the frame_start tag the call adds, if any, comes back in tag_out_*. */
int jrc_frame_sync_work(jrc_frame_sync* f, int noutput_items, int ninput0, int ninput1, const jrc_cf32* in, const jrc_cf32* in_delayed,
                        const uint64_t* tag_offsets, const double* tag_values, int n_tags, jrc_cf32* out, int* n_consumed,
                        uint64_t* tag_out_offset, double* tag_out_value, int* n_tag_out);
/* d_state (0 SYNC, 1 COPY, 2 RESET), d_frame_start and d_freq_offset after the last call */
int jrc_frame_sync_state(const jrc_frame_sync* f, int* state, int* frame_start, float* freq_offset);

/* zero_pad (lib/zero_pad_impl.cc:62-94; make(debug, pad_front, pad_tail)): out = [pad_front | in | pad_tail] where the padding is
 * complex Gaussian noise, N(0, 1e-2) per component (the reference seeds a fresh std::random_device per call; `seed` makes it
 * reproducible here).  Returns n_input + pad_front + pad_tail.  _dev: n_bursts rows [n_input] -> rows [n_out]. */
int jrc_zero_pad(jrc_ctx* ctx, int n_input, unsigned pad_front, unsigned pad_tail, uint64_t seed, const jrc_cf32* in, jrc_cf32* out);
int jrc_zero_pad_dev(jrc_ctx* ctx, int n_bursts, int n_input, unsigned pad_front, unsigned pad_tail, uint64_t seed,
                     const jrc_cf32* d_in, jrc_cf32* d_out, void* stream);
/* the same with row strides (in items): burst b reads d_in + b*in_stride and writes d_out + b*out_stride — one TX port of a batch of
 * precoder / modulator outputs laid out [frame][port][samples] is in_stride = n_ports * n_input apart */
int jrc_zero_pad_strided_dev(jrc_ctx* ctx, int n_bursts, int n_input, unsigned pad_front, unsigned pad_tail, uint64_t seed,
                             const jrc_cf32* d_in, long in_stride, jrc_cf32* d_out, long out_stride, void* stream);
/* the OFDM modulator (jrc_ofdm_mod_dev) and the zero_pad behind each TX port as ONE pass (…radar_sim.grc:801-897, :2184-2188 -> lib/zero_pad_impl.cc:76-90):
 * d_in = the precoder's output [n_frames][n_ports][n_symbols][fft_len]; burst (port t, frame f) = [pad_front noise | n_symbols x (cp | symbol) |
 * pad_tail noise] is written at d_out + t*out_port_stride + f*out_burst_stride (items).  Samples and padding are bit-identical to
 * jrc_ofdm_mod_dev followed by jrc_zero_pad_strided_dev per port with seed + t*seed_port_step; the unpadded time-domain packet is never
 * written.  fft_len: a power of two in [4, 8192] (else JRC_ERR_UNSUPPORTED: use the two calls).  Returns the burst length in items. */
int jrc_ofdm_mod_pad_dev(jrc_ctx* ctx, int fft_len, int cp_len, const float* d_window, int n_frames, int n_ports, int n_symbols,
                         unsigned pad_front, unsigned pad_tail, uint64_t seed, uint64_t seed_port_step,
                         const jrc_cf32* d_in, jrc_cf32* d_out, long out_port_stride, long out_burst_stride, void* stream);

/* batched, device-resident form of the whole front end (detection metrics -> frame_detector -> frame_sync run to completion on
 * one capture): frame k of the capture lands in row k of d_frames ([max_frames][max_symbols * fft_len] time-domain samples,
 * cyclic prefixes removed, de-rotated, first two symbols = the long training field as frame_sync delivers it).  d_work:
 * jrc_sync_frontend_work_bytes(n_samples) bytes of scratch whose contents afterwards are unspecified (the metric streams only pass through it
 * when the windows do not fit an LDS tile; otherwise the peak mask is all that is written). */
typedef struct {
    int fft_len, cp_len;
    double threshold;                 /* frame_detector */
    unsigned min_n_peaks, ignore_gap;
    int sync_length, n_taps;          /* frame_sync */
    const jrc_cf32* d_ltf_taps;       /* [n_taps] on the device */
    int delay, window, power_window;  /* stock metric blocks: blocks_delay, moving_avg length, moving_average_ff length */
    float power_scale;                /* moving_average_ff scale */
} jrc_sync_cfg;
typedef struct {
    int start, len;                   /* first sample of the (delayed) capture the detector copied, samples copied */
    float coarse_cfo;                 /* frame_detector's tag value */
    int frame_start;                  /* frame_sync d_frame_start */
    float fine_cfo;                   /* frame_sync d_freq_offset */
    double tag_value;                 /* frame_sync's frame_start tag: coarse - fine */
    int n_out;                        /* samples written to the row (a multiple of fft_len) */
    int pad_;
} jrc_sync_frame;
size_t jrc_sync_frontend_work_bytes(int n_samples);
/* frame_detector alone, run to completion on metric streams already on the device (d_marks: n_samples/64 + 2 words of scratch) */
int jrc_frame_detector_scan_dev(jrc_ctx* ctx, int fft_len, int cp_len, double threshold, unsigned min_n_peaks, unsigned ignore_gap,
                                int n_samples, const jrc_cf32* d_in_abs, const float* d_in_cor, unsigned long long* d_marks, int max_frames,
                                jrc_sync_frame* d_info, int* d_n_frames, void* stream);
int jrc_sync_frontend_dev(jrc_ctx* ctx, const jrc_sync_cfg* cfg, int n_samples, const jrc_cf32* d_x, jrc_cf32* d_work, int max_frames,
                          int max_symbols, jrc_cf32* d_frames, jrc_sync_frame* d_info, int* d_n_frames, void* stream);

/* ---- ofdm_frame_generator (lib/ofdm_frame_generator_impl.cc:55-216; make(fft_len, occupied_carriers, pilot_carriers, pilot_symbols,
 * sync_words, ltf_len, len_tag_key, output_is_shifted)): the SISO carrier allocator.  Carrier / pilot-symbol sets are passed flattened
 * with per-set sizes, indices as given to make().  create returns NULL for the constructor's std::invalid_argument cases. ---- */
typedef struct jrc_frame_generator jrc_frame_generator;
jrc_frame_generator* jrc_frame_generator_create(jrc_ctx* ctx, int fft_len, int n_occ_sets, const int* occ_sizes, const int* occ_flat,
                                                int n_pil_sets, const int* pil_sizes, const int* pil_flat, int n_psym_sets,
                                                const int* psym_sizes, const jrc_cf32* psym_flat, int n_sync, const jrc_cf32* sync_words,
                                                int output_is_shifted);
void jrc_frame_generator_destroy(jrc_frame_generator* g);
int jrc_frame_generator_output_length(const jrc_frame_generator* g, int ninput_items);      /* calculate_output_stream_length */
/* work(): ninput_items symbols -> output_length vectors of fft_len; returns the vectors written */
int jrc_frame_generator_work(jrc_frame_generator* g, int ninput_items, const jrc_cf32* in, jrc_cf32* out);
int jrc_frame_generator_dev(jrc_frame_generator* g, int n_packets, int ninput_items, const jrc_cf32* d_in, jrc_cf32* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JRC_H */
