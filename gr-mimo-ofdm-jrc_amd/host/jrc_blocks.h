// jrc_blocks.h — host-side blocks with the reference's public interface (namespace, class names, make() signatures
// and setters of /root/reference/include/mimo_ofdm_jrc/*.h), implemented over the C ABI of include/jrc.h.
//
//   reference header                                   class here
//   include/mimo_ofdm_jrc/mimo_ofdm_radar.h:35-64      gr::mimo_ofdm_jrc::mimo_ofdm_radar
//   include/mimo_ofdm_jrc/matrix_transpose.h           gr::mimo_ofdm_jrc::matrix_transpose
//   include/mimo_ofdm_jrc/range_angle_estimator.h      gr::mimo_ofdm_jrc::range_angle_estimator
//   include/mimo_ofdm_jrc/ofdm_cyclic_prefix_remover.h gr::mimo_ofdm_jrc::ofdm_cyclic_prefix_remover
//   include/mimo_ofdm_jrc/fft_peak_detect.h            gr::mimo_ofdm_jrc::fft_peak_detect
//   include/mimo_ofdm_jrc/mimo_ofdm_equalizer.h:64-78  gr::mimo_ofdm_jrc::mimo_ofdm_equalizer
//   include/mimo_ofdm_jrc/mimo_precoder.h              gr::mimo_ofdm_jrc::mimo_precoder
//   include/mimo_ofdm_jrc/target_simulator.h:30-60     gr::mimo_ofdm_jrc::target_simulator
//   include/mimo_ofdm_jrc/stream_encoder.h:56-64       gr::mimo_ofdm_jrc::stream_encoder
//   include/mimo_ofdm_jrc/stream_decoder.h:43-56       gr::mimo_ofdm_jrc::stream_decoder
//   include/mimo_ofdm_jrc/moving_avg.h:44-55           gr::mimo_ofdm_jrc::moving_avg
//   include/mimo_ofdm_jrc/frame_detector.h:44-49       gr::mimo_ofdm_jrc::frame_detector
//   include/mimo_ofdm_jrc/frame_sync.h:44-49           gr::mimo_ofdm_jrc::frame_sync
//   include/mimo_ofdm_jrc/zero_pad.h                   gr::mimo_ofdm_jrc::zero_pad
//   include/mimo_ofdm_jrc/ofdm_frame_generator.h       gr::mimo_ofdm_jrc::ofdm_frame_generator
//
//   (none: the five-block radar branch as one block)   gr::mimo_ofdm_jrc::radar_chain
//
// Built against GNU Radio 3.8 with -DJRC_WITH_GNURADIO; otherwise against the stand-alone test runtime.
#pragma once

#include <string>
#include <vector>

#include "jrc_block_runtime.h"

// export decoration of the reference's public classes (include/mimo_ofdm_jrc/api.h:27-31): the GNU Radio build takes it from
// <gnuradio/attributes.h>, exactly as api.h does; the stand-alone test runtime needs none
#ifndef MIMO_OFDM_JRC_API
#ifdef JRC_WITH_GNURADIO
#include <gnuradio/attributes.h>
#ifdef gnuradio_mimo_ofdm_jrc_EXPORTS
#define MIMO_OFDM_JRC_API __GR_ATTR_EXPORT
#else
#define MIMO_OFDM_JRC_API __GR_ATTR_IMPORT
#endif
#else
#define MIMO_OFDM_JRC_API
#endif
#endif

// the global enums of the reference's public headers (mimo_ofdm_equalizer.h:27-36, stream_encoder.h:27-39)
enum ChannelEstimator { LS = 0, STA = 1 };
enum Modulation { BPSK = 0, QPSK = 1, QAM16 = 2 };
enum MCS : uint8_t { BPSK_1_2 = 0, BPSK_3_4 = 1, QPSK_1_2 = 2, QPSK_3_4 = 3, QAM16_1_2 = 4, QAM16_3_4 = 5 };
enum PACKET_TYPE : uint8_t { NDP = 1, DATA = 2 };

namespace gr {
namespace mimo_ofdm_jrc {

class MIMO_OFDM_JRC_API mimo_ofdm_radar : virtual public jrc_rt::block {
public:
    typedef JRC_SPTR<mimo_ofdm_radar> sptr;
    static sptr make(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, bool background_removal,
                     bool background_recording, int record_len, int interp_factor, bool enable_tx_interleave,
                     const std::string& radar_chan_file, const std::string& len_tag_key = "packet_len", bool debug = false);
    virtual void set_background_record(bool background_record) = 0;
    virtual void capture_radar_data(bool capture_sig) = 0;
};

// The radar branch of the flowgraphs as ONE block: mimo_ofdm_radar -> fft_vxx(reverse) -> matrix_transpose -> fft_vxx(forward,
// shift) -> range_angle_estimator (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2189-2197).  Same input ports and tags as
// mimo_ofdm_radar (N_tx + N_rx streams of fft_len vectors, `packet_len`), same `params` message port and log file as
// range_angle_estimator; no stream output — the range-angle map stays on the device.  Every frame offered in a scheduler turn goes
// through the host-fed pipeline (jrc_chain_feed_*): batches in flight on their own streams, results published in frame order.
// defaults of radar_chain::make (a value <= 0 selects them; the frames per batch are capped so that a slot's pinned staging buffer stays within 64 MiB:
// 64 frames at the .grc's shape and at config B, 7 at config D's 8.4 MB frames): a batch is whatever a scheduler turn offers up to frames_per_batch, so a larger
// value costs nothing at low packet rates; sized on the reference flowgraph's own 27 KB packets (bench.py, host_fed_reference_flowgraph_shape:
// 16 x 3 -> 335 k packets/s, 32 x 3 -> 478 k, 64 x 3 -> 853 k with 64 packets per scheduler turn, round 5; a batch is one H2D + five kernels
// + one D2H, replayed from a hipGraph, whatever it holds)
#define RADAR_CHAIN_DEFAULT_FRAMES_PER_BATCH 64
#define RADAR_CHAIN_DEFAULT_BATCHES_IN_FLIGHT 3
class MIMO_OFDM_JRC_API radar_chain : virtual public jrc_rt::block {
public:
    typedef JRC_SPTR<radar_chain> sptr;
    static sptr make(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int interp_range, int interp_angle, bool enable_tx_interleave,
                     std::vector<float> range_bins, std::vector<float> angle_bins, float noise_discard_range_m,
                     float noise_discard_angle_deg, float snr_threshold, float power_threshold, const std::string& stats_path,
                     bool stats_record,
                     int frames_per_batch = 0,            // 0 = automatic: up to RADAR_CHAIN_DEFAULT_FRAMES_PER_BATCH, a slot of at most 64 MiB
                     int batches_in_flight = RADAR_CHAIN_DEFAULT_BATCHES_IN_FLIGHT,
                     const std::string& len_tag_key = "packet_len", bool debug = false,
                     // mimo_ofdm_radar's background arguments (include/mimo_ofdm_jrc/mimo_ofdm_radar.h:52-56)
                     bool background_removal = false, bool background_recording = false, int record_len = 0);
    virtual int frames_done() const = 0;
    virtual void set_background_record(bool background_record) = 0;      // mimo_ofdm_radar::set_background_record
    virtual int n_devices() const = 0;                                     // GPUs the block feeds (environment JRC_DEVICES=0,1,...)
    // batches stay in flight across general_work calls and are published in frame order as they finish (a batch older than
    // JRC_RADAR_CHAIN_MAX_AGE_US, default 2000, is waited for at the end of a call; stop() publishes the rest): flush() does so now
    virtual void flush() = 0;
    virtual int pending_batches() const = 0;
    virtual long rx_only_batches() const = 0;                              // batches whose TX reference rows were already on the device
    // where general_work's time went so far, ns: 0 staging (TX row compare + copies into the pinned slot), 1 submit calls into the feed,
    // 2 collecting + publishing finished batches, 3 the whole of general_work
    virtual long profile_ns(int what) const = 0;
};

class MIMO_OFDM_JRC_API matrix_transpose : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<matrix_transpose> sptr;
    static sptr make(int input_len, int output_len, int interp_factor, bool debug, std::string len_key = "packet_len");
};

class MIMO_OFDM_JRC_API range_angle_estimator : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<range_angle_estimator> sptr;
    static sptr make(int vlen, std::vector<float> range_bins, std::vector<float> angle_bins, float noise_discard_range_m,
                     float noise_discard_angle_deg, float snr_threshold, float power_threshold,
                     const std::string& stats_path, bool stats_record, const std::string& len_key = "packet_len",
                     bool debug = false);
    virtual void set_snr_threshold(float snr_threshold) = 0;
    virtual void set_power_threshold(float power_threshold) = 0;
    virtual void set_stats_record(bool stats_record) = 0;
};

class MIMO_OFDM_JRC_API ofdm_cyclic_prefix_remover : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<ofdm_cyclic_prefix_remover> sptr;
    static sptr make(int fft_len, int cp_len, std::string len_key = "packet_len");
};

class MIMO_OFDM_JRC_API fft_peak_detect : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<fft_peak_detect> sptr;
    static sptr make(int samp_rate, float interp_factor, float threshold, int samp_protect, std::vector<float> max_freq,
                     bool cut_max_freq, const std::string& len_key);
    virtual void set_threshold(float threshold) = 0;
    virtual void set_samp_protect(int samp) = 0;
    virtual void set_max_freq(std::vector<float> freq) = 0;
};

class MIMO_OFDM_JRC_API mimo_ofdm_equalizer : virtual public jrc_rt::block {
public:
    typedef JRC_SPTR<mimo_ofdm_equalizer> sptr;
    virtual void set_estimator(ChannelEstimator algo) = 0;
    virtual void set_bandwidth(double bw) = 0;
    virtual void set_frequency(double freq) = 0;
    virtual void set_stats_record(bool stats_record) = 0;
    static sptr make(ChannelEstimator estimator_algo, double freq, double bw, int fft_len, int cp_len,
                     std::vector<int> data_carriers, std::vector<int> pilot_carriers,
                     const std::vector<std::vector<gr_complex>>& pilot_symbols, std::vector<gr_complex> long_seq,
                     const std::vector<std::vector<gr_complex>>& mapped_ltf_symbols, int n_mimo_ltf,
                     const std::string& chan_est_file, const std::string& comm_log_file, bool stats_record, bool debug);
};

class MIMO_OFDM_JRC_API mimo_precoder : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<mimo_precoder> sptr;
    static sptr make(int fft_len, int N_tx, int N_ss, const std::vector<int>& data_carriers,
                     const std::vector<int>& pilot_carriers, const std::vector<std::vector<gr_complex>>& pilot_symbols,
                     const std::vector<std::vector<gr_complex>>& sync_words,
                     const std::vector<std::vector<gr_complex>>& mapped_ltf_symbols, const std::string& chan_est_file,
                     bool chan_est_smoothing, const std::string& radar_log_file, bool radar_aided, bool phased_steering,
                     bool use_radar_streams, const std::string& len_tag_key = "packet_len", bool debug = false);
    virtual void set_chan_est_smoothing(bool chan_est_smoothing) = 0;
    virtual void set_radar_aided(bool radar_aided) = 0;
    virtual void set_use_radar_streams(bool use_radar_streams) = 0;
    virtual void set_phased_steering(bool phased_steering) = 0;
};

class MIMO_OFDM_JRC_API target_simulator : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<target_simulator> sptr;
    static sptr make(std::vector<float> range, std::vector<float> velocity, std::vector<float> rcs, std::vector<float> azimuth,
                     std::vector<float> position_rx, int samp_rate, float center_freq, float self_coupling_db,
                     bool rndm_phaseshift = false, bool self_coupling = false, const std::string& len_key = "packet_len",
                     bool debug = false);
    virtual void setup_targets(std::vector<float> range, std::vector<float> velocity, std::vector<float> rcs,
                               std::vector<float> azimuth, std::vector<float> position_rx, int samp_rate, float center_freq,
                               float self_coupling_db, bool rndm_phaseshift, bool self_coupling) = 0;
};

class MIMO_OFDM_JRC_API stream_encoder : virtual public jrc_rt::block {
public:
    typedef JRC_SPTR<stream_encoder> sptr;
    static sptr make(MCS mod_encode, int data_len, int N_ss_radar, bool debug);
    virtual void set_mcs(MCS mod_encode) = 0;
};

class MIMO_OFDM_JRC_API stream_decoder : virtual public jrc_rt::block {
public:
    typedef JRC_SPTR<stream_decoder> sptr;
    static sptr make(int n_data_carriers, const std::string& comm_log_file, bool stats_record, bool debug);
    virtual void set_stats_record(bool stats_record) = 0;
};

class MIMO_OFDM_JRC_API moving_avg : virtual public jrc_rt::sync_block {
public:
    typedef JRC_SPTR<moving_avg> sptr;
    static sptr make(int length, float scale, int max_iter, bool debug);
    virtual int length() const = 0;
    virtual float scale() const = 0;
    virtual void set_length_and_scale(int length, float scale) = 0;
    virtual void set_length(int length) = 0;
    virtual void set_scale(float scale) = 0;
};

class MIMO_OFDM_JRC_API ofdm_frame_generator : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<ofdm_frame_generator> sptr;
    virtual std::string len_tag_key() = 0;
    virtual const int fft_len() = 0;
    virtual std::vector<std::vector<int>> occupied_carriers() = 0;
    static sptr make(int fft_len, const std::vector<std::vector<int>>& occupied_carriers, const std::vector<std::vector<int>>& pilot_carriers,
                     const std::vector<std::vector<gr_complex>>& pilot_symbols, const std::vector<std::vector<gr_complex>>& sync_words,
                     int ltf_len, const std::string& len_tag_key = "packet_len", const bool output_is_shifted = true);
};

class MIMO_OFDM_JRC_API zero_pad : virtual public jrc_rt::tagged_stream_block {
public:
    typedef JRC_SPTR<zero_pad> sptr;
    static sptr make(bool debug, unsigned int pad_front, unsigned int pad_tail);
};

class MIMO_OFDM_JRC_API frame_detector : virtual public jrc_rt::block {
public:
    typedef JRC_SPTR<frame_detector> sptr;
    static sptr make(int fft_len, int cp_len, double threshold, unsigned int min_n_peaks, unsigned int ignore_gap, bool debug);
};

class MIMO_OFDM_JRC_API frame_sync : virtual public jrc_rt::block {
public:
    typedef JRC_SPTR<frame_sync> sptr;
    static sptr make(int fft_len, int cp_len, unsigned int sync_length, std::vector<gr_complex> ltf_seq_time, bool debug);
};

}  // namespace mimo_ofdm_jrc
}  // namespace gr
