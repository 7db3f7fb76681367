// jrc_blocks.cc — work()/general_work() bodies of the reference blocks (the seven hot-path blocks of SURVEY 8a, then the 8(f) blocks) over the C ABI (include/jrc.h).
// Tag, message and file handling is host code that follows the reference line by line (citations are to
// /root/reference/lib); all sample arithmetic happens in the HIP kernels behind the jrc_* calls.
#include "jrc_blocks.h"

#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <thread>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <random>

#include "../../include/jrc.h"

namespace gr {
namespace mimo_ofdm_jrc {

namespace {

int device_from_env()
{
    const char* d = getenv("JRC_DEVICE");
    return d ? atoi(d) : 0;
}

// JRC_DEVICES=0,1,2,...: the GPUs a frame-parallel block deals its batches over (empty / unset: the one of JRC_DEVICE)
std::vector<int> devices_from_env()
{
    std::vector<int> out;
    const char* d = getenv("JRC_DEVICES");
    if (d)
        for (const char* p = d; *p;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            out.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
    if (out.empty()) out.push_back(device_from_env());
    return out;
}

// one context per block = per scheduler thread
struct ctx_holder {
    jrc_ctx* ctx = nullptr;
    ctx_holder()
    {
        int st = jrc_create(device_from_env(), &ctx);
        if (st != JRC_OK) throw std::runtime_error(std::string("jrc_create: ") + jrc_strerror(st));
    }
    ~ctx_holder() { jrc_destroy(ctx); }
    static void raise(int st, std::string msg)
    {
        if (msg.empty()) msg = jrc_strerror(st);
        if (st == JRC_ERR_INVALID_ARG) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }
    void check(int st) const { if (st < 0) raise(st, jrc_last_error(ctx)); }
    // calls on a feed report through the feed (a multi-GPU feed owns its contexts: this block's context knows nothing of their failures)
    static void check_feed(const jrc_chain_feed* fd, int st) { if (st < 0) raise(st, jrc_chain_feed_last_error(fd)); }
};

// Wall-clock stamps of the log files.  Only the two formats are the reference's wire format (lib/utils.cc:302 "%m-%d-%Y %H:%M:%S" for the
// "NEW RECORD" header lines, :316-317 "%H:%M:%S" + ".mmm" for the per-frame lines read back by mimo_precoder, lib/mimo_precoder_impl.cc:913-953).
std::string wall_clock(const char* fmt, bool with_millis)
{
    using namespace std::chrono;
    const system_clock::time_point tp = system_clock::now();
    const std::time_t secs = system_clock::to_time_t(tp);
    std::tm parts;
    localtime_r(&secs, &parts);
    char text[96];
    size_t n = std::strftime(text, sizeof(text) - 8, fmt, &parts);
    if (with_millis) {
        const long ms = (long)(duration_cast<milliseconds>(tp.time_since_epoch()).count() % 1000);
        n += (size_t)std::snprintf(text + n, 8, ".%03ld", ms);
    }
    return std::string(text, n);
}
inline std::string current_date_time() { return wall_clock("%m-%d-%Y %H:%M:%S", false); }
inline std::string current_date_time2() { return wall_clock("%H:%M:%S", true); }

// The payload of a message on a PDU port: either a pmt symbol (its characters) or a pair whose cdr is a blob — the two forms
// stream_encoder accepts (lib/stream_encoder_impl.cc:103-125); anything else is refused with the reference's message.
struct pdu_view {
    std::string owned;                   // keeps a symbol's characters alive
    const uint8_t* bytes = nullptr;
    int size = 0;
    explicit pdu_view(const pmt::pmt_t& msg)
    {
        if (pmt::is_symbol(msg)) {
            owned = pmt::symbol_to_string(msg);
            bytes = reinterpret_cast<const uint8_t*>(owned.data());
            size = (int)owned.size();
        } else if (pmt::is_pair(msg)) {
            const pmt::pmt_t blob = pmt::cdr(msg);
            bytes = static_cast<const uint8_t*>(pmt::blob_data(blob));
            size = (int)pmt::blob_length(blob);
        } else {
            throw std::invalid_argument("[STREAM ENCODER] Encoder expects PDUs or strings");
        }
    }
    int first_byte() const { return size ? bytes[0] : 0; }      // the packet type travels in the first payload byte (:110, :117)
};

}  // namespace

// =================================================================================================
// mimo_ofdm_radar  (lib/mimo_ofdm_radar_impl.cc)
// =================================================================================================
class mimo_ofdm_radar_impl : public mimo_ofdm_radar {
    ctx_holder d_c;
    jrc_radar* d_radar = nullptr;
    int d_fft_len, d_N_tx, d_N_rx, d_N_sym, d_N_pre, d_interp_factor;
    bool new_radar_frame = false;   // uninitialised in the reference (lib/mimo_ofdm_radar_impl.h:59)
    std::string d_radar_chan_file;
    std::vector<gr_complex> d_last_est;   // radar_chan_est of the last frame, [P][fft_len]

public:
    mimo_ofdm_radar_impl(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, bool background_removal,
                         bool background_recording, int record_len, int interp_factor, bool enable_tx_interleave,
                         const std::string& radar_chan_file, const std::string&, bool)
        : jrc_rt::block("mimo_ofdm_radar", jrc_rt::io_signature::make(N_tx + N_rx, N_tx + N_rx, sizeof(gr_complex) * fft_len),
                        jrc_rt::io_signature::make(1, 1, sizeof(gr_complex) * fft_len * interp_factor)),   // :81-83
          d_fft_len(fft_len), d_N_tx(N_tx), d_N_rx(N_rx), d_N_sym(N_sym), d_N_pre(N_pre), d_interp_factor(interp_factor),
          d_radar_chan_file(radar_chan_file), d_last_est((size_t)N_tx * N_rx * fft_len)          // vector::resize value-initialises (:115)
    {
        d_c.check(jrc_radar_create(d_c.ctx, fft_len, N_tx, N_rx, N_sym, N_pre, background_removal, background_recording,
                                   record_len, interp_factor, enable_tx_interleave, &d_radar));
        set_tag_propagation_policy(TPP_DONT);
    }
    ~mimo_ofdm_radar_impl() override { jrc_radar_destroy(d_radar); }

    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override
    {
        std::vector<jrc_rt::tag_t> rx_tags, tx_tags;
        uint64_t rx_packet_len = 0, tx_packet_len = 0, n_tx_samples_discard = 0;
        get_tags_in_range(rx_tags, d_N_tx, nitems_read(d_N_tx), nitems_read(d_N_tx) + ninput_items[d_N_tx],
                          pmt::mp("packet_len"));                                                    // :167
        if (!rx_tags.empty()) {
            get_tags_in_range(tx_tags, 0, nitems_read(0), nitems_read(0) + ninput_items[0], pmt::mp("packet_len"));
            new_radar_frame = true;
            size_t tx_frame_offset = 0;
            if (tx_tags.size() > rx_tags.size()) {                                                    // :191-198
                tx_frame_offset = tx_tags.size() - rx_tags.size();
                for (size_t i = 0; i < tx_frame_offset; i++) n_tx_samples_discard += pmt::to_uint64(tx_tags[i].value);
            }
            if (tx_tags.size() <= tx_frame_offset) throw std::runtime_error("[MIMO OFDM RADAR] no packet_len tag on TX input");
            rx_packet_len = pmt::to_uint64(rx_tags[0].value);
            tx_packet_len = pmt::to_uint64(tx_tags[tx_frame_offset].value);
        }
        if (!new_radar_frame) {                                                                       // :219-234
            for (int i = 0; i < d_N_tx + d_N_rx; i++) consume(i, ninput_items[i]);
            return 0;
        }
        std::vector<const jrc_cf32*> tx(d_N_tx), rx(d_N_rx);
        for (int t = 0; t < d_N_tx; t++) tx[t] = (const jrc_cf32*)input_items[t];
        for (int r = 0; r < d_N_rx; r++) rx[r] = (const jrc_cf32*)input_items[d_N_tx + r];
        int n = jrc_radar_work(d_radar, tx.data(), rx.data(), ninput_items[0], ninput_items[d_N_tx], n_tx_samples_discard,
                               (jrc_cf32*)output_items[0]);
        d_c.check(n);
        (void)noutput_items;
        for (int p = 0; p < d_N_tx * d_N_rx; p++)                                                     // radar_chan_est, kept for capture_radar_data
            memcpy(&d_last_est[(size_t)p * d_fft_len], (const gr_complex*)output_items[0] + (size_t)p * d_fft_len * d_interp_factor,
                   sizeof(gr_complex) * d_fft_len);
        add_item_tag(0, nitems_written(0), pmt::string_to_symbol("packet_len"), pmt::from_long(n),
                     pmt::string_to_symbol(alias()));                                                 // :303-309
        for (int r = 0; r < d_N_rx; r++) consume(r + d_N_tx, (int)rx_packet_len);                    // :326-334
        for (int t = 0; t < d_N_tx; t++) consume(t, (int)(n_tx_samples_discard + tx_packet_len));
        new_radar_frame = false;
        return n;
    }
    void set_background_record(bool b) override { jrc_radar_set_background_record(d_radar, b); }      // :342-346
    // radar_chan.csv (:348-387): one line per capture, "HH:MM:SS.mmm, N_tx, N_rx, fft_len:(re,im);(re,im);...;" + an empty line —
    // Eigen's IOFormat(FullPrecision, DontAlignCols, ";", ":", "", "", "", ";\n") of the P*fft_len row vector.  Eigen prints 6-7
    // significant digits depending on its version; 9 are written here (float round trip), as for chan_est.csv.
    void capture_radar_data(bool capture_sig) override
    {
        if (!capture_sig) return;
        std::ofstream f(d_radar_chan_file, std::ofstream::app);
        if (!f.is_open()) throw std::runtime_error("[MIMO OFDM RADAR] Could not open file!!");
        f << current_date_time2() << ", " << d_N_tx << ", " << d_N_rx << ", " << d_fft_len << ":";
        char buf[96];
        for (size_t i = 0; i < d_last_est.size(); i++) {
            snprintf(buf, sizeof(buf), "%s(%.9g,%.9g)", i ? ";" : "", d_last_est[i].real(), d_last_est[i].imag());
            f << buf;
        }
        f << ";\n" << "\n";
        f.flush();
        std::cout << "[MIMO OFDM RADAR] Radar image captured!" << std::endl;
    }
};

mimo_ofdm_radar::sptr mimo_ofdm_radar::make(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, bool background_removal,
                                            bool background_recording, int record_len, int interp_factor,
                                            bool enable_tx_interleave, const std::string& radar_chan_file,
                                            const std::string& len_tag_key, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new mimo_ofdm_radar_impl(fft_len, N_tx, N_rx, N_sym, N_pre, background_removal,
                                                         background_recording, record_len, interp_factor,
                                                         enable_tx_interleave, radar_chan_file, len_tag_key, debug));
}

// =================================================================================================
// matrix_transpose  (lib/matrix_transpose_impl.cc)
// =================================================================================================
class matrix_transpose_impl : public matrix_transpose {
    ctx_holder d_c;
    int d_input_len, d_output_len, d_interp_factor;

public:
    matrix_transpose_impl(int input_len, int output_len, int interp_factor, bool, const std::string& len_key)
        : jrc_rt::tagged_stream_block("matrix_transpose", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex) * input_len),
                                      jrc_rt::io_signature::make(1, 1, sizeof(gr_complex) * output_len * interp_factor), len_key),
          d_input_len(input_len), d_output_len(output_len), d_interp_factor(interp_factor)
    {
        set_relative_rate((double)input_len / (double)output_len);
        set_tag_propagation_policy(TPP_DONT);
    }
    int calculate_output_stream_length(const gr_vector_int&) override { return d_input_len; }          // :62-67
    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star& out) override
    {
        int n = jrc_matrix_transpose(d_c.ctx, d_input_len, d_output_len, d_interp_factor, ninput_items[0],
                                     (const jrc_cf32*)in[0], (jrc_cf32*)out[0]);
        if (n == JRC_ERR_LENGTH_MISMATCH) throw std::runtime_error(jrc_strerror(n));                 // :82-83
        d_c.check(n);
        return n;
    }
};
matrix_transpose::sptr matrix_transpose::make(int input_len, int output_len, int interp_factor, bool debug, std::string len_key)
{
    return JRC_GET_INITIAL_SPTR(new matrix_transpose_impl(input_len, output_len, interp_factor, debug, len_key));
}

// =================================================================================================
// range_angle_estimator  (lib/range_angle_estimator_impl.cc)
// =================================================================================================
class range_angle_estimator_impl : public range_angle_estimator {
    ctx_holder d_c;
    int d_vlen;
    std::vector<float> d_range_bins, d_angle_bins;
    float d_ndr, d_nda, d_snr_threshold, d_power_threshold;
    std::string d_stats_path;
    bool d_stats_record, d_new_stat_started = false;

    static pmt::pmt_t pack(const char* key, float v) { return pmt::list2(pmt::string_to_symbol(key), pmt::init_f32vector(1, &v)); }

public:
    range_angle_estimator_impl(int vlen, std::vector<float> range_bins, std::vector<float> angle_bins, float ndr, float nda,
                               float snr_threshold, float power_threshold, const std::string& stats_path, bool stats_record,
                               const std::string& len_key, bool)
        : jrc_rt::tagged_stream_block("range_angle_estimator", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex) * vlen),
                                      jrc_rt::io_signature::make(0, 0, 0), len_key),
          d_vlen(vlen), d_range_bins(range_bins), d_angle_bins(angle_bins), d_ndr(ndr), d_nda(nda),
          d_snr_threshold(snr_threshold), d_power_threshold(power_threshold), d_stats_path(stats_path), d_stats_record(stats_record)
    {
        message_port_register_out(pmt::mp("params"));                                                 // :89
        std::ofstream f(d_stats_path, std::ofstream::app);                                            // :93-97
        if (!f.is_open()) std::cerr << "[RANGE-ANGLE ESTIMATOR] Could not open log file at " << d_stats_path << std::endl;
    }
    int calculate_output_stream_length(const gr_vector_int&) override { return 0; }                   // :114-119
    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star&) override
    {
        jrc_ra_result r;
        d_c.check(jrc_ra_estimate(d_c.ctx, d_vlen, ninput_items[0], (const jrc_cf32*)in[0], d_range_bins.data(),
                                  (int)d_range_bins.size(), d_angle_bins.data(), (int)d_angle_bins.size(), d_ndr, d_nda,
                                  d_snr_threshold, d_power_threshold, &r));
        if (r.published) {                                                                            // :234-253
            message_port_pub(pmt::mp("params"), pmt::list4(pack("range", r.range_val), pack("angle", r.angle_val),
                                                           pack("power", r.peak_power), pack("snr", r.snr_est)));
            if (d_stats_record) {                                                                     // :255-279
                std::ofstream f(d_stats_path, std::ofstream::app);
                if (!f.is_open()) throw std::runtime_error("[STREAM DECODER] Could not open file!!");
                if (!d_new_stat_started) { f << "\n NEW RECORD - " << current_date_time() << "\n"; d_new_stat_started = true; }
                f << current_date_time2() << ", \t" << r.peak_power << ", \t" << r.snr_est << ", \t" << r.range_val << ", \t"
                  << r.angle_val << "\n";
            }
        }
        return 0;
    }
    void set_snr_threshold(float v) override { d_snr_threshold = v; }
    void set_power_threshold(float v) override { d_power_threshold = v; }
    void set_stats_record(bool v) override { d_stats_record = v; d_new_stat_started = false; }        // :289-302
};
range_angle_estimator::sptr range_angle_estimator::make(int vlen, std::vector<float> range_bins, std::vector<float> angle_bins,
                                                        float ndr, float nda, float snr_threshold, float power_threshold,
                                                        const std::string& stats_path, bool stats_record,
                                                        const std::string& len_key, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new range_angle_estimator_impl(vlen, range_bins, angle_bins, ndr, nda, snr_threshold,
                                                               power_threshold, stats_path, stats_record, len_key, debug));
}

// =================================================================================================
// radar_chain: mimo_ofdm_radar + fft_vxx + matrix_transpose + fft_vxx + range_angle_estimator in one block, over the host-fed
// pipeline of the device-resident chain (jrc_chain_feed_*).  Input side = lib/mimo_ofdm_radar_impl.cc:160-238 (tags, stale TX
// packets discarded), output side = lib/range_angle_estimator_impl.cc:234-279 (message + log line per published frame).
// =================================================================================================
class radar_chain_impl : public radar_chain {
    ctx_holder d_c;
    jrc_chain_feed* d_feed = nullptr;
    int d_fft_len, d_N_tx, d_N_rx, d_N_sym, d_N_pre, d_n_items, d_fpb, d_slots;
    std::string d_stats_path;
    bool d_stats_record, d_new_stat_started = false;
    int d_n_devices = 1, d_record_len = 0;
    // read by the getters from any thread while the scheduler's thread or the flusher advances them under d_setlock
    std::atomic<int> d_frames_done{0}, d_pending{0};
    std::atomic<long> d_rx_only_batches{0};
    std::atomic<long> d_prof_ns[4] = {{0}, {0}, {0}, {0}};          // profile_ns(): staging, submit, collect + publish, whole general_work
    struct prof_scope {                                            // (two steady_clock reads, ~50 ns, per scope)
        std::atomic<long>& acc; std::chrono::steady_clock::time_point t0;
        explicit prof_scope(std::atomic<long>& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
        ~prof_scope() { acc.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed); }
    };
    bool d_bg_removal = false;
    std::vector<jrc_ra_result> d_res;
    // batches stay in flight across general_work calls (at most d_slots), published in frame order as they complete; one older than
    // d_max_age_us is waited for at the end of a call, stop() / flush() wait for all.  JRC_RADAR_CHAIN_MAX_AGE_US, 0 = every call drains.
    std::deque<std::chrono::steady_clock::time_point> d_submitted;
    long d_max_age_us = 2000;
    // the age bound holds while the scheduler is idle too: a thread of the block's own wakes every half bound and publishes what is overdue
    // (message_port_pub from a block's own thread is what gr::blocks::socket_pdu and friends do); every touch of the feed is under d_setlock
    std::thread d_flusher;
    std::atomic<bool> d_flusher_stop{false};
    void flusher_main()
    {
        const auto nap = std::chrono::microseconds(std::max<long>(100, d_max_age_us / 2));
        while (!d_flusher_stop.load()) {
            std::this_thread::sleep_for(nap);
            if (d_flusher_stop.load()) break;
            jrc_rt::thread::scoped_lock guard(d_setlock);
            // (the lock is held while an OVERDUE batch is waited for — at most the rest of one batch's device time, well under the age bound;
            // batches that are merely in flight are polled, never waited for)
            try { if (d_feed && jrc_chain_feed_pending(d_feed) > 0) collect_ready(false); }
            catch (const std::exception& e) {
                // a failed collect or publish (HIP error, log file not writable) will fail again: say it once, remember it for the scheduler's
                // thread — the next general_work / flush rethrows it, as it would have had it collected the batch itself — and stop retrying
                std::cerr << "[RADAR CHAIN] " << e.what() << std::endl;
                d_flusher_error = e.what();
                d_flusher_failed.store(true);
                break;
            }
        }
    }
    std::atomic<bool> d_flusher_failed{false};
    std::string d_flusher_error;                                    // written once by the flusher before d_flusher_failed, read after it
    void rethrow_flusher_error()
    {
        if (d_flusher_failed.load()) throw std::runtime_error("[RADAR CHAIN] (from the flusher thread) " + d_flusher_error);
    }
    // TX-resident submission: the reference rows the radar correlates with (the N_sym symbols behind N_pre of every TX port) are the MIMO-LTFs
    // in the reference's flowgraph, the same for every packet.  The block keeps the rows of the last full submission; a batch whose frames all
    // carry exactly those rows (memcmp, a few KB per frame) uploads its receive ports only.  Rows that keep changing (data symbols inside the
    // radar's window) switch the comparison off for a while.  JRC_RADAR_CHAIN_TX_RESIDENT=0 disables it.
    bool d_tx_res_enabled = true, d_tx_ref_valid = false, d_tx_ref_on_device = false;
    std::vector<jrc_cf32> d_tx_ref;
    int d_tx_misses = 0, d_tx_backoff = 0;
    long d_full_batches = 0;

    static pmt::pmt_t pack(const char* key, float v) { return pmt::list2(pmt::string_to_symbol(key), pmt::init_f32vector(1, &v)); }

    void publish(int n)
    {
        for (int i = 0; i < n; i++) {
            const jrc_ra_result& r = d_res[i];
            d_frames_done++;
            if (!r.published) continue;
            message_port_pub(pmt::mp("params"), pmt::list4(pack("range", r.range_val), pack("angle", r.angle_val),
                                                           pack("power", r.peak_power), pack("snr", r.snr_est)));
            if (d_stats_record) {
                std::ofstream f(d_stats_path, std::ofstream::app);
                if (!f.is_open()) throw std::runtime_error("[STREAM DECODER] Could not open file!!");
                if (!d_new_stat_started) { f << "\n NEW RECORD - " << current_date_time() << "\n"; d_new_stat_started = true; }
                f << current_date_time2() << ", \t" << r.peak_power << ", \t" << r.snr_est << ", \t" << r.range_val << ", \t"
                  << r.angle_val << "\n";
            }
        }
    }
    void collect_one()
    {
        prof_scope ps(d_prof_ns[2]);
        int n = 0;
        ctx_holder::check_feed(d_feed, jrc_chain_feed_collect(d_feed, d_res.data(), nullptr, &n));
        if (!d_submitted.empty()) d_submitted.pop_front();
        d_pending.store(jrc_chain_feed_pending(d_feed));
        publish(n);
    }
    // finished batches, and (wait) those in flight for longer than the age bound; all: everything
    void collect_ready(bool all)
    {
        while (jrc_chain_feed_pending(d_feed) > 0) {
            if (!all && !d_submitted.empty() && jrc_chain_feed_poll(d_feed) != 1) {
                const auto age = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - d_submitted.front()).count();
                if (age <= d_max_age_us) break;
            }
            collect_one();
        }
    }

public:
    radar_chain_impl(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int interp_range, int interp_angle, bool interleave,
                     const std::vector<float>& range_bins, const std::vector<float>& angle_bins, float ndr, float nda, float snr_threshold,
                     float power_threshold, const std::string& stats_path, bool stats_record, int frames_per_batch, int batches_in_flight,
                     bool background_removal, bool background_recording, int record_len)
        : jrc_rt::block("radar_chain", jrc_rt::io_signature::make(N_tx + N_rx, N_tx + N_rx, sizeof(gr_complex) * fft_len),
                        jrc_rt::io_signature::make(0, 0, 0)),
          d_fft_len(fft_len), d_N_tx(N_tx), d_N_rx(N_rx), d_N_sym(N_sym), d_N_pre(N_pre), d_n_items(N_pre + N_sym), d_fpb(frames_per_batch),
          d_slots(batches_in_flight), d_stats_path(stats_path), d_stats_record(stats_record)
    {
        if ((int)range_bins.size() != fft_len * interp_range || (int)angle_bins.size() != N_tx * N_rx * interp_angle)
            throw std::invalid_argument("[RADAR CHAIN] range_bins / angle_bins do not match the map size");
        // Only the N_sym symbols behind the preamble are read by mimo_ofdm_radar (lib/mimo_ofdm_radar_impl.cc:250-274: N_pre + sym): the
        // frames handed to the device hold just those (N_pre = 0, n_items = N_sym).  At the reference flowgraph's point (N_pre 5, N_sym 4)
        // that is 4 of 9 symbols per port and packet across PCIe.
        jrc_chain_cfg cfg;
        cfg.fft_len = fft_len; cfg.N_tx = N_tx; cfg.N_rx = N_rx; cfg.N_sym = N_sym; cfg.N_pre = 0;
        cfg.interp_range = interp_range; cfg.interp_angle = interp_angle; cfg.enable_tx_interleave = interleave;
        cfg.n_items = N_sym; cfg.noise_discard_range_m = ndr; cfg.noise_discard_angle_deg = nda;
        cfg.snr_threshold = snr_threshold; cfg.power_threshold = power_threshold;
        const std::vector<int> devs = devices_from_env();
        // full batches are replayed from a recorded hipGraph (one launch instead of copy + kernels + copy enqueued one by one; measured neutral at the
        // reference flowgraph's 27 KB packets, round 5: what a batch costs is its size); JRC_RADAR_CHAIN_GRAPH=0 submits directly
        const char* ge = getenv("JRC_RADAR_CHAIN_GRAPH");
        const int feed_flags = (ge && atoi(ge) == 0) ? 0 : JRC_FEED_GRAPH;
        if (devs.size() > 1) {         // one host process, several GPUs: batch k on device k mod n, results in frame order (jrc_chain_feed_create_multi)
            if (background_removal || background_recording)
                throw std::invalid_argument("[RADAR CHAIN] background removal keeps the frames of a stream on one GPU: unset JRC_DEVICES");
            int st = jrc_chain_feed_create_multi(devs.data(), (int)devs.size(), &cfg, range_bins.data(), angle_bins.data(), batches_in_flight, d_fpb, 0, feed_flags, &d_feed);
            if (st != JRC_OK) throw std::runtime_error(std::string("[RADAR CHAIN] jrc_chain_feed_create_multi: ") + jrc_strerror(st));
            d_slots = batches_in_flight * (int)devs.size();
        } else {
            d_c.check(jrc_chain_feed_create(d_c.ctx, &cfg, range_bins.data(), angle_bins.data(), d_slots, d_fpb, 0, feed_flags, &d_feed));
        }
        d_n_devices = (int)devs.size();
        // the block has no stream output: nobody reads the range-angle map, so it is not stored (results are bit-identical); shapes
        // outside the fused kernel keep the map (JRC_ERR_UNSUPPORTED) — as does JRC_CHAIN_WRITE_MAP=1
        if (!getenv("JRC_CHAIN_WRITE_MAP")) (void)jrc_chain_feed_set_write_map(d_feed, 0);
        d_bg_removal = background_removal; d_record_len = record_len;
        if (background_removal || background_recording)
            ctx_holder::check_feed(d_feed, jrc_chain_feed_set_background(d_feed, background_removal, background_recording, record_len));
        if (const char* e = getenv("JRC_RADAR_CHAIN_MAX_AGE_US")) d_max_age_us = atol(e);
        if (const char* e = getenv("JRC_RADAR_CHAIN_TX_RESIDENT")) d_tx_res_enabled = atoi(e) != 0;
        d_tx_ref.resize((size_t)N_tx * N_sym * fft_len);
        d_res.resize((size_t)d_fpb);
        if (d_max_age_us > 0) d_flusher = std::thread(&radar_chain_impl::flusher_main, this);
        message_port_register_out(pmt::mp("params"));
        set_tag_propagation_policy(TPP_DONT);
        std::ofstream f(d_stats_path, std::ofstream::app);
        if (stats_record && !f.is_open()) std::cerr << "[RADAR CHAIN] Could not open log file at " << d_stats_path << std::endl;
    }
    ~radar_chain_impl() override
    {
        d_flusher_stop.store(true);
        if (d_flusher.joinable()) d_flusher.join();
        // a block destroyed without stop() / flush(): publish what is still in flight (a destructor must not throw: a failure is only reported)
        try { if (d_feed && !d_flusher_failed.load() && jrc_chain_feed_pending(d_feed) > 0) collect_ready(true); }
        catch (const std::exception& e) { std::cerr << "[RADAR CHAIN] ~radar_chain: " << e.what() << std::endl; }
        jrc_chain_feed_destroy(d_feed);
    }
    int frames_done() const override { return d_frames_done; }
    int n_devices() const override { return d_n_devices; }
    long rx_only_batches() const override { return d_rx_only_batches; }
    long profile_ns(int what) const override { return what >= 0 && what < 4 ? d_prof_ns[what].load() : 0; }
    int pending_batches() const override { return d_pending.load(); }
    void flush() override
    {
        jrc_rt::thread::scoped_lock guard(d_setlock);
        rethrow_flusher_error();
        collect_ready(true);
    }
    bool stop() override { flush(); return true; }                      // the scheduler is done with the block: publish what is still in flight
    void set_background_record(bool background_record) override
    {
        jrc_rt::thread::scoped_lock guard(d_setlock);      // between two general_work calls; batches of earlier calls are published first
        collect_ready(true);
        ctx_holder::check_feed(d_feed, jrc_chain_feed_set_background(d_feed, d_bg_removal, background_record, d_record_len));
    }

    int general_work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star&) override
    {
        jrc_rt::thread::scoped_lock guard(d_setlock);
        prof_scope ps_all(d_prof_ns[3]);
        rethrow_flusher_error();
        collect_ready(false);                                                                         // what finished since the last call
        std::vector<jrc_rt::tag_t> rx_tags, tx_tags;
        get_tags_in_range(rx_tags, d_N_tx, nitems_read(d_N_tx), nitems_read(d_N_tx) + ninput_items[d_N_tx], pmt::mp("packet_len"));
        if (rx_tags.empty()) {                                                                        // no frame in sight (:219-234)
            for (int i = 0; i < d_N_tx + d_N_rx; i++) consume(i, ninput_items[i]);
            return 0;
        }
        get_tags_in_range(tx_tags, 0, nitems_read(0), nitems_read(0) + ninput_items[0], pmt::mp("packet_len"));
        const size_t tx_skip = tx_tags.size() > rx_tags.size() ? tx_tags.size() - rx_tags.size() : 0;   // stale TX packets (:191-198)
        if (tx_tags.size() <= tx_skip) throw std::runtime_error("[MIMO OFDM RADAR] no packet_len tag on TX input");
        const size_t n_frames = std::min(rx_tags.size(), tx_tags.size() - tx_skip);
        const size_t item = (size_t)d_fft_len, port = (size_t)d_N_sym * item;                          // a port of a staged frame: the N_sym used symbols
        const size_t port_bytes = sizeof(jrc_cf32) * port;
        long rx_end = 0, tx_end = 0;
        size_t f = 0;
        while (f < n_frames) {
            jrc_cf32* stage = nullptr;
            // Batches are staged by this (the scheduler's) thread straight into the slot's PINNED buffer and submitted in place, one after the
            // other, also when JRC_DEVICES lists several GPUs: what jrc_chain_feed_submit_many's per-device threads parallelise is the
            // pageable -> pinned staging copy, and there is none here; an in-place submit only enqueues (H2D + kernels + D2H, asynchronous),
            // so consecutive batches still overlap on their devices.
            if (jrc_chain_feed_pending(d_feed) == d_slots) collect_one();
            ctx_holder::check_feed(d_feed, jrc_chain_feed_acquire(d_feed, &stage));
            int nb = 0;
            auto t_stage = std::chrono::steady_clock::now();
            const bool try_resident = d_tx_res_enabled && d_tx_ref_valid && d_tx_backoff == 0;
            bool tx_same = try_resident;                                                              // all frames of the batch carry the resident rows
            for (; f < n_frames && nb < d_fpb; f++) {
                const long rx0 = (long)(rx_tags[f].offset - nitems_read(d_N_tx)), tx0 = (long)(tx_tags[f + tx_skip].offset - nitems_read(0));
                const long rx_len = (long)pmt::to_uint64(rx_tags[f].value), tx_len = (long)pmt::to_uint64(tx_tags[f + tx_skip].value);
                if (rx_len < d_n_items || tx_len < d_n_items) throw std::runtime_error("[RADAR CHAIN] packet shorter than N_pre + N_sym items");
                if (rx0 + rx_len > ninput_items[d_N_tx] || tx0 + tx_len > ninput_items[0]) break;
                jrc_cf32* dst = stage + (size_t)nb * (d_N_tx + d_N_rx) * port;
                for (int t = 0; t < d_N_tx; t++) {
                    const jrc_cf32* src = (const jrc_cf32*)input_items[t] + (size_t)(tx0 + d_N_pre) * item;
                    if (tx_same && memcmp(src, d_tx_ref.data() + (size_t)t * port, port_bytes) != 0) {
                        tx_same = false;                                                              // this batch goes up whole: fill in the TX ports skipped so far
                        for (int fb = 0; fb < nb; fb++)
                            memcpy(stage + (size_t)fb * (d_N_tx + d_N_rx) * port, d_tx_ref.data(), port_bytes * d_N_tx);
                        for (int tb = 0; tb < t; tb++) memcpy(dst + (size_t)tb * port, d_tx_ref.data() + (size_t)tb * port, port_bytes);
                    }
                    if (!tx_same) memcpy(dst + (size_t)t * port, src, port_bytes);
                }
                for (int r = 0; r < d_N_rx; r++)
                    memcpy(dst + (size_t)(d_N_tx + r) * port, (const jrc_cf32*)input_items[d_N_tx + r] + (size_t)(rx0 + d_N_pre) * item, port_bytes);
                rx_end = rx0 + rx_len; tx_end = tx0 + tx_len;
                nb++;
            }
            d_prof_ns[0].fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_stage).count(), std::memory_order_relaxed);
            if (nb == 0) break;                                                                       // the next frame is not complete yet
            prof_scope ps_submit(d_prof_ns[1]);
            if (tx_same) {
                if (!d_tx_ref_on_device) {                                                            // first use of these rows: hand them to the feed (nothing may be in flight)
                    collect_ready(true);
                    ctx_holder::check_feed(d_feed, jrc_chain_feed_set_tx(d_feed, d_tx_ref.data()));
                    d_tx_ref_on_device = true;
                }
                ctx_holder::check_feed(d_feed, jrc_chain_feed_submit_rx(d_feed, nullptr, nb));
                d_rx_only_batches++; d_tx_misses = 0;
            } else {
                ctx_holder::check_feed(d_feed, jrc_chain_feed_submit(d_feed, nullptr, nb));
                d_full_batches++;
                if (d_tx_res_enabled) {
                    // the rows of this batch's last frame become the candidate for the next batches; rows that never repeat (data symbols in
                    // the radar's window) stop the comparing for 64 batches at a time
                    const jrc_cf32* last = stage + (size_t)(nb - 1) * (d_N_tx + d_N_rx) * port;
                    const bool repeated = d_tx_ref_valid && memcmp(last, d_tx_ref.data(), port_bytes * d_N_tx) == 0;
                    if (!repeated) { memcpy(d_tx_ref.data(), last, port_bytes * d_N_tx); d_tx_ref_on_device = false; }
                    d_tx_ref_valid = true;
                    if (try_resident && ++d_tx_misses >= 4) { d_tx_backoff = 64; d_tx_misses = 0; }
                    else if (d_tx_backoff > 0) d_tx_backoff--;
                }
            }
            d_submitted.push_back(std::chrono::steady_clock::now());
            d_pending.store(jrc_chain_feed_pending(d_feed));
            if (nb < d_fpb) break;
        }
        collect_ready(d_max_age_us <= 0);                                                             // finished or overdue batches, in frame order
        if (rx_end == 0 && tx_end == 0) return 0;                                                      // first frame incomplete: wait for more input
        for (int r = 0; r < d_N_rx; r++) consume(r + d_N_tx, (int)rx_end);
        for (int t = 0; t < d_N_tx; t++) consume(t, (int)tx_end);
        return 0;
    }
};

radar_chain::sptr radar_chain::make(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int interp_range, int interp_angle,
                                    bool enable_tx_interleave, std::vector<float> range_bins, std::vector<float> angle_bins,
                                    float noise_discard_range_m, float noise_discard_angle_deg, float snr_threshold, float power_threshold,
                                    const std::string& stats_path, bool stats_record, int frames_per_batch, int batches_in_flight,
                                    const std::string&, bool, bool background_removal, bool background_recording, int record_len)
{
    if (frames_per_batch <= 0) {                                                                    // 0: automatic — the header's default, within 64 MiB of staging per slot
        const long long frame_bytes = (long long)(N_tx + N_rx) * N_sym * fft_len * (long long)sizeof(gr_complex);
        long long fit = frame_bytes > 0 ? (64LL << 20) / frame_bytes : RADAR_CHAIN_DEFAULT_FRAMES_PER_BATCH;
        // the feed also holds a map and a channel estimate per frame of every slot ON THE DEVICE (jrc_chain_feed_create allocates them whether or
        // not the block asks for maps): kept within 4 GiB per device over the slots — no change at the .grc's shape or at configs B / D (0.8 /
        // 0.35 GB), a bound for shapes whose maps are large beside their frames
        const long long P = (long long)N_tx * N_rx;
        const long long dev_per_frame = ((long long)fft_len * interp_range * P * interp_angle + P * fft_len) * (long long)sizeof(gr_complex);
        const int slots = batches_in_flight > 0 ? batches_in_flight : RADAR_CHAIN_DEFAULT_BATCHES_IN_FLIGHT;
        if (dev_per_frame > 0) fit = std::min<long long>(fit, (4LL << 30) / (dev_per_frame * slots));
        frames_per_batch = (int)std::max<long long>(1, std::min<long long>(RADAR_CHAIN_DEFAULT_FRAMES_PER_BATCH, fit));
    }
    if (batches_in_flight <= 0) batches_in_flight = RADAR_CHAIN_DEFAULT_BATCHES_IN_FLIGHT;
    return JRC_GET_INITIAL_SPTR(new radar_chain_impl(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, enable_tx_interleave,
                                                     range_bins, angle_bins, noise_discard_range_m, noise_discard_angle_deg, snr_threshold,
                                                     power_threshold, stats_path, stats_record, frames_per_batch, batches_in_flight,
                                                     background_removal, background_recording, record_len));
}

// =================================================================================================
// ofdm_cyclic_prefix_remover  (lib/ofdm_cyclic_prefix_remover_impl.cc)
// =================================================================================================
class ofdm_cyclic_prefix_remover_impl : public ofdm_cyclic_prefix_remover {
    ctx_holder d_c;
    int d_fft_len, d_cp_len;
    std::vector<jrc_rt::tag_t> d_tags;

public:
    ofdm_cyclic_prefix_remover_impl(int fft_len, int cp_len, const std::string& len_key)
        : jrc_rt::tagged_stream_block("ofdm_cyclic_prefix_remover", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)),
                                      jrc_rt::io_signature::make(1, 1, sizeof(gr_complex) * fft_len), len_key),
          d_fft_len(fft_len), d_cp_len(cp_len)
    {
        set_tag_propagation_policy(TPP_DONT);
    }
    int calculate_output_stream_length(const gr_vector_int& n) override { return n[0] / (d_fft_len + d_cp_len); }   // :62-67
    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star& out) override
    {
        get_tags_in_range(d_tags, 0, nitems_read(0), nitems_read(0) + 1);                             // :79-83
        for (auto& t : d_tags) add_item_tag(0, nitems_written(0), t.key, t.value, t.srcid);
        int n = jrc_cp_remove(d_c.ctx, d_fft_len, d_cp_len, ninput_items[0], (const jrc_cf32*)in[0], (jrc_cf32*)out[0]);
        d_c.check(n);
        return n;
    }
};
ofdm_cyclic_prefix_remover::sptr ofdm_cyclic_prefix_remover::make(int fft_len, int cp_len, std::string len_key)
{
    return JRC_GET_INITIAL_SPTR(new ofdm_cyclic_prefix_remover_impl(fft_len, cp_len, len_key));
}

// =================================================================================================
// fft_peak_detect  (lib/fft_peak_detect_impl.cc)
// =================================================================================================
class fft_peak_detect_impl : public fft_peak_detect {
    ctx_holder d_c;
    int d_samp_rate, d_samp_protect;
    float d_interp_factor, d_threshold;
    std::vector<float> d_max_freq;

public:
    fft_peak_detect_impl(int samp_rate, float interp_factor, float threshold, int samp_protect, std::vector<float> max_freq,
                         bool, const std::string& len_key)
        : jrc_rt::tagged_stream_block("fft_peak_detect", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)),
                                      jrc_rt::io_signature::make3(3, 3, sizeof(float), sizeof(float), sizeof(float)), len_key),
          d_samp_rate(samp_rate), d_samp_protect(samp_protect), d_interp_factor(interp_factor), d_threshold(threshold),
          d_max_freq(max_freq)
    {
        set_tag_propagation_policy(TPP_DONT);
    }
    int calculate_output_stream_length(const gr_vector_int&) override { return 1; }
    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star& out) override
    {
        int k;
        int n = jrc_fft_peak_detect(d_c.ctx, d_samp_rate, d_interp_factor, d_threshold, d_samp_protect, ninput_items[0],
                                    (const jrc_cf32*)in[0], (float*)out[0], (float*)out[1], (float*)out[2], &k);
        d_c.check(n);
        return n;                                                                                     // always 1 (:110)
    }
    void set_threshold(float t) override { d_threshold = t; }
    void set_samp_protect(int s) override { d_samp_protect = s; }
    void set_max_freq(std::vector<float> f) override { d_max_freq = f; }
};
fft_peak_detect::sptr fft_peak_detect::make(int samp_rate, float interp_factor, float threshold, int samp_protect,
                                            std::vector<float> max_freq, bool cut_max_freq, const std::string& len_key)
{
    return JRC_GET_INITIAL_SPTR(new fft_peak_detect_impl(samp_rate, interp_factor, threshold, samp_protect, max_freq, cut_max_freq, len_key));
}

// =================================================================================================
// mimo_ofdm_equalizer  (lib/mimo_ofdm_equalizer_impl.cc)
// =================================================================================================
static std::vector<gr_complex> flatten(const std::vector<std::vector<gr_complex>>& v)
{
    std::vector<gr_complex> o;
    for (auto& r : v) o.insert(o.end(), r.begin(), r.end());
    return o;
}

class mimo_ofdm_equalizer_impl : public mimo_ofdm_equalizer {
    ctx_holder d_c;
    jrc_equalizer* d_eq = nullptr;
    int d_fft_len, d_N_data, d_N_tx;
    std::string d_chan_est_file;
    bool d_stats_record;
    std::vector<gr_complex> d_chan_est;
    std::vector<jrc_rt::tag_t> tags;

public:
    mimo_ofdm_equalizer_impl(ChannelEstimator algo, double freq, double bw, int fft_len, int cp_len, std::vector<int> data_carriers,
                             std::vector<int> pilot_carriers, const std::vector<std::vector<gr_complex>>& pilot_symbols,
                             std::vector<gr_complex> ltf_seq, const std::vector<std::vector<gr_complex>>& mapped_ltf_symbols,
                             int n_mimo_ltf, const std::string& chan_est_file, const std::string&, bool stats_record, bool)
        : jrc_rt::block("mimo_ofdm_equalizer", jrc_rt::io_signature::make(1, 1, fft_len * sizeof(gr_complex)),
                        jrc_rt::io_signature::make(1, 1, data_carriers.size() * sizeof(gr_complex))),              // :89-91
          d_fft_len(fft_len), d_N_data((int)data_carriers.size()), d_chan_est_file(chan_est_file), d_stats_record(stats_record)
    {
        if (algo != LS && algo != STA) throw std::runtime_error("[OFDM Equalizer] Estimator not implemented");    // :941-943
        std::vector<gr_complex> ps = flatten(pilot_symbols), ml = flatten(mapped_ltf_symbols);
        jrc_eq_cfg c;
        c.estimator = algo; c.freq = freq; c.bw = bw; c.fft_len = fft_len; c.cp_len = cp_len;
        c.n_data = (int)data_carriers.size(); c.n_pilot = (int)pilot_carriers.size();
        c.data_carriers = data_carriers.data(); c.pilot_carriers = pilot_carriers.data();
        c.n_pilot_rows = (int)pilot_symbols.size(); c.pilot_symbols = (const jrc_cf32*)ps.data();
        c.ltf_seq = (const jrc_cf32*)ltf_seq.data(); c.mapped_ltf = (const jrc_cf32*)ml.data();
        c.mapped_cols = (int)mapped_ltf_symbols[0].size(); c.n_mimo_ltf = n_mimo_ltf;
        d_N_tx = c.mapped_cols / n_mimo_ltf;                                                                       // :168
        d_c.check(jrc_equalizer_create(d_c.ctx, &c, 1, &d_eq));
        d_chan_est.resize((size_t)fft_len * d_N_tx);
        set_tag_propagation_policy(TPP_DONT);
    }
    ~mimo_ofdm_equalizer_impl() override { jrc_equalizer_destroy(d_eq); }
    void forecast(int noutput_items, gr_vector_int& req) override { req[0] = noutput_items; }                    // :182-186

    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star& out) override
    {
        jrc_rt::block::set_thread_priority(50);
        jrc_rt::thread::scoped_lock lock(d_setlock);
        get_tags_in_window(tags, 0, 0, ninput_items[0], pmt::string_to_symbol("frame_start"));                   // :221
        std::vector<int64_t> offs; std::vector<double> vals;
        for (auto& t : tags) { offs.push_back((int64_t)(t.offset - nitems_read(0))); vals.push_back(pmt::to_double(t.value)); }
        int consumed = 0, n_ev = 0, ce_written = 0;
        jrc_eq_event ev[4];
        int n = jrc_equalizer_work(d_eq, 0, noutput_items, ninput_items[0], (const jrc_cf32*)in[0], offs.data(), vals.data(),
                                   (int)offs.size(), (jrc_cf32*)out[0], &consumed, ev, 4, &n_ev, (jrc_cf32*)d_chan_est.data(), &ce_written);
        d_c.check(n);
        for (int i = 0; i < n_ev; i++) {
            pmt::pmt_t dict = pmt::make_dict();
            if (ev[i].kind == 1) {                                                                                // :331-337
                dict = pmt::dict_add(dict, pmt::mp("data_bytes"), pmt::from_uint64(ev[i].data_bytes));
                dict = pmt::dict_add(dict, pmt::mp("mcs"), pmt::from_uint64(ev[i].mcs));
                dict = pmt::dict_add(dict, pmt::mp("packet_type"), pmt::from_uint64(ev[i].packet_type));
                dict = pmt::dict_add(dict, pmt::mp("snr"), pmt::from_double(ev[i].snr));
                dict = pmt::dict_add(dict, pmt::mp("freq_offset"), pmt::from_double(ev[i].freq_offset));
                add_item_tag(0, nitems_written(0) + ev[i].offset, pmt::string_to_symbol("stream_start"), dict, pmt::string_to_symbol(alias()));
            } else {                                                                                              // :626-629
                dict = pmt::dict_add(dict, pmt::mp("snr_data"), pmt::from_double(ev[i].snr_data));
                dict = pmt::dict_add(dict, pmt::mp("chan_mean"), pmt::init_c32vector(ev[i].n_chan_mean, (const gr_complex*)ev[i].chan_mean));
                add_item_tag(0, nitems_written(0) + ev[i].offset, pmt::string_to_symbol("stream_end"), dict, pmt::string_to_symbol(alias()));
            }
        }
        if (ce_written) write_chan_est();
        consume(0, consumed);
        return n;
    }
    // chan_est_file wire format "sc:(re,im);(re,im);...\n" (:378-416), read back by mimo_precoder (:806-833).
    // Eigen's FullPrecision = NumTraits<float>::digits10() significant digits through the stream's default float format ("%.<p>g"): 6 in Eigen
    // 3.3 / 3.4 (std::numeric_limits<float>::digits10), 7 in 3.2 (ceil(-log10(eps))) — recollection, no Eigen in this image.  9 are written by
    // default (a float round trip, so the precoder steers with the estimate itself, not its 6-digit rounding); JRC_CSV_DIGITS=6 writes the
    // file byte for byte as the reference's Eigen 3.3 / 3.4 build would.  The precoder's reader takes any of them (:806-833).
    void write_chan_est()
    {
        std::ofstream f(d_chan_est_file, std::ofstream::trunc);
        if (!f.is_open()) throw std::runtime_error("[OFDM Equalizer] Could not open file!!");
        int digits = 9;
        if (const char* e = getenv("JRC_CSV_DIGITS")) { const int v = atoi(e); if (v >= 1 && v <= 17) digits = v; }
        char fmt[32], buf[96];
        snprintf(fmt, sizeof(fmt), "%%s(%%.%dg,%%.%dg)", digits, digits);
        for (int sc = 0; sc < d_fft_len; sc++) {
            f << sc << ":";
            for (int t = 0; t < d_N_tx; t++) {
                const gr_complex h = d_chan_est[(size_t)sc * d_N_tx + t];
                snprintf(buf, sizeof(buf), fmt, t ? ";" : "", h.real(), h.imag());
                f << buf;
            }
            f << "\n";
        }
    }
    void set_estimator(ChannelEstimator algo) override
    {
        jrc_rt::thread::scoped_lock lock(d_setlock);
        if (algo != LS && algo != STA) throw std::runtime_error("[OFDM Equalizer] Estimator not implemented");
        jrc_equalizer_set_estimator(d_eq, algo);
    }
    void set_bandwidth(double bw) override { jrc_rt::thread::scoped_lock lock(d_setlock); jrc_equalizer_set_bandwidth(d_eq, bw); }
    void set_frequency(double f) override { jrc_rt::thread::scoped_lock lock(d_setlock); jrc_equalizer_set_frequency(d_eq, f); }
    void set_stats_record(bool s) override { jrc_rt::thread::scoped_lock lock(d_setlock); d_stats_record = s; }
};
mimo_ofdm_equalizer::sptr mimo_ofdm_equalizer::make(ChannelEstimator estimator_algo, double freq, double bw, int fft_len, int cp_len,
                                                    std::vector<int> data_carriers, std::vector<int> pilot_carriers,
                                                    const std::vector<std::vector<gr_complex>>& pilot_symbols,
                                                    std::vector<gr_complex> long_seq,
                                                    const std::vector<std::vector<gr_complex>>& mapped_ltf_symbols, int n_mimo_ltf,
                                                    const std::string& chan_est_file, const std::string& comm_log_file,
                                                    bool stats_record, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new mimo_ofdm_equalizer_impl(estimator_algo, freq, bw, fft_len, cp_len, data_carriers, pilot_carriers,
                                                             pilot_symbols, long_seq, mapped_ltf_symbols, n_mimo_ltf, chan_est_file,
                                                             comm_log_file, stats_record, debug));
}

// =================================================================================================
// mimo_precoder  (lib/mimo_precoder_impl.cc)
// =================================================================================================
class mimo_precoder_impl : public mimo_precoder {
    ctx_holder d_c;
    jrc_precoder* d_pre = nullptr;
    int d_fft_len, d_N_tx, d_N_data;
    std::vector<int> d_active;   // data then pilot carriers, shifted indices (:197-200)
    std::string d_chan_est_file, d_radar_log_file;
    bool d_chan_est_smoothing, d_radar_aided, d_phased_steering, d_use_radar_streams;
    std::vector<gr_complex> steering_matrix;        // [fft_len][T*T] column-major
    std::vector<gr_complex> steering_matrix_mean;   // [T*T]
    std::vector<gr_complex> chan_est_vector_mean;   // [T]
    bool mean_chan_est_changed = true;
    std::time_t last_chanEst_update_time = 0;
    std::vector<gr_complex> d_radar_streams;

public:
    mimo_precoder_impl(int fft_len, int N_tx, int N_ss, const std::vector<int>& data_carriers, const std::vector<int>& pilot_carriers,
                       const std::vector<std::vector<gr_complex>>& pilot_symbols, const std::vector<std::vector<gr_complex>>& sync_words,
                       const std::vector<std::vector<gr_complex>>& mapped_ltf_symbols, const std::string& chan_est_file,
                       bool chan_est_smoothing, const std::string& radar_log_file, bool radar_aided, bool phased_steering,
                       bool use_radar_streams, const std::string& len_tag_key, bool)
        : jrc_rt::tagged_stream_block("mimo_precoder", jrc_rt::io_signature::make(N_ss, N_ss, sizeof(gr_complex)),
                                      jrc_rt::io_signature::make(N_tx, N_tx, fft_len * sizeof(gr_complex)), len_tag_key),   // :94-96
          d_fft_len(fft_len), d_N_tx(N_tx), d_N_data((int)data_carriers.size()), d_chan_est_file(chan_est_file),
          d_radar_log_file(radar_log_file), d_chan_est_smoothing(chan_est_smoothing), d_radar_aided(radar_aided),
          d_phased_steering(phased_steering), d_use_radar_streams(use_radar_streams)
    {
        if (data_carriers.empty()) throw std::invalid_argument("Data carriers must be of type vector of vector i.e. ().");
        if (pilot_carriers.empty()) throw std::invalid_argument("Pilot carriers must be of type vector of vector i.e. ((),).");
        if (pilot_symbols.empty()) throw std::invalid_argument("Pilot symbols must be of type vector of vector i.e. ((),).");
        for (auto& p : pilot_symbols)
            if (p.size() != pilot_carriers.size()) throw std::invalid_argument("pilot_carriers do not match pilot_symbols");   // :158-164
        for (auto& w : sync_words)
            if (w.size() != (unsigned)fft_len) throw std::invalid_argument("[MIMO PRECODER] sync words must be fft length");   // :170-176
        if (mapped_ltf_symbols.size() != (size_t)fft_len)
            throw std::invalid_argument("[MIMO PRECODER] MIMO LTF symbols should have (fft length x Ntx) rows!!");             // :179-182
        for (auto& r : mapped_ltf_symbols)
            if (r.size() != (size_t)N_tx * N_tx) throw std::invalid_argument("[MIMO PRECODER] MIMO LTF symbols should have (Ntx) columns!!");
        std::vector<gr_complex> ps = flatten(pilot_symbols), sw = flatten(sync_words), ml = flatten(mapped_ltf_symbols);
        jrc_pre_cfg c;
        c.fft_len = fft_len; c.N_tx = N_tx; c.n_data = (int)data_carriers.size(); c.n_pilot = (int)pilot_carriers.size();
        c.data_carriers = data_carriers.data(); c.pilot_carriers = pilot_carriers.data();
        c.n_pilot_rows = (int)pilot_symbols.size(); c.pilot_symbols = (const jrc_cf32*)ps.data();
        c.n_sync = (int)sync_words.size(); c.sync_words = (const jrc_cf32*)sw.data(); c.mapped_ltf = (const jrc_cf32*)ml.data();
        d_c.check(jrc_precoder_create(d_c.ctx, &c, &d_pre));
        auto shift = [&](int v) { if (v < 0) v += fft_len; return (v + fft_len / 2) % fft_len; };
        for (int v : data_carriers) d_active.push_back(shift(v));
        for (int v : pilot_carriers) d_active.push_back(shift(v));
        steering_matrix.assign((size_t)fft_len * N_tx * N_tx, gr_complex(0, 0));
        steering_matrix_mean.assign((size_t)N_tx * N_tx, gr_complex(0, 0));
        chan_est_vector_mean.assign(N_tx, gr_complex(0, 0));
        set_tag_propagation_policy(TPP_DONT);
        set_relative_rate(1, (uint64_t)data_carriers.size());
    }
    ~mimo_precoder_impl() override { jrc_precoder_destroy(d_pre); }
    int calculate_output_stream_length(const gr_vector_int& n) override { return jrc_precoder_output_length(d_pre, n[0]); }   // :265-272

    bool is_active(int line) const { for (int a : d_active) if (a == line) return true; return false; }

    // lib/mimo_precoder_impl.cc:775-898: parse "sc:(re,im);...", per-subcarrier and mean steering matrices
    bool compute_steering_matrix()
    {
        std::ifstream f(d_chan_est_file);
        if (!f.is_open()) {
            if (!d_chan_est_file.empty()) std::cerr << "[MIMO PRECODER] Could not open channel estimate file at " << d_chan_est_file << std::endl;
            return false;
        }
        struct stat sb;
        std::time_t curr = (stat(d_chan_est_file.c_str(), &sb) == 0) ? sb.st_mtime : 0;                             // boost last_write_time :791
        if (curr <= last_chanEst_update_time && !mean_chan_est_changed) return true;                                  // :794-798
        std::fill(chan_est_vector_mean.begin(), chan_est_vector_mean.end(), gr_complex(0, 0));
        const int T = d_N_tx;
        std::vector<gr_complex> rows;   // [n_lines][T]
        std::vector<int> sc_of_line;
        std::string line;
        int n_line_read = 0;
        while (std::getline(f, line)) {
            std::stringstream ls(line);
            std::string entry;
            getline(ls, entry, ':');
            int sc_idx = std::stoi(entry);
            std::vector<gr_complex> v(T);
            int n_col = 0;
            while (getline(ls, entry, ';')) {
                float re = 0, im = 0;
                sscanf(entry.c_str(), "(%f,%f)", &re, &im);
                if (n_col < T) {
                    if (is_active(n_line_read)) chan_est_vector_mean[n_col] += gr_complex(re, im);                   // :814-817
                    v[n_col] = gr_complex(re, im);
                }
                n_col++;
            }
            n_line_read++;
            if (n_col != T) { std::cerr << "[MIMO PRECODER] Steering matrix computation FAILED! --> Line is not correct: " << n_line_read << std::endl; return false; }
            if (sc_idx < 0 || sc_idx >= d_fft_len) return false;
            rows.insert(rows.end(), v.begin(), v.end());
            sc_of_line.push_back(sc_idx);
        }
        if (n_line_read < d_fft_len) { std::cerr << "[MIMO PRECODER] Steering matrix computation FAILED! --> Number of parsed lines not correct: " << n_line_read << std::endl; return false; }
        std::vector<gr_complex> Q((size_t)n_line_read * T * T);
        d_c.check(jrc_steering_from_channel(d_c.ctx, T, n_line_read, (const jrc_cf32*)rows.data(), d_phased_steering, (jrc_cf32*)Q.data()));   // :846-861
        for (int l = 0; l < n_line_read; l++)
            std::copy(Q.begin() + (size_t)l * T * T, Q.begin() + (size_t)(l + 1) * T * T, steering_matrix.begin() + (size_t)sc_of_line[l] * T * T);
        for (int t = 0; t < T; t++) chan_est_vector_mean[t] = chan_est_vector_mean[t] / (gr_complex)(float)d_active.size();   // :872-875
        d_c.check(jrc_steering_from_channel(d_c.ctx, T, 1, (const jrc_cf32*)chan_est_vector_mean.data(), d_phased_steering,
                                            (jrc_cf32*)steering_matrix_mean.data()));                                 // :880-893
        mean_chan_est_changed = false;
        last_chanEst_update_time = curr;
        return true;
    }

    // lib/mimo_precoder_impl.cc:903-983: last line of the radar log, 5th field = angle estimate
    bool compute_radar_aided_steering()
    {
        std::ifstream f(d_radar_log_file);
        if (!f.is_open()) { std::cerr << "[MIMO PRECODER] Could not open radar log file at " << d_radar_log_file << std::endl; return false; }
        std::string line, lastline;
        while (std::getline(f, line)) if (!line.empty()) lastline = line;
        if (lastline.empty()) {
            std::cerr << "[MIMO PRECODER] Radar log file is empty at " << d_radar_log_file << std::endl;
            compute_steering_matrix();                                                                               // :939-941
            return false;
        }
        std::stringstream ls(lastline);
        std::string entry;
        for (int i = 0; i < 4; i++) getline(ls, entry, ',');
        getline(ls, entry, '\n');
        float angle_estimate = std::stof(entry);
        for (int t = 0; t < d_N_tx; t++)
            chan_est_vector_mean[t] = std::exp(gr_complex(0, M_PI * sin(angle_estimate / 180.0 * M_PI) * t));     // :956-959
        d_c.check(jrc_steering_from_channel(d_c.ctx, d_N_tx, 1, (const jrc_cf32*)chan_est_vector_mean.data(), d_phased_steering,
                                            (jrc_cf32*)steering_matrix_mean.data()));                                 // :961-974
        mean_chan_est_changed = true;
        return true;
    }

    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        jrc_rt::block::set_thread_priority(30);
        jrc_rt::thread::scoped_lock lock(d_setlock);
        if (output_items.size() != (size_t)d_N_tx) throw std::runtime_error("[MIMO PRECODER] output_items should contain {N_tx} buffers!!");
        std::vector<jrc_rt::tag_t> tags;
        get_tags_in_range(tags, 0, nitems_read(0), nitems_read(0) + ninput_items[0], pmt::mp("mcs"));               // :305-325
        if (tags.size() != 1) throw std::runtime_error("no mcs tag in input stream!");
        int mcs = (int)pmt::to_long(tags[0].value);
        get_tags_in_range(tags, 0, nitems_read(0), nitems_read(0) + ninput_items[0], pmt::mp("packet_type"));
        if (tags.size() != 1) throw std::runtime_error("no packet_type tag in input stream!");
        int packet_type = (int)pmt::to_uint64(tags[0].value);
        get_tags_in_range(tags, 0, nitems_read(0), nitems_read(0) + ninput_items[0], pmt::mp("pdu_len"));
        if (tags.size() != 1) throw std::runtime_error("no pdu_len tag in input stream!");
        int data_size_crc = (int)pmt::to_long(tags[0].value);

        int steer_mode = 0;
        const gr_complex* rs = nullptr;
        if (packet_type == DATA) {
            bool custom = d_radar_aided ? compute_radar_aided_steering() : compute_steering_matrix();               // :500-532
            if (custom) steer_mode = (!d_chan_est_smoothing && !d_radar_aided) ? 2 : 1;                               // :600
            if (d_use_radar_streams) {                                                                               // :435-493
                const int n_sym = ninput_items[0] / d_N_data;
                d_radar_streams.assign((size_t)(d_N_tx - 1) * n_sym * d_fft_len, gr_complex(0, 0));
                std::random_device rd;
                std::default_random_engine e1(rd());
                std::uniform_int_distribution<unsigned int> uniform_dist(0, 3);
                const float a = 0.707107f / 2.0f;
                for (auto& s : d_radar_streams) { unsigned q = uniform_dist(e1); s = gr_complex((q & 1) ? a : -a, (q & 2) ? a : -a); }
                rs = d_radar_streams.data();
            }
        }
        std::vector<jrc_cf32*> outs(d_N_tx);
        for (int t = 0; t < d_N_tx; t++) outs[t] = (jrc_cf32*)output_items[t];
        int n = jrc_precoder_work(d_pre, ninput_items[0], (const jrc_cf32*)input_items[0], mcs, packet_type, data_size_crc, steer_mode,
                                  (const jrc_cf32*)steering_matrix_mean.data(), (const jrc_cf32*)steering_matrix.data(),
                                  (const jrc_cf32*)rs, outs.data());
        if (n == JRC_ERR_SIG_FIELD) throw std::runtime_error("[MIMO PRECODER] something is wrong!!");               // :327-333
        d_c.check(n);
        return n;                                                                                                    // :740
    }
    void set_chan_est_smoothing(bool v) override { jrc_rt::thread::scoped_lock lock(d_setlock); d_chan_est_smoothing = v; }
    void set_radar_aided(bool v) override { jrc_rt::thread::scoped_lock lock(d_setlock); d_radar_aided = v; }
    void set_use_radar_streams(bool v) override { jrc_rt::thread::scoped_lock lock(d_setlock); d_use_radar_streams = v; }
    void set_phased_steering(bool v) override { jrc_rt::thread::scoped_lock lock(d_setlock); mean_chan_est_changed = true; d_phased_steering = v; }
};
mimo_precoder::sptr mimo_precoder::make(int fft_len, int N_tx, int N_ss, const std::vector<int>& data_carriers,
                                        const std::vector<int>& pilot_carriers, const std::vector<std::vector<gr_complex>>& pilot_symbols,
                                        const std::vector<std::vector<gr_complex>>& sync_words,
                                        const std::vector<std::vector<gr_complex>>& mapped_ltf_symbols, const std::string& chan_est_file,
                                        bool chan_est_smoothing, const std::string& radar_log_file, bool radar_aided, bool phased_steering,
                                        bool use_radar_streams, const std::string& len_tag_key, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new mimo_precoder_impl(fft_len, N_tx, N_ss, data_carriers, pilot_carriers, pilot_symbols, sync_words,
                                                       mapped_ltf_symbols, chan_est_file, chan_est_smoothing, radar_log_file, radar_aided,
                                                       phased_steering, use_radar_streams, len_tag_key, debug));
}

// =================================================================================================
// target_simulator  (lib/target_simulator_impl.cc)
// =================================================================================================
class target_simulator_impl : public target_simulator {
    ctx_holder d_c;
    jrc_tsim* d_sim = nullptr;
    int d_samp_rate = 0, d_num_targets = 0;
    size_t d_n_rx;
    bool d_rndm_phaseshift = false;
    pmt::pmt_t d_key, d_srcid;
    std::vector<gr_complex> d_phase;

public:
    target_simulator_impl(std::vector<float> range, std::vector<float> velocity, std::vector<float> rcs, std::vector<float> azimuth,
                          std::vector<float> position_rx, int samp_rate, float center_freq, float self_coupling_db,
                          bool rndm_phaseshift, bool self_coupling, const std::string& len_key, bool)
        : jrc_rt::tagged_stream_block("target_simulator", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)),
                                      jrc_rt::io_signature::make((int)position_rx.size(), (int)position_rx.size(), sizeof(gr_complex)),
                                      len_key),                                                                 // :80-83
          d_n_rx(position_rx.size())
    {
        setup_targets(range, velocity, rcs, azimuth, position_rx, samp_rate, center_freq, self_coupling_db, rndm_phaseshift,
                      self_coupling);
    }
    ~target_simulator_impl() override { jrc_tsim_destroy(d_sim); }

    void setup_targets(std::vector<float> range, std::vector<float> velocity, std::vector<float> rcs, std::vector<float> azimuth,
                       std::vector<float> position_rx, int samp_rate, float center_freq, float self_coupling_db,
                       bool rndm_phaseshift, bool self_coupling) override                                      // :121-198
    {
        jrc_rt::thread::scoped_lock lock(d_setlock);
        if (velocity.size() != range.size() || rcs.size() != range.size() || azimuth.size() != range.size())
            throw std::invalid_argument("[TARGET SIM] range, velocity, rcs and azimuth must have the same length");   // FIXME at :160
        if (position_rx.size() != d_n_rx)
            throw std::invalid_argument("[TARGET SIM] position_rx must keep the number of output streams");
        jrc_tsim_cfg c;
        c.n_targets = (int)range.size();
        c.range = range.data(); c.velocity = velocity.data(); c.rcs = rcs.data(); c.azimuth = azimuth.data();
        c.n_rx = (int)position_rx.size(); c.position_rx = position_rx.data();
        c.samp_rate = samp_rate; c.center_freq = center_freq; c.self_coupling_db = self_coupling_db;
        c.rndm_phaseshift = rndm_phaseshift; c.self_coupling = self_coupling;
        c.sum_targets = 0;                  // as written in the reference (:354-366)
        c.max_bursts = 1;
        jrc_tsim* fresh = jrc_tsim_create(d_c.ctx, &c);
        if (!fresh) throw std::invalid_argument(jrc_last_error(d_c.ctx));
        jrc_tsim_destroy(d_sim);
        d_sim = fresh;
        d_samp_rate = samp_rate; d_num_targets = c.n_targets; d_rndm_phaseshift = rndm_phaseshift;
        d_key = pmt::string_to_symbol("rx_time");                                                             // :153-154
        d_srcid = pmt::string_to_symbol("stat_targ_sim");
        if (d_rndm_phaseshift) std::srand(std::time(NULL));                                                   // :196
    }

    int calculate_output_stream_length(const gr_vector_int& ninput_items) override { return ninput_items[0]; }   // :113-118

    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star& out) override
    {
        jrc_rt::thread::scoped_lock lock(d_setlock);
        const int n_input = ninput_items[0];
        const gr_complex* phases = nullptr;
        if (d_rndm_phaseshift) {                                                                              // :313-322
            d_phase.resize(d_num_targets);
            for (int k = 0; k < d_num_targets; k++)
                d_phase[k] = std::exp(gr_complex(0, 2 * M_PI * float((std::rand() % 1000 + 1) / 1000.0)));
            phases = d_phase.data();
        }
        for (size_t l = 0; l < d_n_rx; l++) {                                                                 // :331-335
            const uint64_t time_sec = nitems_written(l) / d_samp_rate;
            const double time_frac_sec = nitems_written(l) / (float)d_samp_rate - time_sec;
            add_item_tag(l, nitems_written(l), d_key, pmt::make_tuple(pmt::from_uint64(time_sec), pmt::from_double(time_frac_sec)), d_srcid);
        }
        std::vector<jrc_cf32*> o(d_n_rx);
        for (size_t l = 0; l < d_n_rx; l++) o[l] = (jrc_cf32*)out[l];
        int n = jrc_tsim_work(d_sim, (const jrc_cf32*)in[0], n_input, o.data(), (const jrc_cf32*)phases);
        d_c.check(n);
        return n_input;                                                                                       // :384
    }
};
target_simulator::sptr target_simulator::make(std::vector<float> range, std::vector<float> velocity, std::vector<float> rcs,
                                              std::vector<float> azimuth, std::vector<float> position_rx, int samp_rate,
                                              float center_freq, float self_coupling_db, bool rndm_phaseshift, bool self_coupling,
                                              const std::string& len_key, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new target_simulator_impl(range, velocity, rcs, azimuth, position_rx, samp_rate, center_freq,
                                                          self_coupling_db, rndm_phaseshift, self_coupling, len_key, debug));
}

// =================================================================================================
// stream_encoder  (lib/stream_encoder_impl.cc)
// =================================================================================================
class stream_encoder_impl : public stream_encoder {
    ctx_holder d_c;
    int d_data_len, d_symbol_len = 0, d_offset = 0;
    MCS d_mod_encode;
    uint8_t d_scrambler = 1;                                                                         // :53
    std::vector<gr_complex> d_complex_symbols;
    std::mutex d_mutex;
    static const int MAX_PAYLOAD_SIZE = 3100;                                                        // lib/utils.h

public:
    stream_encoder_impl(MCS mod_encode, int data_len, int, bool)
        : jrc_rt::block("stream_encoder", jrc_rt::io_signature::make(0, 0, 0), jrc_rt::io_signature::make(1, 1, sizeof(gr_complex))),
          d_data_len(data_len), d_mod_encode(mod_encode)
    {
        message_port_register_in(pmt::mp("pdu_in"));                                                 // :51
        if (jrc_stream_n_ofdm_sym(mod_encode, data_len, 4) < 0) throw std::invalid_argument("wrong encoding");
    }
    void set_mcs(MCS mod_encode) override                                                            // :272-278
    {
        std::unique_lock<std::mutex> lock(d_mutex);
        d_mod_encode = mod_encode;
    }
    int general_work(int noutput_items, gr_vector_int&, gr_vector_const_void_star&, gr_vector_void_star& output_items) override
    {
        gr_complex* out = (gr_complex*)output_items[0];
        if (d_offset == 0) {                                      // nothing left of the previous PDU: fetch the next one (:88-100)
            const pmt::pmt_t msg = delete_head_nowait(pmt::intern("pdu_in"));
            if (!msg.get()) return 0;
            std::unique_lock<std::mutex> lock(d_mutex);
            const pdu_view pdu(msg);
            const int packet_size_byte = pdu.size, packet_type = pdu.first_byte();
            if (packet_size_byte + 4 > MAX_PAYLOAD_SIZE) {         // payload + CRC-32 beyond the codec's buffers (:139-143)
                std::cout << "[STREAM ENCODER] Data Packet too Large -> Maximun Packet Size (byte): " << MAX_PAYLOAD_SIZE << std::endl;
                return 0;
            }
            const int n_sym = jrc_stream_n_ofdm_sym(d_mod_encode, d_data_len, packet_size_byte + 4);
            if (n_sym < 0) throw std::invalid_argument("wrong encoding");                            // :206-208
            d_symbol_len = n_sym * d_data_len;                                                       // :186
            d_complex_symbols.assign((size_t)d_symbol_len, gr_complex(0, 0));
            int n = jrc_stream_encode(d_c.ctx, d_mod_encode, d_data_len, pdu.bytes, packet_size_byte, d_scrambler++,
                                      (jrc_cf32*)d_complex_symbols.data(), d_symbol_len);
            if (d_scrambler > 127) d_scrambler = 1;                                                  // :171-175
            d_c.check(n);
            const pmt::pmt_t srcid = pmt::string_to_symbol(alias());                                 // tags (:224-241)
            add_item_tag(0, nitems_written(0), pmt::string_to_symbol("packet_len"), pmt::from_long(d_symbol_len), srcid);
            add_item_tag(0, nitems_written(0), pmt::mp("packet_type"), pmt::from_long(packet_type), srcid);
            add_item_tag(0, nitems_written(0), pmt::mp("mcs"), pmt::from_long(d_mod_encode), srcid);
            add_item_tag(0, nitems_written(0), pmt::mp("pdu_len"), pmt::from_long(packet_size_byte + 4), srcid);
        }
        const int n_out = std::min(noutput_items, d_symbol_len - d_offset);                          // :253-266
        std::memcpy(out, d_complex_symbols.data() + d_offset, n_out * sizeof(gr_complex));
        d_offset += n_out;
        if (d_offset == d_symbol_len) { d_offset = 0; d_complex_symbols.clear(); }
        return n_out;
    }
};
stream_encoder::sptr stream_encoder::make(MCS mod_encode, int data_len, int N_ss_radar, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new stream_encoder_impl(mod_encode, data_len, N_ss_radar, debug));
}

// =================================================================================================
// stream_decoder  (lib/stream_decoder_impl.cc)
// =================================================================================================
class stream_decoder_impl : public stream_decoder {
    ctx_holder d_c;
    int d_n_data_carriers;
    std::string d_comm_log_file;
    bool d_stats_record, d_frame_rx_complete = true, d_start_decoding = false, d_new_stat_started = false;
    std::deque<float> per_stats, snr_data_stats;                     // rolling windows of 25 and 1 (:64-65)
    float d_snr_est = 0, d_snr_data_est = 0;                         // dB (impl.h:83-84)
    int d_data_length = 0, d_mcs = 0, d_packet_type = 0, d_n_ofdm_sym = 0, n_copied = 0;
    std::vector<gr_complex> d_rx_symbols, chan_est_mean;
    std::vector<uint8_t> d_payload;
    std::mutex d_mutex;
    static const int MAX_PAYLOAD_SIZE = 3100, info_bytes = 2 + 2 * sizeof(float);                     // lib/utils.h, impl.h:88
    static const int MAX_SYM = ((16 + 8 * MAX_PAYLOAD_SIZE + 6) / 24) + 1;

    static float rolling_mean(const std::deque<float>& w)
    {
        if (w.empty()) return 0.f;
        double s = 0; for (float v : w) s += v;
        return (float)(s / w.size());
    }
    static void push(std::deque<float>& w, float v, size_t size) { w.push_back(v); if (w.size() > size) w.pop_front(); }

    void publish_stats()                                                                             // :262-274, :311-323
    {
        float per_val = 100.0 * rolling_mean(per_stats);
        pmt::pmt_t per_pack = pmt::list2(pmt::string_to_symbol("per"), pmt::init_f32vector(1, &per_val));
        float snr_val = rolling_mean(snr_data_stats);
        pmt::pmt_t snr_pack = pmt::list2(pmt::string_to_symbol("snr"), pmt::init_f32vector(1, &snr_val));
        message_port_pub(pmt::mp("stats"), pmt::list2(per_pack, snr_pack));
    }
    void publish_sym(bool ok)                                                                        // :249-257, :296-304
    {
        std::vector<uint8_t> sob((size_t)info_bytes + (d_data_length > 4 ? d_data_length - 4 : 0));
        sob[0] = ok ? 1 : 0;
        sob[1] = (uint8_t)d_packet_type;
        std::memcpy(&sob[2], &d_snr_est, sizeof(float));
        std::memcpy(&sob[2 + sizeof(float)], &d_snr_data_est, sizeof(float));
        if (d_data_length > 4) std::memcpy(&sob[info_bytes], d_payload.data(), (size_t)d_data_length - 4);
        pmt::pmt_t dict = pmt::dict_add(pmt::make_dict(), pmt::mp("SNR"), pmt::from_double(d_snr_est));
        message_port_pub(pmt::mp("sym"), pmt::cons(dict, pmt::make_blob(sob.data(), sob.size())));
    }
    void log_line(std::ofstream& f, int ok)                                                          // :282-294, :331-343
    {
        f << current_date_time2() << ", \t" << ok << ", \t" << d_packet_type << ", \t" << d_mcs << ", \t" << d_snr_est << ", \t"
          << d_snr_data_est << ", \t" << d_data_length << ", \t";
        for (size_t i = 0; i < chan_est_mean.size(); i++) f << chan_est_mean[i] << ";";
        f << "\n";
        f.flush();
    }
    void decode()                                                                                     // :205-405
    {
        std::ofstream file_stream(d_comm_log_file, std::ofstream::app);
        if (d_stats_record) {
            if (!file_stream.is_open()) throw std::runtime_error("[STREAM DECODER] Could not open file!!");
            if (!d_new_stat_started) {
                file_stream << "\n NEW RECORD - " << current_date_time() << "\n";
                file_stream.flush();
                d_new_stat_started = true;
            }
        }
        d_payload.assign((size_t)std::max(d_data_length, 8), 0);
        int crc_ok = 0;
        int n = jrc_stream_decode(d_c.ctx, d_mcs, d_n_data_carriers, d_data_length, (const jrc_cf32*)d_rx_symbols.data(),
                                  d_n_ofdm_sym * d_n_data_carriers, d_payload.data(), &crc_ok);
        d_c.check(n);
        if (!crc_ok) {                                                                                // :246-296
            std::cerr << "[STREAM DECODER] Data Checksum is WRONG!!! --> Dropping Packet, bytes:" << d_data_length << std::endl;
            publish_sym(false);
            publish_stats();                                             // statistics before this failure is counted
            push(per_stats, 1, 25);
            push(snr_data_stats, d_snr_data_est, 1);
            if (d_stats_record) log_line(file_stream, 0);
            return;
        }
        push(per_stats, 0, 25);                                                                       // :299-301
        push(snr_data_stats, d_snr_data_est, 1);
        publish_sym(true);
        publish_stats();
        if (d_stats_record) log_line(file_stream, 1);
    }

public:
    stream_decoder_impl(int n_data_carriers, const std::string& comm_log_file, bool stats_record, bool)
        : jrc_rt::block("stream_decoder", jrc_rt::io_signature::make(1, 1, n_data_carriers * sizeof(gr_complex)),
                        jrc_rt::io_signature::make(1, 1, sizeof(float))),                             // :52-55
          d_n_data_carriers(n_data_carriers), d_comm_log_file(comm_log_file), d_stats_record(stats_record)
    {
        message_port_register_out(pmt::mp("sym"));
        message_port_register_out(pmt::mp("stats"));
        set_tag_propagation_policy(TPP_DONT);
        std::ifstream file_stream(d_comm_log_file);
        if (!file_stream.is_open()) std::cerr << "[MIMO PRECODER] Could not open log file!" << std::endl;   // sic (:84-88)
    }
    void set_stats_record(bool stats_record) override                                                 // :469-483
    {
        std::unique_lock<std::mutex> lock(d_mutex);
        d_stats_record = stats_record;
        d_new_stat_started = false;
    }
    int general_work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        const int n_input_items = ninput_items[0];
        const gr_complex* in = (const gr_complex*)input_items[0];
        float* per_out = (float*)output_items[0];
        std::vector<float> per_vector;
        std::vector<jrc_rt::tag_t> tags;
        const uint64_t nread = nitems_read(0);
        jrc_rt::thread::scoped_lock lock(d_setlock);
        int n_in = 0, n_out = 0;
        while (n_in < n_input_items) {
            get_tags_in_range(tags, 0, nread + n_in, nread + n_in + 1, pmt::string_to_symbol("stream_start"));
            if (tags.size()) {                                                                        // :115-147
                d_frame_rx_complete = false;
                pmt::pmt_t dict = tags[0].value;
                d_snr_est = (float)pmt::to_double(pmt::dict_ref(dict, pmt::mp("snr"), pmt::from_double(0)));
                d_data_length = (int)pmt::to_uint64(pmt::dict_ref(dict, pmt::mp("data_bytes"), pmt::from_double(0)));
                d_mcs = (int)pmt::to_uint64(pmt::dict_ref(dict, pmt::mp("mcs"), pmt::from_double(0)));
                d_packet_type = (int)pmt::to_uint64(pmt::dict_ref(dict, pmt::mp("packet_type"), pmt::from_double(0)));
                const int n_sym = jrc_stream_n_ofdm_sym(d_mcs, d_n_data_carriers, d_data_length);
                if (n_sym >= 0 && n_sym <= MAX_SYM && d_data_length <= MAX_PAYLOAD_SIZE) {
                    d_n_ofdm_sym = n_sym;
                    n_copied = 0;
                    d_start_decoding = true;
                    d_rx_symbols.assign((size_t)n_sym * d_n_data_carriers, gr_complex(0, 0));
                } else {
                    d_start_decoding = false;
                }
            }
            get_tags_in_range(tags, 0, nread + n_in, nread + n_in + 1, pmt::mp("stream_end"));
            if (tags.size()) {                                                                        // :149-157
                pmt::pmt_t dict = tags[0].value;
                d_snr_data_est = pmt::to_float(pmt::dict_ref(dict, pmt::mp("snr_data"), pmt::from_double(0)));
                chan_est_mean = pmt::c32vector_elements(pmt::dict_ref(dict, pmt::mp("chan_mean"), pmt::from_double(0)));
            }
            if (n_copied < d_n_ofdm_sym && d_start_decoding) {                                        // :159-185
                // the hard decisions of :166-169 are taken on the device; the symbols are kept until the frame is complete
                std::memcpy(&d_rx_symbols[(size_t)n_copied * d_n_data_carriers], in, d_n_data_carriers * sizeof(gr_complex));
                n_copied++;
                if (n_copied == d_n_ofdm_sym) {
                    decode();
                    in += d_n_data_carriers;
                    n_in++;
                    d_frame_rx_complete = true;
                    n_out++;
                    per_vector.push_back(100.0 * rolling_mean(per_stats));
                    n_copied = 0;
                    break;
                }
            }
            in += d_n_data_carriers;
            n_in++;
        }
        if (n_out) std::memcpy(per_out, per_vector.data(), n_out * sizeof(float));
        consume(0, n_in);
        return n_out;
    }
};
stream_decoder::sptr stream_decoder::make(int n_data_carriers, const std::string& comm_log_file, bool stats_record, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new stream_decoder_impl(n_data_carriers, comm_log_file, stats_record, debug));
}

// =================================================================================================
// moving_avg  (lib/moving_avg_impl.cc)
// =================================================================================================
class moving_avg_impl : public moving_avg {
    ctx_holder d_c;
    int d_length, d_max_iter, d_new_length;
    float d_scale, d_new_scale;
    bool d_updated = false;

public:
    moving_avg_impl(int length, float scale, int max_iter, bool)
        : jrc_rt::sync_block("moving_avg", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)), jrc_rt::io_signature::make(1, 1, sizeof(gr_complex))),
          d_length(length), d_max_iter(max_iter), d_new_length(length), d_scale(scale), d_new_scale(scale)
    {
        set_history(length);                                                                          // :51
    }
    int length() const override { return d_new_length; }
    float scale() const override { return d_new_scale; }
    void set_length_and_scale(int length, float scale) override { d_new_length = length; d_new_scale = scale; d_updated = true; }   // :100-105
    void set_length(int length) override { d_new_length = length; d_updated = true; }
    void set_scale(float scale) override { d_new_scale = scale; d_updated = true; }
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        if (d_updated) {                                                                              // :67-74
            d_length = d_new_length;
            d_scale = d_new_scale;
            set_history(d_length);
            d_updated = false;
            return 0;
        }
        int n = jrc_moving_avg(d_c.ctx, d_length, d_scale, d_max_iter, noutput_items, (const jrc_cf32*)input_items[0], (jrc_cf32*)output_items[0]);
        d_c.check(n);
        return n;                                                                                     // num_iter (:92)
    }
};
moving_avg::sptr moving_avg::make(int length, float scale, int max_iter, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new moving_avg_impl(length, scale, max_iter, debug));
}

// =================================================================================================
// frame_detector  (lib/frame_detector_impl.cc)
// =================================================================================================
class frame_detector_impl : public frame_detector {
    ctx_holder d_c;
    jrc_frame_detector* d_det = nullptr;
    std::mutex d_mutex;

public:
    frame_detector_impl(int fft_len, int cp_len, double threshold, unsigned int min_n_peaks, unsigned int ignore_gap, bool)
        : jrc_rt::block("frame_detector", jrc_rt::io_signature::make3(3, 3, sizeof(gr_complex), sizeof(gr_complex), sizeof(float)),
                        jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)))                         // :42-44
    {
        d_det = jrc_frame_detector_create(d_c.ctx, fft_len, cp_len, threshold, min_n_peaks, ignore_gap);
        if (!d_det) throw std::invalid_argument(jrc_last_error(d_c.ctx));
        set_tag_propagation_policy(TPP_DONT);
    }
    ~frame_detector_impl() override { jrc_frame_detector_destroy(d_det); }
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        std::unique_lock<std::mutex> lock(d_mutex);
        const int ninput = std::min(std::min(ninput_items[0], ninput_items[1]), ninput_items[2]);     // :85
        int consumed = 0, n_tags = 0;
        uint64_t off[8]; double cfo[8];
        int n = jrc_frame_detector_work(d_det, noutput_items, ninput, (const jrc_cf32*)input_items[0], (const jrc_cf32*)input_items[1],
                                        (const float*)input_items[2], (jrc_cf32*)output_items[0], &consumed, off, cfo, 8, &n_tags);
        d_c.check(n);
        for (int i = 0; i < n_tags; i++)                                                              // insert_tag (:197-203)
            add_item_tag(0, off[i], pmt::string_to_symbol("frame_start"), pmt::from_double(cfo[i]), pmt::string_to_symbol(name()));
        consume_each(consumed);
        return n;
    }
};
frame_detector::sptr frame_detector::make(int fft_len, int cp_len, double threshold, unsigned int min_n_peaks, unsigned int ignore_gap, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new frame_detector_impl(fft_len, cp_len, threshold, min_n_peaks, ignore_gap, debug));
}

// =================================================================================================
// frame_sync  (lib/frame_sync_impl.cc)
// =================================================================================================
class frame_sync_impl : public frame_sync {
    ctx_holder d_c;
    jrc_frame_sync* d_sync = nullptr;
    int d_fft_len, d_cp_len;
    std::vector<jrc_rt::tag_t> d_tags;
    std::mutex d_mutex;

public:
    frame_sync_impl(int fft_len, int cp_len, unsigned int sync_length, std::vector<gr_complex> ltf_seq_time, bool)
        : jrc_rt::block("frame_sync", jrc_rt::io_signature::make(2, 2, sizeof(gr_complex)), jrc_rt::io_signature::make(1, 1, sizeof(gr_complex))),
          d_fft_len(fft_len), d_cp_len(cp_len)
    {
        d_sync = jrc_frame_sync_create(d_c.ctx, fft_len, cp_len, sync_length, (const jrc_cf32*)ltf_seq_time.data(), (int)ltf_seq_time.size());
        if (!d_sync) throw std::invalid_argument(jrc_last_error(d_c.ctx));
        set_tag_propagation_policy(TPP_DONT);
    }
    ~frame_sync_impl() override { jrc_frame_sync_destroy(d_sync); }
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override                  // :74-86
    {
        int st = 0;
        jrc_frame_sync_state(d_sync, &st, nullptr, nullptr);
        const int need = st == 0 ? d_fft_len + d_cp_len : noutput_items;
        ninput_items_required[0] = need;
        ninput_items_required[1] = need;
    }
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        std::unique_lock<std::mutex> lock(d_mutex);
        const int ninput = std::min(std::min(ninput_items[0], ninput_items[1]), 8192);               // :111
        const uint64_t nread = nitems_read(0);
        get_tags_in_range(d_tags, 0, nread, nread + ninput);                                         // :120
        std::vector<uint64_t> off; std::vector<double> val;
        for (auto& t : d_tags) { off.push_back(t.offset); val.push_back(pmt::to_double(t.value)); }
        int consumed = 0, n_tag_out = 0;
        uint64_t to = 0; double tv = 0;
        int n = jrc_frame_sync_work(d_sync, noutput_items, ninput_items[0], ninput_items[1], (const jrc_cf32*)input_items[0],
                                    (const jrc_cf32*)input_items[1], off.data(), val.data(), (int)off.size(), (jrc_cf32*)output_items[0],
                                    &consumed, &to, &tv, &n_tag_out);
        d_c.check(n);                                                                                // JRC_ERR_LENGTH_MISMATCH -> std::runtime_error (:135)
        if (n_tag_out) add_item_tag(0, to, pmt::string_to_symbol("frame_start"), pmt::from_double(tv), pmt::string_to_symbol(name()));   // :182-186
        consume(0, consumed);
        consume(1, consumed);
        return n;
    }
};
frame_sync::sptr frame_sync::make(int fft_len, int cp_len, unsigned int sync_length, std::vector<gr_complex> ltf_seq_time, bool debug)
{
    return JRC_GET_INITIAL_SPTR(new frame_sync_impl(fft_len, cp_len, sync_length, ltf_seq_time, debug));
}

// =================================================================================================
// zero_pad  (lib/zero_pad_impl.cc)
// =================================================================================================
class zero_pad_impl : public zero_pad {
    ctx_holder d_c;
    unsigned d_pad_front, d_pad_tail;
    uint64_t d_seed;

public:
    zero_pad_impl(bool, unsigned int pad_front, unsigned int pad_tail)
        : jrc_rt::tagged_stream_block("zero_pad", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)),
                                      jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)), "packet_len"),
          d_pad_front(pad_front), d_pad_tail(pad_tail), d_seed(std::random_device{}())           // std::random_device per call (:70-71)
    {
        set_tag_propagation_policy(TPP_DONT);
    }
    int calculate_output_stream_length(const gr_vector_int& ninput_items) override { return ninput_items[0] + d_pad_front + d_pad_tail; }
    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star& out) override
    {
        int n = jrc_zero_pad(d_c.ctx, ninput_items[0], d_pad_front, d_pad_tail, d_seed++, (const jrc_cf32*)in[0], (jrc_cf32*)out[0]);
        d_c.check(n);
        return n;                                                                               // :91-93
    }
};
zero_pad::sptr zero_pad::make(bool debug, unsigned int pad_front, unsigned int pad_tail)
{
    return JRC_GET_INITIAL_SPTR(new zero_pad_impl(debug, pad_front, pad_tail));
}

// =================================================================================================
// ofdm_frame_generator  (lib/ofdm_frame_generator_impl.cc)
// =================================================================================================
class ofdm_frame_generator_impl : public ofdm_frame_generator {
    ctx_holder d_c;
    jrc_frame_generator* d_gen = nullptr;
    int d_fft_len;
    std::vector<int> d_occ_sizes;
    int d_n_sync;
    std::string d_len_tag_key;
    std::vector<std::vector<int>> d_occupied;          // normalised as the reference keeps them (lib/ofdm_frame_generator_impl.cc:83-96): what its getter returns

public:
    std::string len_tag_key() override { return d_len_tag_key; }
    const int fft_len() override { return d_fft_len; }
    std::vector<std::vector<int>> occupied_carriers() override { return d_occupied; }
    ofdm_frame_generator_impl(int fft_len, const std::vector<std::vector<int>>& occupied_carriers, const std::vector<std::vector<int>>& pilot_carriers,
                              const std::vector<std::vector<gr_complex>>& pilot_symbols, const std::vector<std::vector<gr_complex>>& sync_words,
                              int, const std::string& len_tag_key, const bool output_is_shifted)
        : jrc_rt::tagged_stream_block("ofdm_frame_generator", jrc_rt::io_signature::make(1, 1, sizeof(gr_complex)),
                                      jrc_rt::io_signature::make(1, 1, sizeof(gr_complex) * fft_len), len_tag_key),
          d_fft_len(fft_len), d_n_sync((int)sync_words.size()), d_len_tag_key(len_tag_key), d_occupied(occupied_carriers)
    {
        if (occupied_carriers.empty()) throw std::invalid_argument("Occupied carriers must be of type vector of vector i.e. ((),).");   // :77-82
        for (auto& set : d_occupied)                                                                                                     // :83-96
            for (auto& c : set) {
                if (c < 0) c += fft_len;
                if (c > fft_len || c < 0) throw std::invalid_argument("data carrier index out of bounds");
                if (output_is_shifted) c = (c + fft_len / 2) % fft_len;
            }
        if (pilot_carriers.empty()) throw std::invalid_argument("Pilot carriers must be of type vector of vector i.e. ((),).");
        if (pilot_symbols.empty()) throw std::invalid_argument("Pilot symbols must be of type vector of vector i.e. ((),).");
        for (auto& w : sync_words) if (w.size() != (unsigned)fft_len) throw std::invalid_argument("sync words must be fft length");      // :126-130
        std::vector<int> osz, ofl, psz, pfl, ssz;
        std::vector<gr_complex> sfl, sw;
        for (auto& v : occupied_carriers) { osz.push_back((int)v.size()); ofl.insert(ofl.end(), v.begin(), v.end()); }
        for (auto& v : pilot_carriers) { psz.push_back((int)v.size()); pfl.insert(pfl.end(), v.begin(), v.end()); }
        for (auto& v : pilot_symbols) { ssz.push_back((int)v.size()); sfl.insert(sfl.end(), v.begin(), v.end()); }
        for (auto& v : sync_words) sw.insert(sw.end(), v.begin(), v.end());
        d_occ_sizes = osz;
        ofl.push_back(0); pfl.push_back(0); sfl.push_back(gr_complex(0, 0)); sw.push_back(gr_complex(0, 0));   // never-empty data() pointers
        d_gen = jrc_frame_generator_create(d_c.ctx, fft_len, (int)osz.size(), osz.data(), ofl.data(), (int)psz.size(), psz.data(), pfl.data(),
                                           (int)ssz.size(), ssz.data(), (const jrc_cf32*)sfl.data(), d_n_sync, (const jrc_cf32*)sw.data(),
                                           output_is_shifted ? 1 : 0);
        if (!d_gen) throw std::invalid_argument(jrc_last_error(d_c.ctx));
        set_tag_propagation_policy(TPP_DONT);
    }
    ~ofdm_frame_generator_impl() override { jrc_frame_generator_destroy(d_gen); }
    int calculate_output_stream_length(const gr_vector_int& ninput_items) override { return jrc_frame_generator_output_length(d_gen, ninput_items[0]); }
    int work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& in, gr_vector_void_star& out) override
    {
        // tags travel with the OFDM symbol that carries their item (:177-191)
        std::vector<jrc_rt::tag_t> tags;
        long n_ofdm_symbols = 0;
        int curr_set = 0;
        for (int i = 0; i < ninput_items[0];) {
            const int to_alloc = d_occ_sizes[curr_set];
            get_tags_in_range(tags, 0, nitems_read(0) + i, nitems_read(0) + std::min(i + to_alloc, (int)ninput_items[0]));
            for (auto& t : tags)
                add_item_tag(0, nitems_written(0) + n_ofdm_symbols + (n_ofdm_symbols == 0 ? 0 : d_n_sync), t.key, t.value);
            n_ofdm_symbols++;
            i += to_alloc;
            curr_set = (curr_set + 1) % (int)d_occ_sizes.size();
        }
        int n = jrc_frame_generator_work(d_gen, ninput_items[0], (const jrc_cf32*)in[0], (jrc_cf32*)out[0]);
        d_c.check(n);
        return n;                                                                        // n_ofdm_symbols + sync words (:215)
    }
};
ofdm_frame_generator::sptr ofdm_frame_generator::make(int fft_len, const std::vector<std::vector<int>>& occupied_carriers,
                                                      const std::vector<std::vector<int>>& pilot_carriers,
                                                      const std::vector<std::vector<gr_complex>>& pilot_symbols,
                                                      const std::vector<std::vector<gr_complex>>& sync_words, int ltf_len,
                                                      const std::string& len_tag_key, const bool output_is_shifted)
{
    return JRC_GET_INITIAL_SPTR(new ofdm_frame_generator_impl(fft_len, occupied_carriers, pilot_carriers, pilot_symbols, sync_words, ltf_len,
                                                              len_tag_key, output_is_shifted));
}

}  // namespace mimo_ofdm_jrc
}  // namespace gr
