// jrc_blocks_capi.cc — C test harness around the host-side blocks (stand-alone runtime build only): lets
// tests/test_host_blocks.py construct the blocks through their reference-style make(), feed buffers and tags, run one
// scheduler turn and read back outputs, tags and published messages as JSON.
#ifndef JRC_WITH_GNURADIO
#include <cstring>

#include "jrc_blocks.h"

using namespace gr::mimo_ofdm_jrc;

namespace {
thread_local std::string g_err;
struct handle { std::shared_ptr<jrc_host::block> b; };
template <class F> void* guard_make(F f)
{
    try { auto* h = new handle(); h->b = f(); return h; }
    catch (const std::exception& e) { g_err = std::string(typeid(e).name()) + ": " + e.what(); return nullptr; }
}
std::vector<std::vector<gr_complex>> rows(const float* p, int n_rows, int n_cols)
{
    std::vector<std::vector<gr_complex>> v(n_rows, std::vector<gr_complex>(n_cols));
    for (int r = 0; r < n_rows; r++)
        for (int c = 0; c < n_cols; c++) v[r][c] = gr_complex(p[2 * ((size_t)r * n_cols + c)], p[2 * ((size_t)r * n_cols + c) + 1]);
    return v;
}
}  // namespace

extern "C" {

const char* jrcb_last_error() { return g_err.c_str(); }
void jrcb_destroy(void* h) { delete (handle*)h; }

void* jrcb_make_radar2(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int bg_removal, int bg_recording, int record_len,
                       int interp_factor, int interleave, const char* radar_chan_file)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<mimo_ofdm_radar>(
        mimo_ofdm_radar::make(fft_len, N_tx, N_rx, N_sym, N_pre, bg_removal, bg_recording, record_len, interp_factor, interleave,
                              radar_chan_file ? radar_chan_file : ""))); });
}
void* jrcb_make_radar(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int bg_removal, int bg_recording, int record_len,
                      int interp_factor, int interleave)
{
    return jrcb_make_radar2(fft_len, N_tx, N_rx, N_sym, N_pre, bg_removal, bg_recording, record_len, interp_factor, interleave, "");
}
void* jrcb_make_radar_chain(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int interp_range, int interp_angle, int interleave,
                            const float* rb, int n_rb, const float* ab, int n_ab, float ndr, float nda, float snr_thr, float pow_thr,
                            const char* stats_path, int stats_record, int frames_per_batch, int batches_in_flight)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<radar_chain>(
        radar_chain::make(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, interleave != 0, std::vector<float>(rb, rb + n_rb),
                          std::vector<float>(ab, ab + n_ab), ndr, nda, snr_thr, pow_thr, stats_path, stats_record != 0, frames_per_batch,
                          batches_in_flight))); });
}
void* jrcb_make_radar_chain_bg(int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int interp_range, int interp_angle, int interleave,
                               const float* rb, int n_rb, const float* ab, int n_ab, float ndr, float nda, float snr_thr, float pow_thr,
                               const char* stats_path, int stats_record, int frames_per_batch, int batches_in_flight, int bg_removal,
                               int bg_recording, int record_len)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<radar_chain>(
        radar_chain::make(fft_len, N_tx, N_rx, N_sym, N_pre, interp_range, interp_angle, interleave != 0, std::vector<float>(rb, rb + n_rb),
                          std::vector<float>(ab, ab + n_ab), ndr, nda, snr_thr, pow_thr, stats_path, stats_record != 0, frames_per_batch,
                          batches_in_flight, "packet_len", false, bg_removal != 0, bg_recording != 0, record_len))); });
}
void* jrcb_make_transpose(int input_len, int output_len, int interp)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<matrix_transpose>(
        matrix_transpose::make(input_len, output_len, interp, false))); });
}
void* jrcb_make_estimator(int vlen, const float* rb, int n_rb, const float* ab, int n_ab, float ndr, float nda, float snr_thr,
                          float pow_thr, const char* stats_path, int stats_record)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<range_angle_estimator>(
        range_angle_estimator::make(vlen, std::vector<float>(rb, rb + n_rb), std::vector<float>(ab, ab + n_ab), ndr, nda, snr_thr,
                                    pow_thr, stats_path, stats_record))); });
}
void* jrcb_make_cp_remover(int fft_len, int cp_len)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<ofdm_cyclic_prefix_remover>(
        ofdm_cyclic_prefix_remover::make(fft_len, cp_len))); });
}
void* jrcb_make_peak_detect(int samp_rate, float interp, float threshold, int samp_protect)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<fft_peak_detect>(
        fft_peak_detect::make(samp_rate, interp, threshold, samp_protect, {}, false, "packet_len"))); });
}
void* jrcb_make_equalizer(int algo, double freq, double bw, int fft_len, int cp_len, const int* dc, int n_dc, const int* pc, int n_pc,
                          const float* pilot_symbols, int n_rows, const float* ltf, const float* mapped, int mapped_cols,
                          int n_mimo_ltf, const char* chan_est_file)
{
    return guard_make([&] {
        std::vector<gr_complex> l(fft_len);
        for (int i = 0; i < fft_len; i++) l[i] = gr_complex(ltf[2 * i], ltf[2 * i + 1]);
        return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<mimo_ofdm_equalizer>(mimo_ofdm_equalizer::make(
            (ChannelEstimator)algo, freq, bw, fft_len, cp_len, std::vector<int>(dc, dc + n_dc), std::vector<int>(pc, pc + n_pc),
            rows(pilot_symbols, n_rows, n_pc), l, rows(mapped, fft_len, mapped_cols), n_mimo_ltf, chan_est_file, "", false, false)));
    });
}
void* jrcb_make_precoder(int fft_len, int N_tx, const int* dc, int n_dc, const int* pc, int n_pc, const float* pilot_symbols,
                         int n_rows, const float* sync, int n_sync, const float* mapped, const char* chan_est_file, int smoothing,
                         const char* radar_log_file, int radar_aided, int phased, int radar_streams)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<mimo_precoder>(mimo_precoder::make(
        fft_len, N_tx, 1, std::vector<int>(dc, dc + n_dc), std::vector<int>(pc, pc + n_pc), rows(pilot_symbols, n_rows, n_pc),
        rows(sync, n_sync, fft_len), rows(mapped, fft_len, N_tx * N_tx), chan_est_file, smoothing, radar_log_file, radar_aided, phased,
        radar_streams))); });
}

// kind: 0 long, 1 uint64, 2 double
void* jrcb_make_target_simulator(const float* range, const float* velocity, const float* rcs, const float* azimuth, int n_targets,
                                 const float* position_rx, int n_rx, int samp_rate, float center_freq, float self_coupling_db,
                                 int rndm_phaseshift, int self_coupling)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<target_simulator>(
        target_simulator::make(std::vector<float>(range, range + n_targets), std::vector<float>(velocity, velocity + n_targets),
                               std::vector<float>(rcs, rcs + n_targets), std::vector<float>(azimuth, azimuth + n_targets),
                               std::vector<float>(position_rx, position_rx + n_rx), samp_rate, center_freq, self_coupling_db,
                               rndm_phaseshift != 0, self_coupling != 0))); });
}

// carrier / symbol sets flattened with per-set sizes
void* jrcb_make_frame_generator(int fft_len, int n_occ, const int* occ_sizes, const int* occ_flat, int n_pil, const int* pil_sizes, const int* pil_flat,
                                int n_ps, const int* ps_sizes, const float* ps_flat, int n_sync, const float* sync_words, int shifted)
{
    return guard_make([&] {
        std::vector<std::vector<int>> occ, pil;
        std::vector<std::vector<gr_complex>> ps, sw;
        for (int k = 0, p = 0; k < n_occ; k++) { occ.emplace_back(occ_flat + p, occ_flat + p + occ_sizes[k]); p += occ_sizes[k]; }
        for (int k = 0, p = 0; k < n_pil; k++) { pil.emplace_back(pil_flat + p, pil_flat + p + pil_sizes[k]); p += pil_sizes[k]; }
        for (int k = 0, p = 0; k < n_ps; k++) { ps.emplace_back((const gr_complex*)ps_flat + p, (const gr_complex*)ps_flat + p + ps_sizes[k]); p += ps_sizes[k]; }
        for (int k = 0; k < n_sync; k++) sw.emplace_back((const gr_complex*)sync_words + (size_t)k * fft_len, (const gr_complex*)sync_words + (size_t)(k + 1) * fft_len);
        return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<ofdm_frame_generator>(
            ofdm_frame_generator::make(fft_len, occ, pil, ps, sw, 0, "packet_len", shifted != 0)));
    });
}
void* jrcb_make_zero_pad(unsigned pad_front, unsigned pad_tail)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<zero_pad>(zero_pad::make(false, pad_front, pad_tail))); });
}
void* jrcb_make_moving_avg(int length, float scale, int max_iter)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<moving_avg>(moving_avg::make(length, scale, max_iter, false))); });
}
void* jrcb_make_frame_detector(int fft_len, int cp_len, double threshold, unsigned min_n_peaks, unsigned ignore_gap)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<frame_detector>(
        frame_detector::make(fft_len, cp_len, threshold, min_n_peaks, ignore_gap, false))); });
}
void* jrcb_make_frame_sync(int fft_len, int cp_len, unsigned sync_length, const float* ltf_seq_time, int ntaps)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<frame_sync>(
        frame_sync::make(fft_len, cp_len, sync_length, std::vector<gr_complex>((const gr_complex*)ltf_seq_time, (const gr_complex*)ltf_seq_time + ntaps), false))); });
}
void* jrcb_make_stream_encoder(int mcs, int data_len)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<stream_encoder>(
        stream_encoder::make((MCS)mcs, data_len, 0, false))); });
}
void* jrcb_make_stream_decoder(int n_data_carriers, const char* comm_log_file, int stats_record)
{
    return guard_make([&] { return std::static_pointer_cast<jrc_host::block>(std::shared_ptr<stream_decoder>(
        stream_decoder::make(n_data_carriers, comm_log_file, stats_record != 0, false))); });
}
// message delivery: kind 0 = pmt symbol (string), 1 = PDU pair (dict . blob), 2 = something else (a pmt long)
int jrcb_post_msg(void* h, const char* port, int kind, const uint8_t* data, int len)
{
    auto& b = ((handle*)h)->b;
    pmt::pmt_t m = kind == 0 ? pmt::string_to_symbol(std::string((const char*)data, (size_t)len))
                 : kind == 1 ? pmt::cons(pmt::make_dict(), pmt::make_blob(data, (size_t)len)) : pmt::from_long(len);
    b->t_msg_in[port].push_back(m);
    return 0;
}
// the equalizer's dictionary tags (lib/mimo_ofdm_equalizer_impl.cc:331-337, :626-629) on an input stream
int jrcb_add_stream_start(void* h, int port, uint64_t offset, long data_bytes, long mcs, long packet_type, double snr)
{
    auto& b = ((handle*)h)->b;
    pmt::pmt_t d = pmt::make_dict();
    d = pmt::dict_add(d, pmt::mp("data_bytes"), pmt::from_uint64((uint64_t)data_bytes));
    d = pmt::dict_add(d, pmt::mp("mcs"), pmt::from_uint64((uint64_t)mcs));
    d = pmt::dict_add(d, pmt::mp("packet_type"), pmt::from_uint64((uint64_t)packet_type));
    d = pmt::dict_add(d, pmt::mp("snr"), pmt::from_double(snr));
    jrc_host::tag_t t; t.offset = offset; t.key = pmt::mp("stream_start"); t.value = d;
    b->t_in_tags[port].push_back(t);
    return 0;
}
int jrcb_add_stream_end(void* h, int port, uint64_t offset, double snr_data, const float* chan_mean, int n_chan)
{
    auto& b = ((handle*)h)->b;
    pmt::pmt_t d = pmt::make_dict();
    d = pmt::dict_add(d, pmt::mp("snr_data"), pmt::from_double(snr_data));
    d = pmt::dict_add(d, pmt::mp("chan_mean"), pmt::init_c32vector((size_t)n_chan, (const gr_complex*)chan_mean));
    jrc_host::tag_t t; t.offset = offset; t.key = pmt::mp("stream_end"); t.value = d;
    b->t_in_tags[port].push_back(t);
    return 0;
}

int jrcb_add_in_tag(void* h, int port, uint64_t offset, const char* key, int kind, long lv, double dv)
{
    auto& b = ((handle*)h)->b;
    if (port < 0 || port >= (int)b->t_in_tags.size()) return -1;
    jrc_host::tag_t t;
    t.offset = offset; t.key = pmt::mp(key);
    t.value = kind == 0 ? pmt::from_long(lv) : (kind == 1 ? pmt::from_uint64((uint64_t)lv) : pmt::from_double(dv));
    b->t_in_tags[port].push_back(t);
    return 0;
}

// `count` tags `stride` items apart in one call: what an upstream block's own thread has attached by the time the scheduler calls work()
// (a harness that adds them one ctypes call at a time pays more for the tags than the block pays for the packets)
int jrcb_add_in_tags(void* h, int port, uint64_t first_offset, uint64_t stride, int count, const char* key, int kind, long lv, double dv)
{
    auto& b = ((handle*)h)->b;
    if (port < 0 || port >= (int)b->t_in_tags.size() || count < 0) return -1;
    jrc_host::tag_t t;
    t.key = pmt::mp(key);
    t.value = kind == 0 ? pmt::from_long(lv) : (kind == 1 ? pmt::from_uint64((uint64_t)lv) : pmt::from_double(dv));
    auto& v = b->t_in_tags[port];
    v.reserve(v.size() + (size_t)count);
    for (int i = 0; i < count; i++) { t.offset = first_offset + (uint64_t)i * stride; v.push_back(t); }
    return 0;
}

// one scheduler turn; returns items produced, or -1000 on exception (see jrcb_last_error)
int jrcb_run(void* h, int noutput_items, const int* ninput_items, int n_in, const void* const* in, int n_out, void* const* out)
{
    auto& b = ((handle*)h)->b;
    gr_vector_int nin(ninput_items, ninput_items + n_in);
    gr_vector_const_void_star vin(in, in + n_in);
    gr_vector_void_star vout(out, out + n_out);
    try { return b->t_run(noutput_items, nin, vin, vout); }
    catch (const std::invalid_argument& e) { g_err = std::string("invalid_argument: ") + e.what(); return -1001; }
    catch (const std::exception& e) { g_err = std::string("runtime_error: ") + e.what(); return -1000; }
}
int jrcb_consumed(void* h, int port) { return ((handle*)h)->b->t_consumed[port]; }

// JSON dump of everything the scheduler would see after the turns so far
int jrcb_state_json(void* h, char* buf, int len)
{
    auto& b = ((handle*)h)->b;
    std::ostringstream o;
    o << "{\"out_tags\":[";
    for (size_t p = 0; p < b->t_out_tags.size(); p++) {
        if (p) o << ',';
        o << '[';
        for (size_t i = 0; i < b->t_out_tags[p].size(); i++) {
            auto& t = b->t_out_tags[p][i];
            if (i) o << ',';
            o << "{\"offset\":" << (long long)t.offset << ",\"key\":\"" << t.key->s << "\",\"value\":";
            pmt::to_json(t.value, o);
            o << '}';
        }
        o << ']';
    }
    o << "],\"published\":[";
    std::lock_guard<std::mutex> pub_guard(b->t_pub_lock);
    for (size_t i = 0; i < b->t_published.size(); i++) {
        if (i) o << ',';
        o << "{\"port\":\"" << b->t_published[i].first << "\",\"msg\":";
        pmt::to_json(b->t_published[i].second, o);
        o << '}';
    }
    o << "],\"nitems_read\":[";
    for (size_t i = 0; i < b->t_read.size(); i++) { if (i) o << ','; o << b->t_read[i]; }
    o << "],\"nitems_written\":[";
    for (size_t i = 0; i < b->t_written.size(); i++) { if (i) o << ','; o << b->t_written[i]; }
    o << "]}";
    std::string s = o.str();
    if ((int)s.size() + 1 > len) return -(int)s.size() - 1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// accumulated time inside radar_chain's general_work by phase (0 staging, 1 feed submits, 2 collect + publish, 3 total), nanoseconds, 64 bits:
// what a soak run differences (the "profile_us" name of jrcb_call_setter saturates at INT_MAX microseconds).  -1: not a radar_chain.
long long jrcb_profile_ns(void* h, int what)
{
    auto& b = ((handle*)h)->b;
    if (auto* rc = dynamic_cast<radar_chain*>(b.get())) return (long long)rc->profile_ns(what);
    return -1;
}

int jrcb_call_setter(void* h, const char* name, double v)
{
    auto& b = ((handle*)h)->b;
    std::string n(name);
    try {
        if (auto* r = dynamic_cast<mimo_ofdm_radar*>(b.get())) {
            if (n == "set_background_record") { r->set_background_record(v != 0); return 0; }
            if (n == "capture_radar_data") { r->capture_radar_data(v != 0); return 0; }
        }
        if (auto* rc = dynamic_cast<radar_chain*>(b.get())) {
            if (n == "set_background_record") { rc->set_background_record(v != 0); return 0; }
            if (n == "n_devices") return rc->n_devices();
            if (n == "frames_done") return rc->frames_done();
            if (n == "flush") { rc->flush(); return 0; }
            if (n == "stop") { return rc->stop() ? 0 : -1; }
            if (n == "pending_batches") return rc->pending_batches();
            if (n == "rx_only_batches") return (int)rc->rx_only_batches();
            if (n == "profile_us") {                   // an int holds 35 minutes of microseconds: saturate instead of wrapping (long runs: jrcb_profile_ns)
                const long long us = rc->profile_ns((int)v) / 1000;
                return us > 2147483647LL ? 2147483647 : (int)us;
            }
        }
        if (auto* e = dynamic_cast<range_angle_estimator*>(b.get())) {
            if (n == "set_snr_threshold") { e->set_snr_threshold((float)v); return 0; }
            if (n == "set_power_threshold") { e->set_power_threshold((float)v); return 0; }
            if (n == "set_stats_record") { e->set_stats_record(v != 0); return 0; }
        }
        if (auto* q = dynamic_cast<mimo_ofdm_equalizer*>(b.get())) {
            if (n == "set_estimator") { q->set_estimator((ChannelEstimator)(int)v); return 0; }
            if (n == "set_bandwidth") { q->set_bandwidth(v); return 0; }
            if (n == "set_frequency") { q->set_frequency(v); return 0; }
        }
        if (auto* p = dynamic_cast<mimo_precoder*>(b.get())) {
            if (n == "set_chan_est_smoothing") { p->set_chan_est_smoothing(v != 0); return 0; }
            if (n == "set_radar_aided") { p->set_radar_aided(v != 0); return 0; }
            if (n == "set_use_radar_streams") { p->set_use_radar_streams(v != 0); return 0; }
            if (n == "set_phased_steering") { p->set_phased_steering(v != 0); return 0; }
        }
        if (auto* ma = dynamic_cast<moving_avg*>(b.get())) {
            if (n == "set_length") { ma->set_length((int)v); return 0; }
            if (n == "set_scale") { ma->set_scale((float)v); return 0; }
        }
        if (auto* se = dynamic_cast<stream_encoder*>(b.get())) { if (n == "set_mcs") { se->set_mcs((MCS)(int)v); return 0; } }
        if (auto* sd = dynamic_cast<stream_decoder*>(b.get())) { if (n == "set_stats_record") { sd->set_stats_record(v != 0); return 0; } }
        if (auto* d = dynamic_cast<fft_peak_detect*>(b.get())) {
            if (n == "set_threshold") { d->set_threshold((float)v); return 0; }
            if (n == "set_samp_protect") { d->set_samp_protect((int)v); return 0; }
        }
    } catch (const std::exception& e) { g_err = e.what(); return -1000; }
    return -1;
}

}  // extern "C"
#endif
