// Test double of the part of include/jrc.h that host/jrc_blocks.cc's radar_chain block drives: a host-fed pipeline with slots, batches that
// take a while to finish, results in submission order.  CPU only, no arithmetic of the radar path: each frame's "result" carries numbers the
// driver can recompute from what it staged (a checksum of the frame's receive ports, and of the TX ports the feed saw for it — the staged
// ones after a full submission, the resident copy after a receive-only one).  It exists so that the block's THREADING (scheduler thread,
// flusher thread, getters from a third thread) can run under ThreadSanitizer / AddressSanitizer here, where no GPU and so no libjrc_hip.so
// can run.  Test infrastructure only; nothing under gr-mimo-ofdm-jrc_amd/ links it.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "jrc.h"

using clk = std::chrono::steady_clock;

struct jrc_ctx { int device; };

struct batch { int slot; int n; clk::time_point done; bool rx_only; };

struct jrc_chain_feed {
    jrc_chain_cfg cfg;
    int n_slots, fps, next_slot = 0;
    size_t port, frame;                                   // complex samples per port / per frame
    std::vector<std::vector<jrc_cf32>> stage;
    std::vector<jrc_cf32> tx_resident;
    bool have_tx = false;
    std::deque<batch> flight;
    long latency_us;
    std::string err;
    std::atomic<int> inside{0};                           // > 1 = two threads inside the feed at once: the contract says one feeder thread
    bool overlapped = false;
    long collects = 0, fail_collect_at = -1;              // FEED_DOUBLE_FAIL_COLLECT_AT=k: the k-th collect call and every later one fail
};
// what the block still collected while it was being destroyed is only visible from outside the block: totals over all feeds of the process
std::atomic<long> feed_double_frames_collected{0}, feed_double_collect_calls_after_failure{0};

namespace {
struct in_feed {                                          // the feed is not thread-safe: the block must serialise every call
    jrc_chain_feed* f;
    explicit in_feed(const jrc_chain_feed* p) : f(const_cast<jrc_chain_feed*>(p)) { if (f->inside.fetch_add(1) != 0) f->overlapped = true; }
    ~in_feed() { f->inside.fetch_sub(1); }
};
float checksum(const jrc_cf32* p, size_t n)
{
    double s = 0;
    for (size_t i = 0; i < n; i++) s += (double)p[i].re * (double)((i % 7) + 1) - (double)p[i].im * (double)((i % 5) + 1);
    return (float)s;
}
int fail(jrc_chain_feed* f, int st, const char* msg) { f->err = msg; return st; }
}  // namespace

extern "C" {

int jrc_create(int device, jrc_ctx** ctx) { *ctx = new jrc_ctx{device}; return JRC_OK; }
void jrc_destroy(jrc_ctx* ctx) { delete ctx; }
const char* jrc_last_error(const jrc_ctx*) { return "feed double"; }
const char* jrc_strerror(int) { return "feed double error"; }

int jrc_chain_feed_create(jrc_ctx*, const jrc_chain_cfg* cfg, const float*, const float*, int n_slots, int frames_per_slot, int, int,
                          jrc_chain_feed** feed)
{
    auto* f = new jrc_chain_feed;
    f->cfg = *cfg; f->n_slots = n_slots; f->fps = frames_per_slot;
    f->port = (size_t)cfg->n_items * cfg->fft_len;
    f->frame = f->port * (cfg->N_tx + cfg->N_rx);
    f->stage.assign(n_slots, std::vector<jrc_cf32>(f->frame * frames_per_slot));
    f->tx_resident.resize(f->port * cfg->N_tx);
    const char* e = getenv("FEED_DOUBLE_LATENCY_US");
    f->latency_us = e ? atol(e) : 300;
    if (const char* k = getenv("FEED_DOUBLE_FAIL_COLLECT_AT")) f->fail_collect_at = atol(k);
    *feed = f;
    return JRC_OK;
}
int jrc_chain_feed_create_multi(const int*, int, const jrc_chain_cfg*, const float*, const float*, int, int, int, int, jrc_chain_feed**)
{
    return JRC_ERR_UNSUPPORTED;
}
void jrc_chain_feed_destroy(jrc_chain_feed* f) { delete f; }
const char* jrc_chain_feed_last_error(const jrc_chain_feed* f) { return f->err.c_str(); }
int jrc_chain_feed_pending(const jrc_chain_feed* f) { in_feed g(f); return (int)f->flight.size(); }
int jrc_chain_feed_poll(const jrc_chain_feed* f) { in_feed g(f); return !f->flight.empty() && clk::now() >= f->flight.front().done; }
int jrc_chain_feed_set_write_map(jrc_chain_feed* f, int) { in_feed g(f); return JRC_OK; }
int jrc_chain_feed_set_background(jrc_chain_feed* f, int, int, int)
{
    in_feed g(f);
    return f->flight.empty() ? JRC_OK : fail(f, JRC_ERR_INVALID_ARG, "set_background with batches in flight");
}
int jrc_chain_feed_acquire(jrc_chain_feed* f, jrc_cf32** h)
{
    in_feed g(f);
    if ((int)f->flight.size() == f->n_slots) return fail(f, JRC_ERR_INVALID_ARG, "every slot is in flight");
    *h = f->stage[f->next_slot].data();
    return JRC_OK;
}
static int submit(jrc_chain_feed* f, const jrc_cf32* h, int n, bool rx_only)
{
    in_feed g(f);
    if (h) return fail(f, JRC_ERR_UNSUPPORTED, "the double takes in-place submissions only");
    if (n < 1 || n > f->fps) return fail(f, JRC_ERR_INVALID_ARG, "n_frames");
    if ((int)f->flight.size() == f->n_slots) return fail(f, JRC_ERR_INVALID_ARG, "every slot is in flight");
    if (rx_only && !f->have_tx) return fail(f, JRC_ERR_INVALID_ARG, "submit_rx before set_tx");
    f->flight.push_back(batch{f->next_slot, n, clk::now() + std::chrono::microseconds(f->latency_us), rx_only});
    f->next_slot = (f->next_slot + 1) % f->n_slots;
    return JRC_OK;
}
int jrc_chain_feed_submit(jrc_chain_feed* f, const jrc_cf32* h, int n) { return submit(f, h, n, false); }
int jrc_chain_feed_submit_rx(jrc_chain_feed* f, const jrc_cf32* h, int n) { return submit(f, h, n, true); }
int jrc_chain_feed_set_tx(jrc_chain_feed* f, const jrc_cf32* h_tx)
{
    in_feed g(f);
    if (!f->flight.empty()) return fail(f, JRC_ERR_INVALID_ARG, "set_tx with batches in flight");
    f->have_tx = h_tx != nullptr;
    if (h_tx) memcpy(f->tx_resident.data(), h_tx, sizeof(jrc_cf32) * f->tx_resident.size());
    return JRC_OK;
}
int jrc_chain_feed_collect(jrc_chain_feed* f, jrc_ra_result* results, jrc_cf32*, int* n_frames)
{
    in_feed g(f);
    if (f->overlapped) return fail(f, JRC_ERR_INVALID_ARG, "two threads were inside the feed at once");
    if (f->flight.empty()) { if (n_frames) *n_frames = 0; return 0; }
    if (f->fail_collect_at >= 0 && ++f->collects >= f->fail_collect_at) {
        if (f->collects > f->fail_collect_at) feed_double_collect_calls_after_failure++;
        return fail(f, JRC_ERR_HIP, "feed double: collect fails (FEED_DOUBLE_FAIL_COLLECT_AT)");
    }
    const batch b = f->flight.front();
    std::this_thread::sleep_until(b.done);
    f->flight.pop_front();
    const size_t tx = f->port * f->cfg.N_tx;
    for (int i = 0; i < b.n; i++) {
        const jrc_cf32* fr = f->stage[b.slot].data() + (size_t)i * f->frame;
        memset(&results[i], 0, sizeof(results[i]));
        results[i].published = 1;
        results[i].range_val = checksum(fr + tx, f->frame - tx);                                  // the receive ports as staged
        results[i].angle_val = checksum(b.rx_only ? f->tx_resident.data() : fr, tx);              // the TX ports the device would have used
        results[i].peak_power = b.rx_only ? 1.f : 0.f;
        results[i].snr_est = 30.f;
    }
    if (n_frames) *n_frames = b.n;
    feed_double_frames_collected += b.n;
    return b.n;
}

}  // extern "C"
