// The radar_chain block of host/jrc_blocks.cc driven the way a GNU Radio scheduler drives it, with the feed double (feed_double.cc) in the
// place of libjrc_hip.so, so that the block's own threading can run under ThreadSanitizer / AddressSanitizer + UBSan on the CPU:
//   thread 1 (the "scheduler"): one general_work per turn, with idle gaps longer than the age bound so that the block's flusher thread publishes;
//   thread 2 (an observer, like a GUI probe or a control-port getter): frames_done / pending_batches / rx_only_batches and the published
//            messages while the other two run;
//   thread 3: the block's flusher.
// Checks (exit code 1 + a line on stderr when one fails): every frame published once, in frame order; each message carries the checksum of
// that frame's receive ports, and of the TX rows that frame had (whether it went up whole or receive-only against the resident rows);
// receive-only batches happened; nothing pending after stop(); the feed never saw two threads at once.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "jrc_blocks.h"

using namespace gr::mimo_ofdm_jrc;

static float checksum(const gr_complex* p, size_t n, size_t i0 = 0)
{
    double s = 0;
    for (size_t i = 0; i < n; i++) s += (double)p[i].real() * (double)(((i + i0) % 7) + 1) - (double)p[i].imag() * (double)(((i + i0) % 5) + 1);
    return (float)s;
}
#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } } while (0)

extern std::atomic<long> feed_double_frames_collected, feed_double_collect_calls_after_failure;

// (a) a block destroyed without stop() / flush() still publishes what is in flight — the destructor collects; (b) a collect that fails
// in the flusher thread is reported once, remembered, rethrown by the scheduler's next general_work / flush, and not retried for ever
static int scenario(const std::string& which)
{
    // (destructor: six packets = a full batch and a receive-only one, whose first use of the resident rows collects the first inside the turn;
    //  flusher_error: four packets = one batch, so that the first collect call is the flusher's)
    const int N = 64, T = 4, R = 2, S = 4, NPRE = 5, n_items = NPRE + S, Ir = 8, Ia = 16, fpb = 4, slots = 3, nf = which == "destructor" ? 6 : 4;
    std::vector<float> rb((size_t)N * Ir, 0.f), ab((size_t)T * R * Ia, 0.f);
    auto blk = radar_chain::make(N, T, R, S, NPRE, Ir, Ia, false, rb, ab, 2.4f, 28.96f, -100.f, 0.f, "", false, fpb, slots);
    std::vector<std::vector<gr_complex>> ports(T + R, std::vector<gr_complex>((size_t)2 * nf * n_items * N, gr_complex(1.f, -1.f)));
    auto turn = [&](int k) {
        for (int i = 0; i < nf; i++) {
            jrc_rt::tag_t t;
            t.key = pmt::mp("packet_len"); t.value = pmt::from_long(n_items);
            t.offset = (uint64_t)(k * nf + i) * n_items;
            blk->t_in_tags[0].push_back(t);
            blk->t_in_tags[T].push_back(t);
        }
        gr_vector_int nin(T + R, nf * n_items);
        gr_vector_const_void_star in;
        for (int p = 0; p < T + R; p++) in.push_back(ports[p].data() + (size_t)k * nf * n_items * N);
        gr_vector_void_star out;
        return blk->t_run(0, nin, in, out);
    };
    if (which == "destructor") {
        CHECK(turn(0) == 0, "general_work");
        const int left = blk->pending_batches();
        CHECK(left > 0, "nothing left in flight to destroy the block over");
        const long before = feed_double_frames_collected.load();
        CHECK(before < nf, "every frame was collected before the destructor ran");
        blk.reset();                                                    // no stop(), no flush()
        CHECK(feed_double_frames_collected.load() == nf, "the destructor collected %ld of %d frames", feed_double_frames_collected.load(), nf);
        printf("ok: destructor collected %ld frames that were in flight\n", feed_double_frames_collected.load() - before);
        return 0;
    }
    // "flusher_error": FEED_DOUBLE_FAIL_COLLECT_AT makes a collect fail while the scheduler idles, i.e. in the flusher thread
    CHECK(turn(0) == 0, "general_work");
    std::this_thread::sleep_for(std::chrono::milliseconds(60));         // many flusher periods: it must fail once and stop, not retry every half bound
    const long retries = feed_double_collect_calls_after_failure.load();
    CHECK(retries <= 1, "the flusher kept calling a failing collect (%ld calls after the failure)", retries);
    bool thrown = false;
    try { turn(1); } catch (const std::runtime_error& e) { thrown = std::string(e.what()).find("flusher") != std::string::npos; }
    CHECK(thrown, "general_work after a failed flusher collect did not rethrow it");
    thrown = false;
    try { blk->flush(); } catch (const std::runtime_error&) { thrown = true; }
    CHECK(thrown, "flush() after a failed flusher collect did not rethrow it");
    blk.reset();                                                        // the destructor neither throws nor retries
    printf("ok: flusher error was sticky, %ld collect calls after the failure\n", feed_double_collect_calls_after_failure.load());
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 2) return scenario(argv[2]);
    const int N = 64, T = 4, R = 2, S = 4, NPRE = 5, n_items = NPRE + S, Ir = 8, Ia = 16;
    const int F = argc > 1 ? atoi(argv[1]) : 240, per_turn = 6, fpb = 4, slots = 3;
    std::vector<float> rb((size_t)N * Ir, 0.f), ab((size_t)T * R * Ia, 0.f);
    auto blk = radar_chain::make(N, T, R, S, NPRE, Ir, Ia, false, rb, ab, 2.4f, 28.96f, -100.f, 0.f, "", false, fpb, slots);

    // F packets: the TX rows behind the preamble repeat (the MIMO-LTFs) except in packets 17 and 18 and from packet 150 on (a new set)
    std::mt19937 rng(5);
    std::normal_distribution<float> nd;
    const size_t item = N, pkt = (size_t)n_items * item;
    std::vector<std::vector<gr_complex>> ports(T + R, std::vector<gr_complex>((size_t)F * pkt));
    for (auto& p : ports) for (auto& v : p) v = gr_complex(nd(rng), nd(rng));
    for (int f = 1; f < F; f++) {
        const int like = (f == 17 || f == 18) ? f : (f >= 150 ? 150 : 0);
        if (like == f) continue;
        for (int t = 0; t < T; t++)
            std::copy(ports[t].begin() + like * pkt + NPRE * item, ports[t].begin() + (like + 1) * pkt, ports[t].begin() + f * pkt + NPRE * item);
    }
    std::vector<float> want_rx(F), want_tx(F);
    for (int f = 0; f < F; f++) {
        std::vector<gr_complex> tx, rx;
        for (int t = 0; t < T; t++) tx.insert(tx.end(), ports[t].begin() + f * pkt + NPRE * item, ports[t].begin() + (f + 1) * pkt);
        for (int r = 0; r < R; r++) rx.insert(rx.end(), ports[T + r].begin() + f * pkt + NPRE * item, ports[T + r].begin() + (f + 1) * pkt);
        want_tx[f] = checksum(tx.data(), tx.size());
        want_rx[f] = checksum(rx.data(), rx.size());
    }

    std::atomic<bool> done{false};
    std::atomic<long> observations{0};
    std::thread observer([&] {
        int last_done = 0;
        while (!done.load()) {
            const int d = blk->frames_done(), p = blk->pending_batches();
            const long ro = blk->rx_only_batches();
            CHECK(d >= last_done && d <= F, "frames_done went from %d to %d", last_done, d);
            CHECK(p >= 0 && p <= slots && ro >= 0, "pending %d rx_only %ld", p, ro);
            last_done = d;
            size_t n_pub;
            { std::lock_guard<std::mutex> g(blk->t_pub_lock); n_pub = blk->t_published.size(); }
            CHECK((int)n_pub <= F, "more messages than frames");
            observations++;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    });

    const char* age_env = getenv("JRC_RADAR_CHAIN_MAX_AGE_US");
    const bool drains = age_env && atol(age_env) == 0;                 // bound 0: every turn publishes all it submitted
    std::atomic<int> turns_left_in_flight{0};
    std::thread scheduler([&] {
        int turn = 0;
        for (int f0 = 0; f0 < F; f0 += per_turn, turn++) {
            const int nf = std::min(per_turn, F - f0);
            for (int k = 0; k < nf; k++) {
                jrc_rt::tag_t t;
                t.key = pmt::mp("packet_len"); t.value = pmt::from_long(n_items);
                t.offset = (uint64_t)(f0 + k) * n_items;
                blk->t_in_tags[0].push_back(t);
                blk->t_in_tags[T].push_back(t);
            }
            gr_vector_int nin(T + R, nf * n_items);
            gr_vector_const_void_star in;
            for (int p = 0; p < T + R; p++) in.push_back(ports[p].data() + (size_t)f0 * pkt);
            gr_vector_void_star out;
            CHECK(blk->t_run(0, nin, in, out) == 0, "general_work");
            for (int p = 0; p < T + R; p++) CHECK(blk->t_consumed[p] == nf * n_items, "turn %d consumed %d on port %d", turn, blk->t_consumed[p], p);
            // the double's batches take 300 us: the turn's last one cannot be back yet — it stays in flight into the next turn unless the bound is 0
            const int left = blk->pending_batches();
            if (drains) CHECK(left == 0, "turn %d: bound 0 but %d batches in flight after general_work", turn, left);
            else if (left > 0) turns_left_in_flight++;
            if (turn % 5 == 4) std::this_thread::sleep_for(std::chrono::milliseconds(3));      // idle scheduler: the flusher publishes what is overdue
            if (turn == 20) blk->flush();                                                       // a setter's path, between two turns
        }
    });
    scheduler.join();
    std::this_thread::sleep_for(std::chrono::milliseconds(4));          // > age bound: the flusher alone empties the pipeline
    CHECK(blk->pending_batches() == 0, "the flusher left %d batches in flight past the age bound", blk->pending_batches());
    blk->stop();
    done.store(true);
    observer.join();

    CHECK(blk->frames_done() == F && blk->pending_batches() == 0, "frames_done %d pending %d", blk->frames_done(), blk->pending_batches());
    CHECK((int)blk->t_published.size() == F, "%zu messages for %d frames", blk->t_published.size(), F);
    int rx_only_frames = 0;
    for (int f = 0; f < F; f++) {
        const auto& m = blk->t_published[f].second;                     // ((range [v]) (angle [v]) (power [v]) (snr [v]))
        const float range = m->list[0]->list[1]->f[0], angle = m->list[1]->list[1]->f[0], power = m->list[2]->list[1]->f[0];
        CHECK(range == want_rx[f], "frame %d: receive ports %g, staged %g (out of order or overwritten while in flight)", f, want_rx[f], range);
        CHECK(angle == want_tx[f], "frame %d: TX rows %g, the feed used %g (%s submission)", f, want_tx[f], angle, power ? "receive-only" : "full");
        rx_only_frames += power != 0.f;
    }
    CHECK(blk->rx_only_batches() > 0 && rx_only_frames > F / 2, "rx-only batches %ld, frames %d of %d", blk->rx_only_batches(), rx_only_frames, F);
    CHECK(observations.load() > 10, "the observer ran %ld times", observations.load());
    CHECK(drains || turns_left_in_flight.load() > 0, "no turn left a batch in flight: the block drained every call although the age bound is not 0");
    printf("ok: %d frames, %ld receive-only batches (%d frames), %ld observations\n", F, blk->rx_only_batches(), rx_only_frames, observations.load());
    blk.reset();                                                        // joins the flusher, destroys the feed
    return 0;
}
