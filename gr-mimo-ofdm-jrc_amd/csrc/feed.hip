// feed.hip — host-fed pipeline over the radar chain (jrc_chain_feed_* of include/jrc.h).
//
// A GNU Radio flowgraph hands work() HOST buffers: the reference's radar branch (mimo_ofdm_radar -> fft -> transpose -> fft ->
// range_angle_estimator, examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2189-2197) consumes T+R tagged streams from
// pageable ring buffers and publishes one small PDU per frame (lib/range_angle_estimator_impl.cc:199-235).  The per-block
// host entry points copy in, run, copy out and wait; a stream of frames is better served by keeping `n_slots` batches in
// flight: each slot owns pinned staging, device buffers, its own stream and its own jrc_chain (partial-maximum scratch is per
// chain), so the H2D copy of batch k+1, the kernels of batch k and the D2H of batch k-1's results overlap on the copy
// engines and the CUs.  With JRC_FEED_GRAPH a full slot is replayed as one hipGraph (copy-in, A1, fused A2-A4, finalize,
// copy-out: one submit instead of six).  Measured on MI355X / ROCm 7.2 (profiles/r01_i_feed_probe_configB.jsonl): the replay
// shortens the idle-pipeline latency of a batch by ~16 us (1 frame: 116 us against 133 us) but hipGraphLaunch costs more host
// time than the six direct submits, so a saturated feed of small batches is slower with it (1-frame batches: 15.9 k against
// 18.6 k frames/s); from 4 frames per batch up the direct path runs at the PCIe limit (56 GB/s = 50 k config-B frames/s).
// Hence a flag, off by default: latency-critical callers set it, throughput-bound ones do not.
//
// Order is preserved: slots are submitted and collected round-robin, results come back in submission order.
//
// Several GPUs from one host process (jrc_chain_feed_create_multi): the slots are dealt round-robin over the devices — batch k runs on
// device k mod n — each device with its own context, streams and buffers, and ONE HOST THREAD PER GPU: jrc_chain_feed_submit_many hands
// one batch to every device's thread, which stages it (pageable -> pinned copy, the host-side cost that does not overlap otherwise),
// enqueues it and returns the caller's buffers; submission order, hence result order, is that of the batches.  Frames are independent
// (SURVEY.md §8(e)), so there is no exchange between the devices.
#include "jrc_internal.h"
#include "radar_kernels.h"

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>


struct feed_slot {
    jrc_ctx* ctx = nullptr;              // the context (GPU) this slot lives on
    jrc_chain* chain = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float2* h_frames = nullptr;          // pinned [fps] frames
    float2* d_frames = nullptr;
    float2* d_chanest = nullptr;
    float2* d_map = nullptr;
    jrc_ra_result* d_results = nullptr;
    jrc_ra_result* h_results = nullptr;  // pinned
    float2* h_maps = nullptr;            // pinned [maps_per_slot] maps (optional)
    hipGraphExec_t graph = nullptr;      // full-slot replay
    hipGraphExec_t graph_rx = nullptr;   // full-slot replay of the TX-resident sequence (receive ports up, kernels, results down)
    bool graph_failed = false;
    bool warm = false;                   // one direct pass done (tables cached, nothing left to allocate inside a capture)
    int n_frames = 0;                    // frames of the batch in flight
    int state = 0;                       // 0 free, 1 acquired, 2 in flight
    float2* d_tx = nullptr;              // resident TX reference ports [T][n_items][N] (jrc_chain_feed_set_tx)
    bool tx_in_frames = false;           // every frame of d_frames holds the resident TX ports (nothing overwrote them since the broadcast)
};

struct jrc_chain_feed {
    jrc_ctx* ctx = nullptr;
    jrc_chain_cfg cfg;
    int n_slots = 0, fps = 0, maps_per_slot = 0, flags = 0;
    size_t frame_elems = 0, chanest_elems = 0, map_elems = 0;
    std::vector<feed_slot> slots;
    int head = 0;       // next slot to acquire / submit
    int tail = 0;       // oldest slot in flight
    int in_flight = 0;
    long graph_replays = 0, direct_submits = 0;
    // multi-device feeds own their contexts and one worker thread per device
    std::vector<jrc_ctx*> owned_ctx;
    struct worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<int()> job;        // at most one job at a time per device
        bool has_job = false, done = false, stop = false;
        int status = JRC_OK;
    };
    std::vector<worker*> workers;
    int n_devices = 1;
    bool tx_resident = false;            // jrc_chain_feed_set_tx: the slots hold a copy of the TX reference ports
    size_t tx_elems = 0;                 // T * n_items * N
};

// a failure inside one slot's context (a multi-device feed has one per GPU) is repeated in the feed's own, which is where
// jrc_chain_feed_last_error looks: the caller of a feed never holds the slot contexts
static int feed_relay(jrc_chain_feed* fd, jrc_ctx* where, int st)
{
    if (st < 0 && where && where != fd->ctx) jrc_fail(fd->ctx, st, "%s", jrc_last_error(where));
    return st;
}
#define FEED_TRY(fd, where, expr) do { const int st__ = feed_relay((fd), (where), (expr)); if (st__ < 0) return st__; } while (0)

static void feed_worker_main(jrc_chain_feed::worker* w)
{
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->stop; });
        if (w->stop) return;
        std::function<int()> job = std::move(w->job);
        lk.unlock();
        const int st = job();
        lk.lock();
        w->status = st; w->has_job = false; w->done = true;
        w->cv.notify_all();
    }
}

static void feed_free_slot(feed_slot& s)
{
    if (s.graph) (void)hipGraphExecDestroy(s.graph);
    if (s.graph_rx) (void)hipGraphExecDestroy(s.graph_rx);
    if (s.chain) jrc_chain_destroy(s.chain);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.h_frames) (void)hipHostFree(s.h_frames);
    if (s.h_results) (void)hipHostFree(s.h_results);
    if (s.h_maps) (void)hipHostFree(s.h_maps);
    if (s.d_frames) (void)hipFree(s.d_frames);
    if (s.d_chanest) (void)hipFree(s.d_chanest);
    if (s.d_map) (void)hipFree(s.d_map);
    if (s.d_results) (void)hipFree(s.d_results);
    if (s.d_tx) (void)hipFree(s.d_tx);
    s = feed_slot();
}

extern "C" void jrc_chain_feed_destroy(jrc_chain_feed* fd)
{
    if (!fd) return;
    for (auto* w : fd->workers) {
        { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    for (auto& s : fd->slots) {
        if (s.ctx) (void)hipSetDevice(s.ctx->device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    for (auto& s : fd->slots) { if (s.ctx) (void)hipSetDevice(s.ctx->device); feed_free_slot(s); }
    for (auto* c : fd->owned_ctx) jrc_destroy(c);
    delete fd;
}

static int feed_create(const std::vector<jrc_ctx*>& ctxs, bool own, const jrc_chain_cfg* cfg, const float* range_bins, const float* angle_bins,
                       int n_slots, int frames_per_slot, int maps_per_slot, int flags, jrc_chain_feed** out)
{
    jrc_ctx* ctx = ctxs[0];
    jrc_chain_feed* fd = new jrc_chain_feed();
    fd->ctx = ctx; fd->cfg = *cfg; fd->n_slots = n_slots; fd->fps = frames_per_slot; fd->maps_per_slot = maps_per_slot; fd->flags = flags;
    fd->n_devices = (int)ctxs.size();
    if (own) fd->owned_ctx = ctxs;
    fd->slots.resize((size_t)n_slots);
    int st = JRC_OK;
    for (int i = 0; i < n_slots && st == JRC_OK; i++) {
        feed_slot& s = fd->slots[(size_t)i];
        s.ctx = ctxs[(size_t)i % ctxs.size()];                 // batch k -> device k mod n
        hipError_t e = hipSetDevice(s.ctx->device);
        if (e != hipSuccess) { st = jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_feed_create: %s", hipGetErrorString(e)); break; }
        st = jrc_chain_create(s.ctx, cfg, range_bins, angle_bins, frames_per_slot, &s.chain);
        if (st != JRC_OK) { if (s.ctx != ctx) jrc_fail(ctx, st, "%s", jrc_last_error(s.ctx)); break; }
        if (i == 0) {
            fd->tx_elems = (size_t)cfg->N_tx * cfg->n_items * cfg->fft_len;
            fd->frame_elems = jrc_chain_frame_bytes(s.chain) / sizeof(float2);
            fd->chanest_elems = jrc_chain_chanest_bytes(s.chain) / sizeof(float2);
            fd->map_elems = jrc_chain_map_bytes(s.chain) / sizeof(float2);
        }
        const size_t F = (size_t)frames_per_slot;
        e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_frames, sizeof(float2) * F * fd->frame_elems, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_results, sizeof(jrc_ra_result) * F, hipHostMallocDefault);
        if (e == hipSuccess && maps_per_slot)
            e = hipHostMalloc((void**)&s.h_maps, sizeof(float2) * (size_t)maps_per_slot * fd->map_elems, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_frames, sizeof(float2) * F * fd->frame_elems);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_chanest, sizeof(float2) * F * fd->chanest_elems);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_map, sizeof(float2) * F * fd->map_elems);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_results, sizeof(jrc_ra_result) * F);
        if (e != hipSuccess) st = jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_feed_create: %s", hipGetErrorString(e));
    }
    if (st == JRC_OK && fd->n_devices > 1)
        for (int d = 0; d < fd->n_devices; d++) {
            auto* w = new jrc_chain_feed::worker();
            w->th = std::thread(feed_worker_main, w);
            fd->workers.push_back(w);
        }
    if (st != JRC_OK) {
        std::string msg = ctx->last_error;
        if (own) fd->owned_ctx.clear();                         // the caller of feed_create destroys them after reading the message
        jrc_chain_feed_destroy(fd);
        ctx->last_error = msg;
        return st;
    }
    *out = fd;
    return JRC_OK;
}

extern "C" int jrc_chain_feed_create(jrc_ctx* ctx, const jrc_chain_cfg* cfg, const float* range_bins, const float* angle_bins,
                                     int n_slots, int frames_per_slot, int maps_per_slot, int flags, jrc_chain_feed** out)
{
    if (!ctx || !cfg || !range_bins || !angle_bins || !out) return JRC_ERR_INVALID_ARG;
    if (n_slots < 1 || n_slots > 16 || frames_per_slot < 1 || maps_per_slot < 0 || maps_per_slot > frames_per_slot)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_create: need 1..16 slots, >= 1 frame per slot, 0 <= maps_per_slot <= frames_per_slot");
    return feed_create(std::vector<jrc_ctx*>{ctx}, false, cfg, range_bins, angle_bins, n_slots, frames_per_slot, maps_per_slot, flags, out);
}

extern "C" int jrc_chain_feed_create_multi(const int* devices, int n_devices, const jrc_chain_cfg* cfg, const float* range_bins,
                                           const float* angle_bins, int slots_per_device, int frames_per_slot, int maps_per_slot, int flags,
                                           jrc_chain_feed** out)
{
    if (!devices || n_devices < 1 || n_devices > 64 || !cfg || !range_bins || !angle_bins || !out) return JRC_ERR_INVALID_ARG;
    if (slots_per_device < 1 || slots_per_device * n_devices > 128 || frames_per_slot < 1 || maps_per_slot < 0 || maps_per_slot > frames_per_slot)
        return JRC_ERR_INVALID_ARG;
    std::vector<jrc_ctx*> ctxs;
    int st = JRC_OK;
    for (int d = 0; d < n_devices && st == JRC_OK; d++) {
        jrc_ctx* c = nullptr;
        st = jrc_create(devices[d], &c);
        if (st == JRC_OK) ctxs.push_back(c);
    }
    if (st == JRC_OK) st = feed_create(ctxs, true, cfg, range_bins, angle_bins, slots_per_device * n_devices, frames_per_slot, maps_per_slot, flags, out);
    if (st != JRC_OK) for (auto* c : ctxs) jrc_destroy(c);
    return st;
}

extern "C" int jrc_chain_feed_n_devices(const jrc_chain_feed* fd) { return fd ? fd->n_devices : JRC_ERR_INVALID_ARG; }
extern "C" const char* jrc_chain_feed_last_error(const jrc_chain_feed* fd) { return fd ? jrc_last_error(fd->ctx) : ""; }

extern "C" size_t jrc_chain_feed_frame_bytes(const jrc_chain_feed* fd) { return fd ? fd->frame_elems * sizeof(float2) : 0; }
extern "C" size_t jrc_chain_feed_map_bytes(const jrc_chain_feed* fd) { return fd ? fd->map_elems * sizeof(float2) : 0; }
extern "C" int jrc_chain_feed_pending(const jrc_chain_feed* fd) { return fd ? fd->in_flight : JRC_ERR_INVALID_ARG; }
extern "C" int jrc_chain_feed_stats(const jrc_chain_feed* fd, long* graph_replays, long* direct_submits)
{
    if (!fd) return JRC_ERR_INVALID_ARG;
    if (graph_replays) *graph_replays = fd->graph_replays;
    if (direct_submits) *direct_submits = fd->direct_submits;
    return JRC_OK;
}

extern "C" int jrc_chain_feed_set_background(jrc_chain_feed* fd, int background_removal, int background_recording, int record_len)
{
    if (!fd) return JRC_ERR_INVALID_ARG;
    if (fd->in_flight) return jrc_fail(fd->ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_set_background: collect the batches in flight first");
    if (fd->n_devices > 1 && (background_removal || background_recording))
        return jrc_fail(fd->ctx, JRC_ERR_UNSUPPORTED, "background removal: the frames of one radar stream stay in order on one GPU (one feed per stream; "
                        "shard streams, or blocks primed with jrc_chain_prime_background_dev, over GPUs)");
    FEED_TRY(fd, fd->slots[0].ctx, jrc_chain_set_background(fd->slots[0].chain, background_removal, background_recording, record_len));
    if (!background_removal && !background_recording && jrc_chain_background_size(fd->slots[0].chain) == 0) return JRC_OK;
    for (int i = 1; i < fd->n_slots; i++) FEED_TRY(fd, fd->slots[(size_t)i].ctx, jrc_chain_share_background(fd->slots[(size_t)i].chain, fd->slots[0].chain));
    for (auto& s : fd->slots) {          // the history buffers alternate from batch to batch: no fixed graph
        if (s.graph) { (void)hipGraphExecDestroy(s.graph); s.graph = nullptr; }
        if (s.graph_rx) { (void)hipGraphExecDestroy(s.graph_rx); s.graph_rx = nullptr; }
        s.graph_failed = true;
    }
    return JRC_OK;
}

extern "C" int jrc_chain_feed_set_write_map(jrc_chain_feed* fd, int write_map)
{
    if (!fd) return JRC_ERR_INVALID_ARG;
    if (fd->in_flight) return jrc_fail(fd->ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_set_write_map: collect the batches in flight first");
    if (!write_map && fd->maps_per_slot) return jrc_fail(fd->ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_set_write_map: the feed copies maps back (maps_per_slot > 0)");
    for (auto& s : fd->slots) {
        FEED_TRY(fd, s.ctx, jrc_chain_set_write_map(s.chain, write_map));
        if (s.graph) { (void)hipGraphExecDestroy(s.graph); s.graph = nullptr; }      // recorded with the other kernels: record again
        if (s.graph_rx) { (void)hipGraphExecDestroy(s.graph_rx); s.graph_rx = nullptr; }
    }
    return JRC_OK;
}

extern "C" int jrc_chain_feed_acquire(jrc_chain_feed* fd, jrc_cf32** h_frames)
{
    if (!fd || !h_frames) return JRC_ERR_INVALID_ARG;
    feed_slot& s = fd->slots[(size_t)fd->head];
    if (s.state == 2)
        return jrc_fail(fd->ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_acquire: all %d slots are in flight, collect one first", fd->n_slots);
    s.state = 1;
    *h_frames = (jrc_cf32*)s.h_frames;
    return JRC_OK;
}

// the resident TX reference ports into the TX part of every frame of a slot's device buffer
__global__ __launch_bounds__(256) void feed_broadcast_tx_kernel(const float4* __restrict__ tx, float4* __restrict__ frames, size_t tx4, size_t frame4)
{
    float4* dst = frames + (size_t)blockIdx.y * frame4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tx4; i += (size_t)gridDim.x * blockDim.x) dst[i] = tx[i];
}

// the whole slot on its stream: copy in, A1 -> A5, results (and the first maps) out.  rx_only: the T reference ports of every frame are the
// resident copy (broadcast into the slot's frames once, again after a full submission overwrote them); only the R receive ports cross PCIe
static int feed_enqueue(jrc_chain_feed* fd, feed_slot& s, int n, bool rx_only = false)
{
    jrc_ctx* ctx = s.ctx;
    if (rx_only) {
        if (!s.tx_in_frames) {
            if ((fd->tx_elems % 2) || (fd->frame_elems % 2)) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "jrc_chain_feed_submit_rx: odd port size");
            const unsigned gx = (unsigned)std::min<size_t>((fd->tx_elems / 2 + 255) / 256, 64);
            hipLaunchKernelGGL(feed_broadcast_tx_kernel, dim3(gx, (unsigned)fd->fps), dim3(256), 0, s.stream, (const float4*)s.d_tx, (float4*)s.d_frames,
                               fd->tx_elems / 2, fd->frame_elems / 2);
            JRC_HIP(ctx, hipGetLastError());
            s.tx_in_frames = true;
        }
        const size_t pitch = sizeof(float2) * fd->frame_elems, width = sizeof(float2) * (fd->frame_elems - fd->tx_elems);
        JRC_HIP(ctx, hipMemcpy2DAsync(s.d_frames + fd->tx_elems, pitch, s.h_frames + fd->tx_elems, pitch, width, (size_t)n, hipMemcpyHostToDevice, s.stream));
    } else {
        JRC_HIP(ctx, hipMemcpyAsync(s.d_frames, s.h_frames, sizeof(float2) * (size_t)n * fd->frame_elems, hipMemcpyHostToDevice, s.stream));
        s.tx_in_frames = false;
    }
    JRC_TRY(jrc_chain_run_dev(s.chain, n, (const jrc_cf32*)s.d_frames, (jrc_cf32*)s.d_chanest, (jrc_cf32*)s.d_map, s.d_results, (void*)s.stream));
    JRC_HIP(ctx, hipMemcpyAsync(s.h_results, s.d_results, sizeof(jrc_ra_result) * (size_t)n, hipMemcpyDeviceToHost, s.stream));
    const int nm = n < fd->maps_per_slot ? n : fd->maps_per_slot;
    if (nm > 0)
        JRC_HIP(ctx, hipMemcpyAsync(s.h_maps, s.d_map, sizeof(float2) * (size_t)nm * fd->map_elems, hipMemcpyDeviceToHost, s.stream));
    return JRC_OK;
}

// stage (when the frames come from pageable memory) and enqueue one slot; runs on the calling thread or on the slot's device thread
static int feed_launch_slot(jrc_chain_feed* fd, feed_slot& s, const jrc_cf32* h_frames, int n_frames, bool threaded_copy, bool* replayed, bool rx_only = false)
{
    jrc_ctx* ctx = s.ctx;
    if (h_frames && (const float2*)h_frames != s.h_frames) {    // pageable source (a GNU Radio buffer): stage it
        if (rx_only) {                                          // only the receive ports of each frame
            for (int f = 0; f < n_frames; f++)
                memcpy(s.h_frames + (size_t)f * fd->frame_elems + fd->tx_elems, (const float2*)h_frames + (size_t)f * fd->frame_elems + fd->tx_elems,
                       sizeof(float2) * (fd->frame_elems - fd->tx_elems));
        } else {
            const size_t bytes = sizeof(float2) * (size_t)n_frames * fd->frame_elems;
            if (threaded_copy) jrc_host_copy(s.h_frames, h_frames, bytes);     // one feeder thread: split large copies
            else memcpy(s.h_frames, h_frames, bytes);                           // a thread per device is already copying
        }
    }
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    // A full slot is replayed from a recorded graph once the slot is warm (pointers and sizes of a full slot never change): one graph for the
    // full-copy sequence, one for the TX-resident one.  The broadcast of the resident rows into the slot's frames is not part of the recorded
    // sequence: a slot whose frames do not hold them yet (first rx-only batch, or a full submission overwrote them) is submitted directly once.
    hipGraphExec_t& gx = rx_only ? s.graph_rx : s.graph;
    const bool want_graph = (fd->flags & JRC_FEED_GRAPH) && n_frames == fd->fps && !s.graph_failed && s.warm && (!rx_only || s.tx_in_frames);
    if (want_graph && !gx) {
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamBeginCapture(s.stream, hipStreamCaptureModeRelaxed);
        int st = JRC_OK;
        if (e == hipSuccess) {
            st = feed_enqueue(fd, s, n_frames, rx_only);
            e = hipStreamEndCapture(s.stream, &g);
        }
        if (e == hipSuccess && st == JRC_OK) e = hipGraphInstantiate(&gx, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (e != hipSuccess || st != JRC_OK) {       // fall back to direct submission on this slot, loudly in last_error
            (void)hipGetLastError();
            gx = nullptr; s.graph_failed = true;
            jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_feed: graph capture failed (%s), submitting directly", hipGetErrorString(e));
        }
    }
    *replayed = false;
    if (want_graph && gx) {
        JRC_HIP(ctx, hipGraphLaunch(gx, s.stream));
        if (!rx_only) s.tx_in_frames = false;
        *replayed = true;
    } else {
        JRC_TRY(feed_enqueue(fd, s, n_frames, rx_only));
        s.warm = true;
    }
    JRC_HIP(ctx, hipEventRecord(s.done, s.stream));
    return JRC_OK;
}

static int feed_check_submit(jrc_chain_feed* fd, const feed_slot& s, const jrc_cf32* h_frames, int n_frames)
{
    jrc_ctx* ctx = fd->ctx;
    if (n_frames < 1 || n_frames > fd->fps)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_submit: n_frames %d outside [1, %d]", n_frames, fd->fps);
    if (s.state == 2)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_submit: all %d slots are in flight, collect one first", fd->n_slots);
    if (!h_frames && s.state != 1)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_submit: no host frames given and no acquired buffer to take them from");
    return JRC_OK;
}

static void feed_mark_submitted(jrc_chain_feed* fd, feed_slot& s, int n_frames, bool replayed)
{
    if (replayed) fd->graph_replays++; else fd->direct_submits++;
    s.n_frames = n_frames;
    s.state = 2;
    fd->head = (fd->head + 1) % fd->n_slots;
    fd->in_flight++;
}

extern "C" int jrc_chain_feed_submit(jrc_chain_feed* fd, const jrc_cf32* h_frames, int n_frames)
{
    JRC_TRACE("jrc_chain_feed_submit");
    if (!fd) return JRC_ERR_INVALID_ARG;
    feed_slot& s = fd->slots[(size_t)fd->head];
    JRC_TRY(feed_check_submit(fd, s, h_frames, n_frames));
    bool replayed = false;
    const int st = feed_launch_slot(fd, s, h_frames, n_frames, true, &replayed);
    if (st != JRC_OK) { if (s.ctx != fd->ctx) jrc_fail(fd->ctx, st, "%s", jrc_last_error(s.ctx)); return st; }
    feed_mark_submitted(fd, s, n_frames, replayed);
    return JRC_OK;
}

// TX-resident submission (round 4).  In the reference's flowgraph the radar block is pointed at the MIMO-LTF symbols (N_pre = 5, N_sym = N_tx,
// examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:1292-1295): the T reference ports it correlates with are the same rows, packet after
// packet.  jrc_chain_feed_set_tx hands the feed those rows once; jrc_chain_feed_submit_rx then uploads only the R receive ports of each frame
// (half of a frame's bytes at T = R) — the caller's promise is that the frames' own TX ports equal the resident rows, which a block checks with
// a memcmp per frame (radar_chain, host/jrc_blocks.cc).  Full submissions stay possible at any time; they overwrite the slot's TX ports, which
// the next receive-only batch on that slot restores from the resident copy first.
extern "C" int jrc_chain_feed_set_tx(jrc_chain_feed* fd, const jrc_cf32* h_tx)
{
    if (!fd) return JRC_ERR_INVALID_ARG;
    if (fd->in_flight) return jrc_fail(fd->ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_set_tx: collect the batches in flight first");
    fd->tx_resident = h_tx != nullptr;
    for (auto& s : fd->slots) {
        s.tx_in_frames = false;
        if (!h_tx) continue;
        jrc_ctx* ctx = s.ctx;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess && !s.d_tx) e = hipMalloc((void**)&s.d_tx, sizeof(float2) * fd->tx_elems);
        if (e == hipSuccess) e = hipMemcpy(s.d_tx, h_tx, sizeof(float2) * fd->tx_elems, hipMemcpyHostToDevice);
        if (e != hipSuccess) { fd->tx_resident = false; return feed_relay(fd, ctx, jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_feed_set_tx: %s", hipGetErrorString(e))); }
    }
    return JRC_OK;
}

extern "C" int jrc_chain_feed_submit_rx(jrc_chain_feed* fd, const jrc_cf32* h_frames, int n_frames)
{
    JRC_TRACE("jrc_chain_feed_submit_rx");
    if (!fd) return JRC_ERR_INVALID_ARG;
    if (!fd->tx_resident) return jrc_fail(fd->ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_submit_rx: no resident TX ports (jrc_chain_feed_set_tx)");
    feed_slot& s = fd->slots[(size_t)fd->head];
    JRC_TRY(feed_check_submit(fd, s, h_frames, n_frames));
    bool replayed = false;
    const int st = feed_launch_slot(fd, s, h_frames, n_frames, true, &replayed, true);
    if (st != JRC_OK) { if (s.ctx != fd->ctx) jrc_fail(fd->ctx, st, "%s", jrc_last_error(s.ctx)); return st; }
    feed_mark_submitted(fd, s, n_frames, replayed);
    return JRC_OK;
}

// 1 when the oldest batch in flight has finished (jrc_chain_feed_collect will not block), 0 when it has not or nothing is in flight
extern "C" int jrc_chain_feed_poll(const jrc_chain_feed* fd)
{
    if (!fd) return JRC_ERR_INVALID_ARG;
    if (fd->in_flight == 0) return 0;
    const feed_slot& s = fd->slots[(size_t)fd->tail];
    if (hipSetDevice(s.ctx->device) != hipSuccess) return 0;
    return hipEventQuery(s.done) == hipSuccess ? 1 : 0;
}

// up to one batch per device in one call: batch k goes to the next slot (device (head + k) mod n) and is staged and enqueued by that
// device's own host thread, all of them at once; returns when every batch has left the caller's buffers.  n_batches <= free slots.
extern "C" int jrc_chain_feed_submit_many(jrc_chain_feed* fd, const jrc_cf32* const* h_frames, const int* n_frames, int n_batches)
{
    JRC_TRACE("jrc_chain_feed_submit_many");
    if (!fd || !h_frames || !n_frames || n_batches < 0) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = fd->ctx;
    if (n_batches > fd->n_slots - fd->in_flight)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_submit_many: %d batches but only %d free slots", n_batches, fd->n_slots - fd->in_flight);
    for (int k = 0; k < n_batches; k++) {
        if (!h_frames[k]) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_feed_submit_many: batch %d has no frames", k);
        JRC_TRY(feed_check_submit(fd, fd->slots[(size_t)((fd->head + k) % fd->n_slots)], h_frames[k], n_frames[k]));
    }
    if (fd->workers.empty()) {                                   // single device: one after the other on this thread
        for (int k = 0; k < n_batches; k++) JRC_TRY(jrc_chain_feed_submit(fd, h_frames[k], n_frames[k]));
        return JRC_OK;
    }
    int done = 0, st_all = JRC_OK;
    while (done < n_batches) {
        // a wave of at most one batch per device, so that every device thread holds one job
        const int wave = n_batches - done < fd->n_devices ? n_batches - done : fd->n_devices;
        std::vector<bool> replayed((size_t)wave, false);
        std::vector<char> rp((size_t)wave, 0);
        for (int k = 0; k < wave; k++) {
            const int si = (fd->head + k) % fd->n_slots;
            feed_slot* sp = &fd->slots[(size_t)si];
            auto* w = fd->workers[(size_t)si % fd->workers.size()];
            const jrc_cf32* src = h_frames[done + k];
            const int n = n_frames[done + k];
            char* flag = &rp[(size_t)k];
            std::lock_guard<std::mutex> lk(w->m);
            w->job = [fd, sp, src, n, flag]() { bool r = false; const int st = feed_launch_slot(fd, *sp, src, n, false, &r); *flag = r ? 1 : 0; return st; };
            w->has_job = true; w->done = false;
            w->cv.notify_all();
        }
        const int head0 = fd->head;
        int first_bad = wave;
        std::vector<int> sts((size_t)wave, JRC_OK);
        for (int k = 0; k < wave; k++) {
            const int si = (head0 + k) % fd->n_slots;
            auto* w = fd->workers[(size_t)si % fd->workers.size()];
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->done; });
            sts[(size_t)k] = w->status;
            if (w->status != JRC_OK && st_all == JRC_OK) { st_all = w->status; first_bad = k; jrc_fail(ctx, st_all, "%s", jrc_last_error(fd->slots[(size_t)si].ctx)); }
        }
        // batches ahead of the first failure are in flight like any other: they are marked and will be collected, in order.  A batch BEHIND
        // it that did launch cannot be (results come back in submission order, and its predecessor never ran): its stream is drained so that
        // the pinned staging buffer and the slot are free again, and its work is dropped.  The failing batches themselves are drained as well.
        for (int k = 0; k < first_bad; k++) {
            feed_slot& s = fd->slots[(size_t)fd->head];
            feed_mark_submitted(fd, s, n_frames[done + k], rp[(size_t)k] != 0);
        }
        for (int k = first_bad; k < wave; k++) {       // failed launches too: their copy from the pinned buffer may already be queued
            feed_slot& s = fd->slots[(size_t)((head0 + k) % fd->n_slots)];
            if (hipSetDevice(s.ctx->device) == hipSuccess) (void)hipStreamSynchronize(s.stream);
        }
        if (st_all != JRC_OK) return st_all;
        done += wave;
    }
    return JRC_OK;
}

extern "C" int jrc_chain_feed_collect(jrc_chain_feed* fd, jrc_ra_result* results, jrc_cf32* maps, int* n_frames)
{
    JRC_TRACE("jrc_chain_feed_collect");
    if (!fd || !results) return JRC_ERR_INVALID_ARG;
    if (fd->in_flight == 0) { if (n_frames) *n_frames = 0; return 0; }
    feed_slot& s = fd->slots[(size_t)fd->tail];
    jrc_ctx* ctx = s.ctx;
    {
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipEventSynchronize(s.done);
        if (e != hipSuccess) return feed_relay(fd, ctx, jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_feed_collect: %s", hipGetErrorString(e)));
    }
    for (int i = 0; i < s.n_frames; i++) {
        results[i] = s.h_results[i];
        // snr_est / published with the host libm's log10f, as jrc_chain_fetch_results does (range_angle_estimator_impl.cc:227)
        ra_finish_host(&results[i], fd->cfg.snr_threshold, fd->cfg.power_threshold);
    }
    const int nm = s.n_frames < fd->maps_per_slot ? s.n_frames : fd->maps_per_slot;
    if (maps && nm > 0) memcpy(maps, s.h_maps, sizeof(float2) * (size_t)nm * fd->map_elems);
    if (n_frames) *n_frames = s.n_frames;
    const int n = s.n_frames;
    s.state = 0; s.n_frames = 0;
    fd->tail = (fd->tail + 1) % fd->n_slots;
    fd->in_flight--;
    return n;
}
