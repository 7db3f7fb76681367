// chain.hip — fused, device-resident radar chain A1 -> A2 -> A3 -> A4 -> A5 over a batch of frames
//
// Replaces, in one launch sequence, the reference's
//   mimo_ofdm_radar (lib/mimo_ofdm_radar_impl.cc:131-340)
//   -> fft_vxx reverse/no-shift, size N*Ir      (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:940-962)
//   -> matrix_transpose (lib/matrix_transpose_impl.cc:69-110)
//   -> fft_vxx forward/shift, size P*Ia         (...radar_sim.grc:963-985)
//   -> range_angle_estimator (lib/range_angle_estimator_impl.cc:121-284)
//
// range_angle_fused_kernel is the roofline kernel: it reads the P x N channel estimate (L2 resident)
// and streams the (N*Ir) x (P*Ia) complex map to HBM exactly once, doing both zero-padded FFTs and the
// estimator's arg-max scan on chip.
//
//   Range axis.  R[p][k] = sum_{n<N} H[p][n] e^{+j2pi nk/NR}, NR = N*Ir, is needed only through its
//   N non-zero inputs.  A workgroup owns the residue class k = C*q + c (C = NR/64, q < 64):
//       R[p][C q + c] = IFFT_64( g_c[p] )[q],   g_c[p][n'] = sum_m H[p][n'+64m] e^{+j2pi (n'+64m) c / NR}
//   i.e. a twiddled fold of the N inputs down to 64 points followed by one 64-point transform per
//   virtual-array pair, done by one wavefront with one point per lane (cross-lane shuffles, no LDS).
//   Angle axis.  For each of the workgroup's 64 range bins the P inputs x[p] are zero-padded to
//   NA = P*Ia: out[Ia u + r] = FFT_P( x[p] e^{-j2pi p r / NA} )[u].  One lane computes two adjacent
//   residues r = 2i, 2i+1 (two P-point FFTs in registers) so that it owns adjacent output bins and
//   stores them as one 16-byte access; 8 lanes cover a full 128-byte line of the map row.
//   fftshift is a rotation of u.  The transpose + zero padding of matrix_transpose never touches
//   memory.
#include "radar_kernels.h"

#include <cmath>

#define RA_L 64   // range bins (and fold length) per workgroup

// ---- tiny in-register forward FFT, P in {1,2,4,8,16}, natural order in / out --------------------
template <int P, int K>
struct TwMul {   // multiply by w_P^K = exp(-j 2 pi K / P)
    static __device__ __forceinline__ float2 mul(float2 v)
    {
        constexpr int idx = K * (16 / P);   // sixteenths of a turn, 0..7
        constexpr float R2 = 0.70710678118654752440f;
        constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
        if constexpr (idx == 0) return v;
        else if constexpr (idx == 4) return make_float2(v.y, -v.x);
        else if constexpr (idx == 2) return make_float2((v.x + v.y) * R2, (v.y - v.x) * R2);
        else if constexpr (idx == 6) return make_float2((v.y - v.x) * R2, -(v.x + v.y) * R2);
        else if constexpr (idx == 1) return make_float2(v.x * C1 + v.y * S1, v.y * C1 - v.x * S1);
        else if constexpr (idx == 3) return make_float2(v.x * S1 + v.y * C1, v.y * S1 - v.x * C1);
        else if constexpr (idx == 5) return make_float2(v.y * C1 - v.x * S1, -(v.x * C1 + v.y * S1));
        else return make_float2(v.y * S1 - v.x * C1, -(v.x * S1 + v.y * C1));   // idx == 7
    }
};

template <int P, int K>
struct Bfly {
    static __device__ __forceinline__ void run(float2* x, const float2* e, const float2* o)
    {
        float2 t = TwMul<P, K>::mul(o[K]);
        x[K] = cadd(e[K], t);
        x[K + P / 2] = csub(e[K], t);
        if constexpr (K + 1 < P / 2) Bfly<P, K + 1>::run(x, e, o);
    }
};

template <int P>
__device__ __forceinline__ void fft_fwd_small(float2 (&x)[P])
{
    if constexpr (P > 1) {
        float2 e[P / 2], o[P / 2];
#pragma unroll
        for (int k = 0; k < P / 2; k++) { e[k] = x[2 * k]; o[k] = x[2 * k + 1]; }
        fft_fwd_small<P / 2>(e);
        fft_fwd_small<P / 2>(o);
        Bfly<P, 0>::run(x, e, o);
    }
}

// ---- the fused kernel ------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void range_angle_fused_kernel(
    const float2* __restrict__ H,        // [F][P][N]
    float2* __restrict__ map,            // [F][NR][NA]
    PeakPartial* __restrict__ partials,  // [F][C]
    const float2* __restrict__ twR,      // [NR]  exp(+j 2 pi i / NR)
    const float2* __restrict__ twA,      // [NA]  exp(-j 2 pi i / NA)
    int N, int NR, int Ia, int F)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int NA = P * Ia;
    const int C = NR / RA_L;
    // XCD-aware decode: block b runs on XCD b%8; all residue classes of a frame share that frame's H,
    // so keep a frame's workgroups on one XCD (one L2).
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int f = (j / C) * 8 + xcd;
    const int c = j % C;
    if (f >= F) return;

    float2* s_twc = smem;                       // [N]   exp(+j 2 pi n c / NR)
    float2* s_g = s_twc + N;                    // [P][65]
    float2* s_twA = s_g + P * (RA_L + 1);       // [NA]
    float2* s_tw64 = s_twA + NA;                // [32]  exp(+j 2 pi k / 64)
    __shared__ PeakPartial red[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int n = tid; n < N; n += 256) s_twc[n] = twR[(int)(((long)n * c) % NR)];
    for (int a = tid; a < NA; a += 256) s_twA[a] = twA[a];
    if (tid < 32) s_tw64[tid] = twR[tid * (NR / 64)];
    __syncthreads();

    // ---- range axis: fold to 64 points, 64-point inverse FFT across the wavefront -----------------
    const float2* Hf = H + (size_t)f * P * N;
    for (int p = wave; p < P; p += 4) {
        const float2* Hp = Hf + (size_t)p * N;
        float2 v = make_float2(0.f, 0.f);
        for (int n = lane; n < N; n += RA_L) {
            float2 h = Hp[n], w = s_twc[n];
            v.x = fmaf(h.x, w.x, fmaf(-h.y, w.y, v.x));
            v.y = fmaf(h.x, w.y, fmaf(h.y, w.x, v.y));
        }
#pragma unroll
        for (int half = 32; half >= 1; half >>= 1) {     // radix-2 DIF, twiddle exp(+j 2 pi k / (2 half))
            float2 o = make_float2(__shfl_xor(v.x, half), __shfl_xor(v.y, half));
            if (lane & half) {
                float2 d = csub(o, v);
                v = cmul(d, s_tw64[(lane & (half - 1)) * (32 / half)]);
            } else {
                v = cadd(v, o);
            }
        }
        s_g[p * (RA_L + 1) + (__brev((unsigned)lane) >> 26)] = v;   // lane holds X[bitrev6(lane)]
    }
    __syncthreads();

    // ---- angle axis + fftshift + store + arg-max -------------------------------------------------
    PeakTracker trk;
    trk.init();
    const int ipr = Ia >> 1;                 // residue pairs per range bin
    const int items = RA_L * ipr;
    const int amask = NA - 1, ahalf = NA >> 1;
    float2* mapf = map + (size_t)f * NR * NA;
    for (int w0 = 0; w0 < items; w0 += 256) {     // items is a multiple of 64: whole waves are in or out
        const int w = w0 + tid;
        if (w >= items) break;
        const int i = w % ipr, ql = w / ipr;
        const int k = C * ql + c;            // global range bin
        float2 x[P];
#pragma unroll
        for (int p = 0; p < P; p++) x[p] = s_g[p * (RA_L + 1) + ql];
        float2 o0[P], o1[P];
        {
            const int r = 2 * i;
#pragma unroll
            for (int p = 0; p < P; p++) o0[p] = (p == 0) ? x[0] : cmul(x[p], s_twA[(p * r) & amask]);
            fft_fwd_small<P>(o0);
        }
        {
            const int r = 2 * i + 1;
#pragma unroll
            for (int p = 0; p < P; p++) o1[p] = (p == 0) ? x[0] : cmul(x[p], s_twA[(p * r) & amask]);
            fft_fwd_small<P>(o1);
        }
        float2* row = mapf + (size_t)k * NA;
        const unsigned flat0 = (unsigned)k * (unsigned)NA;
#pragma unroll
        for (int u = 0; u < P; u++) {
            const int a = (Ia * u + 2 * i + ahalf) & amask;   // fftshift: out'[a'] = out[(a' + NA/2) % NA]
            float4 val = make_float4(o0[u].x, o0[u].y, o1[u].x, o1[u].y);
            *reinterpret_cast<float4*>(row + a) = val;
        }
        // estimator arg-max (lib/range_angle_estimator_impl.cc:137-151) on the values still in registers
        float m = -1.0f;
#pragma unroll
        for (int u = 0; u < P; u++) m = fmaxf(m, fmaxf(fast_power(o0[u]), fast_power(o1[u])));
        const float thr = trk.raise(m);
        if (m >= thr) {
#pragma unroll
            for (int u = 0; u < P; u++) {
                const int a = (Ia * u + 2 * i + ahalf) & amask;
                if (fast_power(o0[u]) >= thr) trk.exact(o0[u], flat0 + a);
                if (fast_power(o1[u]) >= thr) trk.exact(o1[u], flat0 + a + 1);
            }
        }
    }
    block_reduce_peak(trk, red);
    if (tid == 0) { partials[(size_t)f * C + c].best = trk.best; partials[(size_t)f * C + c].idx = trk.idx; }
}

// ------------------------------------------------------------------------------------------------
struct jrc_chain {
    jrc_ctx* ctx;
    jrc_chain_cfg cfg;
    int P, NR, NA, C, max_frames;
    float* d_bins = nullptr;          // range_bins (NR) then angle_bins (NA)
    PeakPartial* d_partials = nullptr;
    const float2* twR = nullptr;
    const float2* twA = nullptr;
    size_t lds_bytes = 0;
    // timing
    bool timing = false;
    static const int kPool = 512;
    std::vector<hipEvent_t> ev;       // 4 per run
    int ev_used = 0;
    double ms_acc[3] = {0, 0, 0};
    int launches = 0;
    jrc_ra_result* h_pinned = nullptr;
};

template <int P>
static int launch_fused(jrc_chain* ch, int n_frames, const float2* d_H, float2* d_map, hipStream_t s)
{
    const int groups = (n_frames + 7) / 8;
    dim3 grid((unsigned)(groups * 8 * ch->C));
    static size_t attr_bytes = 64 * 1024;   // dynamic LDS above 64 KiB must be opted into per kernel
    if (ch->lds_bytes > attr_bytes) {
        JRC_HIP(ch->ctx, hipFuncSetAttribute((const void*)range_angle_fused_kernel<P>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ch->lds_bytes));
        attr_bytes = ch->lds_bytes;
    }
    hipLaunchKernelGGL((range_angle_fused_kernel<P>), grid, dim3(256), ch->lds_bytes, s, d_H, d_map, ch->d_partials,
                       ch->twR, ch->twA, ch->cfg.fft_len, ch->NR, ch->cfg.interp_angle, n_frames);
    JRC_HIP(ch->ctx, hipGetLastError());
    return JRC_OK;
}

extern "C" int jrc_chain_create(jrc_ctx* ctx, const jrc_chain_cfg* cfg, const float* range_bins,
                                const float* angle_bins, int max_frames, jrc_chain** out)
{
    if (!ctx || !cfg || !range_bins || !angle_bins || !out || max_frames <= 0) return JRC_ERR_INVALID_ARG;
    const int N = cfg->fft_len, T = cfg->N_tx, R = cfg->N_rx, P = T * R;
    if (N <= 0 || T <= 0 || R <= 0 || cfg->N_sym < 0 || cfg->N_pre < 0 || cfg->interp_range <= 0 ||
        cfg->interp_angle <= 0 || cfg->n_items < cfg->N_pre + cfg->N_sym)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_create: inconsistent sizes");
    const long NR = (long)N * cfg->interp_range, NA = (long)P * cfg->interp_angle;
    if (!jrc_is_pow2(N) || N < RA_L || !jrc_is_pow2(cfg->interp_range) || !jrc_is_pow2(P) || P > 16 ||
        !jrc_is_pow2(cfg->interp_angle) || cfg->interp_angle < 2 || NA < 4 || NR * NA >= (1L << 32) || N > 8192)
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED,
                        "fused radar chain needs power-of-two fft_len in [64, 8192], power-of-two N_tx*N_rx <= 16, "
                        "power-of-two interpolation factors (angle >= 2); got N=%d P=%d Ir=%d Ia=%d",
                        N, P, cfg->interp_range, cfg->interp_angle);
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    jrc_chain* ch = new jrc_chain();
    ch->ctx = ctx; ch->cfg = *cfg; ch->P = P; ch->NR = (int)NR; ch->NA = (int)NA; ch->C = (int)(NR / RA_L);
    ch->max_frames = max_frames;
    ch->lds_bytes = sizeof(float2) * ((size_t)N + (size_t)P * (RA_L + 1) + (size_t)NA + 32);
    hipError_t e = hipMalloc((void**)&ch->d_bins, sizeof(float) * (size_t)(NR + NA));
    if (e == hipSuccess) e = hipMemcpy(ch->d_bins, range_bins, sizeof(float) * NR, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ch->d_bins + NR, angle_bins, sizeof(float) * NA, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&ch->d_partials, sizeof(PeakPartial) * (size_t)max_frames * ch->C);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ch->h_pinned, sizeof(jrc_ra_result) * (size_t)max_frames, hipHostMallocDefault);
    int st = JRC_OK;
    if (e != hipSuccess) st = jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_create: %s", hipGetErrorString(e));
    if (st == JRC_OK) st = jrc_get_twiddles(ctx, (int)NR, +1, &ch->twR);
    if (st == JRC_OK) st = jrc_get_twiddles(ctx, (int)NA, -1, &ch->twA);
    if (st != JRC_OK) { jrc_chain_destroy(ch); return st; }
    *out = ch;
    return JRC_OK;
}

extern "C" void jrc_chain_destroy(jrc_chain* ch)
{
    if (!ch) return;
    (void)hipDeviceSynchronize();
    for (auto& e : ch->ev) (void)hipEventDestroy(e);
    if (ch->d_bins) (void)hipFree(ch->d_bins);
    if (ch->d_partials) (void)hipFree(ch->d_partials);
    if (ch->h_pinned) (void)hipHostFree(ch->h_pinned);
    delete ch;
}

extern "C" size_t jrc_chain_frame_bytes(const jrc_chain* ch)
{
    return ch ? sizeof(float2) * (size_t)(ch->cfg.N_tx + ch->cfg.N_rx) * ch->cfg.n_items * ch->cfg.fft_len : 0;
}
extern "C" size_t jrc_chain_chanest_bytes(const jrc_chain* ch) { return ch ? sizeof(float2) * (size_t)ch->P * ch->cfg.fft_len : 0; }
extern "C" size_t jrc_chain_map_bytes(const jrc_chain* ch) { return ch ? sizeof(float2) * (size_t)ch->NR * ch->NA : 0; }

extern "C" int jrc_chain_set_timing(jrc_chain* ch, int enabled)
{
    if (!ch) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (enabled && ch->ev.empty()) {
        ch->ev.resize((size_t)jrc_chain::kPool * 4);
        for (auto& e : ch->ev) JRC_HIP(ctx, hipEventCreate(&e));
    }
    ch->timing = enabled != 0;
    ch->ev_used = 0; ch->launches = 0;
    ch->ms_acc[0] = ch->ms_acc[1] = ch->ms_acc[2] = 0;
    return JRC_OK;
}

static int chain_drain_events(jrc_chain* ch)
{
    jrc_ctx* ctx = ch->ctx;
    for (int i = 0; i < ch->ev_used; i++) {
        hipEvent_t* e = &ch->ev[(size_t)i * 4];
        JRC_HIP(ctx, hipEventSynchronize(e[3]));
        for (int k = 0; k < 3; k++) {
            float ms = 0.f;
            JRC_HIP(ctx, hipEventElapsedTime(&ms, e[k], e[k + 1]));
            ch->ms_acc[k] += ms;
        }
        ch->launches++;
    }
    ch->ev_used = 0;
    return JRC_OK;
}

extern "C" int jrc_chain_get_timing(jrc_chain* ch, float ms[3], int* launches)
{
    if (!ch || !ms) return JRC_ERR_INVALID_ARG;
    JRC_TRY(chain_drain_events(ch));
    for (int k = 0; k < 3; k++) ms[k] = ch->launches ? (float)(ch->ms_acc[k] / ch->launches) : 0.f;
    if (launches) *launches = ch->launches;
    return JRC_OK;
}

extern "C" int jrc_chain_run_dev(jrc_chain* ch, int n_frames, const jrc_cf32* d_frames, jrc_cf32* d_chanest,
                                 jrc_cf32* d_map, jrc_ra_result* d_results, void* stream)
{
    if (!ch || !d_frames || !d_chanest || !d_map || !d_results) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (n_frames <= 0 || n_frames > ch->max_frames)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_run_dev: n_frames %d outside (0, %d]", n_frames, ch->max_frames);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const jrc_chain_cfg& c = ch->cfg;
    hipEvent_t* ev = nullptr;
    if (ch->timing) {
        if (ch->ev_used == jrc_chain::kPool) JRC_TRY(chain_drain_events(ch));
        ev = &ch->ev[(size_t)ch->ev_used * 4];
        ch->ev_used++;
        JRC_HIP(ctx, hipEventRecord(ev[0], s));
    }
    // A1
    ChanestGeom g;
    g.N = c.fft_len; g.S = c.N_sym;
    g.port_stride = (long)c.n_items * c.fft_len;
    g.frame_stride = g.port_stride * (c.N_tx + c.N_rx);
    g.tx_item0 = c.N_pre; g.rx_item0 = c.N_pre; g.interleave = c.enable_tx_interleave;
    JRC_TRY(launch_radar_chanest(ctx, c.N_tx, c.N_rx, (const float2*)d_frames, (float2*)d_chanest, g, n_frames, s));
    if (ev) JRC_HIP(ctx, hipEventRecord(ev[1], s));
    // A2 + A3 + A4 + arg-max half of A5
    int st;
    switch (ch->P) {
        case 1: st = launch_fused<1>(ch, n_frames, (const float2*)d_chanest, (float2*)d_map, s); break;
        case 2: st = launch_fused<2>(ch, n_frames, (const float2*)d_chanest, (float2*)d_map, s); break;
        case 4: st = launch_fused<4>(ch, n_frames, (const float2*)d_chanest, (float2*)d_map, s); break;
        case 8: st = launch_fused<8>(ch, n_frames, (const float2*)d_chanest, (float2*)d_map, s); break;
        default: st = launch_fused<16>(ch, n_frames, (const float2*)d_chanest, (float2*)d_map, s); break;
    }
    JRC_TRY(st);
    if (ev) JRC_HIP(ctx, hipEventRecord(ev[2], s));
    // rest of A5
    RaParams prm;
    prm.vlen = ch->NA; prm.n_inputs = ch->NR; prm.n_range_bins = ch->NR; prm.n_angle_bins = ch->NA;
    prm.noise_discard_range_m = c.noise_discard_range_m; prm.noise_discard_angle_deg = c.noise_discard_angle_deg;
    JRC_TRY(launch_ra_finalize(ctx, (const float2*)d_map, (size_t)ch->NR * ch->NA, ch->d_partials, ch->C, prm, ch->d_bins,
                               ch->d_bins + ch->NR, d_results, n_frames, s));
    if (ev) JRC_HIP(ctx, hipEventRecord(ev[3], s));
    return JRC_OK;
}

extern "C" int jrc_chain_fetch_results(jrc_chain* ch, int n_frames, const jrc_ra_result* d_results,
                                       jrc_ra_result* h_results, void* stream)
{
    if (!ch || !d_results || !h_results) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (n_frames <= 0 || n_frames > ch->max_frames) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_fetch_results: bad n_frames");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    JRC_HIP(ctx, hipMemcpyAsync(ch->h_pinned, d_results, sizeof(jrc_ra_result) * (size_t)n_frames, hipMemcpyDeviceToHost, s));
    JRC_HIP(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < n_frames; i++) {
        h_results[i] = ch->h_pinned[i];
        ra_finish_host(&h_results[i], ch->cfg.snr_threshold, ch->cfg.power_threshold);
    }
    return JRC_OK;
}
