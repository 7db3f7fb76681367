// onchip.hip — kernels written in round 6, while the GPU pool was closed to this repository: checked on the CPU emulation of the kernels
// (tests/hipcpu) against the oracle and against the kernels they replace, NEVER run or timed on a device.  Both are opt-in; the defaults are the
// kernels of rounds 1-5, whose device code this round left byte-identical (tools/device_code_diff.py).
//   * td_onchip_kernel        target_simulator (lib/target_simulator_impl.cc:202-385), bursts that fit a workgroup's LDS, one kernel (JRC_TSIM_ONCHIP=1)
//   * ofdm_mod_burst_kernel   fft_vxx(reverse, shift, window) + cyclic prefixer + zero_pad (lib/zero_pad_impl.cc:76-90) behind every TX port, one kernel
//                             (jrc_ofdm_mod_pad_dev; examples/radar_sim_device_resident.py with JRC_DRF_FUSED_MOD=1)
#include "tsim_device.h"

#include <cmath>

// ---- the whole burst on chip: bursts short enough for (2 + R) x n cells of LDS — the 64-carrier flowgraphs' 2400-sample
//      bursts (77 KB at two RX antennas) — go through ONE kernel, one workgroup per burst: per (simulator, target) pair the input . doppler is
//      loaded, transformed as a single n-point mixed-radix Stockham transform in LDS (the column passes above with a tile one column wide, radices =
//      the factors of n, twiddles from the context's n-entry table), multiplied by the pair's timeshift (and phase) into one LDS accumulator per RX
//      antenna; then every antenna's sum goes through the inverse transform (conj, forward, conj) and out, with the self-coupling term and the
//      accumulate option of the column pass.  One read of every input, R writes, nothing else touches HBM; 1 launch instead of 3.
//      The timeshift table is the direct route's (row-pass order [k1][pos(k2)], n = n1 x n2): natural k = k1 + n1 k2 reads entry k1 n2 + pos(k2).
//      Same algebra as the three passes, another factorisation: results agree with them to rounding, not bit for bit.
//      Written in round 6 without a device: opt-in (JRC_TSIM_ONCHIP=1), never timed.
__global__ __launch_bounds__(256) void td_onchip_kernel(td_srcs srcs, long in_stride, td_ts ts, long ts_l_stride, int V, int R,
                                                        float2* __restrict__ out, long out_burst_stride, long out_rx_stride, td_self self,
                                                        float self_coupling, int accumulate, const float2* __restrict__ wn, td_plan pl /* n1 = n: radices of n */,
                                                        int d_n1, int d_n2 /* the direct route's split of n: order of the timeshift table */)
{
    extern __shared__ __attribute__((aligned(16))) float2 td_lds[];
    const int n = pl.n1, tid = threadIdx.x;
    float2* buf0 = td_lds;
    float2* buf1 = buf0 + n;
    float2* acc = buf1 + n;                                                  // [R][n]
    const size_t b = blockIdx.x;
    const int m = d_n2 > 256 ? d_n2 / 256 : 1;
    for (int v = 0; v < V; v++) {
        const float2* __restrict__ src = srcs.in[v] + b * (size_t)in_stride;
        const float2* __restrict__ dz = srcs.dop[v];
        for (int i = tid; i < n; i += 256) buf0[i] = cmul(src[i], dz[i]);    // volk_32fc_x2_multiply_32fc (:345)
        __syncthreads();
        const float2* X = td_col_transform<1>(buf0, buf1, wn, pl, 0, tid, 256);
        for (int k = tid; k < n; k += 256) {
            const int k1 = k % d_n1, k2 = k / d_n1;
            const int pos = m > 1 ? (k2 % m) * 256 + k2 / m : k2;
            float2 x = X[k];
            if (ts.use_phase) x = cmul(x, ts.phase[v]);
            const float2* __restrict__ tr = ts.tsp[v] + (size_t)k1 * d_n2 + pos;
            for (int l = 0; l < R; l++) {
                const float2 y = cmul(x, tr[(size_t)l * ts_l_stride]);
                acc[(size_t)l * n + k] = v ? cadd(acc[(size_t)l * n + k], y) : y;
            }
        }
        __syncthreads();
    }
    for (int l = 0; l < R; l++) {
        for (int k = tid; k < n; k += 256) { const float2 a = acc[(size_t)l * n + k]; buf0[k] = make_float2(a.x, -a.y); }   // conjugated: the inverse runs on the forward passes
        __syncthreads();
        const float2* y = td_col_transform<1>(buf0, buf1, wn, pl, 0, tid, 256);
        float2* o = out + b * (size_t)out_burst_stride + (size_t)l * out_rx_stride;
        for (int i = tid; i < n; i += 256) {
            float2 r = make_float2(y[i].x, -y[i].y);
            if (accumulate) r = cadd(o[i], r);
            for (int q = 0; q < self.n; q++) {                               // out += (gr_complex)pow(10, db/20) * in  (:376), per simulator
                const float2 xi = self.in[q][b * (size_t)in_stride + i];
                r = cadd(r, make_float2(self_coupling * xi.x - 0.0f * xi.y, self_coupling * xi.y + 0.0f * xi.x));
            }
            o[i] = r;
        }
        __syncthreads();                                                     // buf0 / buf1 are the next antenna's
    }
}

int td_onchip_launch(jrc_ctx* ctx, hipStream_t s, const td_srcs& srcs, const td_ts& ts, long ts_l_stride, const td_self& self, int V, int R, int n,
                     int d_n1, int d_n2, int n_bursts, float2* d_out, float self_coupling, int accumulate)
{
    const size_t lds_on = sizeof(float2) * (size_t)(2 + R) * n;
    td_plan pn;
    td_factor(n, &pn);
    const float2* wn = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, n, -1, &wn));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)td_onchip_kernel, lds_on));
    hipLaunchKernelGGL(td_onchip_kernel, dim3((unsigned)n_bursts), dim3(256), lds_on, s, srcs, (long)n, ts, ts_l_stride, V, R, d_out,
                       (long)R * n, (long)n, self, self_coupling, accumulate, wn, pn, d_n1, d_n2);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// ---- OFDM modulator + zero_pad as ONE kernel: the fft_vxx(reverse, shift, window) -> ofdm_cyclic_prefixer -> zero_pad chain behind
// every TX port of the simulation flowgraph (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:801-897, :2184-2188; lib/zero_pad_impl.cc:76-90)
// writes, block by block, the time-domain packet [F][T][n_sym (N + cp)] to HBM, reads it again and writes the padded burst; here symbol
// (f, t, k) of the precoder's output goes through the SAME Stockham passes as jrc_ofdm_mod_dev (fft_stockham_kernel: bit-identical samples) and
// lands at bursts[t][f][pad_front + k (N + cp)] directly, and the workgroups that hold a burst's first / last symbol write its pad noise with the
// generator of zero_pad_kernel (same key: seed of the port, burst, sample -> bit-identical padding).  One launch instead of 1 + T, the unpadded
// time-domain packet never exists.
__global__ __launch_bounds__(256) void ofdm_mod_burst_kernel(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw,
                                                             const float* __restrict__ window, int n, int logn, size_t batch, int tp,
                                                             int n_ports, int n_sym, int cp, int pad_front, int pad_tail,
                                                             unsigned long long seed, unsigned long long seed_port_step, float sigma,
                                                             long out_port_stride, long out_burst_stride)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int per_block = blockDim.x / tp;
    const int lt = threadIdx.x % tp, lb = threadIdx.x / tp;
    const size_t b = (size_t)blockIdx.x * per_block + lb;
    const bool live = b < batch;
    float2* buf0 = lds + (size_t)lb * 2 * n;
    float2* buf1 = buf0 + n;
    const size_t f = b / ((size_t)n_ports * n_sym);
    const int t = (int)((b / n_sym) % n_ports), k = (int)(b % n_sym);
    const float2* src_g = in + b * (size_t)n;
    float2* burst = out + (size_t)t * out_port_stride + f * (size_t)out_burst_stride;
    float2* dst_g = burst + pad_front + (size_t)k * (n + cp) + cp;

    int Ns = 1;
    const float2* cur = nullptr;                  // nullptr = still in global memory
    float2* nxt = buf0;
    bool first = true;
    while (Ns < n) {                              // the passes of fft_stockham_kernel with forward = 0, shift = 1, a cyclic prefix to prepend
        const int R = ((logn & 1) && first) ? 2 : 4;
        if (live) {
            if (R == 2) stockham_pass<2>(first ? src_g : nullptr, n, first ? window : nullptr, cur, nxt, nullptr, 0, tw, n, Ns, 1, lt, tp);
            else stockham_pass<4>(first ? src_g : nullptr, n, first ? window : nullptr, cur, nxt, nullptr, 0, tw, n, Ns, 1, lt, tp);
        }
        __syncthreads();
        cur = nxt; nxt = (nxt == buf0) ? buf1 : buf0;
        Ns *= R; first = false;
    }
    if (!live) return;
    for (int pos = lt; pos < n; pos += tp) dst_g[pos] = cur[pos];
    for (int jj = lt; jj < cp; jj += tp) dst_g[jj - cp] = cur[n - cp + jj];
    // pad noise of burst (t, f): the front by the transform of its first symbol, the tail by that of its last
    const unsigned long long port_seed = seed + seed_port_step * (unsigned long long)t;
    const int n_in = n_sym * (n + cp);
    auto noise = [&](int i) {
        const unsigned long long r = splitmix64(port_seed ^ splitmix64(((unsigned long long)f << 32) | (unsigned)i));
        const float u1 = ((float)(unsigned)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
        const float u2 = (float)(unsigned)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f);     // [0, 1)
        const float rad = sigma * sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);
        burst[i] = make_float2(rad * cs, rad * sn);
    };
    if (k == 0) for (int i = lt; i < pad_front; i += tp) noise(i);
    if (k == n_sym - 1) for (int i = pad_front + n_in + lt; i < pad_front + n_in + pad_tail; i += tp) noise(i);
}

extern "C" int jrc_ofdm_mod_pad_dev(jrc_ctx* ctx, int fft_len, int cp_len, const float* d_window, int n_frames, int n_ports, int n_symbols,
                                    unsigned pad_front, unsigned pad_tail, uint64_t seed, uint64_t seed_port_step,
                                    const jrc_cf32* d_in, jrc_cf32* d_out, long out_port_stride, long out_burst_stride, void* stream)
{
    JRC_TRACE("jrc_ofdm_mod_pad_dev");
    if (!ctx || n_frames < 0 || n_ports < 1 || n_symbols < 1) return JRC_ERR_INVALID_ARG;
    if (cp_len < 0 || cp_len > fft_len) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "ofdm_mod_pad: bad cp_len");
    if (!jrc_is_pow2(fft_len) || fft_len < 4 || fft_len > 8192)
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "ofdm_mod_pad: fft_len %d is not a power of two in [4, 8192] (use jrc_ofdm_mod_dev + jrc_zero_pad_strided_dev)", fft_len);
    const long n_out = (long)n_symbols * (fft_len + cp_len) + pad_front + pad_tail;
    if (n_frames == 0) return (int)n_out;
    if (!d_in || !d_out) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "ofdm_mod_pad: null buffers");
    if (out_burst_stride < n_out || (n_ports > 1 && out_port_stride < out_burst_stride * (long)n_frames))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "ofdm_mod_pad: output strides shorter than the bursts they hold");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int n = fft_len;
    const float2* tw = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, n, +1, &tw));
    int tp = n / 4; if (tp > 256) tp = 256; if (tp < 1) tp = 1;                 // the geometry of launch_fft_vcc_ex's Stockham branch
    const int per_block = 256 / tp;
    const size_t batch = (size_t)n_frames * n_ports * n_symbols;
    const size_t blocks = (batch + per_block - 1) / per_block;
    const size_t lds_bytes = sizeof(float2) * 2 * (size_t)n * per_block;
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)ofdm_mod_burst_kernel, lds_bytes));
    hipLaunchKernelGGL(ofdm_mod_burst_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, s, (const float2*)d_in, (float2*)d_out, tw, d_window, n,
                       jrc_ilog2(n), batch, tp, n_ports, n_symbols, cp_len, (int)pad_front, (int)pad_tail, (unsigned long long)seed,
                       (unsigned long long)seed_port_step, 1e-2f, out_port_stride, out_burst_stride);
    JRC_HIP(ctx, hipGetLastError());
    return (int)n_out;
}
