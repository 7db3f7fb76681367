// radar.hip — A1: MIMO-OFDM radar channel estimate (matched filter over OFDM symbols)
//
// Replaces mimo_ofdm_radar_impl::general_work (reference lib/mimo_ofdm_radar_impl.cc:131-340):
//   H[p][sc] = sum_{sym<S} rx_r[sym][sc] * conj(tx_t[sym][sc]),  p = r*T+t (or t*R+r when interleaved)
// HBM-bound: (T+R)*S*N*8 bytes read, P*N*8 written per frame; one lane per subcarrier so every load is
// a coalesced 512-byte wave access, all P accumulators live in registers, symbols are accumulated in
// order with individually rounded products (no FMA contraction) so the result is bit-identical to
// the reference's scalar loop.
#include "radar_kernels.h"
#include "fft_device.h"

#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// One lane per (subcarrier, RX antenna): T accumulators in registers, symbols consumed in batches of U
// so that U*(T+1) independent 8-byte loads are in flight per lane before the first dependent use.  The R
// waves of a workgroup walk the same TX rows, so the T re-reads of each TX row hit L1/L2, not HBM.
template <int T, int U>
__global__ __launch_bounds__(256) void radar_chanest_kernel(const float2* __restrict__ frames,
                                                            float2* __restrict__ H, ChanestGeom g, int R)
{
#pragma clang fp contract(off)
    const int sc = blockIdx.x * 64 + threadIdx.x;
    const int r = threadIdx.y;
    const int f = blockIdx.y;
    if (sc >= g.N) return;
    const float2* fb = frames + (size_t)f * g.frame_stride;
    const float2* rxp = fb + (size_t)(T + r) * g.port_stride + (size_t)g.rx_item0 * g.N + sc;
    const float2* txp = fb + (size_t)g.tx_item0 * g.N + sc;

    float2 acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) acc[t] = make_float2(0.f, 0.f);

    int sym = 0;
    for (; sym + U <= g.S; sym += U) {
        float2 rx[U], tx[U][T];
#pragma unroll
        for (int u = 0; u < U; u++) {
            rx[u] = rxp[(size_t)(sym + u) * g.N];
#pragma unroll
            for (int t = 0; t < T; t++) tx[u][t] = txp[(size_t)t * g.port_stride + (size_t)(sym + u) * g.N];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < T; t++) {
                // rx * conj(tx) = (ac + bd) + j(bc - ad), products rounded individually (:273)
                float pr = rx[u].x * tx[u][t].x + rx[u].y * tx[u][t].y;
                float pi = rx[u].y * tx[u][t].x - rx[u].x * tx[u][t].y;
                acc[t].x = acc[t].x + pr;
                acc[t].y = acc[t].y + pi;
            }
    }
    for (; sym < g.S; sym++) {
        float2 rx = rxp[(size_t)sym * g.N];
#pragma unroll
        for (int t = 0; t < T; t++) {
            float2 tx = txp[(size_t)t * g.port_stride + (size_t)sym * g.N];
            float pr = rx.x * tx.x + rx.y * tx.y;
            float pi = rx.y * tx.x - rx.x * tx.y;
            acc[t].x = acc[t].x + pr;
            acc[t].y = acc[t].y + pi;
        }
    }
    float2* Hf = H + (size_t)f * T * R * g.N;
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int p = g.interleave ? (t * R + r) : (r * T + t);   // :262-269
        Hf[(size_t)p * g.N + sc] = acc[t];
    }
}

// Same arithmetic, two adjacent subcarriers per lane (16-byte loads): half the load instructions per byte.
template <int T, int U>
__global__ __launch_bounds__(256) void radar_chanest_x2_kernel(const float2* __restrict__ frames,
                                                               float2* __restrict__ H, ChanestGeom g, int R)
{
#pragma clang fp contract(off)
    const int sc = (blockIdx.x * 64 + threadIdx.x) * 2;
    const int r = threadIdx.y;
    const int f = blockIdx.y;
    if (sc >= g.N) return;
    const float2* fb = frames + (size_t)f * g.frame_stride;
    const float4* rxp = reinterpret_cast<const float4*>(fb + (size_t)(T + r) * g.port_stride + (size_t)g.rx_item0 * g.N + sc);
    const float2* txb = fb + (size_t)g.tx_item0 * g.N + sc;
    const size_t row4 = (size_t)g.N / 2;      // float4 per symbol row

    float4 acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto mac = [&](float4& a, const float4 rx, const float4 tx) {
        // rx * conj(tx) = (ac + bd) + j(bc - ad) for both subcarriers, products rounded individually (:273)
        const float pr0 = rx.x * tx.x + rx.y * tx.y, pi0 = rx.y * tx.x - rx.x * tx.y;
        const float pr1 = rx.z * tx.z + rx.w * tx.w, pi1 = rx.w * tx.z - rx.z * tx.w;
        a.x = a.x + pr0; a.y = a.y + pi0; a.z = a.z + pr1; a.w = a.w + pi1;
    };
    int sym = 0;
    for (; sym + U <= g.S; sym += U) {
        float4 rx[U], tx[U][T];
#pragma unroll
        for (int u = 0; u < U; u++) {
            {   // each RX symbol is read exactly once: non-temporal, so it does not push the TX rows (read by every receiver) out of cache
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(rxp + (size_t)(sym + u) * row4));
                rx[u] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int t = 0; t < T; t++)
                tx[u][t] = *reinterpret_cast<const float4*>(txb + (size_t)t * g.port_stride + (size_t)(sym + u) * g.N);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < T; t++) mac(acc[t], rx[u], tx[u][t]);
    }
    for (; sym < g.S; sym++) {
        const float4 rx = rxp[(size_t)sym * row4];
#pragma unroll
        for (int t = 0; t < T; t++)
            mac(acc[t], rx, *reinterpret_cast<const float4*>(txb + (size_t)t * g.port_stride + (size_t)sym * g.N));
    }
    float2* Hf = H + (size_t)f * T * R * g.N;
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int p = g.interleave ? (t * R + r) : (r * T + t);   // :262-269
        *reinterpret_cast<float4*>(Hf + (size_t)p * g.N + sc) = acc[t];
    }
}

// any T, R: one lane per (pair, subcarrier)
__global__ __launch_bounds__(256) void radar_chanest_generic_kernel(const float2* __restrict__ frames,
                                                                    float2* __restrict__ H, ChanestGeom g,
                                                                    int T, int R)
{
#pragma clang fp contract(off)
    const int sc = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const int rt = blockIdx.z;
    if (sc >= g.N) return;
    const int r = rt / T, t = rt % T;
    const float2* fb = frames + (size_t)f * g.frame_stride;
    const float2* txp = fb + (size_t)t * g.port_stride + (size_t)g.tx_item0 * g.N + sc;
    const float2* rxp = fb + (size_t)(T + r) * g.port_stride + (size_t)g.rx_item0 * g.N + sc;
    float2 acc = make_float2(0.f, 0.f);
    for (int sym = 0; sym < g.S; sym++) {
        float2 tx = txp[(size_t)sym * g.N], rx = rxp[(size_t)sym * g.N];
        float pr = rx.x * tx.x + rx.y * tx.y;
        float pi = rx.y * tx.x - rx.x * tx.y;
        acc.x = acc.x + pr;
        acc.y = acc.y + pi;
    }
    const int p = g.interleave ? (t * R + r) : (r * T + t);
    H[((size_t)f * T * R + p) * g.N + sc] = acc;
}

int launch_radar_chanest(jrc_ctx* ctx, int T, int R, const float2* d_frames, float2* d_H,
                         const ChanestGeom& g, int n_frames, hipStream_t stream)
{
    if (n_frames <= 0 || g.N <= 0) return JRC_OK;
    const bool aligned16 = (g.N % 2 == 0) && (g.port_stride % 2 == 0) && (g.frame_stride % 2 == 0) &&
                           ((reinterpret_cast<size_t>(d_frames) | reinterpret_cast<size_t>(d_H)) & 15) == 0;
    if (R <= 4 && (T == 1 || T == 2 || T == 4) && aligned16 && g.N >= 128 && !ctx->tune.chanest_x1) {
        // launched in chunks of two workgroups per CU: with every workgroup resident from the start the frame reads advance evenly; a
        // grid beyond one resident wave loses 10-15 % to its second, ragged wave of workgroups.  Two per CU against four: the same at
        // fft_len 256 (0.093 ms per 512 config-B frames), 10 % faster at fft_len 1024 (0.362-0.372 against 0.402 ms per 256 config-D
        // frames, 5.8 TB/s; any chunk that is not a whole number of workgroups per CU loses: 0.46-0.49 ms)
        const int wg_per_frame = (g.N / 2 + 63) / 64;
        int chunk = ctx->tune.chanest_chunk > 0 ? ctx->tune.chanest_chunk : (2 * ctx->n_cus) / wg_per_frame;
        if (chunk < 1) chunk = 1;
        const dim3 block(64, R, 1);
        for (int f0 = 0; f0 < n_frames; f0 += chunk) {
            const int nf = n_frames - f0 < chunk ? n_frames - f0 : chunk;
            const dim3 grid(wg_per_frame, nf, 1);
            const float2* fr = d_frames + (size_t)f0 * g.frame_stride;
            float2* Hc = d_H + (size_t)f0 * T * R * g.N;
            if (T == 1) hipLaunchKernelGGL((radar_chanest_x2_kernel<1, 8>), grid, block, 0, stream, fr, Hc, g, R);
            else if (T == 2) hipLaunchKernelGGL((radar_chanest_x2_kernel<2, 4>), grid, block, 0, stream, fr, Hc, g, R);
            else if (ctx->tune.chanest_u2) hipLaunchKernelGGL((radar_chanest_x2_kernel<4, 2>), grid, block, 0, stream, fr, Hc, g, R);
            else hipLaunchKernelGGL((radar_chanest_x2_kernel<4, 4>), grid, block, 0, stream, fr, Hc, g, R);
        }
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    if (R <= 4 && (T == 1 || T == 2 || T == 4)) {
        dim3 grid((g.N + 63) / 64, n_frames, 1), block(64, R, 1);
        if (T == 1) hipLaunchKernelGGL((radar_chanest_kernel<1, 8>), grid, block, 0, stream, d_frames, d_H, g, R);
        else if (T == 2) hipLaunchKernelGGL((radar_chanest_kernel<2, 8>), grid, block, 0, stream, d_frames, d_H, g, R);
        else hipLaunchKernelGGL((radar_chanest_kernel<4, 8>), grid, block, 0, stream, d_frames, d_H, g, R);
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    const int threads = g.N >= 256 ? 256 : (g.N <= 64 ? 64 : ((g.N + 63) / 64) * 64);
    dim3 grid((g.N + threads - 1) / threads, n_frames, T * R);
    hipLaunchKernelGGL(radar_chanest_generic_kernel, grid, dim3(threads), 0, stream, d_frames, d_H, g, T, R);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// A6 + A7 + A1 in one kernel: RX handed over in the TIME domain (the input of ofdm_cyclic_prefix_remover,
// lib/ofdm_cyclic_prefix_remover_impl.cc:69-99), each symbol transformed in LDS (the stock fft_vxx forward + shift of the
// flowgraph, examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc `fft_vxx_0_0`) and multiplied into the T accumulators of
// mimo_ofdm_radar (lib/mimo_ofdm_radar_impl.cc:250-274) straight from registers: the frequency-domain RX symbols never go
// to HBM (unfused: written once, read once), and the N_pre preamble symbols the radar block skips are not transformed at all.
//
// One (frame, receiver) stream per group of N threads (N >= 256: one stream per workgroup; smaller N: 256/N streams): the group
// transforms FOUR symbols at a time, n/4 lanes each, then every thread owns one subcarrier and adds the four products to its T
// accumulators in symbol order (the accumulation order of the reference).  The butterflies are the Stockham passes of
// fft_stockham_kernel (same helper, same twiddle table) and the sums run in the same order, so the result agrees with
// jrc_cp_remove_fft_dev + jrc_radar_chanest_dev to the last bits (1e-7 relative: only the compiler's choice of which product of a
// complex multiply it fuses into an FMA differs between the two kernels); the next four symbols' samples and this round's TX rows are in flight while the current four go through their LDS passes.
__device__ __forceinline__ void chanest_mac(float2& a, const float2 rx, const float2 tx)
{
#pragma clang fp contract(off)
    // rx * conj(tx) = (ac + bd) + j(bc - ad), products rounded individually (:273)
    const float pr = rx.x * tx.x + rx.y * tx.y, pi = rx.y * tx.x - rx.x * tx.y;
    a.x = a.x + pr; a.y = a.y + pi;
}

// LDS index hook: padding the Stockham scatter (runs of Ns points at stride 4*Ns) against bank conflicts — i + (i >> 2) — was measured
// and does not pay: the kernel is bound by instruction issue, not by LDS, and the two extra integer operations per access cost 7 %
#define DC_PAD(i) (i)

// SPR = symbols a stream's thread group transforms per round (n/4 lanes each): 4 -> n threads per stream and one subcarrier per thread in
// the accumulation; 2 -> n/2 threads and two subcarriers; 1 -> n/4 threads and four.  At fft_len 1024 the 4-symbol form is one 1024-thread
// workgroup with 72 KiB of LDS — a single resident workgroup per CU, so nothing runs while it sits in one of its six barriers per round
// (0.868 ms per 256 config-D frames, 2.5 TB/s).  The 1-symbol form is a 256-thread workgroup with 24 KiB and 112 registers: four per CU,
// whose barrier waits overlap (0.651 ms, 3.35 TB/s; 2 symbols: 0.707 ms).  At fft_len 256 the 4-symbol form already has 8 workgroups per CU
// and stays best (3.43 TB/s against 3.32 / 2.81).  Register budgets below the natural ones (more workgroups per CU) spill and lose.
template <int T, int SPR>
__global__ __launch_bounds__(SPR == 4 ? 1024 : (SPR == 2 ? 512 : 256), SPR == 4 ? 1 : (SPR == 2 ? 2 : 4)) void demod_chanest_kernel(const float2* __restrict__ tx, const float2* __restrict__ rx_td,
                                                                             float2* __restrict__ H, const float2* __restrict__ tw_g,
                                                                             DemodGeom g, int n_frames)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr int SPT = 4 / SPR;                               // subcarriers per thread in the accumulation
    const int n = g.N, tp = n >> 2, ns = SPR * tp;             // lanes per transform, threads per stream
    const int per_block = blockDim.x / ns;
    const int ls = threadIdx.x / ns, ts = threadIdx.x % ns;    // stream within the workgroup, thread within the stream
    const int u = ts / tp, lt = ts % tp;                       // symbol slot 0..SPR-1, lane within the transform
    float2* tw = lds;
    const int np = n;                                          // buffer length (see DC_PAD)
    float2* sbase = lds + n + (size_t)ls * (2 * SPR) * np;     // per stream: SPR slots x 2 buffers x np
    float2* bufA = sbase + (size_t)u * 2 * np;
    float2* bufB = bufA + np;
    for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];

    // workgroup -> streams; with several workgroups per frame, those of one frame share blockIdx % n_xcd (one XCD, so the TX rows
    // every receiver multiplies with are fetched into one L2)
    long blk = blockIdx.x;
    if (g.blocks_per_frame > 1) {
        const long q = blk / g.n_xcd, x = blk % g.n_xcd;
        blk = ((q / g.blocks_per_frame) * g.n_xcd + x) * g.blocks_per_frame + (q % g.blocks_per_frame);
    }
    const long b = blk * per_block + ls;
    const bool live = b < (long)n_frames * g.R;
    const long f = live ? b / g.R : 0;
    const int r = live ? (int)(b % g.R) : 0;
    const float2* src = rx_td + f * g.rx_frame_stride + (long)r * g.rx_stream_stride + (long)g.rx_sym0 * (n + g.cp) + g.cp;
    const float2* txb = tx + f * g.tx_frame_stride + (long)g.tx_item0 * n + ts;
    const int half_n = n >> 1;

    float2 acc[SPT][T], pn[4] = {};
#pragma unroll
    for (int j = 0; j < SPT; j++)
#pragma unroll
        for (int t = 0; t < T; t++) acc[j][t] = make_float2(0.f, 0.f);

    auto fetch_rx = [&](int sym0) {            // this slot's symbol of the round starting at sym0: read once, non-temporal
        typedef float v2f __attribute__((ext_vector_type(2)));
        if (sym0 + u < g.S) {
            const float2* sp = src + (long)(sym0 + u) * (n + g.cp);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(sp + lt + q * tp));
                pn[q] = make_float2(t.x, t.y);
            }
        }
    };
    if (live && g.S > 0) fetch_rx(0);
    __syncthreads();

    const bool odd = g.logn & 1;
    float2* wr = bufA;                 // buffer this slot's next pass writes
    for (int sym0 = 0; sym0 < g.S; sym0 += SPR) {
        float2 p[4], tc[SPT][T][SPR] = {};
#pragma unroll
        for (int q = 0; q < 4; q++) p[q] = pn[q];
        if (live && sym0 + SPR < g.S) fetch_rx(sym0 + SPR);  // next round's samples: in flight during this round's passes
        if (live) {                                          // this round's TX rows: needed only after the passes
#pragma unroll
            for (int q = 0; q < SPR; q++)
                if (sym0 + q < g.S) {
#pragma unroll
                    for (int j = 0; j < SPT; j++)
#pragma unroll
                        for (int t = 0; t < T; t++) tc[j][t][q] = txb[(long)t * g.tx_port_stride + (long)(sym0 + q) * n + j * ns];
                }
        }

        // first pass, from registers (Ns = 1: no twiddles); p[q] = x[lt + q*tp]
        int Ns;
        if (odd) {         // radix 2: j = lt pairs (p0, p2), j = lt + tp pairs (p1, p3); out[2j], out[2j+1]
            wr[DC_PAD(2 * lt)] = cadd(p[0], p[2]); wr[DC_PAD(2 * lt + 1)] = csub(p[0], p[2]);
            wr[DC_PAD(2 * (lt + tp))] = cadd(p[1], p[3]); wr[DC_PAD(2 * (lt + tp) + 1)] = csub(p[1], p[3]);
            Ns = 2;
        } else {           // radix 4: out[4j + r]
            const float2 a = cadd(p[0], p[2]), bb = csub(p[0], p[2]), c = cadd(p[1], p[3]), d = csub(p[1], p[3]);
            const float2 jd = make_float2(d.y, -d.x);
            wr[DC_PAD(4 * lt)] = cadd(a, c); wr[DC_PAD(4 * lt + 1)] = cadd(bb, jd);
            wr[DC_PAD(4 * lt + 2)] = csub(a, c); wr[DC_PAD(4 * lt + 3)] = csub(bb, jd);
            Ns = 4;
        }
        __syncthreads();
        float2* rd = wr;
        wr = (wr == bufA) ? bufB : bufA;
        while (Ns * 4 < n) {           // middle passes, LDS -> LDS: stockham_pass<4> of fft_device.h on the padded layout
            const int k = lt & (Ns - 1), tws = n / (Ns * 4);
            float2 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v[q] = rd[DC_PAD(lt + q * tp)];
                if (q && k) v[q] = cmul(v[q], tw[(k * q * tws) & (n - 1)]);
            }
            const float2 a = cadd(v[0], v[2]), bb = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
            const float2 jd = make_float2(d.y, -d.x);
            const int j0 = ((lt - k) << 2) + k;
            wr[DC_PAD(j0)] = cadd(a, c); wr[DC_PAD(j0 + Ns)] = cadd(bb, jd);
            wr[DC_PAD(j0 + 2 * Ns)] = csub(a, c); wr[DC_PAD(j0 + 3 * Ns)] = csub(bb, jd);
            __syncthreads();
            float2* t2 = rd; rd = wr; wr = t2;
            Ns *= 4;
        }
        // last pass (Ns = n/4, k = lt): X[lt + q*Ns] goes where fft_vxx's shift puts it, subcarrier (lt + q*Ns + n/2) mod n
        {
            float2 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v[q] = rd[DC_PAD(lt + q * tp)];
                if (q && lt) v[q] = cmul(v[q], tw[(lt * q) & (n - 1)]);
            }
            const float2 a = cadd(v[0], v[2]), bb = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
            const float2 jd = make_float2(d.y, -d.x);
            wr[DC_PAD((lt + half_n) & (n - 1))] = cadd(a, c);
            wr[DC_PAD((lt + tp + half_n) & (n - 1))] = cadd(bb, jd);
            wr[DC_PAD((lt + 2 * tp + half_n) & (n - 1))] = csub(a, c);
            wr[DC_PAD((lt + 3 * tp + half_n) & (n - 1))] = csub(bb, jd);
        }
        __syncthreads();
        // every thread owns subcarriers ts (+ ns): the symbols of this round, in order.  Slot q's result sits in the buffer that slot
        // just wrote (the same one of the pair for every slot); the next round's first pass writes the other one.
        const int off = (int)(wr - bufA);
#pragma unroll
        for (int q = 0; q < SPR; q++)
            if (sym0 + q < g.S) {
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    const float2 xv = sbase[(size_t)q * 2 * np + off + DC_PAD(ts + j * ns)];
#pragma unroll
                    for (int t = 0; t < T; t++) chanest_mac(acc[j][t], xv, tc[j][t][q]);
                }
            }
        wr = rd;
    }
    if (!live) return;
    float2* Hf = H + (size_t)f * T * g.R * n;
#pragma unroll
    for (int j = 0; j < SPT; j++)
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int pidx = g.interleave ? (t * g.R + r) : (r * T + t);   // :262-269
            Hf[(size_t)pidx * n + ts + j * ns] = acc[j][t];
        }
}

// ---- the same block for fft_len 256 and 1024 with radix-16 passes in registers -------------------------------------------------------
// demod_chanest_kernel moves a symbol through log4(n) radix-4 passes, LDS -> LDS with a workgroup barrier after each; what bounds it is
// instruction issue (~110 vector instructions per point at fft_len 1024).  Here one symbol belongs to n/16 lanes of ONE wavefront — 16
// points per lane — and goes through Stockham passes of radix 16, 16 (and 4 at fft_len 1024) with the 16-point transforms in registers
// (fft_fwd_small<16>): two (three) LDS exchanges per symbol, synchronised inside the wavefront only, then the workgroup meets once per
// round to add the round's symbols — 4096 / n of them — to the accumulators in symbol order, every thread owning n / 256 subcarriers as
// before.  A symbol's buffer is padded by one element per sixteen (the first pass writes at a stride of sixteen).  Same transform, same
// accumulation order; the factorisation differs from fft_stockham_kernel's, so the estimate agrees with the unfused path to rounding
// (1e-6 relative), as the radix-4 kernel's does.
template <int T, int LOGN>
// register budgets and TX batch sizes measured on configs B / D (256 frames x 4 streams, tools/td_kernel_probe.py): fft_len 256: 4 waves per
// SIMD, TX rows of 2 symbols per batch 0.134 ms per 512 frames (3: 0.137; batches of 8: 0.142, of 16: spills); fft_len 1024: 2 waves per SIMD,
// batches of 2 0.560 ms per 256 frames (of 4: 0.590 at 256 VGPRs; 3 waves per SIMD spill: 0.73-2.2 ms)
__global__ __launch_bounds__(256, LOGN == 10 ? 2 : 4) void demod_chanest16_kernel(const float2* __restrict__ tx, const float2* __restrict__ rx_td,
                                                                 float2* __restrict__ H, const float2* __restrict__ tw_g,
                                                                 DemodGeom g, int n_frames)
{
    constexpr int N = 1 << LOGN, LPS = N / 16, SPW = 64 / LPS, NSYM = 4 * SPW, SPT = N / 256, NP = N + N / 16;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2* tw = lds;                                          // [N] exp(-j 2 pi i / N)
    float2* buf = lds + N;                                     // [NSYM][NP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = lane % LPS, slot = wave * SPW + lane / LPS;  // lane within the symbol's transform, symbol slot of the round
    float2* my = buf + (size_t)slot * NP;
    auto ph = [](int i) { return i + (i >> 4); };
    for (int i = tid; i < N; i += 256) tw[i] = tw_g[i];

    // workgroup -> (frame, receiver); the receivers of a frame share blockIdx % n_xcd, so its TX rows are fetched into one L2
    long blk = blockIdx.x;
    {
        const long q = blk / g.n_xcd, x = blk % g.n_xcd;
        blk = ((q / g.R) * g.n_xcd + x) * g.R + (q % g.R);
    }
    if (blk >= (long)n_frames * g.R) return;
    const long f = blk / g.R;
    const int r = (int)(blk % g.R);
    const float2* src = rx_td + f * g.rx_frame_stride + (long)r * g.rx_stream_stride + (long)g.rx_sym0 * (N + g.cp) + g.cp + l;
    const float2* txb = tx + f * g.tx_frame_stride + (long)g.tx_item0 * N + tid;

    float2 acc[SPT][T], pn[16];
#pragma unroll
    for (int m = 0; m < SPT; m++)
#pragma unroll
        for (int t = 0; t < T; t++) acc[m][t] = make_float2(0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 16; q++) pn[q] = make_float2(0.f, 0.f);
    auto fetch_rx = [&](int sym0) {            // this slot's symbol of the round starting at sym0: read once, non-temporal
        typedef float v2f __attribute__((ext_vector_type(2)));
        if (sym0 + slot < g.S) {
            const float2* sp = src + (long)(sym0 + slot) * (N + g.cp);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(sp + q * LPS));
                pn[q] = make_float2(t.x, t.y);
            }
        }
    };
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    fetch_rx(0);
    __syncthreads();

    for (int sym0 = 0; sym0 < g.S; sym0 += NSYM) {
        float2 v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = pn[q];             // v[q] = x[l + q n/16]
        if (sym0 + NSYM < g.S) fetch_rx(sym0 + NSYM);          // next round's samples: in flight during this round's passes
        // pass 1 (radix 16, Ns = 1: no twiddles): out[16 l + q]
        fft_fwd_small<16>(v);
#pragma unroll
        for (int q = 0; q < 16; q++) my[ph(16 * l + q)] = v[q];
        wave_sync();
        // pass 2 (radix 16, Ns = 16): k = l mod 16, twiddles exp(-j 2 pi k q / 256), out[16 (l - k + q) + k]
        const int k = l & 15;
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = my[ph(l + q * LPS)];
#pragma unroll
        for (int q = 1; q < 16; q++) v[q] = cmul(v[q], tw[(k * q * (N / 256)) & (N - 1)]);
        fft_fwd_small<16>(v);
        JRC_LOCKSTEP();                                        // every lane has read its sixteen points before any lane overwrites the buffer
        if constexpr (LOGN == 8) {                             // done: X[l + 16 q] goes where fft_vxx's shift puts it
#pragma unroll
            for (int q = 0; q < 16; q++) my[ph((l + 16 * q + N / 2) & (N - 1))] = v[q];
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) my[ph(16 * ((l - k) + q) + k)] = v[q];
            wave_sync();
            // pass 3 (radix 4, Ns = 256): four butterflies per lane, j = l + 64 i: twiddles exp(-j 2 pi j q / 1024), X[j + 256 q], shifted
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = l + 64 * i;
#pragma unroll
                for (int q = 0; q < 4; q++) v[4 * i + q] = my[ph(j + 256 * q)];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = l + 64 * i;
                float2* a = v + 4 * i;
#pragma unroll
                for (int q = 1; q < 4; q++) a[q] = cmul(a[q], tw[(j * q) & (N - 1)]);
                const float2 s0 = cadd(a[0], a[2]), d0 = csub(a[0], a[2]), s1 = cadd(a[1], a[3]), d1 = csub(a[1], a[3]);
                const float2 jd = make_float2(d1.y, -d1.x);
                a[0] = cadd(s0, s1); a[1] = cadd(d0, jd); a[2] = csub(s0, s1); a[3] = csub(d0, jd);
            }
            JRC_LOCKSTEP();                                    // (as above: reads of the pass before its writes)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int q = 0; q < 4; q++) my[ph((l + 64 * i + 256 * q + N / 2) & (N - 1))] = v[4 * i + q];
        }
        // every thread owns subcarriers tid + 256 m: the symbols of this round, in order (mimo_ofdm_radar_impl.cc:250-274).  The TX rows
        // of TXD symbols are requested together — the first batch before the workgroup meets, so the barrier wait and the L2 round trip
        // overlap — and multiplied in once they are all here: one exposed round trip per batch instead of one per symbol.
        constexpr int TXD = 2;
        float2 tc[TXD][SPT][T];
        auto fetch_tx = [&](int q0) {
#pragma unroll
            for (int q = 0; q < TXD; q++)
                if (sym0 + q0 + q < g.S) {
#pragma unroll
                    for (int m = 0; m < SPT; m++)
#pragma unroll
                        for (int t = 0; t < T; t++) tc[q][m][t] = txb[(long)t * g.tx_port_stride + (long)(sym0 + q0 + q) * N + 256 * m];
                }
        };
        fetch_tx(0);
        __syncthreads();
#pragma unroll
        for (int q0 = 0; q0 < NSYM; q0 += TXD) {
            if (q0) fetch_tx(q0);
#pragma unroll
            for (int q = 0; q < TXD; q++)
                if (sym0 + q0 + q < g.S) {
#pragma unroll
                    for (int m = 0; m < SPT; m++) {
                        const float2 xv = buf[(size_t)(q0 + q) * NP + ph(tid + 256 * m)];
#pragma unroll
                        for (int t = 0; t < T; t++) chanest_mac(acc[m][t], xv, tc[q][m][t]);
                    }
                }
        }
        __syncthreads();
    }
    float2* Hf = H + (size_t)f * T * g.R * N;
#pragma unroll
    for (int m = 0; m < SPT; m++)
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int pidx = g.interleave ? (t * g.R + r) : (r * T + t);   // :262-269
            Hf[(size_t)pidx * N + tid + 256 * m] = acc[m][t];
        }
}

template <int T, int LOGN>
static int launch_demod_chanest16(jrc_ctx* ctx, const float2* d_tx, const float2* d_rx_td, float2* d_H, const float2* tw, DemodGeom g, int n_frames,
                                  hipStream_t stream)
{
    constexpr int N = 1 << LOGN, NSYM = 4096 / N;
    const long grp = (long)g.n_xcd * g.R;
    const size_t lds_bytes = sizeof(float2) * ((size_t)N + (size_t)NSYM * (N + N / 16));
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)demod_chanest16_kernel<T, LOGN>, lds_bytes));
    // one launch for the whole batch: chunks of 1 ... 8 workgroups per CU were measured and are the same or slower
    const long blocks = (((long)n_frames * g.R + grp - 1) / grp) * grp;          // whole frames per group of R workgroups of an XCD
    hipLaunchKernelGGL((demod_chanest16_kernel<T, LOGN>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, d_tx, d_rx_td, d_H, tw, g, n_frames);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

bool demod_chanest_supported(int N, int T)
{
    return jrc_is_pow2(N) && N >= 16 && N <= 1024 && (T == 1 || T == 2 || T == 3 || T == 4 || T == 8);
}

template <int T, int SPR>
static int launch_demod_chanest_t(jrc_ctx* ctx, const float2* d_tx, const float2* d_rx_td, float2* d_H, const float2* tw, DemodGeom g, int n_frames,
                                  hipStream_t stream)
{
    const int ns = SPR * (g.N / 4);                                             // threads per stream
    const int threads = ns >= 256 ? ns : 256, per_block = threads / ns;         // streams per workgroup
    g.blocks_per_frame = per_block < g.R ? (g.R + per_block - 1) / per_block : 1;
    long blocks = ((long)n_frames * g.R + per_block - 1) / per_block;
    if (g.blocks_per_frame > 1) {
        // a frame's receivers must not straddle the remapped groups: whole frames per group of blocks_per_frame workgroups
        if (g.R % per_block) g.blocks_per_frame = 1;
        else { const long grp = (long)g.n_xcd * g.blocks_per_frame; blocks = (blocks + grp - 1) / grp * grp; }
    }
    const size_t lds_bytes = sizeof(float2) * ((size_t)g.N + (size_t)per_block * (2 * SPR) * g.N);
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)demod_chanest_kernel<T, SPR>, lds_bytes));
    hipLaunchKernelGGL((demod_chanest_kernel<T, SPR>), dim3((unsigned)blocks), dim3((unsigned)threads), lds_bytes, stream, d_tx, d_rx_td, d_H, tw, g, n_frames);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

int launch_demod_chanest(jrc_ctx* ctx, int T, const float2* d_tx, const float2* d_rx_td, float2* d_H, DemodGeom g, int n_frames,
                         hipStream_t stream)
{
    if (n_frames <= 0) return JRC_OK;
    if (!demod_chanest_supported(g.N, T))
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "time-domain channel estimate: fft_len %d / N_tx %d outside the fused kernel "
                        "(powers of two 16..1024, N_tx in {1,2,3,4,8}); use jrc_cp_remove_fft_dev + jrc_radar_chanest_dev", g.N, T);
    const float2* tw = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, g.N, -1, &tw));
    g.logn = jrc_ilog2(g.N);
    g.n_xcd = ctx->n_xcd;
    // fewer symbols per round (fewer threads per stream, more subcarriers per thread) as fft_len grows: more, smaller workgroups per CU;
    // JRC_DEMOD_SPR overrides
    if ((g.N == 256 || g.N == 1024) && (T == 1 || T == 2 || T == 4) && ctx->tune.demod_spr != 1 && ctx->tune.demod_spr != 2 && ctx->tune.demod_spr != 4) {
        // radix-16 passes in registers (JRC_DEMOD_SPR = 1 / 2 / 4 selects the radix-4 kernel instead)
        if (g.N == 256) {
            if (T == 1) return launch_demod_chanest16<1, 8>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
            if (T == 2) return launch_demod_chanest16<2, 8>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
            return launch_demod_chanest16<4, 8>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
        }
        if (T == 1) return launch_demod_chanest16<1, 10>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
        if (T == 2) return launch_demod_chanest16<2, 10>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
        return launch_demod_chanest16<4, 10>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
    }
    int spr = g.N >= 1024 ? 1 : (g.N >= 512 ? 2 : 4);
    if (ctx->tune.demod_spr == 1 || ctx->tune.demod_spr == 2 || ctx->tune.demod_spr == 4) spr = ctx->tune.demod_spr;
    if (g.N < 8) spr = 4;
#define JRC_DEMOD_CASE(TT)                                                                                               \
    case TT: return spr == 1 ? launch_demod_chanest_t<TT, 1>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream)          \
                  : spr == 2 ? launch_demod_chanest_t<TT, 2>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream)          \
                             : launch_demod_chanest_t<TT, 4>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
    switch (T) {
        JRC_DEMOD_CASE(1)
        JRC_DEMOD_CASE(2)
        JRC_DEMOD_CASE(3)
        JRC_DEMOD_CASE(4)
        default: break;
    }
    return spr == 2 ? launch_demod_chanest_t<8, 2>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream)
                    : launch_demod_chanest_t<8, 4>(ctx, d_tx, d_rx_td, d_H, tw, g, n_frames, stream);
#undef JRC_DEMOD_CASE
}

// ------------------------------------------------------------------------------------------------
// background recording / removal (lib/mimo_ofdm_radar_impl.cc:276-293)
__global__ void radar_background_kernel(float2* __restrict__ est, float2* __restrict__ temp,
                                        const float2* __restrict__ ring, int pn, int ring_size,
                                        int ring_head, int record_len, int recording, int removal)
{
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pn) return;
    float2 e = est[idx];
    if (recording) temp[idx] = e;
    if (removal) {
        float2 m = make_float2(0.f, 0.f);
        const float n = (float)ring_size;
        for (int i = 0; i < ring_size; i++) {
            const int slot = (ring_head + i) % record_len;
            float2 v = ring[(size_t)slot * pn + idx];
            m.x = m.x + v.x / n;      // complex / float divides each component (:289)
            m.y = m.y + v.y / n;
        }
        est[idx] = make_float2(e.x - m.x, e.y - m.y);
    }
}

// ------------------------------------------------------------------------------------------------
struct jrc_radar {
    jrc_ctx* ctx;
    int N, T, R, S, Npre, Ir, interleave;
    int bg_removal, bg_recording, record_len;
    float2* d_in = nullptr;    // [T+R][S][N]
    float2* d_est = nullptr;   // [P][N]
    float2* d_temp = nullptr;  // radar_chan_est_temp
    float2* d_ring = nullptr;  // [record_len][P*N]
    int ring_size = 0, ring_head = 0;
};

extern "C" int jrc_radar_create(jrc_ctx* ctx, int fft_len, int N_tx, int N_rx, int N_sym, int N_pre,
                                int background_removal, int background_recording, int record_len,
                                int interp_factor, int enable_tx_interleave, jrc_radar** out)
{
    if (!ctx || !out) return JRC_ERR_INVALID_ARG;
    if (fft_len <= 0 || N_tx <= 0 || N_rx <= 0 || N_sym < 0 || N_pre < 0 || interp_factor <= 0 || record_len < 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "mimo_ofdm_radar: non-positive size parameter");
    jrc_radar* r = new jrc_radar();
    r->ctx = ctx;
    r->N = fft_len; r->T = N_tx; r->R = N_rx; r->S = N_sym; r->Npre = N_pre; r->Ir = interp_factor;
    r->interleave = enable_tx_interleave;
    r->bg_removal = background_removal; r->bg_recording = background_recording; r->record_len = record_len;
    const size_t pn = (size_t)N_tx * N_rx * fft_len;
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_in, sizeof(float2) * (size_t)(N_tx + N_rx) * (N_sym ? N_sym : 1) * fft_len);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_est, sizeof(float2) * pn);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_temp, sizeof(float2) * pn);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_ring, sizeof(float2) * pn * (record_len ? record_len : 1));
    if (e == hipSuccess) e = hipMemset(r->d_temp, 0, sizeof(float2) * pn);   // vector::resize value-initialises (:115)
    // hipMemset runs on the null stream and may return before it has run; the work below is queued on the context's NON-BLOCKING stream, which the
    // null stream does not order itself against: finish it here (found by the radar block's packet-sequence fuzz under load)
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        jrc_radar_destroy(r);
        return jrc_fail(ctx, JRC_ERR_HIP, "jrc_radar_create: %s", hipGetErrorString(e));
    }
    *out = r;
    return JRC_OK;
}

extern "C" void jrc_radar_destroy(jrc_radar* r)
{
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    (void)hipStreamSynchronize(r->ctx->stream);
    if (r->d_in) (void)hipFree(r->d_in);
    if (r->d_est) (void)hipFree(r->d_est);
    if (r->d_temp) (void)hipFree(r->d_temp);
    if (r->d_ring) (void)hipFree(r->d_ring);
    delete r;
}

extern "C" int jrc_radar_set_background_record(jrc_radar* r, int on)
{
    if (!r) return JRC_ERR_INVALID_ARG;
    r->bg_recording = on;
    return JRC_OK;
}

extern "C" int jrc_radar_ring_size(const jrc_radar* r) { return r ? r->ring_size : JRC_ERR_INVALID_ARG; }

extern "C" int jrc_radar_work(jrc_radar* r, const jrc_cf32* const* tx, const jrc_cf32* const* rx,
                              size_t n_items_tx, size_t n_items_rx, size_t tx_discard, jrc_cf32* out)
{
    if (!r || !tx || !rx || !out) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = r->ctx;
    const int N = r->N, T = r->T, R = r->R, S = r->S, P = T * R;
    // The reference reads N_pre+N_sym items (+ discard on TX) without checking (:250-274); fail loudly instead.
    if (n_items_rx < (size_t)(r->Npre + S) || n_items_tx < tx_discard + (size_t)(r->Npre + S))
        return jrc_fail(ctx, JRC_ERR_SHORT_INPUT, "mimo_ofdm_radar: need %d items (+%zu discard) per port, got tx=%zu rx=%zu",
                        r->Npre + S, tx_discard, n_items_tx, n_items_rx);
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t port_items = (size_t)S * N;
    const size_t in_bytes = sizeof(float2) * (size_t)(T + R) * port_items;
    const size_t est_bytes = sizeof(float2) * (size_t)P * N;
    JRC_TRY(jrc_ensure_pinned(ctx, in_bytes + est_bytes));
    float2* h_in = (float2*)ctx->pinned;
    float2* h_est = (float2*)((char*)ctx->pinned + in_bytes);
    for (int t = 0; t < T; t++)
        memcpy(h_in + (size_t)t * port_items, (const float2*)tx[t] + ((size_t)r->Npre + tx_discard) * N,
               sizeof(float2) * port_items);
    for (int q = 0; q < R; q++)
        memcpy(h_in + (size_t)(T + q) * port_items, (const float2*)rx[q] + (size_t)r->Npre * N,
               sizeof(float2) * port_items);
    if (in_bytes) JRC_HIP(ctx, hipMemcpyAsync(r->d_in, h_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));

    ChanestGeom g;
    g.N = N; g.S = S; g.frame_stride = (long)(T + R) * (long)port_items; g.port_stride = (long)port_items;
    g.tx_item0 = 0; g.rx_item0 = 0; g.interleave = r->interleave;
    JRC_TRY(launch_radar_chanest(ctx, T, R, r->d_in, r->d_est, g, 1, ctx->stream));

    const int pn = P * N;
    if (r->bg_recording || r->bg_removal) {
        hipLaunchKernelGGL(radar_background_kernel, dim3((pn + 255) / 256), dim3(256), 0, ctx->stream, r->d_est,
                           r->d_temp, r->d_ring, pn, r->ring_size, r->ring_head, r->record_len ? r->record_len : 1,
                           r->bg_recording, r->bg_removal);
        JRC_HIP(ctx, hipGetLastError());
    }
    if (r->bg_removal && r->record_len > 0) {   // :297-300 circular_buffer::push_back
        int slot;
        if (r->ring_size < r->record_len) {
            slot = (r->ring_head + r->ring_size) % r->record_len;
            r->ring_size++;
        } else {
            slot = r->ring_head;
            r->ring_head = (r->ring_head + 1) % r->record_len;
        }
        JRC_HIP(ctx, hipMemcpyAsync(r->d_ring + (size_t)slot * pn, r->d_temp, sizeof(float2) * pn,
                                    hipMemcpyDeviceToDevice, ctx->stream));
    }
    JRC_HIP(ctx, hipMemcpyAsync(h_est, r->d_est, est_bytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // zero-padded rows on the host side of the PCIe link (:243, :312-315): only P*N values cross it
    memset(out, 0, sizeof(float2) * (size_t)P * N * r->Ir);
    for (int p = 0; p < P; p++)
        memcpy((float2*)out + (size_t)p * N * r->Ir, h_est + (size_t)p * N, sizeof(float2) * N);
    return P;
}

// batched, device-resident A1 alone (the first stage of jrc_chain_run_dev), for callers that keep their own pipeline
extern "C" int jrc_radar_chanest_dev(jrc_ctx* ctx, int fft_len, int N_tx, int N_rx, int N_sym, int N_pre, int n_items,
                                     int enable_tx_interleave, int n_frames, const jrc_cf32* d_frames, jrc_cf32* d_chanest,
                                     void* stream)
{
    if (!ctx || !d_frames || !d_chanest) return JRC_ERR_INVALID_ARG;
    if (fft_len <= 0 || N_tx <= 0 || N_rx <= 0 || N_sym < 0 || N_pre < 0 || n_items < N_pre + N_sym || n_frames < 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_radar_chanest_dev: inconsistent sizes");
    JRC_BIND(ctx);
    ChanestGeom g;
    g.N = fft_len; g.S = N_sym; g.port_stride = (long)n_items * fft_len; g.frame_stride = g.port_stride * (N_tx + N_rx);
    g.tx_item0 = N_pre; g.rx_item0 = N_pre; g.interleave = enable_tx_interleave;
    return launch_radar_chanest(ctx, N_tx, N_rx, (const float2*)d_frames, (float2*)d_chanest, g, n_frames,
                                stream ? (hipStream_t)stream : ctx->stream);
}

extern "C" int jrc_radar_chanest_td_dev(jrc_ctx* ctx, int fft_len, int cp_len, int N_tx, int N_rx, int N_sym, int N_pre, int n_items,
                                        long rx_stream_len, int enable_tx_interleave, int n_frames, const jrc_cf32* d_tx,
                                        const jrc_cf32* d_rx_td, jrc_cf32* d_chanest, void* stream)
{
    JRC_TRACE("jrc_radar_chanest_td_dev");
    if (!ctx || !d_tx || !d_rx_td || !d_chanest) return JRC_ERR_INVALID_ARG;
    if (fft_len <= 0 || cp_len < 0 || N_tx <= 0 || N_rx <= 0 || N_sym < 0 || N_pre < 0 || n_items < N_pre + N_sym || n_frames < 0 ||
        rx_stream_len < (long)n_items * (fft_len + cp_len))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_radar_chanest_td_dev: inconsistent sizes");
    JRC_BIND(ctx);
    DemodGeom g;
    g.N = fft_len; g.cp = cp_len; g.S = N_sym; g.R = N_rx; g.logn = 0;
    g.tx_port_stride = (long)n_items * fft_len; g.tx_frame_stride = g.tx_port_stride * N_tx;
    g.rx_stream_stride = rx_stream_len; g.rx_frame_stride = rx_stream_len * N_rx;
    g.tx_item0 = N_pre; g.rx_sym0 = N_pre; g.interleave = enable_tx_interleave; g.blocks_per_frame = 1;
    return launch_demod_chanest(ctx, N_tx, (const float2*)d_tx, (const float2*)d_rx_td, (float2*)d_chanest, g, n_frames,
                                stream ? (hipStream_t)stream : ctx->stream);
}
