// chain.hip — fused, device-resident radar chain A1 -> A2 -> A3 -> A4 -> A5 over a batch of frames
//
// Replaces, in one launch sequence, the reference's
//   mimo_ofdm_radar (lib/mimo_ofdm_radar_impl.cc:131-340)
//   -> fft_vxx reverse/no-shift, size N*Ir      (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:940-962)
//   -> matrix_transpose (lib/matrix_transpose_impl.cc:69-110)
//   -> fft_vxx forward/shift, size P*Ia         (...radar_sim.grc:963-985)
//   -> range_angle_estimator (lib/range_angle_estimator_impl.cc:121-284)
//
// range_angle_fused_kernel / range_angle_wide_kernel (configs B and D) are the roofline kernels: they read the P x N channel estimate (L2 resident)
// and stream the (N*Ir) x (P*Ia) complex map to HBM exactly once, doing both zero-padded FFTs and the
// estimator's arg-max scan on chip.
//
//   Range axis.  R[p][k] = sum_{n<N} H[p][n] e^{+j2pi nk/NR}, NR = N*Ir, is needed only through its
//   N non-zero inputs.  A workgroup owns the residue class k = C*q + c (C = NR/64, q < 64):
//       R[p][C q + c] = IFFT_64( g_c[p] )[q],   g_c[p][n'] = sum_m H[p][n'+64m] e^{+j2pi (n'+64m) c / NR}
//   i.e. a twiddled fold of the N inputs down to 64 points followed by one 64-point transform per
//   virtual-array pair, done by one wavefront with one point per lane (cross-lane shuffles, no LDS).
//   Angle axis.  For each of the workgroup's 64 range bins the P inputs x[p] are zero-padded to
//   NA = P*Ia: out[Ia u + r] = FFT_P( x[p] e^{-j2pi p r / NA} )[u].  One lane owns one residue r (its P-1
//   twiddles live in registers) and computes one P-point FFT in registers; the Ia = 16 lanes of a range
//   bin cover one full 128-byte line of the map row per store, u after u.  fftshift is a rotation of u.
//   The transpose + zero padding of matrix_transpose never touches memory.
#include "radar_kernels.h"
#include "fft_device.h"

#include <cmath>
#include <cstdlib>

#define RA_GROUP 8   // map stores a wave issues back to back; then it waits until all but one have completed (docs/history.md §3.1)
#define RA_L 64   // range bins (and fold length) per workgroup

// ---- the fused kernel ------------------------------------------------------------------------
// Grid: one workgroup per (frame, slice); a slice owns the residue classes c = slice + WPF*i, i < C/WPF.
// The frame's channel estimate H (P x N) is staged in LDS once per workgroup and reused for every class;
// the class twiddles of the NEXT class are prefetched into registers while the current class is being
// stored, so the only exposed global-memory latency is the one H fetch per workgroup.
// LDS = P*N*8 (H) + P*64*8 (range bins of the current class) [+ N*8 class twiddles for N > 256]: 40 KiB for config B ->
// 256-thread workgroups, two resident per CU (more, shorter-lived ones measured slower); 144 KiB for config D -> one
// 512-thread workgroup per CU.
//
// MODE 0: the map is stored and scanned (the roofline kernel).
// MODE 1, "detect only" (jrc_chain_set_write_map(chain, 0)): the same transforms and the same arg-max on the values in registers,
//         but the map is never stored — a consumer that only takes range_angle_estimator's message (lib/range_angle_estimator_impl.cc:
//         234-253) does not pay the 4-16 MiB per frame.
//         Both map-less modes (1 and 3) also write the range profiles they pass through LDS — s_g, the 64 range bins of the class for every
//         pair, 8 KiB per class, 6 % of the map's bytes — to a buffer rng[frame][class][pair][64]: ra_window_rows_kernel (below) re-computes
//         the rows of the estimator's noise window (range bins [peak + NR/2 - dr, peak + NR/2 + dr), :197-226) from them with the same
//         angle-axis code (same twiddles, cmul_pin, fft_fwd_small_pin), as full rows of a compact buffer win[frame][2 dr][NA] that
//         ra_finalize_kernel reads instead of the map: same instructions on the same inputs, so the noise sum, and with it every field of
//         the result, is bit-identical to MODE 0.
// MODE 3, power map (jrc_chain_set_map_format(chain, JRC_MAP_POWER)): MODE 1 plus the map as float |z|^2 — what the flowgraph's display
//         branch consumes (blocks_complex_to_mag_squared -> gui_heatmap_plot, examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:2192) —
//         at half the bytes of the complex map.  A lane's 4-byte cells would make 64-byte store segments, so each wave turns its cells
//         through a private LDS tile ([rows of this trip][NA] floats) and stores whole 16-byte pieces of full rows; the estimator reads
//         the MODE 2 window rows as in detect-only mode.  ROWS1: the tile holds one row at a time (when LDS is short: fft_len 1024).
// IA: interp_angle known at compile time (16, the flowgraphs' interp_factor_angle), or 0 for the runtime argument
template <int P, int NT, int MMAX, bool TWC_LDS, int MODE, int IA, bool ROWS1 = false>
#ifndef JRC_WPS256
#define JRC_WPS256 3
#endif
__global__ __launch_bounds__(NT, (NT == 1024 ? 4 : (NT == 512 ? 2 : JRC_WPS256))) void range_angle_fused_kernel(
    const float2* __restrict__ H,        // [F][P][N]
    float2* __restrict__ map,            // MODE 0: [F][NR][NA]; MODE 3: float [F][NR][NA]; MODE 1: unused
    PeakPartial* __restrict__ partials,  // [F][pstride]
    const float2* __restrict__ twR,      // [NR]  exp(+j 2 pi i / NR)
    const float2* __restrict__ twA,      // [NA]  exp(-j 2 pi i / NA)
    int N, int NR, int Ia_arg, int F, int WPF,
    int pstride,                         // partial maxima per frame in `partials` (>= WPF; unused slots hold the neutral element)
    float2* __restrict__ rng_out,        // MODE 1 / 3: [F][C][P][64] range profiles for the window pass
    int nx,                              // XCDs the hardware deals consecutive workgroups over (jrc_ctx::n_xcd)
    int pace)                            // MODE 0 store pacing: bits 0-11 ticks of 10 ns between a wave's groups of eight stores (0 = off), 12-15 groups it may catch up, 16-17 wave priorities
{
#pragma clang fp contract(off)          // every rounding of this kernel is spelled out (fmaf / cmul_pin / fft_fwd_small_pin): the three MODEs agree bit for bit
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    constexpr int NW = NT / 64;
    const int Ia = IA > 0 ? IA : Ia_arg;
    const int NA = P * Ia;
    const int C = NR / RA_L;
    // XCD-aware decode: block b runs on XCD b % nx; the slices of a frame share that frame's H,
    // so keep a frame's workgroups on one XCD (one L2).
    const int xcd = blockIdx.x % nx;
    const int j = blockIdx.x / nx;
    const int f = (j / WPF) * nx + xcd;
    const int slice = j % WPF;
    if (f >= F) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* s_H = smem;                         // [P][N]
    float2* s_g = s_H + (size_t)P * N;          // [P][64]; reused as reduction scratch at the end
    float2* s_twc = s_g + P * RA_L;             // [N] class twiddles (TWC_LDS only)
    constexpr int NPT = TWC_LDS ? 4 : 1;        // class twiddles prefetched per thread (N <= NT*NPT)
    const int M = N / RA_L;                     // fold length per lane, <= MMAX

    const int n_iter = (C - slice + WPF - 1) / WPF;
    auto class_of = [&](int it) -> int { return slice + it * WPF; };     // the class this workgroup works on in its it-th trip
    const int c_first = class_of(0);

    // class twiddles for this lane's fold inputs n = lane + 64 m:  exp(+j 2 pi n c / NR)
    // (large N: staged through LDS one class ahead instead of living in registers across the store phase)
    float2 tc[TWC_LDS ? 1 : MMAX];
    float2 tn[NPT];
    if constexpr (TWC_LDS) {
#pragma unroll
        for (int q = 0; q < NPT; q++) { const int n = tid + NT * q; if (n < N) tn[q] = twR[(n * c_first) & (NR - 1)]; }
    } else {
#pragma unroll
        for (int m = 0; m < MMAX; m++)
            if (m < M) tc[m] = twR[((lane + RA_L * m) * c_first) & (NR - 1)];     // n*c < 2^31
    }
    {
        // stage H: 16-byte loads, fully coalesced
        const float4* Hf4 = reinterpret_cast<const float4*>(H + (size_t)f * P * N);
        float4* sH4 = reinterpret_cast<float4*>(s_H);
        for (int i = tid; i < (P * N) / 2; i += NT) sH4[i] = Hf4[i];
    }
    // 64-point inverse FFT twiddles of this lane, one per radix-2 stage: exp(+j 2 pi k / (2 half))
    float2 t64[6];
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int half = 32 >> st;
        t64[st] = twR[((lane & (half - 1)) * (32 / half)) * (NR / 64)];
    }
    // angle twiddles of this lane's residue r = tid % Ia:  exp(-j 2 pi p r / NA)
    const int r = tid % Ia;
    float2 ta[P];
#pragma unroll
    for (int p = 1; p < P; p++) ta[p] = twA[(p * r) & (NA - 1)];

    PeakTracker trk;
    trk.init();
    const int items = RA_L * Ia;                // (range bin, residue) pairs per class; a multiple of 64
    const int ahalf = NA >> 1, amask = NA - 1;
    float2* mapf = map + (size_t)f * NR * NA;
    float* mapp = reinterpret_cast<float*>(map) + (size_t)f * NR * NA;          // MODE 3
    float* s_pw = reinterpret_cast<float*>(s_twc + (TWC_LDS ? N : 0)) + (size_t)wave * (ROWS1 ? NA : RA_L * P);   // MODE 3: this wave's tile

    typedef float v2f __attribute__((ext_vector_type(2)));
    // store pacing (MODE 0, docs/history.md §3.1): release time of the wave's next group of stores, in ticks of the 100 MHz real-time counter
    const int pace_T = pace & 0xfff, pace_K = (pace >> 12) & 0xf;
    long long t_next = 0;
    if constexpr (MODE == 0 && IA > 0) {
        if (pace_T) t_next = (long long)wall_clock64();
        if ((pace >> 16) & 3) {            // the waves that share a SIMD get different priorities, so that one computes while the other waits on its stores
            const bool first = NT >= 512 ? (wave < NW / 2) : (((j >> 5) & 1) == 0);
            if (first) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        }
    }
#pragma unroll 1
    for (int it = 0; it < n_iter; it++) {
        const int c = class_of(it);
        if constexpr (TWC_LDS) {
#pragma unroll
            for (int q = 0; q < NPT; q++) { const int n = tid + NT * q; if (n < N) s_twc[n] = tn[q]; }
        }
        __syncthreads();                        // s_H staged (first trip) / previous class's s_g reads done
        // ---- range axis: fold to 64 points, 64-point inverse FFT across the wavefront -------------
        {   // the wave's pairs p = wave + NW j side by side: they share the class twiddle of a fold term, and their butterfly chains interleave
            constexpr int PPW = (P + NW - 1) / NW;
            constexpr bool full = (P % NW) == 0;    // every wave has PPW pairs
            float2 v[PPW];
#pragma unroll
            for (int j = 0; j < PPW; j++) v[j] = make_float2(0.f, 0.f);
#pragma unroll
            for (int m = 0; m < MMAX; m++)
                if (m < M) {
                    float2 w;
                    if constexpr (TWC_LDS) w = s_twc[lane + RA_L * m]; else w = tc[m];
#pragma unroll
                    for (int j = 0; j < PPW; j++)
                        if (full || wave + NW * j < P) {
                            const float2 h = s_H[(size_t)(wave + NW * j) * N + lane + RA_L * m];
                            v[j].x = fmaf(h.x, w.x, fmaf(-h.y, w.y, v[j].x));
                            v[j].y = fmaf(h.x, w.y, fmaf(h.y, w.x, v[j].y));
                        }
                }
            // radix-2 DIF across the wavefront, stage st pairs lanes that differ in bit 5 - st: v + o below, (o - v) t above
#pragma unroll
            for (int j = 0; j < PPW; j++) v[j] = wave_dif_stage<5>(v[j], t64[0], lane);
#pragma unroll
            for (int j = 0; j < PPW; j++) v[j] = wave_dif_stage<4>(v[j], t64[1], lane);
#pragma unroll
            for (int j = 0; j < PPW; j++) v[j] = wave_dif_stage<3>(v[j], t64[2], lane);
#pragma unroll
            for (int j = 0; j < PPW; j++) v[j] = wave_dif_stage<2>(v[j], t64[3], lane);
#pragma unroll
            for (int j = 0; j < PPW; j++) v[j] = wave_dif_stage<1>(v[j], t64[4], lane);
#pragma unroll
            for (int j = 0; j < PPW; j++) v[j] = wave_dif_stage<0>(v[j], t64[5], lane);
#pragma unroll
            for (int j = 0; j < PPW; j++)
                if (full || wave + NW * j < P) s_g[(wave + NW * j) * RA_L + (__brev((unsigned)lane) >> 26)] = v[j];   // lane holds X[bitrev6(lane)]
        }
        if (it + 1 < n_iter) {                  // prefetch the next class's twiddles; they land during the stores
            const int cn = class_of(it + 1);
            if constexpr (TWC_LDS) {
#pragma unroll
                for (int q = 0; q < NPT; q++) { const int n = tid + NT * q; if (n < N) tn[q] = twR[(n * cn) & (NR - 1)]; }
            } else {
#pragma unroll
                for (int m = 0; m < MMAX; m++)
                    if (m < M) tc[m] = twR[((lane + RA_L * m) * cn) & (NR - 1)];
            }
        }
        __syncthreads();

        if constexpr (MODE == 1 || MODE == 3) {      // the class's range profiles for the estimator's window pass: 8 KiB, coalesced 16-byte pieces
            float4* dst = reinterpret_cast<float4*>(rng_out + ((size_t)f * C + c) * (P * RA_L));
            const float4* src = reinterpret_cast<const float4*>(s_g);
            for (int i = tid; i < (P * RA_L) / 2; i += NT) dst[i] = src[i];
        }
        // ---- angle axis + fftshift + store + arg-max ---------------------------------------------
#pragma unroll 1
        for (int w0 = 0; w0 < items; w0 += NT) {     // whole waves are in or out
            const int w = w0 + tid;
            if (w >= items) break;
            const int ql = w / Ia;               // (w % Ia == r because Ia divides NT)
            const int k = C * ql + c;            // global range bin
            float2 y[P];
            y[0] = s_g[ql];
#pragma unroll
            for (int p = 1; p < P; p++) y[p] = cmul_pin(s_g[p * RA_L + ql], ta[p]);
            fft_fwd_small_pin<P>(y);
            float m = -1.0f;
#pragma unroll
            for (int u = 0; u < P; u++) m = fmaxf(m, fast_power(y[u]));
            if constexpr (MODE == 3) {
                // |z|^2 as blocks_complex_to_mag_squared computes it (volk_32fc_magnitude_squared_32f, generic: re*re + im*im, unfused)
                const int rw = 64 / Ia;                       // range rows of this wave in this trip: ql0 .. ql0 + rw - 1
                const int rb = lane / Ia;                     // this lane's row among them
                const int ql0 = ql - rb;
                typedef float v4f __attribute__((ext_vector_type(4)));
                if constexpr (!ROWS1) {
                    float* trow = s_pw + rb * NA;
#pragma unroll
                    for (int u = 0; u < P; u++) trow[(Ia * u + r + ahalf) & amask] = y[u].x * y[u].x + y[u].y * y[u].y;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int n4 = rw * NA / 4;               // 16-byte pieces in the tile = 16 P
                    for (int i = lane; i < n4; i += 64) {
                        const int row = (4 * i) / NA, col = (4 * i) % NA;
                        const float4 v = *reinterpret_cast<const float4*>(s_pw + 4 * i);
                        const v4f t = {v.x, v.y, v.z, v.w};
                        __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(mapp + (size_t)(C * (ql0 + row) + c) * NA + col));
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                    for (int row = 0; row < rw; row++) {
                        if (rb == row) {
#pragma unroll
                            for (int u = 0; u < P; u++) s_pw[(Ia * u + r + ahalf) & amask] = y[u].x * y[u].x + y[u].y * y[u].y;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        for (int i = lane; i < NA / 4; i += 64) {
                            const float4 v = *reinterpret_cast<const float4*>(s_pw + 4 * i);
                            const v4f t = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(mapp + (size_t)(C * (ql0 + row) + c) * NA + 4 * i));
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if constexpr (MODE == 0) {
                // The map is write-once data nothing on the GPU reads back except the estimator's few cells: it is stored non-temporally, so
                // that no dirty lines are left behind for the next (read-bound) kernel to compete with (docs/history.md §3.1).
                // How a lane's P stores are issued is measured, not cosmetic (tools/ra_variants.py; docs/history.md §3.1 has the table).  The memory
                // system rewards a wave that keeps FEW stores in flight: with interp_angle compiled in (row offsets as immediates, no
                // spills) the stores go out back to back with an `s_waitcnt vmcnt(1)` after every eighth — 0.362 ms per 512 config-B frames
                // (74.6 % of the HBM peak) against 0.488 ms unthrottled and 0.405 ms for the round-1 shape.  That shape — kept for the
                // runtime interp_angle — puts every store behind a wave-uniform branch on a kernel argument (always taken): the taken
                // branches, and the vmcnt(0) waits of its spill reloads, throttle a wave in the same way, by accident.
                float2* row = mapf + (size_t)k * NA;
#pragma unroll
                for (int u = 0; u < P; u++) {
                    const int a = (Ia * u + r + ahalf) & amask;   // fftshift: out'[a'] = out[(a' + NA/2) % NA]
                    if constexpr (IA > 0) {
                        if ((u % RA_GROUP) == 0 && pace_T) {
                            long long now = (long long)wall_clock64();
                            t_next += pace_T;
                            if (now - t_next > (long long)pace_K * pace_T) t_next = now - (long long)pace_K * pace_T;
                            while (now < t_next) { asm volatile("s_sleep 1"); now = (long long)wall_clock64(); }
                        }
                        const v2f t = {y[u].x, y[u].y};
                        __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(row + a));
                        if ((u % RA_GROUP) == RA_GROUP - 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    } else {
                        if (WPF > 0) { const v2f t = {y[u].x, y[u].y}; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(row + a)); }
                        else row[a] = y[u];
                    }
                }
            }
            // estimator arg-max (lib/range_angle_estimator_impl.cc:137-151) on the values still in registers
            const float thr = trk.raise(m);
            if (m >= thr) {
                const unsigned flat0 = (unsigned)k * (unsigned)NA;
#pragma unroll
                for (int u = 0; u < P; u++)
                    if (fast_power(y[u]) >= thr) trk.exact(y[u], flat0 + ((Ia * u + r + ahalf) & amask));
            }
        }
    }
    __syncthreads();
    block_reduce_peak(trk, reinterpret_cast<PeakPartial*>(s_g));
    if (tid == 0) { partials[(size_t)f * pstride + slice].best = trk.best; partials[(size_t)f * pstride + slice].idx = trk.idx; }
}

// ---- the roofline kernel of configs B and D: classes of 256 range bins, H in registers ------------------------------------------------
// (8 or 16 pairs x interp_angle 16 at fft_len 256 / 512 / 1024; everything else runs the 64-bin kernel above.)
// With classes of 64 range bins the rows a wave stores together lie NR / 64 rows apart — 64 KiB at config B, 256 KiB at config D — and at
// fft_len 1024 the 128 KiB of H in LDS leave one workgroup per CU with two barriers around the range phase of every 128 KiB class: 80 % of
// the HBM peak at config B only with paced stores, 65-70 % at config D or with longer range axes (docs/history.md §3.1).  Here a class is
// k = C q + c with C = NR / 256 and q < 256:
//     R[p][C q + c] = IFFT_256( g_c[p] )[q],   g_c[p][n'] = sum_{m < fft_len/256} H[p][n' + 256 m] e^{+j 2 pi (n' + 256 m) c / NR}
// The fold inputs of a lane — H[p][lane + 64 j + 256 m], the same for every class — live in registers, so H needs no LDS at all; the
// 256-point transform is one radix-4 step across a lane's four points (q = 4 a + b: twiddle e^{+j 2 pi lane b / 256}) and four 64-point
// transforms across the wavefront (a = bitrev6(lane)).  The range bins of a class (32 KiB for 16 pairs) are double-buffered, so ONE barrier
// per 512 KiB class separates the waves that fill a buffer from those that read it, the waves drift apart and the range phase of some
// overlaps the stores of others; rows stored together are NR / 256 rows apart (16 KiB at config B, 64 KiB at config D).
// Geometry: a wave takes P / (NT / 64) pairs.  fft_len 256 / 512: 256 threads, four (two) pairs per wave, 16-32 VGPRs of H, two workgroups
// per CU, stores paced (chain_pace); 16 pairs at fft_len 1024: 512 threads, two pairs per wave, 64 VGPRs of H, one workgroup per CU.
// Same angle axis, arg-max and MODEs as above; rng[frame][class][pair][(q & 3) * 64 + (q >> 2)] for the window pass.
#define RW_L 256
// the workgroup-shared word of MODE 1 lives in a function of its own, so that only the kernels that use it carry the static LDS: the
// MODE 3 form at fft_len 256 / 512 takes 80 KiB of dynamic LDS, two workgroups fill the CU's 160 KiB to the last allocation unit, and four
// more static bytes left one workgroup per CU (0.28 -> 0.39 ms per 512 config-B frames, measured)
template <bool ON> __device__ __forceinline__ unsigned* wide_run_max()        // [0] running maximum, [1], [2] pruning-mode votes for odd / even classes
{
    if constexpr (ON) { __shared__ unsigned v[3]; return v; }
    else return nullptr;
}
template <bool ON> __device__ __forceinline__ float2* wide_sample_twiddles()      // MODE 1: angle twiddles of the three sampled residues, [3][16]
{
    if constexpr (ON) { __shared__ float2 v[3 * 16]; return v; }
    else return nullptr;
}
template <int P, int MODE, int IA, int LOGN, int NT_ = 512>
__global__ __launch_bounds__(NT_, 2) void range_angle_wide_kernel(
    const float2* __restrict__ H, float2* __restrict__ map, PeakPartial* __restrict__ partials,
    const float2* __restrict__ twR, const float2* __restrict__ twA,
    int NR, int F, int WPF, int pstride, float2* __restrict__ rng_out, int nx, int pace)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    constexpr int NT = NT_, NW = NT / 64, PPW = P / NW, N = 1 << LOGN, MM = N >= RW_L ? N / RW_L : 1, Ia = IA, NA = P * Ia;
    constexpr int NJ = N >= RW_L ? 4 : N / 64;                  // fft_len 64 / 128: only the first one / two of a lane's four points are inputs, the rest zero
    static_assert(P % NW == 0, "whole pairs per wave");
    const int C = NR / RW_L;
    const int xcd = blockIdx.x % nx;
    const int jb = blockIdx.x / nx;
    const int f = (jb / WPF) * nx + xcd;
    const int slice = jb % WPF;
    if (f >= F) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* s_g = smem;                                         // [2][P][256]
    float* s_pw = reinterpret_cast<float*>(s_g + 2 * P * RW_L) + (size_t)wave * (64 * P);   // MODE 3: this wave's tile (its four rows)
    unsigned* const s_run = wide_run_max<MODE == 1>();          // MODE 1: the workgroup's running maximum (float bits), see the angle axis
    if constexpr (MODE == 1) { if (tid < 3) s_run[tid] = 0u; }                 // ordered before their first use by the barrier of the first class
    float2* const s_tw1 = wide_sample_twiddles<MODE == 1>();    // exp(-j 2 pi p r / NA) for r = Ia/4, Ia/2, 3 Ia/4 (the sampling bound of the angle axis); same barrier
    if constexpr (MODE == 1) { if (tid < 3 * 16) s_tw1[tid] = twA[((tid & 15) * ((tid >> 4) + 1) * (IA / 4)) & (P * IA - 1)]; }

    // this wave's share of H, for good
    float2 h[PPW][4][MM];
    {
        const float2* Hf = H + (size_t)f * P * N;
#pragma unroll
        for (int jj = 0; jj < PPW; jj++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int m = 0; m < MM; m++) h[jj][j][m] = j < NJ ? Hf[(size_t)(wave + NW * jj) * N + lane + 64 * j + RW_L * m] : make_float2(0.f, 0.f);
    }
    const int n_iter = (C - slice + WPF - 1) / WPF;
    auto class_of = [&](int it) -> int { return slice + it * WPF; };
    float2 tc[4][MM];                                           // class twiddles exp(+j 2 pi n c / NR), n = lane + 64 j + 256 m
    auto fetch_tc = [&](int c) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int m = 0; m < MM; m++) tc[j][m] = j < NJ ? twR[((lane + 64 * j + RW_L * m) * c) & (NR - 1)] : make_float2(0.f, 0.f);
    };
    fetch_tc(class_of(0));
    float2 t64[6];                                              // 64-point inverse FFT twiddles of this lane, one per radix-2 stage
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int half = 32 >> st;
        t64[st] = twR[((lane & (half - 1)) * (32 / half)) * (NR / 64)];
    }
    float2 t256[4];                                             // exp(+j 2 pi lane b / 256)
#pragma unroll
    for (int b = 1; b < 4; b++) t256[b] = twR[(lane * b) * (NR / RW_L)];
    const int r = tid % Ia;
    float2 ta[P];
#pragma unroll
    for (int p = 1; p < P; p++) ta[p] = twA[(p * r) & (NA - 1)];

    PeakTracker trk;
    trk.init();
    bool prune_on = MODE == 1 && !(pace & 8);                   // MODE 1: rows are skipped by their bounds (angle axis below) until that stops paying in this frame
    // The way a class is pruned is the same for every wave of the workgroup (the sampling bound deals the rows to the waves differently from
    // the trips, so a class is covered only if all waves work it the same way): 0 = sum bound, trip by trip; 1 = sampling bound, row by row;
    // 2 = not at all.  A wave that finds its bound useless votes for the next one in the word of the NEXT class's parity (s_run[1 + parity]):
    // the votes of a class are all cast before the next class's barrier and read after it, and the mode only ever goes up.
    unsigned pmode = 0u;
    unsigned seen = 0u;                                         // s_run as last read
    constexpr int items = RW_L * Ia;
    constexpr int ahalf = NA >> 1, amask = NA - 1;
    float2* mapf = map + (size_t)f * NR * NA;
    float* mapp = reinterpret_cast<float*>(map) + (size_t)f * NR * NA;          // MODE 3
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int pace_T = pace & 0xfff, pace_K = (pace >> 12) & 0xf;
    long long t_next = 0;
    if constexpr (MODE == 0) { if (pace_T) t_next = (long long)wall_clock64(); }

#pragma unroll 1
    for (int it = 0; it < n_iter; it++) {
        const int c = class_of(it);
        float2* sg = s_g + (size_t)(it & 1) * (P * RW_L);
        // ---- range axis ---------------------------------------------------------------------------------------------------------
        {
            float2 v[PPW][4];
#pragma unroll
            for (int jj = 0; jj < PPW; jj++) {
                float2 g[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    g[j] = make_float2(0.f, 0.f);
#pragma unroll
                    for (int m = 0; m < MM; m++) {
                        const float2 hh = h[jj][j][m], w = tc[j][m];
                        g[j].x = fmaf(hh.x, w.x, fmaf(-hh.y, w.y, g[j].x));
                        g[j].y = fmaf(hh.x, w.y, fmaf(hh.y, w.x, g[j].y));
                    }
                }
                // inverse radix-4 across the lane's four points: y_b = sum_j g[j] (+j)^(j b)
                const float2 a0 = make_float2(g[0].x + g[2].x, g[0].y + g[2].y), a1 = make_float2(g[0].x - g[2].x, g[0].y - g[2].y);
                const float2 b0 = make_float2(g[1].x + g[3].x, g[1].y + g[3].y), b1 = make_float2(g[1].x - g[3].x, g[1].y - g[3].y);
                v[jj][0] = make_float2(a0.x + b0.x, a0.y + b0.y);
                v[jj][1] = cmul_pin(make_float2(a1.x - b1.y, a1.y + b1.x), t256[1]);      // a1 + j b1
                v[jj][2] = cmul_pin(make_float2(a0.x - b0.x, a0.y - b0.y), t256[2]);
                v[jj][3] = cmul_pin(make_float2(a1.x + b1.y, a1.y - b1.x), t256[3]);      // a1 - j b1
            }
#define RW_STAGE(B, ST)                                                                                   \
            _Pragma("unroll") for (int jj = 0; jj < PPW; jj++)                                            \
                _Pragma("unroll") for (int b = 0; b < 4; b++) v[jj][b] = wave_dif_stage<B>(v[jj][b], t64[ST], lane);
            RW_STAGE(5, 0) RW_STAGE(4, 1) RW_STAGE(3, 2) RW_STAGE(2, 3) RW_STAGE(1, 4) RW_STAGE(0, 5)
#undef RW_STAGE
            const int a = (int)(__brev((unsigned)lane) >> 26);   // lane holds X[4 a + b]
#pragma unroll
            for (int jj = 0; jj < PPW; jj++)
#pragma unroll
                for (int b = 0; b < 4; b++) sg[(wave + NW * jj) * RW_L + b * 64 + a] = v[jj][b];
        }
        if (it + 1 < n_iter) fetch_tc(class_of(it + 1));          // they land during the stores
        __syncthreads();                                          // the one barrier of the class: its range bins are complete; the buffer the next
                                                                  // range phase fills was last read before this barrier
        if constexpr (MODE == 1 || MODE == 3) {
            // write-once for the window pass: around the caches — a cached store would sit dirty in the Infinity Cache and be written back
            // while the NEXT step's A1 streams its input (docs/history.md §6: A1 0.098 -> 0.13 ms per 512 config-B frames behind 128 MB of cached stores)
            // (MODE 3: cached or non-temporal measures the same, 0.284 ms per 512 config-B frames, and the A1 behind it 0.127 ms either way)
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f* dst = reinterpret_cast<v4f*>(rng_out + ((size_t)f * C + c) * (P * RW_L));
            const float4* src = reinterpret_cast<const float4*>(sg);
            if (!(MODE == 1 && (pace & 2)))
            for (int i = tid; i < (P * RW_L) / 2; i += NT) {
                const float4 vv = src[i];
                const v4f t = {vv.x, vv.y, vv.z, vv.w};
                if constexpr (MODE == 1) __builtin_nontemporal_store(t, dst + i);
                else dst[i] = t;
            }
        }
        if (MODE == 1 && (pace & 1)) continue;                    // experiment (JRC_DETECT_EXP bit 0): no angle stage at all — the floor of the range phase
        // ---- angle axis + fftshift + store + arg-max -------------------------------------------------------------------------
        // MODE 1 (nothing stored): a range bin whose bound B_k = (sum_p |R[p][k]|)^2 >= every |cell|^2 of its row lies below the running
        // maximum cannot hold the arg-max, and its Ia P-point transforms are skipped.  Exact: a row is skipped only when
        // B_k (1 + 1e-4) < run_max (1 - 1e-5) with run_max <= the final maximum, so every cell the unpruned scan evaluates exactly near
        // the final maximum is still evaluated, and PeakTracker resolves ties by flat index, not by visiting order — the records stay
        // byte-identical to map mode (tests/test_gpu_chain_modes.py).  Per class a wave computes the bounds of its TRIPS x 4 rows once
        // (one LDS read, a square root and a DPP row sum per lane and trip, all trips in flight together) and turns them into a bit mask
        // of trips still worth computing against the workgroup's running maximum (s_run, shared through LDS).  The workgroup's FIRST
        // class has nothing to compare with: every wave computes the trip with its largest bound, the maxima meet at one extra barrier,
        // and only then is the mask formed.  A frame with a target is left with the main-lobe rows; in a frame of noise alone (or of
        // targets too weak to stand out) the bound is ~P / ln(cells) above the maximum and most rows stay: a wave that finds more than
        // half of a class's trips still to do stops computing bounds for the rest of the frame, so such frames pay them once.
        constexpr int TRIPS = items / NT;
        auto row_bound = [&](int w0) -> float {
            const int q = (w0 + tid) / Ia;
            const int qi = (q & 3) * 64 + (q >> 2);
            float sum = 0.f;
            for (int p = r; p < P; p += Ia) { const float2 v = sg[p * RW_L + qi]; sum += __fsqrt_rn(fmaf(v.x, v.x, v.y * v.y)); }
            if constexpr (Ia == 16) {          // the Ia lanes of a range bin are one DPP row: quad xor 1, xor 2, half-row mirror, row mirror
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, false));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, false));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x141, 0xf, 0xf, false));
                sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x140, 0xf, 0xf, false));
            } else {
                for (int off = Ia >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            }
            return (sum * sum) * (1.0f + 1e-4f);
        };
        unsigned long long rows = 0ull;                           // MODE 1 after the sampling bound: the wave's rows still to compute (bit = trip * 4 + row of the trip)
        bool use_rows = false;
        float bnd[TRIPS];                                         // MODE 1: bound of this lane's row in trip t (the same in the Ia lanes of the row)
        if constexpr (MODE == 1) pmode = max(pmode, *reinterpret_cast<volatile unsigned*>(s_run + 1 + (it & 1)));
        if ((pace & 16) && pmode == 1u) pmode = 2u;               // JRC_DETECT_EXP bit 4: sum bound only
        const bool l1_live = pmode == 0u;                         // ... formed for this class
        unsigned* const vote = MODE == 1 ? s_run + 1 + ((it + 1) & 1) : nullptr;
        unsigned todo = TRIPS >= 32 ? 0xffffffffu : (1u << TRIPS) - 1u;
        int t_first = -1;
        auto candidates = [&]() -> unsigned {
            trk.run_max = fmaxf(trk.run_max, __uint_as_float(*reinterpret_cast<volatile unsigned*>(s_run)));
            const float thr = trk.run_max * (1.0f - 1e-5f);
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < TRIPS; t++) m |= (__ballot(bnd[t] >= thr) != 0ull ? 1u : 0u) << t;
            return m;
        };
        // Second bound, for frames in which nothing stands out (noise, weak or many targets) and sum_p |R[p][k]| says little: the angle axis of a
        // row is a trigonometric polynomial f(theta) = sum_p R[p][k] e^{-j p theta} of degree P - 1, and the cells r = 0, Ia/4, Ia/2, 3 Ia/4 (mod Ia)
        // are its values at 4 P equispaced angles.  By the Ehlich-Zeller inequality (a trigonometric polynomial of degree n is bounded between
        // N > 2 n equidistant nodes by its largest node value over cos(pi n / N); applied to e^{j (P-1) phi} f(2 phi), degree P - 1 on 8 P nodes)
        //     max over all theta of |f|^2  <=  max over those 4 P cells of |f|^2 / cos^2(pi (P - 1) / (8 P))          (P = 16: x 1.1488)
        // so four of the Ia P-point transforms of a row decide whether the other Ia - 4 can hold the maximum (tests/test_oracle_radar.py checks the
        // constant numerically).  A lane takes one row of the wave's TRIPS x 4 (both halves of the wave share the rows when there are only 32);
        // the values are bounds with 1e-4 of slack, so this pass runs on the plain butterflies.
        // rows of the class as the sampling pass deals them to the waves (not the trips' dealing: a row's 16 lanes read one LDS address there; here
        // 32 lanes read 32 rows, which must fall into 32 different banks — q >> 2 runs over 32 consecutive values): bit i of a wave's row mask <-> row_of(i)
        auto row_of = [&](int i) -> int {
            constexpr int RPW_ = TRIPS * 4;
            const int rl = i & 31, hi = i >> 5;
            return RPW_ >= 64 ? 4 * (rl + 32 * (wave >> 1)) + 2 * (wave & 1) + hi : 4 * (rl + 32 * (wave & 1)) + (wave >> 1);
        };
        auto refine = [&](unsigned cand) -> unsigned {
            constexpr int RPW = TRIPS * 4, PASSES = RPW >= 64 ? 4 : 2;
            static_assert(RPW == 64 || RPW == 32, "a wave's rows fill the wave or half of it");
            const int half = RPW >= 64 ? 0 : lane >> 5;
            const int q = row_of(lane);                                          // a row per lane, 32 consecutive values of q >> 2 per half wave: no LDS bank conflicts
            const int qi = (q & 3) * 64 + (q >> 2);
            float rowmax = 0.f;
#pragma unroll 1
            for (int ps = 0; ps < PASSES; ps++) {
                const int ri = RPW >= 64 ? ps : ps + 2 * half;     // residue r = ri Ia / 4
                float2 y[P];
                y[0] = sg[qi];
                if (ri == 0) {
#pragma unroll
                    for (int p = 1; p < P; p++) y[p] = sg[p * RW_L + qi];
                } else {
                    const float2* tw = s_tw1 + (ri - 1) * 16;
#pragma unroll
                    for (int p = 1; p < P; p++) y[p] = cmul(sg[p * RW_L + qi], tw[p]);
                }
                fft_fwd_small<P>(y);
#pragma unroll
                for (int u = 0; u < P; u++) rowmax = fmaxf(rowmax, fast_power(y[u]));
            }
            if constexpr (RPW < 64) rowmax = fmaxf(rowmax, __shfl_xor(rowmax, 32));
            static_assert(P == 16 || P == 8, "1 / cos^2(pi (P - 1) / (8 P)), rounded up, is tabulated for the pair counts of this kernel");
            constexpr float EZ = P == 16 ? 1.14880f : 1.12803f;
            {   // the sampled cells ARE cells of the map (to the last bits: plain instead of pinned butterflies, < 2e-6), so the largest of them is a
                // lower bound of the maximum: it raises the filter threshold (never the tracked candidates) before anything else of the class is computed
                const float wm = wave_max_f32(rowmax) * (1.0f - 4e-6f);
                if (wm > trk.run_max) { trk.run_max = wm; if (lane == 0) atomicMax(s_run, __float_as_uint(wm)); }
                JRC_LOCKSTEP();                                   // lane 0's store precedes every lane's next read of s_run (one instruction stream)
            }
            unsigned long long surv = __ballot(rowmax * (EZ * (1.0f + 1e-4f)) >= trk.run_max * (1.0f - 1e-5f));
            if constexpr (RPW < 64) surv &= 0xffffffffull;
            rows = (pace & 32) ? 0ull : surv;                     // from here on the class is worked row by row: four surviving rows to a trip, whichever they are (JRC_DETECT_EXP bit 5: none - timing only)
            use_rows = true;
            return cand;
        };
        constexpr int S1_COST = TRIPS * 4 >= 64 ? 4 : 2;          // the sampling pass costs as much as this many trips
        if (MODE == 1 && prune_on) {
            if (pmode == 0u) {
#pragma unroll
                for (int t = 0; t < TRIPS; t++) bnd[t] = row_bound(t * NT);
                if (it == 0) {
                    float bm = bnd[0];
#pragma unroll
                    for (int t = 1; t < TRIPS; t++) bm = fmaxf(bm, bnd[t]);
                    bm = wave_max_f32(bm);
                    t_first = 0;
#pragma unroll
                    for (int t = TRIPS - 1; t >= 0; t--) if (__ballot(bnd[t] == bm) != 0ull) t_first = t;
                    todo = 1u << t_first;
                } else {
                    todo = candidates();
                    seen = *reinterpret_cast<volatile unsigned*>(s_run);
                    if (__popc(todo) > S1_COST + 1 && lane == 0) atomicMax(vote, 1u);     // the sums say little here: sample the rows from the next class on
                }
            } else if (pmode == 1u) {
                trk.run_max = fmaxf(trk.run_max, __uint_as_float(*reinterpret_cast<volatile unsigned*>(s_run)));
                refine(todo);
                if (4 * __popcll(rows) > 3 * TRIPS * 4 && lane == 0) atomicMax(vote, 2u); // most rows survive the sampling too: leave the rest of the frame alone
            }
        }
#pragma unroll 1
        for (int tt = 0; ; tt++) {
            int q;                                                // this lane's range bin of the class ((w0 + tid) % Ia == r; a wave's four bins q .. q + 3 are the four b of one a)
            if constexpr (MODE == 1) {
                if (tt > 0 && t_first < 0 && prune_on && l1_live) {   // a trip was computed: has the workgroup's maximum moved?
                    const unsigned now = *reinterpret_cast<volatile unsigned*>(s_run);
                    if (now != seen) { seen = now; todo &= candidates(); }
                }
                if (!use_rows && todo == 0u) {
                    if (t_first < 0) break;
                    __syncthreads();                              // first class of the workgroup: every wave has published the maximum of its strongest trip
                    todo = candidates() & ~(1u << t_first);
                    seen = *reinterpret_cast<volatile unsigned*>(s_run);
                    t_first = -1;
                    // the first class can still change its mind, all waves together: votes into this class's own word (read by everybody before the
                    // barrier above, not read again), a second barrier, and whoever finds a vote samples its rows instead of computing its trips
                    if (__popc(todo) > S1_COST + 1 && lane == 0) atomicMax(s_run + 1 + (it & 1), 1u);
                    __syncthreads();
                    if (!(pace & 16) && *reinterpret_cast<volatile unsigned*>(s_run + 1 + (it & 1)) >= 1u) {
                        pmode = 1u;
                        trk.run_max = fmaxf(trk.run_max, __uint_as_float(*reinterpret_cast<volatile unsigned*>(s_run)));
                        refine(todo);
                        if (4 * __popcll(rows) > 3 * TRIPS * 4 && lane == 0) atomicMax(vote, 2u);
                    }
                    if (!use_rows && todo == 0u) break;
                }
                if (use_rows) {                                   // four surviving rows, one per 16 lanes (a short last group repeats a row: the tracker does not mind)
                    if (rows == 0ull) break;
                    int rsel[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        rsel[g] = rows ? __ffsll(rows) - 1 : rsel[g > 0 ? g - 1 : 0];
                        if (rows) rows &= rows - 1ull;
                    }
                    const int g = lane >> 4;
                    q = row_of(g == 0 ? rsel[0] : (g == 1 ? rsel[1] : (g == 2 ? rsel[2] : rsel[3])));
                } else {
                    const int w0 = (__ffs((int)todo) - 1) * NT;
                    todo &= todo - 1u;
                    q = (w0 + tid) / Ia;
                }
            } else {
                if (tt >= TRIPS) break;
                q = (tt * NT + tid) / Ia;
            }
            const int qi = (q & 3) * 64 + (q >> 2);
            const int k = C * q + c;
            float2 y[P];
            y[0] = sg[qi];
#pragma unroll
            for (int p = 1; p < P; p++) y[p] = cmul_pin(sg[p * RW_L + qi], ta[p]);
            fft_fwd_small_pin<P>(y);
            float m = -1.0f;
#pragma unroll
            for (int u = 0; u < P; u++) m = fmaxf(m, fast_power(y[u]));
            if constexpr (MODE == 3) {
                constexpr int rw = 64 / Ia;
                const int rb = lane / Ia, q0 = q - rb;
                typedef float v4f __attribute__((ext_vector_type(4)));
                float* trow = s_pw + rb * NA;
#pragma unroll
                for (int u = 0; u < P; u++) trow[(Ia * u + r + ahalf) & amask] = y[u].x * y[u].x + y[u].y * y[u].y;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                constexpr int n4 = rw * NA / 4;
                for (int i = lane; i < n4; i += 64) {
                    const int row = (4 * i) / NA, col = (4 * i) % NA;
                    const float4 vv = *reinterpret_cast<const float4*>(s_pw + 4 * i);
                    const v4f t = {vv.x, vv.y, vv.z, vv.w};
                    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(mapp + (size_t)(C * (q0 + row) + c) * NA + col));
                }
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (MODE == 0) {
                float2* row = mapf + (size_t)k * NA;
#pragma unroll
                for (int u = 0; u < P; u++) {
                    const int aa = (Ia * u + r + ahalf) & amask;  // fftshift
                    if ((u % RA_GROUP) == 0 && pace_T) {
                        long long now = (long long)wall_clock64();
                        t_next += pace_T;
                        if (now - t_next > (long long)pace_K * pace_T) t_next = now - (long long)pace_K * pace_T;
                        while (now < t_next) { asm volatile("s_sleep 1"); now = (long long)wall_clock64(); }
                    }
                    const v2f t = {y[u].x, y[u].y};
                    __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(row + aa));
                    if ((u % RA_GROUP) == RA_GROUP - 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                }
            }
            const float before = trk.run_max;
            const float thr = trk.raise(m);
            if constexpr (MODE == 1) { if (trk.run_max > before && lane == 0) atomicMax(s_run, __float_as_uint(trk.run_max)); }   // non-negative floats order as their bit patterns
            if constexpr (MODE == 1) JRC_LOCKSTEP();          // (as above: the wave reads s_run again at its next trip)
            if (m >= thr) {
                const unsigned flat0 = (unsigned)k * (unsigned)NA;
#pragma unroll
                for (int u = 0; u < P; u++)
                    if (fast_power(y[u]) >= thr) trk.exact(y[u], flat0 + ((Ia * u + r + ahalf) & amask));
            }
        }
    }
    __syncthreads();
    block_reduce_peak(trk, reinterpret_cast<PeakPartial*>(s_g));
    if (tid == 0) { partials[(size_t)f * pstride + slice].best = trk.best; partials[(size_t)f * pstride + slice].idx = trk.idx; }
}

// The estimator's noise-window rows for the map-less modes: one workgroup per frame merges the frame's partial maxima into the peak, and
// for each of the 2 dr range bins of the window takes the P range bins the fused kernel left in rng and runs the angle axis exactly as
// the fused kernel does (a lane per residue r: twiddle, P-point transform, fftshift) into win[frame][row][NA].
template <int P>
__global__ __launch_bounds__(256) void ra_window_rows_kernel(const float2* __restrict__ rng, const PeakPartial* __restrict__ partials, int pstride,
                                                             float2* __restrict__ win, const float2* __restrict__ twA, int NR, int Ia,
                                                             int win_rows, int win_off, int L /* range bins per class: 64, or 256 (range_angle_wide_kernel) */)
{
#pragma clang fp contract(off)
    __shared__ PeakPartial red[4];
    __shared__ int s_row0;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int NA = P * Ia, C = NR / L;
    PeakTracker pk;
    pk.init();
    for (int i = tid; i < pstride; i += blockDim.x) pk.merge(partials[(size_t)f * pstride + i].best, partials[(size_t)f * pstride + i].idx);
    block_reduce_peak(pk, red);
    if (tid == 0) s_row0 = (int)(pk.idx / (unsigned)NA) + win_off;
    __syncthreads();
    const int row0 = s_row0;
    const int r = tid % Ia;
    float2 ta[P];
#pragma unroll
    for (int p = 1; p < P; p++) ta[p] = twA[(p * r) & (NA - 1)];
    const int ahalf = NA >> 1, amask = NA - 1;
    for (int row = tid / Ia; row < win_rows; row += blockDim.x / Ia) {
        const int k = ((row0 + row) % NR + NR) % NR;
        const int c = k % C, ql = k / C;                   // k = C ql + c
        const float2* g = rng + ((size_t)f * C + c) * (size_t)(P * L) + (L == RA_L ? ql : (ql & 3) * 64 + (ql >> 2));
        float2 y[P];
        y[0] = g[0];
#pragma unroll
        for (int p = 1; p < P; p++) y[p] = cmul_pin(g[p * L], ta[p]);
        fft_fwd_small_pin<P>(y);
        float2* out = win + ((size_t)f * win_rows + row) * NA;
#pragma unroll
        for (int u = 0; u < P; u++) {                           // read once by ra_finalize, never again: around the caches (see the range profiles above)
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f t = {y[u].x, y[u].y};
            __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(out + ((Ia * u + r + ahalf) & amask)));
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct jrc_bg_state;
struct jrc_chain {
    jrc_ctx* ctx;
    jrc_chain_cfg cfg;
    int P, NR, NA, C, threads, wg_per_cu, n_cus, wpf_override, max_frames;
    bool generic = false;             // shapes the fused kernel does not cover: block-by-block kernels on the device
    bool wide = false;                // range_angle_wide_kernel: 16 pairs x interp_angle 16 at fft_len 256 / 512 / 1024 (configs B, D): classes of 256 range bins, H in registers
    float2* d_pad = nullptr;          // generic mode: [max_frames][P][NR] zero-padded rows / range profiles
    int gen_blocks = 0;               // generic mode: partial-maximum blocks per frame
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
    // detect-only mode (jrc_chain_set_write_map): the map is not stored; the estimator's noise-window rows are re-computed
    bool write_map = true;
    int map_format = JRC_MAP_COMPLEX;  // JRC_MAP_POWER: float |z|^2 map (MODE 3)
    size_t lds_power = 0;             // dynamic LDS of the MODE 3 instantiation (fused-kernel LDS + a tile per wave)
    bool power_rows1 = false;         // the tile holds one row at a time (LDS short)
    int win_dr = 0;                   // the estimator's discard_range_idx (:189), computed as the device does
    float2* d_win = nullptr;          // [max_frames][2 win_dr][NA]
    float2* d_rng = nullptr;          // [max_frames][C][P][64] range profiles the map-less modes leave for the window pass
    // background recording / removal (jrc_chain_set_background)
    jrc_bg_state* bg = nullptr;
    float2* d_raw = nullptr;          // [max_frames][P][N] estimates before the subtraction
    // detect-only pipeline (chain_run): the batch in slices, A1 of slice i+1 (HBM-bound) beside the transforms of slice i (issue-bound)
    static const int kMaxSlices = 16;
    int slices = 1;                   // JRC_DETECT_SLICES: slices of the detect-only pipeline (default 1: none, see chain_pick_slices)
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_a1[kMaxSlices] = {};
    hipEvent_t ev_side[2] = {nullptr, nullptr};
    // results in flight (jrc_chain_fetch_results_begin / _end): a copy stream, two pinned buffers, FIFO of at most two
    hipStream_t copy_stream = nullptr;
    jrc_ra_result* h_ring[2] = {nullptr, nullptr};
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    int ring_n[2] = {0, 0}, ring_head = 0, ring_count = 0;
};

// ---- background state of one radar stream (lib/mimo_ofdm_radar_impl.cc:276-300) on the device --------------------------------
// The reference keeps radar_chan_est_temp (the last RECORDED raw estimate, zero at construction :115) and a boost::circular_buffer of
// record_len past temps; frame f's mean runs over the buffer as it stood before f, oldest entry first, each term divided by the
// entry count (:281-292), and temp is pushed after every frame while removal is on (:297-300).  Here the history lives in HBM oldest
// first ([count][P*N], double buffered), a batch of frames is handled by one kernel (a thread per (frame, element) walks its own
// window: entries older than the batch come from the history, the others from the batch's raw estimates), and a second kernel
// writes the history the next batch will see.  One state may be shared by several chains (the slots of a jrc_chain_feed): an event
// orders consecutive batches across their streams.
struct jrc_bg_state {
    int removal = 0, recording = 0, record_len = 0;
    size_t pn = 0;
    float2* hist[2] = {nullptr, nullptr};
    int cur = 0, count = 0;
    float2* temp = nullptr;
    hipEvent_t updated = nullptr;
    bool pending = false;
    int refs = 1;
};

__global__ __launch_bounds__(256) void chain_background_kernel(const float2* __restrict__ raw, float2* __restrict__ est,
                                                               const float2* __restrict__ hist, const float2* __restrict__ temp,
                                                               int pn, int n0, int L, int recording)
{
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (idx >= pn) return;
    const float2 e = raw[(size_t)f * pn + idx];
    const int c = min(L, n0 + f);                      // entries in the buffer before frame f
    float2 m = make_float2(0.f, 0.f);
    const float n = (float)c;
    for (int j = f - c; j < f; j++) {                  // oldest first (:286-290)
        const float2 v = j < 0 ? hist[(size_t)(n0 + j) * pn + idx] : (recording ? raw[(size_t)j * pn + idx] : temp[idx]);
        m.x = m.x + v.x / n;                           // complex / float divides each component (:289)
        m.y = m.y + v.y / n;
    }
    est[(size_t)f * pn + idx] = make_float2(e.x - m.x, e.y - m.y);   // :292
}

// history after a batch of F frames: the last min(L, n0 + F) entries of (old history, then one push per frame)
__global__ __launch_bounds__(256) void chain_background_push_kernel(const float2* __restrict__ raw, const float2* __restrict__ hist,
                                                                    float2* __restrict__ hist_new, const float2* __restrict__ temp,
                                                                    int pn, int n0, int F, int n1, int recording)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = blockIdx.y;
    if (idx >= pn) return;
    const int j = F - n1 + slot;                       // position in the push sequence of this batch (negative: old history)
    hist_new[(size_t)slot * pn + idx] = j < 0 ? hist[(size_t)(n0 + j) * pn + idx] : (recording ? raw[(size_t)j * pn + idx] : temp[idx]);
}

static void bg_release(jrc_bg_state* b)
{
    if (!b || --b->refs > 0) return;
    if (b->hist[0]) (void)hipFree(b->hist[0]);
    if (b->hist[1]) (void)hipFree(b->hist[1]);
    if (b->temp) (void)hipFree(b->temp);
    if (b->updated) (void)hipEventDestroy(b->updated);
    delete b;
}

// generic mode helper: rows of H -> zero-padded rows of length NR (the padding mimo_ofdm_radar emits, :243, :312-315)
__global__ void pad_rows_kernel(const float2* __restrict__ H, float2* __restrict__ out, int N, int NR, size_t rows)
{
    const size_t total = rows * (size_t)NR;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / NR;
        const int k = (int)(i - row * NR);
        out[i] = k < N ? H[row * N + k] : make_float2(0.f, 0.f);
    }
}

// slices per frame: enough workgroups to fill every CU once (they are persistent over their classes), not more
static int chain_pick_wpf(const jrc_chain* ch, int n_frames)
{
    if (ch->wpf_override > 0) return ch->wpf_override < ch->C ? ch->wpf_override : ch->C;
    const int target = ch->n_cus * ch->wg_per_cu;
    int wpf = 1;
    while (wpf * 2 <= ch->C && (long)wpf * 2 * n_frames <= target) wpf *= 2;
    return wpf;
}

// frames per launch: one resident wave of workgroups, a multiple of the XCD count (the decode deals frames over the XCDs in groups)
static int chain_chunk(const jrc_chain* ch, int wpf)
{
    const int nx = ch->ctx->n_xcd;
    int chunk = (ch->n_cus * ch->wg_per_cu) / wpf;
    if (chunk < nx) chunk = nx;
    return chunk - chunk % nx;
}

// Store pacing of the map-writing kernels (docs/history.md §3.1): every wave releases its map stores in groups of RA_GROUP x 512 B, one group per T
// ticks of wall_clock64().  What the sweeps (tools/pace_sweep.sh) found is a property of the memory system, an OFFERED BYTE RATE at which the map
// stream runs at the pace of a pure store stream — not a tick count: so the word is derived from that rate, the bytes of a group, the waves
// actually resident for this launch and the clock of the counter,
//     T = group_bytes x resident_waves / offered_rate   [ticks],
// and a partition with fewer CUs (CPX / DPX), another SKU or a last, smaller launch of a batch gets the word that offers the same rate.
//   wide kernel as two 256-thread workgroups per CU (fft_len 256 / 512): 7.77 TB/s offered, no catching up — the plateau measured at config B on 256
//     CUs was T = 102 ... 112 ticks of 10 ns (0.328-0.332 ms per 512 frames, 82 % of the HBM peak; 0.352 unpaced), which this gives as 108;
//   64-bin kernel, 16 pairs x interp_angle 16, two 256-thread workgroups per CU: 6.93 TB/s, one group of catching up (T = 121: 0.341 against 0.366 ms);
//   fft_len 1024 on the 64-bin kernel (JRC_NO_WIDE; one 512-thread workgroup per CU): 8.39 TB/s (T = 100: 0.777 ms per 256 frames against 0.85).
// The wide kernel at fft_len 1024 runs the same paced or not (0.689-0.695 ms for T = 60 ... 100) and every other geometry was never found to gain: 0.
// JRC_RA_OFFERED_TBPS replaces the rate, JRC_RA_PACE the whole word.
static int chain_pace(const jrc_chain* ch, int resident_workgroups)
{
    if (ch->ctx->tune.ra_pace >= 0) return ch->ctx->tune.ra_pace;
    double offered = 0;
    int catch_up = 0;
    if (ch->wide && ch->threads == 256) offered = 7.77e12;
    else if (!ch->wide && ch->P == 16 && ch->cfg.interp_angle == 16 && ch->threads == 256 && ch->wg_per_cu == 2 && ch->cfg.interp_range <= 8) { offered = 6.93e12; catch_up = 1; }
    else if (!ch->wide && ch->P == 16 && ch->cfg.interp_angle == 16 && ch->threads == 512 && ch->cfg.fft_len == 1024 && ch->cfg.interp_range <= 8) offered = 8.39e12;
    if (offered == 0) return 0;
    if (ch->ctx->tune.ra_offered_tbps > 0) offered = ch->ctx->tune.ra_offered_tbps * 1e12;
    const double group_bytes = (double)RA_GROUP * 64 * sizeof(float2);
    const double waves = (double)resident_workgroups * (ch->threads / 64);
    long T = lround(group_bytes * waves / offered * (ch->ctx->wall_clock_khz * 1e3));
    if (T < 0) T = 0;
    if (T > 0xfff) T = 0xfff;
    return (catch_up << 12) | (int)T;
}

template <int P, int NT, int MMAX, bool TWC_LDS, int MODE, int IA, bool ROWS1 = false>
static int launch_fused_mode(jrc_chain* ch, int n_frames, int wpf, int pstride, const float2* d_H, float2* d_map, hipStream_t s)
{
    const size_t lds_bytes = MODE == 3 ? ch->lds_power : ch->lds_bytes;
    JRC_TRY(jrc_ensure_dyn_lds(ch->ctx, (const void*)range_angle_fused_kernel<P, NT, MMAX, TWC_LDS, MODE, IA, ROWS1>, lds_bytes));
    // One resident wave of workgroups per launch: a batch that needs more is launched in chunks of that size, and a last,
    // smaller chunk gets more slices per frame so that it fills the machine as well (a grid twice the resident size runs 20 %
    // slower than two launches because its second wave of workgroups starts ragged).  `pstride` partial maxima per frame.
    const int nx = ch->ctx->n_xcd;
    const int chunk = chain_chunk(ch, wpf);
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = n_frames - f0 < chunk ? n_frames - f0 : chunk;
        int w = chain_pick_wpf(ch, nf);
        if (w > pstride) w = pstride;
        const dim3 grid((unsigned)(((nf + nx - 1) / nx) * nx * w));
        float2* mp = MODE == 0 ? d_map + (size_t)f0 * ch->NR * ch->NA
                   : (MODE == 3 ? reinterpret_cast<float2*>(reinterpret_cast<float*>(d_map) + (size_t)f0 * ch->NR * ch->NA) : nullptr);
        hipLaunchKernelGGL((range_angle_fused_kernel<P, NT, MMAX, TWC_LDS, MODE, IA, ROWS1>), grid, dim3(NT), lds_bytes, s,
                           d_H + (size_t)f0 * P * ch->cfg.fft_len, mp,
                           ch->d_partials + (size_t)f0 * pstride,
                           ch->twR, ch->twA, ch->cfg.fft_len, ch->NR, ch->cfg.interp_angle, nf, w, pstride,
                           (MODE == 1 || MODE == 3) ? ch->d_rng + (size_t)f0 * ch->NR * P : nullptr, nx,
                           chain_pace(ch, (int)grid.x < ch->n_cus * ch->wg_per_cu ? (int)grid.x : ch->n_cus * ch->wg_per_cu));
    }
    JRC_HIP(ch->ctx, hipGetLastError());
    return JRC_OK;
}

template <int P, int MODE, int LOGN, int NT_>
static int launch_fused_wide(jrc_chain* ch, int n_frames, int wpf, int pstride, const float2* d_H, float2* d_map, hipStream_t s)
{
    const size_t lds_bytes = MODE == 3 ? ch->lds_power : ch->lds_bytes;
    JRC_TRY(jrc_ensure_dyn_lds(ch->ctx, (const void*)range_angle_wide_kernel<P, MODE, 16, LOGN, NT_>, lds_bytes));
    const int nx = ch->ctx->n_xcd;
    const int chunk = chain_chunk(ch, wpf);
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = n_frames - f0 < chunk ? n_frames - f0 : chunk;
        int w = chain_pick_wpf(ch, nf);
        if (w > pstride) w = pstride;
        const dim3 grid((unsigned)(((nf + nx - 1) / nx) * nx * w));
        float2* mp = MODE == 0 ? d_map + (size_t)f0 * ch->NR * ch->NA
                   : (MODE == 3 ? reinterpret_cast<float2*>(reinterpret_cast<float*>(d_map) + (size_t)f0 * ch->NR * ch->NA) : nullptr);
        hipLaunchKernelGGL((range_angle_wide_kernel<P, MODE, 16, LOGN, NT_>), grid, dim3(NT_), lds_bytes, s,
                           d_H + (size_t)f0 * P * ch->cfg.fft_len, mp, ch->d_partials + (size_t)f0 * pstride, ch->twR, ch->twA,
                           ch->NR, nf, w, pstride, (MODE == 1 || MODE == 3) ? ch->d_rng + (size_t)f0 * ch->NR * P : nullptr, nx,
                           MODE == 1 ? ch->ctx->tune.detect_exp : chain_pace(ch, (int)grid.x < ch->n_cus * ch->wg_per_cu ? (int)grid.x : ch->n_cus * ch->wg_per_cu));
    }
    JRC_HIP(ch->ctx, hipGetLastError());
    return JRC_OK;
}

template <int P, int NT, int MMAX, bool TWC_LDS>
static int launch_fused_nt(jrc_chain* ch, int mode, int n_frames, int wpf, int pstride, const float2* d_H, float2* d_map, hipStream_t s)
{
    // interp_angle = 16 (the flowgraphs' interp_factor_angle, examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc) is compiled in; any other
    // value takes the runtime argument
    if (ch->cfg.interp_angle == 16) {
        if (mode == 0) return launch_fused_mode<P, NT, MMAX, TWC_LDS, 0, 16>(ch, n_frames, wpf, pstride, d_H, d_map, s);
        if (mode == 1) return launch_fused_mode<P, NT, MMAX, TWC_LDS, 1, 16>(ch, n_frames, wpf, pstride, d_H, d_map, s);
        return ch->power_rows1 ? launch_fused_mode<P, NT, MMAX, TWC_LDS, 3, 16, true>(ch, n_frames, wpf, pstride, d_H, d_map, s)
                               : launch_fused_mode<P, NT, MMAX, TWC_LDS, 3, 16, false>(ch, n_frames, wpf, pstride, d_H, d_map, s);
    }
    if (mode == 0) return launch_fused_mode<P, NT, MMAX, TWC_LDS, 0, 0>(ch, n_frames, wpf, pstride, d_H, d_map, s);
    if (mode == 1) return launch_fused_mode<P, NT, MMAX, TWC_LDS, 1, 0>(ch, n_frames, wpf, pstride, d_H, d_map, s);
    return ch->power_rows1 ? launch_fused_mode<P, NT, MMAX, TWC_LDS, 3, 0, true>(ch, n_frames, wpf, pstride, d_H, d_map, s)
                           : launch_fused_mode<P, NT, MMAX, TWC_LDS, 3, 0, false>(ch, n_frames, wpf, pstride, d_H, d_map, s);
}

template <int P>
static int launch_fused(jrc_chain* ch, int mode, int n_frames, int wpf, int pstride, const float2* d_H, float2* d_map, hipStream_t s)
{
    // small frames: 256-thread workgroups, up to three per CU; large frames (H fills most of the LDS): one
    // 512-thread workgroup per CU
    if (ch->threads == 1024) return launch_fused_nt<P, 1024, 16, true>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
    if (ch->threads == 512 && ch->cfg.fft_len <= 256) return launch_fused_nt<P, 512, 4, false>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
    if (ch->threads == 512) return launch_fused_nt<P, 512, 16, true>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
    if (ch->cfg.fft_len > 256) return launch_fused_nt<P, 256, 16, true>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
    return launch_fused_nt<P, 256, 4, false>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
}

// the window pass of the map-less modes: 2 dr rows per frame from the range profiles in d_rng into d_win
static int launch_window_rows(jrc_chain* ch, int n_frames, int pstride, hipStream_t s)
{
    const int rows = 2 * ch->win_dr;
    if (rows <= 0) return JRC_OK;
#define JRC_WIN_CASE(PP) case PP: hipLaunchKernelGGL(ra_window_rows_kernel<PP>, dim3(n_frames), dim3(256), 0, s, ch->d_rng, ch->d_partials, pstride, ch->d_win, \
                                                     ch->twA, ch->NR, ch->cfg.interp_angle, rows, ch->NR / 2 - ch->win_dr, ch->wide ? RW_L : RA_L); break;
    switch (ch->P) {
        JRC_WIN_CASE(1) JRC_WIN_CASE(2) JRC_WIN_CASE(4) JRC_WIN_CASE(8)
        default: hipLaunchKernelGGL(ra_window_rows_kernel<16>, dim3(n_frames), dim3(256), 0, s, ch->d_rng, ch->d_partials, pstride, ch->d_win, ch->twA,
                                    ch->NR, ch->cfg.interp_angle, rows, ch->NR / 2 - ch->win_dr, ch->wide ? RW_L : RA_L);
    }
#undef JRC_WIN_CASE
    JRC_HIP(ch->ctx, hipGetLastError());
    return JRC_OK;
}

static int launch_fused_any(jrc_chain* ch, int mode, int n_frames, int wpf, int pstride, const float2* d_H, float2* d_map, hipStream_t s)
{
    if (ch->wide) {
#define JRC_WIDE_CASE(PP, LG, THREADS)                                                                              \
        { if (mode == 0) return launch_fused_wide<PP, 0, LG, THREADS>(ch, n_frames, wpf, pstride, d_H, d_map, s);    \
          if (mode == 1) return launch_fused_wide<PP, 1, LG, THREADS>(ch, n_frames, wpf, pstride, d_H, d_map, s);    \
          return launch_fused_wide<PP, 3, LG, THREADS>(ch, n_frames, wpf, pstride, d_H, d_map, s); }
        const int lg = jrc_ilog2(ch->cfg.fft_len);
        if (ch->P == 16) {
            if (lg == 10) JRC_WIDE_CASE(16, 10, 512)
            if (lg == 9) JRC_WIDE_CASE(16, 9, 256)
            JRC_WIDE_CASE(16, 8, 256)
        }
        if (lg == 10) JRC_WIDE_CASE(8, 10, 256)
        if (lg == 9) JRC_WIDE_CASE(8, 9, 256)
        JRC_WIDE_CASE(8, 8, 256)
#undef JRC_WIDE_CASE
    }
    switch (ch->P) {
        case 1: return launch_fused<1>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
        case 2: return launch_fused<2>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
        case 4: return launch_fused<4>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
        case 8: return launch_fused<8>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
        default: return launch_fused<16>(ch, mode, n_frames, wpf, pstride, d_H, d_map, s);
    }
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
    const bool fused_ok = jrc_is_pow2(N) && N >= RA_L && N <= 1024 && jrc_is_pow2(cfg->interp_range) && jrc_is_pow2(P) &&
                          P <= 16 && jrc_is_pow2(cfg->interp_angle) && cfg->interp_angle >= 2 && cfg->interp_angle <= 64 &&
                          NA >= 4 && (P * N) % 2 == 0 &&
                          sizeof(float2) * ((size_t)P * N + (size_t)P * RA_L + (N > 256 ? (size_t)N : 0)) + 64 <= 160 * 1024;
    // everything else runs block by block (A1, pad, A2, A3, A4, A5 kernels): powers of two up to 16384, any other size up
    // to 4096 (chirp-z fft_vcc), e.g. 3 TX x 2 RX -> 96 angle bins
    auto size_ok = [](long n) { return n >= 1 && (jrc_is_pow2(n) ? n <= 16384 : n <= 4096); };
    const bool generic_ok = size_ok(NR) && size_ok(NA);
    if (NR * NA >= (1L << 32) || (!fused_ok && !generic_ok))
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED,
                        "radar chain: transform sizes fft_len*interp_range and N_tx*N_rx*interp_angle must be <= 16384 (powers of "
                        "two) or <= 4096 (other sizes); got N=%d P=%d Ir=%d Ia=%d",
                        N, P, cfg->interp_range, cfg->interp_angle);
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    jrc_chain* ch = new jrc_chain();
    ch->ctx = ctx; ch->cfg = *cfg; ch->P = P; ch->NR = (int)NR; ch->NA = (int)NA; ch->C = (int)(NR / RA_L);
    ch->max_frames = max_frames;
    ch->generic = !fused_ok || getenv("JRC_CHAIN_GENERIC") != nullptr;
    ch->lds_bytes = sizeof(float2) * ((size_t)P * N + (size_t)P * RA_L + (N > 256 ? (size_t)N : 0));
    ch->threads = (ch->lds_bytes > 80 * 1024) ? 512 : 256;
    if (getenv("JRC_THREADS")) { int t = atoi(getenv("JRC_THREADS")); if (t == 512 || (t == 1024 && N > 256)) ch->threads = t; else ch->threads = 256; }
    // range_angle_wide_kernel for 8 or 16 pairs x interp_angle 16 (the flowgraphs' 4 x 2 and the benchmark's 4 x 4) at fft_len 256 / 512 / 1024
    // (JRC_NO_WIDE: the 64-bin kernel; fft_len 64 / 128 were measured on it as well — zero inputs beyond fft_len — and stay on the 64-bin
    // kernel: 4 x 2 at fft_len 64 0.60-0.62 against 0.58-0.59, but 4 x 4 at fft_len 64 / 128 0.44 / 0.72 against 0.66 / 0.74).  Two 256-thread workgroups per CU while a lane's share of H
    // stays within 64 VGPRs (pairs per wave x fold terms <= 8), else one 512-thread workgroup (16 pairs at fft_len 1024).  Measured against
    // the 64-bin kernel, of the HBM peak (docs/history.md §3.1): 16 pairs, fft_len 256, interp_range 4 / 8 / 16 / 32: 0.75 / 0.82 / 0.82 / 0.82
    // against 0.72 / 0.80 / 0.60 / 0.65; fft_len 512 with 4 / 8: 0.78 / 0.80 against 0.74 / 0.71; 8 pairs, interp_range 8: fft_len 256 0.76
    // against 0.65, fft_len 1024 0.68-0.75 against 0.56.
    ch->wide = !ch->generic && (P == 16 || P == 8) && cfg->interp_angle == 16 && !getenv("JRC_NO_WIDE") && (N == 256 || N == 512 || N == 1024) && NR >= RW_L;
    if (ch->wide) {               // no H in LDS: two buffers of range bins
        const int fold = N >= RW_L ? N / RW_L : 1;
        ch->threads = (P / 4) * fold <= 8 ? 256 : 512;
        ch->lds_bytes = sizeof(float2) * 2 * (size_t)P * RW_L;
        ch->C = (int)(NR / RW_L);
    }
    ch->wpf_override = getenv("JRC_WPF") ? atoi(getenv("JRC_WPF")) : 0;
    ch->slices = getenv("JRC_DETECT_SLICES") ? atoi(getenv("JRC_DETECT_SLICES")) : 1;
    if (ch->slices < 1) ch->slices = 1;
    if (ch->slices > jrc_chain::kMaxSlices) ch->slices = jrc_chain::kMaxSlices;
    {
        ch->n_cus = ctx->n_cus;
        int by_lds = (int)((160 * 1024) / (ch->lds_bytes + 64));
        int by_regs = ch->threads == 256 ? 3 : 1;          // __launch_bounds__ of the kernel variants
        ch->wg_per_cu = by_lds < by_regs ? by_lds : by_regs;
        if (ch->wg_per_cu < 1) ch->wg_per_cu = 1;
        if (ch->wg_per_cu > 2) ch->wg_per_cu = 2;          // measured: 2 long-lived workgroups per CU beat 3-4 short ones
        if (getenv("JRC_WG_PER_CU")) { const int v = atoi(getenv("JRC_WG_PER_CU")); if (v >= 1 && v <= by_lds && v <= by_regs) ch->wg_per_cu = v; }
    }
    ch->win_dr = NR >= 2 ? (int)(cfg->noise_discard_range_m / (range_bins[1] - range_bins[0])) : 0;   // :189, float arithmetic as on the device
    if (ch->generic) {
        if (!generic_ok) { delete ch; return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "JRC_CHAIN_GENERIC: transform sizes out of range"); }
        ch->C = 1;
        long nb = ((long)NR * NA + 2047) / 2048;
        ch->gen_blocks = (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
    }
    hipError_t e = hipMalloc((void**)&ch->d_bins, sizeof(float) * (size_t)(NR + NA));
    if (e == hipSuccess) e = hipMemcpy(ch->d_bins, range_bins, sizeof(float) * NR, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ch->d_bins + NR, angle_bins, sizeof(float) * NA, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&ch->d_partials, sizeof(PeakPartial) * (size_t)max_frames * (ch->generic ? ch->gen_blocks : ch->C));
    if (e == hipSuccess && ch->generic) e = hipMalloc((void**)&ch->d_pad, sizeof(float2) * (size_t)max_frames * P * NR);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ch->h_pinned, sizeof(jrc_ra_result) * (size_t)max_frames, hipHostMallocDefault);
    int st = JRC_OK;
    if (e != hipSuccess) st = jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_create: %s", hipGetErrorString(e));
    if (st == JRC_OK && !ch->generic) st = jrc_get_twiddles(ctx, (int)NR, +1, &ch->twR);
    if (st == JRC_OK && !ch->generic) st = jrc_get_twiddles(ctx, (int)NA, -1, &ch->twA);
    if (st != JRC_OK) { jrc_chain_destroy(ch); return st; }
    *out = ch;
    return JRC_OK;
}

extern "C" void jrc_chain_destroy(jrc_chain* ch)
{
    if (!ch) return;
    (void)hipSetDevice(ch->ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& e : ch->ev) (void)hipEventDestroy(e);
    for (auto& e : ch->ev_a1) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; k++) {
        if (ch->ev_ready[k]) (void)hipEventDestroy(ch->ev_ready[k]);
        if (ch->ev_copied[k]) (void)hipEventDestroy(ch->ev_copied[k]);
        if (ch->h_ring[k]) (void)hipHostFree(ch->h_ring[k]);
    }
    if (ch->copy_stream) (void)hipStreamDestroy(ch->copy_stream);
    for (int k = 0; k < 2; k++) {
        if (ch->ev_side[k]) (void)hipEventDestroy(ch->ev_side[k]);
        if (ch->side[k]) (void)hipStreamDestroy(ch->side[k]);
    }
    if (ch->d_bins) (void)hipFree(ch->d_bins);
    if (ch->d_partials) (void)hipFree(ch->d_partials);
    if (ch->d_pad) (void)hipFree(ch->d_pad);
    if (ch->d_win) (void)hipFree(ch->d_win);
    if (ch->d_rng) (void)hipFree(ch->d_rng);
    if (ch->d_raw) (void)hipFree(ch->d_raw);
    bg_release(ch->bg);
    if (ch->h_pinned) (void)hipHostFree(ch->h_pinned);
    delete ch;
}

extern "C" size_t jrc_chain_frame_bytes(const jrc_chain* ch)
{
    return ch ? sizeof(float2) * (size_t)(ch->cfg.N_tx + ch->cfg.N_rx) * ch->cfg.n_items * ch->cfg.fft_len : 0;
}
extern "C" size_t jrc_chain_chanest_bytes(const jrc_chain* ch) { return ch ? sizeof(float2) * (size_t)ch->P * ch->cfg.fft_len : 0; }
extern "C" size_t jrc_chain_map_bytes(const jrc_chain* ch)
{
    return ch ? (ch->map_format == JRC_MAP_POWER ? sizeof(float) : sizeof(float2)) * (size_t)ch->NR * ch->NA : 0;
}

// how many launches of the dominant kernel one jrc_chain_run_dev of n_frames makes (batches beyond one resident wave of
// workgroups are launched in chunks; bench.py reports the roofline per launch)
extern "C" int jrc_chain_launches_per_run(const jrc_chain* ch, int n_frames)
{
    if (!ch || n_frames <= 0) return JRC_ERR_INVALID_ARG;
    if (ch->generic) return 1;
    const int resident = ch->n_cus * ch->wg_per_cu;
    const int wpf = chain_pick_wpf(ch, n_frames < resident ? n_frames : resident);
    const int chunk = chain_chunk(ch, wpf);
    return (n_frames + chunk - 1) / chunk;
}

extern "C" int jrc_chain_set_timing(jrc_chain* ch, int enabled)
{
    if (!ch) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    JRC_BIND(ctx);
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
    JRC_BIND(ctx);
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

// one batch through the background state: est = raw - mean(buffer before the frame) when removal is on, then the pushes of the
// batch (:297-300) and the new radar_chan_est_temp (:276-279).  est == nullptr: state update only (jrc_chain_prime_background_dev).
static int chain_background_step(jrc_chain* ch, int n_frames, const float2* raw, float2* est, hipStream_t s)
{
    jrc_ctx* ctx = ch->ctx;
    jrc_bg_state* b = ch->bg;
    const int pn = (int)b->pn;
    if (b->pending) JRC_HIP(ctx, hipStreamWaitEvent(s, b->updated, 0));      // the batch before this one may be on another stream
    const dim3 blk(256);
    if (b->removal) {
        if (n_frames > 65535) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "background removal: at most 65535 frames per batch");
        if (est)
            hipLaunchKernelGGL(chain_background_kernel, dim3((pn + 255) / 256, n_frames), blk, 0, s, raw, est, b->hist[b->cur], b->temp, pn, b->count,
                               b->record_len, b->recording);
        const int n1 = b->record_len < b->count + n_frames ? b->record_len : b->count + n_frames;
        if (n1 > 0) {
            hipLaunchKernelGGL(chain_background_push_kernel, dim3((pn + 255) / 256, n1), blk, 0, s, raw, b->hist[b->cur], b->hist[b->cur ^ 1], b->temp,
                               pn, b->count, n_frames, n1, b->recording);
            b->cur ^= 1;
        }
        b->count = n1;
    }
    JRC_HIP(ctx, hipGetLastError());
    if (b->recording)      // radar_chan_est_temp <- this frame's raw estimate (:276-279): after the batch it holds the last frame's
        JRC_HIP(ctx, hipMemcpyAsync(b->temp, raw + (size_t)(n_frames - 1) * pn, sizeof(float2) * pn, hipMemcpyDeviceToDevice, s));
    JRC_HIP(ctx, hipEventRecord(b->updated, s));
    b->pending = true;
    return JRC_OK;
}

// the raw-estimate buffer of a chain that subtracts the background.  The switch (bg->removal) belongs to the shared state, so any chain of
// the group may have turned it on: each chain makes sure of its own buffer where it is about to use it (a recording-only owner has none)
static int chain_ensure_raw(jrc_chain* ch)
{
    if (ch->d_raw) return JRC_OK;
    hipError_t e = hipMalloc((void**)&ch->d_raw, sizeof(float2) * (size_t)ch->P * ch->cfg.fft_len * (size_t)ch->max_frames);
    if (e != hipSuccess) return jrc_fail(ch->ctx, JRC_ERR_HIP, "background removal: %s", hipGetErrorString(e));
    return JRC_OK;
}

extern "C" int jrc_chain_set_background(jrc_chain* ch, int background_removal, int background_recording, int record_len)
{
    if (!ch || record_len < 0) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    JRC_BIND(ctx);
    const size_t pn = (size_t)ch->P * ch->cfg.fft_len;
    if (!ch->bg) {
        if (!background_removal && !background_recording) return JRC_OK;
        jrc_bg_state* b = new jrc_bg_state();
        b->pn = pn; b->record_len = record_len;
        const size_t hb = sizeof(float2) * pn * (size_t)(record_len ? record_len : 1);
        hipError_t e = hipMalloc((void**)&b->hist[0], hb);
        if (e == hipSuccess) e = hipMalloc((void**)&b->hist[1], hb);
        if (e == hipSuccess) e = hipMalloc((void**)&b->temp, sizeof(float2) * pn);
        if (e == hipSuccess) e = hipMemset(b->temp, 0, sizeof(float2) * pn);            // vector::resize value-initialises (:115)
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);                         // the memset is on the null stream, the chain's work on non-blocking streams: no order between them otherwise
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->updated, hipEventDisableTiming);
        if (e != hipSuccess) { bg_release(b); return jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_set_background: %s", hipGetErrorString(e)); }
        ch->bg = b;
    } else if (record_len != ch->bg->record_len) {
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_set_background: record_len is fixed once the state exists (%d)", ch->bg->record_len);
    }
    JRC_TRY(chain_ensure_raw(ch));             // whoever holds the state may see removal switched on through a chain that shares it: no allocation on the run path
    ch->bg->removal = background_removal != 0;
    ch->bg->recording = background_recording != 0;
    return JRC_OK;
}

// several chains, one radar stream (the slots of a feed): `ch` uses — and advances — the background state of `owner`
extern "C" int jrc_chain_share_background(jrc_chain* ch, jrc_chain* owner)
{
    if (!ch || !owner || !owner->bg || ch == owner) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (ch->P != owner->P || ch->cfg.fft_len != owner->cfg.fft_len || ch->ctx->device != owner->ctx->device)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_share_background: chains of different shape or device");
    JRC_BIND(ctx);
    if (!ch->d_raw) {
        hipError_t e = hipMalloc((void**)&ch->d_raw, sizeof(float2) * owner->bg->pn * (size_t)ch->max_frames);
        if (e != hipSuccess) return jrc_fail(ctx, JRC_ERR_HIP, "jrc_chain_share_background: %s", hipGetErrorString(e));
    }
    bg_release(ch->bg);
    ch->bg = owner->bg;
    ch->bg->refs++;
    return JRC_OK;
}

extern "C" int jrc_chain_background_size(const jrc_chain* ch) { return ch ? (ch->bg ? ch->bg->count : 0) : JRC_ERR_INVALID_ARG; }

// A1 + state update only, outputs dropped: what a GPU replays in front of its block of a sharded stream (the
// <= record_len estimates at the block boundary), and what `background_recording` without consumers amounts to
extern "C" int jrc_chain_prime_background_dev(jrc_chain* ch, int n_frames, const jrc_cf32* d_frames, void* stream)
{
    if (!ch || !d_frames) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (!ch->bg || !ch->bg->removal) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_prime_background_dev: background removal is not enabled");
    if (n_frames <= 0 || n_frames > ch->max_frames) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_prime_background_dev: bad n_frames");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const jrc_chain_cfg& c = ch->cfg;
    ChanestGeom g;
    g.N = c.fft_len; g.S = c.N_sym;
    g.port_stride = (long)c.n_items * c.fft_len;
    g.frame_stride = g.port_stride * (c.N_tx + c.N_rx);
    g.tx_item0 = c.N_pre; g.rx_item0 = c.N_pre; g.interleave = c.enable_tx_interleave;
    JRC_TRY(chain_ensure_raw(ch));
    JRC_TRY(launch_radar_chanest(ctx, c.N_tx, c.N_rx, (const float2*)d_frames, ch->d_raw, g, n_frames, s));
    return chain_background_step(ch, n_frames, ch->d_raw, nullptr, s);
}

// the estimator's window buffer, needed whenever no complex map is stored (detect-only and power-map modes)
static int chain_need_window(jrc_chain* ch, const char* what)
{
    jrc_ctx* ctx = ch->ctx;
    if (ch->generic) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "%s: this shape runs block by block and needs the complex map", what);
    if (2 * ch->win_dr > ch->NR || ch->win_dr < 0)
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "%s: the noise window (%d range bins) is larger than the map", what, 2 * ch->win_dr);
    JRC_BIND(ctx);
    if (!ch->d_win) {
        const size_t bytes = sizeof(float2) * (size_t)ch->max_frames * (size_t)(2 * ch->win_dr ? 2 * ch->win_dr : 1) * ch->NA;
        hipError_t e = hipMalloc((void**)&ch->d_win, bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&ch->d_rng, sizeof(float2) * (size_t)ch->max_frames * ch->NR * ch->P);
        if (e != hipSuccess) return jrc_fail(ctx, JRC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    }
    return JRC_OK;
}

// detect-only mode: write_map = 0 -> jrc_chain_run_dev ignores d_map (may be NULL) and stores no map; results are bit-identical
extern "C" int jrc_chain_set_write_map(jrc_chain* ch, int write_map)
{
    if (!ch) return JRC_ERR_INVALID_ARG;
    if (write_map) { ch->write_map = true; return JRC_OK; }
    JRC_TRY(chain_need_window(ch, "detect-only mode"));
    ch->write_map = false;
    return JRC_OK;
}

// map format: JRC_MAP_COMPLEX (the estimator's input, default) or JRC_MAP_POWER (float |z|^2, the heat-map branch's input)
extern "C" int jrc_chain_set_map_format(jrc_chain* ch, int format)
{
    if (!ch || (format != JRC_MAP_COMPLEX && format != JRC_MAP_POWER)) return JRC_ERR_INVALID_ARG;
    if (format == JRC_MAP_COMPLEX) { ch->map_format = format; return JRC_OK; }
    JRC_TRY(chain_need_window(ch, "power-map mode"));
    // a tile of floats per wave on top of the fused kernel's LDS: all rows of a trip (64 P floats) when that fits, else one row (NA floats)
    const size_t waves = (size_t)ch->threads / 64, cap = 160 * 1024 - 256;
    const size_t full = ch->lds_bytes + sizeof(float) * waves * RA_L * ch->P, one = ch->lds_bytes + sizeof(float) * waves * ch->NA;
    if (full <= cap) { ch->lds_power = full; ch->power_rows1 = false; }
    else if (one <= cap) { ch->lds_power = one; ch->power_rows1 = true; }
    else return jrc_fail(ch->ctx, JRC_ERR_UNSUPPORTED, "power-map mode: no LDS left for the store tile at this shape (%zu bytes needed)", one);
    ch->map_format = format;
    return JRC_OK;
}

// A1 comes in two forms: frequency-domain frames (d_frames), or TX rows + time-domain RX streams (A6 + A7 + A1 fused)
// A1 of frames [f0, f0 + nf) of a batch on stream s (frequency-domain frames, or the RX side in the time domain when d_frames is NULL)
static int chain_a1_stage(jrc_chain* ch, int f0, int nf, const jrc_cf32* d_frames, const jrc_cf32* d_tx, const jrc_cf32* d_rx_td, int cp_len,
                          long rx_stream_len, jrc_cf32* d_chanest, hipStream_t s)
{
    jrc_ctx* ctx = ch->ctx;
    const jrc_chain_cfg& c = ch->cfg;
    float2* out = (float2*)d_chanest + (size_t)f0 * ch->P * c.fft_len;
    if (d_frames) {
        ChanestGeom g;
        g.N = c.fft_len; g.S = c.N_sym;
        g.port_stride = (long)c.n_items * c.fft_len;
        g.frame_stride = g.port_stride * (c.N_tx + c.N_rx);
        g.tx_item0 = c.N_pre; g.rx_item0 = c.N_pre; g.interleave = c.enable_tx_interleave;
        return launch_radar_chanest(ctx, c.N_tx, c.N_rx, (const float2*)d_frames + (size_t)f0 * g.frame_stride, out, g, nf, s);
    }
    DemodGeom g;
    g.N = c.fft_len; g.cp = cp_len; g.S = c.N_sym; g.R = c.N_rx; g.logn = 0;
    g.tx_port_stride = (long)c.n_items * c.fft_len; g.tx_frame_stride = g.tx_port_stride * c.N_tx;
    g.rx_stream_stride = rx_stream_len; g.rx_frame_stride = rx_stream_len * c.N_rx;
    g.tx_item0 = c.N_pre; g.rx_sym0 = c.N_pre; g.interleave = c.enable_tx_interleave; g.blocks_per_frame = 1;
    return launch_demod_chanest(ctx, c.N_tx, (const float2*)d_tx + (size_t)f0 * g.tx_frame_stride, (const float2*)d_rx_td + (size_t)f0 * g.rx_frame_stride,
                                out, g, nf, s);
}

// A2 + A3 + A4 + A5 of frames [f0, f0 + nf) on stream s.  The per-frame work buffers (partial maxima, range profiles, window rows) are
// addressed from the slice's first frame: a slice's partial maxima start at f0 x C (the most slices a frame can have), whatever stride the
// slice itself uses, so slices never overlap.  ev2: recorded between the transforms and the estimator epilogue.
static int chain_transform_stage(jrc_chain* ch, int f0, int nf, const jrc_cf32* d_chanest_all, jrc_cf32* d_map_all, jrc_ra_result* d_results_all,
                                 hipStream_t s, hipEvent_t ev2)
{
    jrc_ctx* ctx = ch->ctx;
    const jrc_chain_cfg& c = ch->cfg;
    const float2* d_chanest = (const float2*)d_chanest_all + (size_t)f0 * ch->P * c.fft_len;
    const bool power = ch->map_format == JRC_MAP_POWER;
    float2* d_map = !d_map_all ? nullptr
                  : (power ? reinterpret_cast<float2*>(reinterpret_cast<float*>(d_map_all) + (size_t)f0 * ch->NR * ch->NA)
                           : (float2*)d_map_all + (size_t)f0 * ch->NR * ch->NA);
    jrc_ra_result* d_results = d_results_all + f0;
    // the launch helpers address the work buffers from their base: hand them the slice's view for the duration of this call
    struct view {
        jrc_chain* ch; PeakPartial* p; float2* r; float2* w; float2* pad;
        explicit view(jrc_chain* c_) : ch(c_), p(c_->d_partials), r(c_->d_rng), w(c_->d_win), pad(c_->d_pad) {}
        ~view() { ch->d_partials = p; ch->d_rng = r; ch->d_win = w; ch->d_pad = pad; }
    } keep(ch);
    int partials_per_frame;
    if (ch->generic) {
        ch->d_partials += (size_t)f0 * ch->gen_blocks;
        if (ch->d_pad) ch->d_pad += (size_t)f0 * ch->P * ch->NR;
        // A1 output -> zero pad -> A2 fft_vxx reverse (NR) -> A3 matrix_transpose -> A4 fft_vxx forward+shift (NA), in place
        const size_t rows = (size_t)nf * ch->P;
        unsigned pb = (unsigned)((rows * ch->NR + 255) / 256); if (pb > 8192) pb = 8192;
        hipLaunchKernelGGL(pad_rows_kernel, dim3(pb), dim3(256), 0, s, d_chanest, ch->d_pad, c.fft_len, ch->NR, rows);
        JRC_HIP(ctx, hipGetLastError());
        JRC_TRY(launch_fft_vcc(ctx, ch->NR, 0, 0, nullptr, rows, ch->d_pad, ch->d_pad, ch->NR, 0, s));
        int tr = jrc_matrix_transpose_dev(ctx, ch->NR, ch->P, c.interp_angle, ch->P, (size_t)nf, (const jrc_cf32*)ch->d_pad, (jrc_cf32*)d_map, (void*)s);
        if (tr < 0) return tr;
        JRC_TRY(launch_fft_vcc(ctx, ch->NA, 1, 1, nullptr, (size_t)nf * ch->NR, (const float2*)d_map, (float2*)d_map, ch->NA, 0, s));
        if (ev2) JRC_HIP(ctx, hipEventRecord(ev2, s));
        for (int f = 0; f < nf; f++)
            JRC_TRY(launch_ra_partial(ctx, (const float2*)d_map + (size_t)f * ch->NR * ch->NA, (size_t)ch->NR * ch->NA,
                                      ch->d_partials + (size_t)f * ch->gen_blocks, ch->gen_blocks, s));
        partials_per_frame = ch->gen_blocks;
    } else {
        // slices per frame of a full chunk, and of the last (smaller) chunk, which is the most any frame of this batch gets
        const int resident = ch->n_cus * ch->wg_per_cu;
        const int wpf = chain_pick_wpf(ch, nf < resident ? nf : resident);
        const int chunk = chain_chunk(ch, wpf);
        const int tail = nf % chunk;
        const int pstride = tail ? chain_pick_wpf(ch, tail) : wpf;
        // (base f0 x C, the per-frame size d_partials was allocated with: slices run side by side on two streams and each picks its own
        // pstride <= C from its own frame count, so a base of f0 x pstride let a later slice with a SMALLER stride land inside an earlier one)
        ch->d_partials += (size_t)f0 * ch->C;
        if (ch->d_rng) ch->d_rng += (size_t)f0 * ch->NR * ch->P;
        if (ch->d_win) ch->d_win += (size_t)f0 * 2 * ch->win_dr * ch->NA;
        if (pstride != wpf)      // frames of full chunks leave slots unused: all-ones = NaN power, never wins a merge
            JRC_HIP(ctx, hipMemsetAsync(ch->d_partials, 0xFF, sizeof(PeakPartial) * (size_t)nf * pstride, s));
        const int mode = !ch->write_map ? 1 : (power ? 3 : 0);
        JRC_TRY(launch_fused_any(ch, mode, nf, wpf, pstride, d_chanest, d_map, s));
        if (mode != 0)           // no complex map to read: the noise-window rows, through the same angle-axis code, into the compact window buffer
            JRC_TRY(launch_window_rows(ch, nf, pstride, s));
        if (ev2) JRC_HIP(ctx, hipEventRecord(ev2, s));
        partials_per_frame = pstride;
    }
    // rest of A5
    RaParams prm;
    prm.vlen = ch->NA; prm.n_inputs = ch->NR; prm.n_range_bins = ch->NR; prm.n_angle_bins = ch->NA;
    prm.noise_discard_range_m = c.noise_discard_range_m; prm.noise_discard_angle_deg = c.noise_discard_angle_deg;
    if (ch->write_map && !power)
        JRC_TRY(launch_ra_finalize(ctx, (const float2*)d_map, (size_t)ch->NR * ch->NA, ch->d_partials, partials_per_frame, prm, ch->d_bins,
                                   ch->d_bins + ch->NR, d_results, nf, 0, s));
    else
        JRC_TRY(launch_ra_finalize(ctx, ch->d_win, (size_t)2 * ch->win_dr * ch->NA, ch->d_partials, partials_per_frame, prm, ch->d_bins,
                                   ch->d_bins + ch->NR, d_results, nf, 2 * ch->win_dr, s));
    return JRC_OK;
}

// how many slices the detect-only pipeline cuts a batch into (1 = the kernels one after the other on the caller's stream: the default).
// Measured at round 4 (tools/detect_slices.sh, profiles/r04_detect_slices.txt): per 512 config-B frames 0.219 ms unsliced against 0.232 / 0.270 /
// 0.377 ms for 2 / 4 / 8 slices (config D, 256 frames: 0.691 against 0.773 / 0.760 / 1.057) — the detect kernel's two workgroups per CU hold
// 2 x 216 of a SIMD's 512 VGPRs, A1's waves (104) do not fit beside them, and an A1 form that does (two symbols in flight, 72 VGPRs:
// JRC_CHANEST_U2) runs at 0.110 ms alone and gains nothing beside it either, while every slice pays its own launch ramp, a smaller pruning
// horizon and two cross-stream event hand-offs.  The pipeline stays behind JRC_DETECT_SLICES for the record and is covered by the switch tests.
static int chain_pick_slices(const jrc_chain* ch, int n_frames)
{
    if (ch->slices <= 1) return 1;
    int n = ch->slices;
    const int nx = ch->ctx->n_xcd;
    while (n > 1 && n_frames / n < nx) n--;
    return n < 1 ? 1 : n;
}

static int chain_pipeline_resources(jrc_chain* ch)
{
    jrc_ctx* ctx = ch->ctx;
    if (ch->side[0]) return JRC_OK;
    for (int k = 0; k < 2; k++) {
        JRC_HIP(ctx, hipStreamCreateWithFlags(&ch->side[k], hipStreamNonBlocking));
        JRC_HIP(ctx, hipEventCreateWithFlags(&ch->ev_side[k], hipEventDisableTiming));
    }
    for (auto& e : ch->ev_a1) JRC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return JRC_OK;
}

static int chain_run(jrc_chain* ch, int n_frames, const jrc_cf32* d_frames, const jrc_cf32* d_tx, const jrc_cf32* d_rx_td, int cp_len,
                     long rx_stream_len, jrc_cf32* d_chanest, jrc_cf32* d_map, jrc_ra_result* d_results, void* stream)
{
    JRC_TRACE("jrc_chain_run");
    jrc_ctx* ctx = ch->ctx;
    if (n_frames <= 0 || n_frames > ch->max_frames)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_run_dev: n_frames %d outside (0, %d]", n_frames, ch->max_frames);
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const jrc_chain_cfg& c = ch->cfg;
    if (!d_frames && (cp_len < 0 || rx_stream_len < (long)c.n_items * (c.fft_len + cp_len)))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_run_td_dev: rx_stream_len %ld shorter than n_items*(fft_len+cp_len)", rx_stream_len);
    if ((!ch->write_map || ch->map_format != JRC_MAP_COMPLEX) && ch->generic) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "detect-only / power-map mode needs the fused kernel");
    if (ch->write_map && !d_map) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_run_dev: d_map is NULL but the chain stores the map (jrc_chain_set_write_map)");
    jrc_bg_state* bg = ch->bg;
    const bool bg_on = bg && (bg->removal || bg->recording);
    // Detect-only mode, no per-kernel timing, no background state (its kernels walk the whole batch), JRC_DETECT_SLICES > 1: the batch runs as a
    // pipeline of slices — A1 of slice i+1 on the caller's stream beside A2..A5 of slice i on a side stream (A1 is HBM-bound, the detect kernel
    // instruction-issue-bound, docs/history.md §3.6).  The caller's stream waits for the side streams before this call returns control of it: the
    // stream-order contract of the entry point is unchanged.  Off by default: measured slower than the kernels in series (chain_pick_slices).
    int n_slices = 1;
    if (!ch->write_map && !ch->generic && !ch->timing && !bg_on) {
        n_slices = chain_pick_slices(ch, n_frames);
        if (n_slices > 1) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) n_slices = 1;   // a capturing caller keeps one stream
        }
    }
    if (n_slices > 1) {
        JRC_TRY(chain_pipeline_resources(ch));
        const int nx = ctx->n_xcd;
        int per = (n_frames + n_slices - 1) / n_slices;
        per = ((per + nx - 1) / nx) * nx;                   // slices in whole XCD groups of frames, a smaller one last
        int i = 0;
        bool used[2] = {false, false};
        for (int f0 = 0; f0 < n_frames; f0 += per, i++) {
            const int nf = n_frames - f0 < per ? n_frames - f0 : per;
            JRC_TRY(chain_a1_stage(ch, f0, nf, d_frames, d_tx, d_rx_td, cp_len, rx_stream_len, d_chanest, s));
            JRC_HIP(ctx, hipEventRecord(ch->ev_a1[i], s));
            hipStream_t t = ch->side[i & 1];
            JRC_HIP(ctx, hipStreamWaitEvent(t, ch->ev_a1[i], 0));
            JRC_TRY(chain_transform_stage(ch, f0, nf, d_chanest, d_map, d_results, t, nullptr));
            used[i & 1] = true;
        }
        for (int k = 0; k < 2; k++)
            if (used[k]) {
                JRC_HIP(ctx, hipEventRecord(ch->ev_side[k], ch->side[k]));
                JRC_HIP(ctx, hipStreamWaitEvent(s, ch->ev_side[k], 0));
            }
        return JRC_OK;
    }
    hipEvent_t* ev = nullptr;
    if (ch->timing) {
        if (ch->ev_used == jrc_chain::kPool) JRC_TRY(chain_drain_events(ch));
        ev = &ch->ev[(size_t)ch->ev_used * 4];
        ch->ev_used++;
        JRC_HIP(ctx, hipEventRecord(ev[0], s));
    }
    // A1 (into the raw-estimate buffer when the background mean is subtracted afterwards)
    const bool bg_sub = bg && bg->removal;
    jrc_cf32* d_est_out = d_chanest;
    if (bg_sub) {
        if (!ch->d_raw) {                              // every chain of a background group gets the buffer when it joins (set / share): only a state handed over by hand ends up here
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
                return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_run_dev: background removal was switched on while this stream is capturing; run the chain once outside the capture");
            JRC_TRY(chain_ensure_raw(ch));
        }
        d_chanest = (jrc_cf32*)ch->d_raw;
    }
    JRC_TRY(chain_a1_stage(ch, 0, n_frames, d_frames, d_tx, d_rx_td, cp_len, rx_stream_len, d_chanest, s));
    if (bg_on) {
        JRC_TRY(chain_background_step(ch, n_frames, (const float2*)d_chanest, (float2*)d_est_out, s));
        d_chanest = d_est_out;
    }
    if (ev) JRC_HIP(ctx, hipEventRecord(ev[1], s));
    JRC_TRY(chain_transform_stage(ch, 0, n_frames, d_chanest, d_map, d_results, s, ev ? ev[2] : nullptr));
    if (ev) JRC_HIP(ctx, hipEventRecord(ev[3], s));
    return JRC_OK;
}

extern "C" int jrc_chain_run_dev(jrc_chain* ch, int n_frames, const jrc_cf32* d_frames, jrc_cf32* d_chanest,
                                 jrc_cf32* d_map, jrc_ra_result* d_results, void* stream)
{
    if (!ch || !d_frames || !d_chanest || !d_results) return JRC_ERR_INVALID_ARG;
    return chain_run(ch, n_frames, d_frames, nullptr, nullptr, 0, 0, d_chanest, d_map, d_results, stream);
}

extern "C" int jrc_chain_run_td_dev(jrc_chain* ch, int n_frames, const jrc_cf32* d_tx, const jrc_cf32* d_rx_td, int cp_len,
                                    long rx_stream_len, jrc_cf32* d_chanest, jrc_cf32* d_map, jrc_ra_result* d_results, void* stream)
{
    if (!ch || !d_tx || !d_rx_td || !d_chanest || !d_results) return JRC_ERR_INVALID_ARG;
    return chain_run(ch, n_frames, nullptr, d_tx, d_rx_td, cp_len, rx_stream_len, d_chanest, d_map, d_results, stream);
}

extern "C" int jrc_chain_fetch_results(jrc_chain* ch, int n_frames, const jrc_ra_result* d_results,
                                       jrc_ra_result* h_results, void* stream)
{
    if (!ch || !d_results || !h_results) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (n_frames <= 0 || n_frames > ch->max_frames) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_fetch_results: bad n_frames");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    JRC_HIP(ctx, hipMemcpyAsync(ch->h_pinned, d_results, sizeof(jrc_ra_result) * (size_t)n_frames, hipMemcpyDeviceToHost, s));
    JRC_HIP(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < n_frames; i++) {
        h_results[i] = ch->h_pinned[i];
        ra_finish_host(&h_results[i], ch->cfg.snr_threshold, ch->cfg.power_threshold);
    }
    return JRC_OK;
}

extern "C" int jrc_chain_fetch_results_begin(jrc_chain* ch, int n_frames, const jrc_ra_result* d_results, void* stream)
{
    if (!ch || !d_results) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (n_frames <= 0 || n_frames > ch->max_frames) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_fetch_results_begin: bad n_frames");
    if (ch->ring_count == 2) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_fetch_results_begin: two copies already in flight (call _end first)");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (!ch->copy_stream) {
        // all-or-nothing: everything is created into locals and handed to the chain only when complete, so a failed allocation leaves the
        // chain as it was and the next call tries again (a half-made set behind a non-null copy_stream used to fail every later call)
        hipStream_t cs = nullptr;
        jrc_ra_result* hr[2] = {nullptr, nullptr};
        hipEvent_t er[2] = {nullptr, nullptr}, ec[2] = {nullptr, nullptr};
        hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; k++) {
            e = hipHostMalloc((void**)&hr[k], sizeof(jrc_ra_result) * (size_t)ch->max_frames, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&er[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ec[k], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            for (int k = 0; k < 2; k++) {
                if (ec[k]) (void)hipEventDestroy(ec[k]);
                if (er[k]) (void)hipEventDestroy(er[k]);
                if (hr[k]) (void)hipHostFree(hr[k]);
            }
            if (cs) (void)hipStreamDestroy(cs);
            JRC_HIP(ctx, e);
        }
        for (int k = 0; k < 2; k++) { ch->h_ring[k] = hr[k]; ch->ev_ready[k] = er[k]; ch->ev_copied[k] = ec[k]; }
        ch->copy_stream = cs;
    }
    const int slot = (ch->ring_head + ch->ring_count) & 1;
    JRC_HIP(ctx, hipEventRecord(ch->ev_ready[slot], s));
    JRC_HIP(ctx, hipStreamWaitEvent(ch->copy_stream, ch->ev_ready[slot], 0));
    JRC_HIP(ctx, hipMemcpyAsync(ch->h_ring[slot], d_results, sizeof(jrc_ra_result) * (size_t)n_frames, hipMemcpyDeviceToHost, ch->copy_stream));
    JRC_HIP(ctx, hipEventRecord(ch->ev_copied[slot], ch->copy_stream));
    ch->ring_n[slot] = n_frames;
    ch->ring_count++;
    return JRC_OK;
}

extern "C" int jrc_chain_fetch_results_end(jrc_chain* ch, jrc_ra_result* h_results, int* n_frames)
{
    if (!ch || !h_results) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = ch->ctx;
    if (ch->ring_count == 0) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "jrc_chain_fetch_results_end: no copy in flight");
    JRC_BIND(ctx);
    const int slot = ch->ring_head;
    JRC_HIP(ctx, hipEventSynchronize(ch->ev_copied[slot]));
    const int n = ch->ring_n[slot];
    for (int i = 0; i < n; i++) {
        h_results[i] = ch->h_ring[slot][i];
        ra_finish_host(&h_results[i], ch->cfg.snr_threshold, ch->cfg.power_threshold);
    }
    if (n_frames) *n_frames = n;
    ch->ring_head ^= 1;
    ch->ring_count--;
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// Row D: range-Doppler map.  NO REFERENCE COUNTERPART — the reference collapses the symbol axis
// (lib/mimo_ofdm_radar_impl.cc:271-274) and has no FFT over symbols; BASELINE.json lists the configuration, so this is
// the build's own definition (oracle: numpy, tests/test_gpu_chain.py), parity unpinned by construction:
//   D[p][sym][sc] = rx_r[sym][sc] * conj(tx_t[sym][sc])                      (the summand of A1, p = r*T+t)
//   RD[p][k][d]   = fftshift_d FFT_{S*Id}( IFFT_{N*Ir}( D[p][.][.] zero-padded )[.][k] )   (unnormalised, like fft_vxx)
// Built from the same kernels as the generic chain: product+pad, Stockham FFT over range, tiled transpose+pad,
// Stockham FFT (+shift) over Doppler.
__global__ void rd_product_pad_kernel(const float2* __restrict__ frames, float2* __restrict__ out, ChanestGeom g, int T, int R,
                                      int NR, size_t total)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % NR);
        size_t q = i / NR;
        const int sym = (int)(q % g.S); q /= g.S;
        const int p = (int)(q % (T * R));
        const size_t f = q / (T * R);
        float2 v = make_float2(0.f, 0.f);
        if (k < g.N) {
            const int r = g.interleave ? p % R : p / T, t = g.interleave ? p / R : p % T;
            const float2* fb = frames + f * g.frame_stride;
            const float2 a = fb[(size_t)(T + r) * g.port_stride + (size_t)(g.rx_item0 + sym) * g.N + k];
            const float2 b = fb[(size_t)t * g.port_stride + (size_t)(g.tx_item0 + sym) * g.N + k];
            v = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
        }
        out[i] = v;
    }
}

// ---- row D, fused: Doppler first (compact), range last (streamed) ------------------------------------------------------------
// The 2-D transform is separable, so the order is chosen for traffic: (1) rd_product_t_kernel forms D and transposes it to
// [pair][subcarrier][symbol] (zero-padded to ND), (2) the stock FFT runs over the symbol axis in place (forward, shifted) — both on
// S*Id*N points per pair, 1/Ir of the output — and (3) range_doppler_fused_kernel does the zero-padded range IFFT exactly like the
// range-angle kernel does (twiddled fold to 64 points + 64-point inverse FFT across the wavefront, one residue class of range bins
// at a time) for a tile of 16 Doppler bins held in LDS, and streams the [range][Doppler] map out once, non-temporally.
#define RD_DT 16   // Doppler bins per workgroup tile (= one 128-byte segment of an output row)

__global__ __launch_bounds__(256) void rd_product_t_kernel(const float2* __restrict__ frames, float2* __restrict__ Dt, ChanestGeom g, int T, int R, int ND,
                                                           size_t z0)
{
    __shared__ float2 tile[64][65];
    const int P = T * R;
    const size_t fp = z0 + blockIdx.z;
    const int p = (int)(fp % P);
    const size_t f = fp / P;
    const int r = g.interleave ? p % R : p / T, t = g.interleave ? p / R : p % T;
    const float2* fb = frames + f * g.frame_stride;
    const float2* rxp = fb + (size_t)(T + r) * g.port_stride + (size_t)g.rx_item0 * g.N;
    const float2* txp = fb + (size_t)t * g.port_stride + (size_t)g.tx_item0 * g.N;
    const int n0 = blockIdx.x * 64, s0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int ss = ly; ss < 64; ss += 4) {
        const int sym = s0 + ss, n = n0 + lx;
        float2 v = make_float2(0.f, 0.f);
        if (sym < g.S && n < g.N) {
            const float2 a = rxp[(size_t)sym * g.N + n], b = txp[(size_t)sym * g.N + n];
            v = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);               // rx * conj(tx)
        }
        tile[ss][lx] = v;
    }
    __syncthreads();
    float2* dst = Dt + fp * (size_t)g.N * ND;
    for (int ll = ly; ll < 64; ll += 4) {
        const int n = n0 + ll, sym = s0 + lx;
        if (n < g.N && sym < ND) dst[(size_t)n * ND + sym] = tile[lx][ll];
    }
}

// Product and Doppler transform as one kernel, for ND = S * Id in {32, 64, 128, 256}: a workgroup takes 4096 / ND subcarriers of one
// (frame, pair), forms rx conj(tx) symbol by symbol (coalesced along the subcarriers) into a [subcarrier][symbol] tile in LDS — zero-padded to
// ND — and transforms every row in two in-register passes, ND = 16 * R2 with n = R2 a + b, k = k1 + 16 k2:
//     X[k1 + 16 k2] = sum_b W_R2^(b k2) W_ND^(b k1) ( sum_a x[R2 a + b] W_16^(a k1) )
// (a 16-point transform per (row, b), in place; then an R2-point transform per (row, k1), stored fftshifted as 128-byte pieces of row
// [subcarrier] of E).  The compact array is written once and never read back by this stage: 402 MB instead of 938 MB of traffic per 16
// config-D frames against rd_product_t_kernel + the stock FFT in place.
template <int ND>
__global__ __launch_bounds__(256) void rd_product_doppler_kernel(const float2* __restrict__ frames, float2* __restrict__ E, ChanestGeom g, int T, int R,
                                                                 const float2* __restrict__ twD /* exp(-j 2 pi i / ND) */, int nblk)
{
    constexpr int R2 = ND / 16, SCB = 4096 / ND, PITCH = ND + R2 + 1, LY = 256 / SCB;
    static_assert(ND == 32 || ND == 64 || ND == 128 || ND == 256, "16 x {2, 4, 8, 16} points");
    __shared__ float2 tile[SCB * PITCH];
    __shared__ float2 s_tw[ND];
    const int tid = threadIdx.x;
    const int P = T * R;
    const size_t fp = blockIdx.x / (unsigned)nblk;
    const int n0 = (int)(blockIdx.x % (unsigned)nblk) * SCB;
    const int p = (int)(fp % P);
    const size_t f = fp / P;
    const int r = g.interleave ? p % R : p / T, t = g.interleave ? p / R : p % T;
    const float2* fb = frames + f * g.frame_stride;
    const float2* rxp = fb + (size_t)(T + r) * g.port_stride + (size_t)g.rx_item0 * g.N;
    const float2* txp = fb + (size_t)t * g.port_stride + (size_t)g.tx_item0 * g.N;
    for (int i = tid; i < ND; i += 256) s_tw[i] = twD[i];
    {   // all of a lane's loads go out before the first product: no branch around a load (symbols past S and subcarriers past fft_len repeat the
        // last one and are zeroed afterwards), so ND / LY = 16 symbols x two streams are in flight per lane
        constexpr int NIT = ND / LY;
        const int lx = tid % SCB, ly = tid / SCB, n = n0 + lx, nc = n < g.N ? n : g.N - 1;
        float2 a[NIT], b[NIT];
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int sym = ly + i * LY, sc = sym < g.S ? sym : g.S - 1;
            a[i] = rxp[(size_t)sc * g.N + nc];
            b[i] = txp[(size_t)sc * g.N + nc];
        }
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int sym = ly + i * LY;
            const bool on = sym < g.S && n < g.N;
            tile[lx * PITCH + sym] = on ? make_float2(a[i].x * b[i].x + a[i].y * b[i].y, a[i].y * b[i].x - a[i].x * b[i].y)      // rx * conj(tx)
                                        : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    {   // 16-point transforms over a, one (row, b) per lane, in place; twiddled for the second pass
        const int b = tid % R2, col = tid / R2;
        float2* x0 = tile + col * PITCH + b;
        float2 x[16];
#pragma unroll
        for (int a = 0; a < 16; a++) x[a] = x0[R2 * a];
        fft_fwd_small<16>(x);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++) x[k1] = cmul(x[k1], s_tw[b * k1]);
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) x0[R2 * k1] = x[k1];
    }
    __syncthreads();
#pragma unroll 1
    for (int task = tid; task < SCB * 16; task += 256) {                   // R2-point transforms over b, one (row, k1) per lane
        const int k1 = task & 15, col = task >> 4;
        const float2* y0 = tile + col * PITCH + R2 * k1;
        float2 y[R2];
#pragma unroll
        for (int b = 0; b < R2; b++) y[b] = y0[b];
        fft_fwd_small<R2>(y);
        if (n0 + col < g.N) {
            float2* dst = E + (fp * (size_t)g.N + n0 + col) * ND;
#pragma unroll
            for (int k2 = 0; k2 < R2; k2++) dst[(k1 + 16 * k2 + ND / 2) & (ND - 1)] = y[k2];     // fftshift; cached: the range kernel reads E next
        }
    }
}

template <int ND>
static int launch_rd_product_doppler(jrc_ctx* ctx, const float2* frames, float2* E, const ChanestGeom& g, int T, int R, size_t fp, hipStream_t s)
{
    const float2* twD = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, ND, -1, &twD));
    const int nblk = (g.N + 4096 / ND - 1) / (4096 / ND);
    hipLaunchKernelGGL(rd_product_doppler_kernel<ND>, dim3((unsigned)(fp * nblk)), dim3(256), 0, s, frames, E, g, T, R, twD, nblk);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

template <int NT, int MMAX, bool TWC_LDS>
__global__ __launch_bounds__(NT) void range_doppler_fused_kernel(const float2* __restrict__ E,     // [units/(ND/16)][N][ND]
                                                                 float2* __restrict__ out,         // [units/(ND/16)][NR][ND]
                                                                 const float2* __restrict__ twR, int N, int NR, int ND, long n_units, int WPF, int nx)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    constexpr int NW = NT / 64, P = RD_DT;
    const int C = NR / RA_L;
    const int xcd = blockIdx.x % nx;
    const long j = blockIdx.x / nx;
    const long u = (j / WPF) * nx + xcd;
    const int slice = (int)(j % WPF);
    if (u >= n_units) return;
    const int tiles = ND / RD_DT;
    const long fp = u / tiles;
    const int d0 = (int)(u % tiles) * RD_DT;

    float2* s_H = smem;                         // [16][N]: the tile's Doppler bins, subcarrier-contiguous
    float2* s_g = s_H + (size_t)P * N;          // [16][64] range bins of the current class
    float2* s_twc = s_g + P * RA_L;             // [N] class twiddles (TWC_LDS only)
    constexpr int NPT = TWC_LDS ? 4 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = N / RA_L;

    float2 tc[TWC_LDS ? 1 : MMAX];
    float2 tn[NPT];
    if constexpr (TWC_LDS) {
#pragma unroll
        for (int q = 0; q < NPT; q++) { const int n = tid + NT * q; if (n < N) tn[q] = twR[(n * slice) & (NR - 1)]; }
    } else {
#pragma unroll
        for (int m = 0; m < MMAX; m++)
            if (m < M) tc[m] = twR[((lane + RA_L * m) * slice) & (NR - 1)];
    }
    {   // stage the tile: row n of E holds the 16 bins as one 128-byte segment; eight lanes fetch it as 16-byte pieces
        const float2* Eb = E + (size_t)fp * N * ND + d0;
        for (int i = tid; i < N * 8; i += NT) {
            const int n = i >> 3, part = i & 7;
            const float4 v = *reinterpret_cast<const float4*>(Eb + (size_t)n * ND + 2 * part);
            s_H[(size_t)(2 * part) * N + n] = make_float2(v.x, v.y);
            s_H[(size_t)(2 * part + 1) * N + n] = make_float2(v.z, v.w);
        }
    }
    float2 t64[6];
#pragma unroll
    for (int st = 0; st < 6; st++) {
        const int half = 32 >> st;
        t64[st] = twR[((lane & (half - 1)) * (32 / half)) * (NR / 64)];
    }
    float2* outp = out + (size_t)fp * NR * ND + d0;
    typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll 1
    for (int c = slice; c < C; c += WPF) {
        if constexpr (TWC_LDS) {
#pragma unroll
            for (int q = 0; q < NPT; q++) { const int n = tid + NT * q; if (n < N) s_twc[n] = tn[q]; }
        }
        __syncthreads();
#pragma unroll 1
        for (int p = wave; p < P; p += NW) {    // range axis: fold to 64 points, 64-point inverse FFT across the wavefront
            const float2* Hp = s_H + (size_t)p * N + lane;
            float2 v = make_float2(0.f, 0.f);
#pragma unroll
            for (int m = 0; m < MMAX; m++)
                if (m < M) {
                    float2 w;
                    if constexpr (TWC_LDS) w = s_twc[lane + RA_L * m]; else w = tc[m];
                    const float2 h = Hp[RA_L * m];
                    v.x = fmaf(h.x, w.x, fmaf(-h.y, w.y, v.x));
                    v.y = fmaf(h.x, w.y, fmaf(h.y, w.x, v.y));
                }
#pragma unroll
            for (int st = 0; st < 6; st++) {
                const int half = 32 >> st;
                const float2 o = make_float2(__shfl_xor(v.x, half), __shfl_xor(v.y, half));
                v = (lane & half) ? cmul(csub(o, v), t64[st]) : cadd(v, o);
            }
            s_g[p * RA_L + (__brev((unsigned)lane) >> 26)] = v;
        }
        {
            const int cn = c + WPF;
            if (cn < C) {
                if constexpr (TWC_LDS) {
#pragma unroll
                    for (int q = 0; q < NPT; q++) { const int n = tid + NT * q; if (n < N) tn[q] = twR[(n * cn) & (NR - 1)]; }
                } else {
#pragma unroll
                    for (int m = 0; m < MMAX; m++)
                        if (m < M) tc[m] = twR[((lane + RA_L * m) * cn) & (NR - 1)];
                }
            }
        }
        __syncthreads();
        // 64 range bins x 16 Doppler bins: 16 lanes write one 128-byte segment of a map row
        for (int w0 = tid; w0 < RA_L * RD_DT; w0 += NT) {
            const int ql = w0 >> 4, di = w0 & 15;
            const float2 y = s_g[di * RA_L + ql];
            const v2f tv = {y.x, y.y};
            __builtin_nontemporal_store(tv, reinterpret_cast<v2f*>(outp + (size_t)(C * ql + c) * ND + di));
        }
    }
}

// Range axis as a pruned FFT, for fft_len = 64 ... 1024 (powers of two): output bin Ir*q + c of the zero-padded inverse transform is bin q of
// the fft_len-point inverse FFT of x[n] * exp(+j 2 pi n c / NR), so a workgroup keeps its tile (fft_len rows x 16 Doppler bins) in
// registers, and per residue c < Ir twiddles it into LDS, transforms it in place along the rows by decimation in frequency — radix-16
// passes in registers (one lane = one bin of one 16-point butterfly), then radix-4 / radix-2 passes on two-bin segments — and stores row
// i, which then holds the bin whose mixed-radix digits are those of i reversed, as the 128-byte line of its output row.  The tile is kept
// conjugated so that the inverse transform runs on the forward butterflies of fft_device.h.  5 log2(fft_len) + 6 flops per output
// instead of the fold's 8 fft_len / 64 + 30.
// row -> bin of the in-place decimation-in-frequency transform: row = sum_s d_s * N / (r_0 ... r_s) holds bin k = sum_s d_s * (r_0 ... r_{s-1})
template <int N, int R16, int R4, int R2>
__device__ __forceinline__ int rd_bin_of_row(int row)
{
    int rem = row, k = 0, span = N, mult = 1;
#pragma unroll
    for (int ps = 0; ps < R16; ps++) { span >>= 4; const int d = rem / span; rem -= d * span; k += d * mult; mult <<= 4; }
#pragma unroll
    for (int ps = 0; ps < R4; ps++) { span >>= 2; const int d = rem / span; rem -= d * span; k += d * mult; mult <<= 2; }
    if constexpr (R2 > 0) k += rem * mult;                                // the last digit (span 1)
    return k;
}

template <int NT /* == fft_len */>
__global__ __launch_bounds__(NT) void range_doppler_pruned_kernel(const float2* __restrict__ E,     // [units/(ND/16)][N][ND]
                                                                  float2* __restrict__ out,         // [units/(ND/16)][NR][ND]
                                                                  const float2* __restrict__ twR, int NR, int ND, long n_units, int WPF, int nx, int exp)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_t[];          // [N][8]: row n = 16 bins
    constexpr int N = NT, LOG2 = (NT == 64 ? 6 : (NT == 128 ? 7 : (NT == 256 ? 8 : (NT == 512 ? 9 : 10)))), RSTEP = NT / 8;
    // radix-16 passes, then radix-4 passes.  At 1024 threads (128 VGPRs) the 16-point butterfly next to the resident tile spills
    // (measured: 0.393 ms per 8 config-D frames against 0.345 ms with radix-4 passes only), so fft_len 1024 stays on radix 4; a
    // 512-thread variant with two 16-point butterflies per lane (243 VGPRs, three passes = six sweeps of the tile instead of ten)
    // measured the same as this one (0.316 against 0.319 ms): the sweeps are not what bounds it.
    constexpr int R16 = NT == 1024 ? 0 : LOG2 / 4;
    constexpr int R4 = (LOG2 - 4 * R16) / 2;
    constexpr int R2 = (LOG2 - 4 * R16) & 1;                              // fft_len 128, 512: one last radix-2 pass
    const int Ir = NR / N;
    const int xcd = blockIdx.x % nx;
    const long jb = blockIdx.x / nx;
    const long u = (jb / WPF) * nx + xcd;
    const int slice = (int)(jb % WPF);
    if (u >= n_units) return;
    const int tiles = ND / RD_DT;
    const long fp = u / tiles;
    const int d0 = (int)(u % tiles) * RD_DT;
    const int tid = threadIdx.x, seg = tid & 7, r0 = tid >> 3;           // this lane's rows: r0 + j * NT/8, its two bins: 2 seg, 2 seg + 1

    // The tile stays in registers in the layout of the FIRST pass's butterflies, which therefore runs straight from them (no LDS
    // round trip for the class twiddle): radix 16 — one bin of rows b0 + l N/16; radix 4 — two bins of rows r0 + j N/8 (butterfly h of
    // the lane takes j = h, h + 2, h + 4, h + 6).
    constexpr int NE = R16 > 0 ? 16 : 8;
    constexpr int ESTEP = R16 > 0 ? N / 16 : N / 8;
    const int bin = tid & 15, b0 = tid >> 4;
    const int erow0 = R16 > 0 ? b0 : r0;
    float4 e[8];            // radix-4 layout
    float2 e16[16];         // radix-16 layout
    if constexpr (R16 > 0) {
        const float2* Eb = E + (size_t)fp * N * ND + d0 + bin;
#pragma unroll
        for (int l = 0; l < 16; l++) e16[l] = Eb[(size_t)(b0 + l * ESTEP) * ND];
    } else {
        const float2* Eb = E + (size_t)fp * N * ND + d0 + 2 * seg;
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = *reinterpret_cast<const float4*>(Eb + (size_t)(r0 + j * RSTEP) * ND);
    }
    (void)NE;
    // pass twiddles: W[i] = exp(+j 2 pi i / N) in LDS (used conjugated)
    float2* s_f = reinterpret_cast<float2*>(s_t);                         // the tile as [N][16]
    float2* s_w = s_f + (size_t)N * 16;
    for (int i = tid; i < N; i += NT) s_w[i] = twR[(size_t)i * Ir];
    __syncthreads();
    // class twiddles exp(+j 2 pi row c / NR) of the lane's rows: the first row from the table, the others by the (uniform) step
    // exp(+j 2 pi ESTEP c / NR); both are fetched one class ahead
    float2 cw0 = twR[(erow0 * slice) & (NR - 1)], cstep = twR[(ESTEP * slice) & (NR - 1)];
    float2* outb = out + (size_t)fp * NR * ND + d0;
    float2* outp = outb + 2 * seg;
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    // which pass is the last one: it has span 1 (no twiddles) and stores its butterflies' outputs straight from registers — row i holds the
    // bin rd_bin_of_row(i) — so a class's 128 KiB (fft_len 1024) leave while the NEXT class's passes run: nothing waits for a store, and
    // the barrier that frees the tile for the next class's first pass sits between that pass's arithmetic and its LDS writes
    constexpr int LAST = R2 > 0 ? 2 : (R4 > 0 ? 4 : 16);
    static_assert(LAST != 16 || R16 >= 2, "a lone radix-16 pass would be first and last pass at once");
    static_assert(LAST != 4 || R16 > 0 || R4 >= 2, "a lone radix-4 pass would be first and last pass at once");
#pragma unroll 1
    for (int c = slice; c < Ir; c += WPF) {
        if constexpr (R16 > 0) {                                          // first pass: radix 16 from registers
            constexpr int q = N / 16;
            float2 v[16];
            {
                float2 w = cw0;
#pragma unroll
                for (int l = 0; l < 16; l++) { const float2 a = cmul(e16[l], w); v[l] = make_float2(a.x, -a.y); w = cmul(w, cstep); }   // conjugated
            }
            fft_fwd_small<16>(v);
            if (q > 1) {
#pragma unroll
                for (int m = 1; m < 16; m++) {
                    const float2 w = s_w[b0 * m];
                    v[m] = make_float2(v[m].x * w.x + v[m].y * w.y, v[m].y * w.x - v[m].x * w.y);
                }
            }
            __syncthreads();                                              // the previous class's last pass has read the tile
            float2* p0 = s_f + (size_t)b0 * 16 + bin;
#pragma unroll
            for (int m = 0; m < 16; m++) p0[(size_t)m * q * 16] = v[m];
        } else {                                                          // first pass: radix 4 from registers (fft_len 1024), one butterfly at a time
            constexpr int q = N / 4;
            __syncthreads();                                              // the previous class's last pass has read the tile
            const float2 cs2 = cmul(cstep, cstep);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int jj = r0 + h * RSTEP;
                float4 a[4];
                {
                    float2 w = h == 0 ? cw0 : cmul(cw0, cstep);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float4 ee = e[h + 2 * j];
                        const float2 x = cmul(make_float2(ee.x, ee.y), w), y = cmul(make_float2(ee.z, ee.w), w);
                        a[j] = make_float4(x.x, -x.y, y.x, -y.y);        // conjugated
                        w = cmul(w, cs2);
                    }
                }
                const float4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
                const float4 t0 = make_float4(a0.x + a2.x, a0.y + a2.y, a0.z + a2.z, a0.w + a2.w);
                const float4 t1 = make_float4(a0.x - a2.x, a0.y - a2.y, a0.z - a2.z, a0.w - a2.w);
                const float4 t2 = make_float4(a1.x + a3.x, a1.y + a3.y, a1.z + a3.z, a1.w + a3.w);
                const float4 t3 = make_float4(a1.x - a3.x, a1.y - a3.y, a1.z - a3.z, a1.w - a3.w);
                float4 y0 = make_float4(t0.x + t2.x, t0.y + t2.y, t0.z + t2.z, t0.w + t2.w);
                float4 y2 = make_float4(t0.x - t2.x, t0.y - t2.y, t0.z - t2.z, t0.w - t2.w);
                float4 y1 = make_float4(t1.x + t3.y, t1.y - t3.x, t1.z + t3.w, t1.w - t3.z);     // t1 - j t3 (forward)
                float4 y3 = make_float4(t1.x - t3.y, t1.y + t3.x, t1.z - t3.w, t1.w + t3.z);     // t1 + j t3
                const float2 w1 = s_w[jj], w2 = cmul(w1, w1), w3 = cmul(w2, w1);                 // used conjugated
                auto cmulc = [](float x, float y, float2 w) { return make_float2(x * w.x + y * w.y, y * w.x - x * w.y); };
                float2 v;
                v = cmulc(y1.x, y1.y, w1); y1.x = v.x; y1.y = v.y; v = cmulc(y1.z, y1.w, w1); y1.z = v.x; y1.w = v.y;
                v = cmulc(y2.x, y2.y, w2); y2.x = v.x; y2.y = v.y; v = cmulc(y2.z, y2.w, w2); y2.z = v.x; y2.w = v.y;
                v = cmulc(y3.x, y3.y, w3); y3.x = v.x; y3.y = v.y; v = cmulc(y3.z, y3.w, w3); y3.z = v.x; y3.w = v.y;
                float4* p0 = s_t + (size_t)jj * 8 + seg;
                p0[0] = y0; p0[(size_t)q * 8] = y1; p0[(size_t)2 * q * 8] = y2; p0[(size_t)3 * q * 8] = y3;
            }
        }
        if (c + WPF < Ir) { cw0 = twR[(erow0 * (c + WPF)) & (NR - 1)]; cstep = twR[(ESTEP * (c + WPF)) & (NR - 1)]; }
        __syncthreads();
#pragma unroll
        for (int ps = 1; ps < R16; ps++) {
            const int L = N >> (4 * ps), q = L >> 4;
            const int b = tid >> 4;                                       // N/16 butterflies x 16 bins: one per lane
            const int jj = b & (q - 1);
            float2* p0 = s_f + ((size_t)(b / q) * L + jj) * 16 + bin;
            float2 v[16];
#pragma unroll
            for (int l = 0; l < 16; l++) v[l] = p0[(size_t)l * q * 16];
            fft_fwd_small<16>(v);
            if ((exp & 2) && !(LAST == 16 && ps == R16 - 1)) continue;
            if (LAST == 16 && ps == R16 - 1) {                            // span 1: rows 16 b + m leave as eight-byte pieces, 16 lanes to a 128-byte line
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    const int k = rd_bin_of_row<N, R16, R4, R2>(16 * b + m);
                    const v2f t = {v[m].x, -v[m].y};
                    if (!(exp & 1)) __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(outb + ((size_t)Ir * k + c) * ND + bin));
                }
            } else {
                if (q > 1) {
#pragma unroll
                    for (int m = 1; m < 16; m++) {
                        const float2 w = s_w[(jj * m) << (4 * ps)];       // exp(+j 2 pi jj m / L); the forward pass needs its conjugate
                        v[m] = make_float2(v[m].x * w.x + v[m].y * w.y, v[m].y * w.x - v[m].x * w.y);
                    }
                }
#pragma unroll
                for (int m = 0; m < 16; m++) p0[(size_t)m * q * 16] = v[m];
                __syncthreads();
            }
        }
#pragma unroll
        for (int ps = (R16 > 0 ? 0 : 1); ps < R4; ps++) {                 // two-bin segments, two butterflies per lane
            const int L = (N >> (4 * R16)) >> (2 * ps), q = L >> 2;
            const bool last = LAST == 4 && ps == R4 - 1;                  // then q == 1
            if ((exp & 2) && !last) continue;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int b = (tid + h * NT) >> 3, jj = b & (q - 1);
                const int row0 = (b / q) * L + jj;
                float4* p0 = s_t + (size_t)row0 * 8 + seg;
                const float4 a0 = p0[0], a1 = p0[(size_t)q * 8], a2 = p0[(size_t)2 * q * 8], a3 = p0[(size_t)3 * q * 8];
                const float4 t0 = make_float4(a0.x + a2.x, a0.y + a2.y, a0.z + a2.z, a0.w + a2.w);
                const float4 t1 = make_float4(a0.x - a2.x, a0.y - a2.y, a0.z - a2.z, a0.w - a2.w);
                const float4 t2 = make_float4(a1.x + a3.x, a1.y + a3.y, a1.z + a3.z, a1.w + a3.w);
                const float4 t3 = make_float4(a1.x - a3.x, a1.y - a3.y, a1.z - a3.z, a1.w - a3.w);
                float4 y0 = make_float4(t0.x + t2.x, t0.y + t2.y, t0.z + t2.z, t0.w + t2.w);
                float4 y2 = make_float4(t0.x - t2.x, t0.y - t2.y, t0.z - t2.z, t0.w - t2.w);
                float4 y1 = make_float4(t1.x + t3.y, t1.y - t3.x, t1.z + t3.w, t1.w - t3.z);     // t1 - j t3 (forward)
                float4 y3 = make_float4(t1.x - t3.y, t1.y + t3.x, t1.z - t3.w, t1.w + t3.z);     // t1 + j t3
                if (last) {
                    const float4 ys[4] = {y0, y1, y2, y3};
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const int k = rd_bin_of_row<N, R16, R4, R2>(row0 + m);
                        const v4f t = {ys[m].x, -ys[m].y, ys[m].z, -ys[m].w};
                        if (!(exp & 1)) __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(outp + ((size_t)Ir * k + c) * ND));
                    }
                } else {
                    if (q > 1) {
                        const float2 w1 = s_w[jj * (N / L)], w2 = cmul(w1, w1), w3 = cmul(w2, w1);   // used conjugated
                        auto cmulc = [](float x, float y, float2 w) { return make_float2(x * w.x + y * w.y, y * w.x - x * w.y); };
                        float2 v;
                        v = cmulc(y1.x, y1.y, w1); y1.x = v.x; y1.y = v.y; v = cmulc(y1.z, y1.w, w1); y1.z = v.x; y1.w = v.y;
                        v = cmulc(y2.x, y2.y, w2); y2.x = v.x; y2.y = v.y; v = cmulc(y2.z, y2.w, w2); y2.z = v.x; y2.w = v.y;
                        v = cmulc(y3.x, y3.y, w3); y3.x = v.x; y3.y = v.y; v = cmulc(y3.z, y3.w, w3); y3.z = v.x; y3.w = v.y;
                    }
                    p0[0] = y0; p0[(size_t)q * 8] = y1; p0[(size_t)2 * q * 8] = y2; p0[(size_t)3 * q * 8] = y3;
                }
            }
            if (!last) __syncthreads();
        }
        if constexpr (R2 > 0) {                                           // span 2: no twiddles; two-bin segments, four butterflies per lane; the last pass
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const int row0 = ((tid + h * NT) >> 3) * 2;
                const float4* p0 = s_t + (size_t)row0 * 8 + seg;
                const float4 a0 = p0[0], a1 = p0[8];
                const v4f lo = {a0.x + a1.x, -(a0.y + a1.y), a0.z + a1.z, -(a0.w + a1.w)};
                const v4f hi = {a0.x - a1.x, -(a0.y - a1.y), a0.z - a1.z, -(a0.w - a1.w)};
                if (!(exp & 1)) __builtin_nontemporal_store(lo, reinterpret_cast<v4f*>(outp + ((size_t)Ir * rd_bin_of_row<N, R16, R4, R2>(row0) + c) * ND));
                if (!(exp & 1)) __builtin_nontemporal_store(hi, reinterpret_cast<v4f*>(outp + ((size_t)Ir * rd_bin_of_row<N, R16, R4, R2>(row0 + 1) + c) * ND));
            }
        }
    }
}

template <int NT>
static int launch_rd_pruned(jrc_ctx* ctx, const float2* E, float2* out, const float2* twR, int NR, int ND, long n_units, hipStream_t s)
{
    const size_t lds = sizeof(float4) * 8 * (size_t)NT + sizeof(float2) * NT;
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)range_doppler_pruned_kernel<NT>, lds));
    const int Ir = NR / NT;
    const long target = (long)ctx->n_cus * (NT == 1024 ? 1 : (NT == 512 ? 2 : 4));
    int wpf = 1;
    while (wpf * 2 <= Ir && (long)wpf * 2 * n_units <= target) wpf *= 2;
    const int nx = ctx->n_xcd;
    const long groups = (n_units + nx - 1) / nx;
    hipLaunchKernelGGL((range_doppler_pruned_kernel<NT>), dim3((unsigned)(groups * wpf * nx)), dim3(NT), lds, s, E, out, twR, NR, ND, n_units, wpf, nx, ctx->tune.rd_exp);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

template <int NT, int MMAX, bool TWC_LDS>
static int launch_rd_fused(jrc_ctx* ctx, const float2* E, float2* out, const float2* twR, int N, int NR, int ND, long n_units, size_t lds, hipStream_t s)
{
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)range_doppler_fused_kernel<NT, MMAX, TWC_LDS>, lds));
    const int C = NR / RA_L;
    const long target = (long)ctx->n_cus * (lds > 80 * 1024 ? 1 : 2);
    int wpf = 1;
    while (wpf * 2 <= C && (long)wpf * 2 * n_units <= target) wpf *= 2;
    const int nx = ctx->n_xcd;
    const long groups = (n_units + nx - 1) / nx;
    hipLaunchKernelGGL((range_doppler_fused_kernel<NT, MMAX, TWC_LDS>), dim3((unsigned)(groups * wpf * nx)), dim3(NT), lds, s, E, out, twR, N, NR, ND, n_units, wpf, nx);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

extern "C" int jrc_range_doppler_dev(jrc_ctx* ctx, const jrc_chain_cfg* c, int interp_doppler, int n_frames,
                                     const jrc_cf32* d_frames, jrc_cf32* d_work, jrc_cf32* d_out, void* stream)
{
    JRC_TRACE("jrc_range_doppler_dev");
    if (!ctx || !c || !d_frames || !d_work || !d_out || n_frames <= 0 || interp_doppler <= 0) return JRC_ERR_INVALID_ARG;
    const int N = c->fft_len, T = c->N_tx, R = c->N_rx, P = T * R, S = c->N_sym;
    const long NR = (long)N * c->interp_range, ND = (long)S * interp_doppler;
    auto size_ok = [](long n) { return n >= 1 && (jrc_is_pow2(n) ? n <= 16384 : n <= 4096); };
    if (!size_ok(NR) || !size_ok(ND) || c->n_items < c->N_pre + S)
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "range-Doppler: fft_len*interp_range and N_sym*interp_doppler must be <= 16384 (powers of two) or <= 4096");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    ChanestGeom g;
    g.N = N; g.S = S; g.port_stride = (long)c->n_items * N; g.frame_stride = g.port_stride * (T + R);
    g.tx_item0 = c->N_pre; g.rx_item0 = c->N_pre; g.interleave = c->enable_tx_interleave;
    // fused path: Doppler FFT on the compact data, zero-padded range IFFT streamed out once (d_work holds [pair][N][ND] <= its size)
    if (jrc_is_pow2(N) && N >= RA_L && N <= 1024 && jrc_is_pow2(NR) && NR >= RA_L && jrc_is_pow2(ND) && ND >= RD_DT && ND <= 8192 &&
        interp_doppler <= c->interp_range && ((reinterpret_cast<size_t>(d_work) & 15) == 0) && !ctx->tune.rd_generic) {
        // frames go through in chunks whose compact array stays in the Infinity Cache between the kernel that writes it and the one that reads it
        // (the chunks reuse the start of d_work; JRC_RD_CHUNK_MB, 0 = one chunk)
        {
            const size_t per_frame = sizeof(float2) * (size_t)P * N * ND;
            const long chunk = ctx->tune.rd_chunk_mb > 0 ? (long)(((size_t)ctx->tune.rd_chunk_mb << 20) / per_frame) : 0;
            if (chunk >= 1 && n_frames > chunk) {
                for (long f0 = 0; f0 < n_frames; f0 += chunk) {
                    const int nf = (int)(n_frames - f0 < chunk ? n_frames - f0 : chunk);
                    const int rc = jrc_range_doppler_dev(ctx, c, interp_doppler, nf, d_frames + (size_t)f0 * g.frame_stride, d_work,
                                                         d_out + (size_t)f0 * P * NR * ND, stream);
                    if (rc < 0) return rc;
                }
                return JRC_OK;
            }
        }
        const size_t fp = (size_t)n_frames * P;
        const bool one_kernel = !ctx->tune.rd_two_step && fp * ((size_t)N * ND / 4096 + 1) < 0x7fffffffull;
        if (one_kernel && ND == 128) { JRC_TRY(launch_rd_product_doppler<128>(ctx, (const float2*)d_frames, (float2*)d_work, g, T, R, fp, s)); }
        else if (one_kernel && ND == 64) { JRC_TRY(launch_rd_product_doppler<64>(ctx, (const float2*)d_frames, (float2*)d_work, g, T, R, fp, s)); }
        else if (one_kernel && ND == 256) { JRC_TRY(launch_rd_product_doppler<256>(ctx, (const float2*)d_frames, (float2*)d_work, g, T, R, fp, s)); }
        else if (one_kernel && ND == 32) { JRC_TRY(launch_rd_product_doppler<32>(ctx, (const float2*)d_frames, (float2*)d_work, g, T, R, fp, s)); }
        else {
            for (size_t z0 = 0; z0 < fp; z0 += 65535) {      // gridDim.z (frame, pair) is limited to 65535: chunks
                const size_t nz = fp - z0 < 65535 ? fp - z0 : 65535;
                hipLaunchKernelGGL(rd_product_t_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((ND + 63) / 64), (unsigned)nz), dim3(256), 0, s,
                                   (const float2*)d_frames, (float2*)d_work, g, T, R, (int)ND, z0);
            }
            JRC_HIP(ctx, hipGetLastError());
            JRC_TRY(launch_fft_vcc(ctx, (int)ND, 1, 1, nullptr, fp * N, (const float2*)d_work, (float2*)d_work, ND, 0, s));        // Doppler
        }
        const float2* twR = nullptr;
        JRC_TRY(jrc_get_twiddles(ctx, (int)NR, +1, &twR));
        const size_t lds = sizeof(float2) * ((size_t)RD_DT * N + (size_t)RD_DT * RA_L + (N > 256 ? (size_t)N : 0));
        const long n_units = (long)fp * (ND / RD_DT);
        if (!ctx->tune.rd_fold) {      // pruned FFT instead of the fold
            if (N == 1024) return launch_rd_pruned<1024>(ctx, (const float2*)d_work, (float2*)d_out, twR, (int)NR, (int)ND, n_units, s);
            if (N == 256) return launch_rd_pruned<256>(ctx, (const float2*)d_work, (float2*)d_out, twR, (int)NR, (int)ND, n_units, s);
            if (N == 512) return launch_rd_pruned<512>(ctx, (const float2*)d_work, (float2*)d_out, twR, (int)NR, (int)ND, n_units, s);
            if (N == 128) return launch_rd_pruned<128>(ctx, (const float2*)d_work, (float2*)d_out, twR, (int)NR, (int)ND, n_units, s);
            if (N == 64) return launch_rd_pruned<64>(ctx, (const float2*)d_work, (float2*)d_out, twR, (int)NR, (int)ND, n_units, s);
        }
        if (lds > 80 * 1024) return launch_rd_fused<512, 16, true>(ctx, (const float2*)d_work, (float2*)d_out, twR, N, (int)NR, (int)ND, n_units, lds, s);
        if (N > 256) return launch_rd_fused<256, 16, true>(ctx, (const float2*)d_work, (float2*)d_out, twR, N, (int)NR, (int)ND, n_units, lds, s);
        return launch_rd_fused<256, 4, false>(ctx, (const float2*)d_work, (float2*)d_out, twR, N, (int)NR, (int)ND, n_units, lds, s);
    }
    const size_t rows = (size_t)n_frames * P * S, total = rows * NR;
    unsigned pb = (unsigned)((total + 255) / 256); if (pb > 16384) pb = 16384;
    hipLaunchKernelGGL(rd_product_pad_kernel, dim3(pb), dim3(256), 0, s, (const float2*)d_frames, (float2*)d_work, g, T, R, (int)NR, total);
    JRC_HIP(ctx, hipGetLastError());
    JRC_TRY(launch_fft_vcc(ctx, (int)NR, 0, 0, nullptr, rows, (const float2*)d_work, (float2*)d_work, NR, 0, s));     // range
    int tr = jrc_matrix_transpose_dev(ctx, (int)NR, S, interp_doppler, S, (size_t)n_frames * P, d_work, d_out, (void*)s);
    if (tr < 0) return tr;
    JRC_TRY(launch_fft_vcc(ctx, (int)ND, 1, 1, nullptr, (size_t)n_frames * P * NR, (const float2*)d_out, (float2*)d_out, ND, 0, s));   // Doppler
    return JRC_OK;
}
