// jrc_internal.h — shared host-side state and device helpers for the gfx950 kernels behind include/jrc.h
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/jrc.h"

struct jrc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    // pinned staging for the host-buffer entry points (GNU Radio buffers are pageable ring buffers)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // device scratch for the host-buffer entry points
    void* scratch[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[4] = {0, 0, 0, 0};
    // twiddle tables exp(sign*j*2*pi*k/n), k < n, keyed by sign*n
    std::map<long, float2*> twiddles;
    // chirp-z tables for fft sizes that are not powers of two, keyed by sign*n
    struct bluestein_tab { float2* chirp; float2* bhat; int M; };
    std::map<long, bluestein_tab> bluestein;
    // dynamic-LDS opt-in granted so far, per kernel (hipFuncAttributeMaxDynamicSharedMemorySize is per device: kept per context)
    std::map<const void*, size_t> dyn_lds;
    int n_cus = 0;
    size_t max_lds_per_block = 64 * 1024;   // what a dynamic-LDS opt-in may ask for: 160 KB on gfx950, else hipDeviceAttributeMaxSharedMemoryPerBlock (at least 64 KB)
    // XCDs (L2 domains) workgroups are dealt over round-robin: 8 on an MI355X in SPX mode (256 CUs), 1 per 32 CUs in the smaller
    // partition modes; JRC_XCDS overrides.  Only locality depends on it (a frame's workgroups share one L2), never results.
    int n_xcd = 8;
    int wall_clock_khz = 100000;     // rate of wall_clock64() (hipDeviceAttributeWallClockRate): the unit of the store-pacing word
    // experiment switches, read once in jrc_create (environment JRC_*)
    struct {
        int chanest_chunk = 0;       // JRC_CHANEST_CHUNK: frames per A1 launch (0 = two workgroups per CU)
        bool chanest_x1 = false;     // JRC_CHANEST_X1: one subcarrier per lane in A1
        bool fd_serial = false;      // JRC_FD_SERIAL: single-wave detector scan
        bool sync_naive = false;     // JRC_SYNC_NAIVE: detection metrics without the LDS tile
        bool sync_tile = false;      // JRC_SYNC_TILE: the front end's peak mask from the one-sample-per-lane tile kernel
        bool sync_streams = false;   // JRC_SYNC_STREAMS: the front end writes the three metric streams and reads them back (the form before round 4)
        bool dec_single = false;     // JRC_DEC_SINGLE: the first-generation Viterbi decoder kernel (one frame per wave, LDS path ring)
        int dec_frames_per_wave = 0; // JRC_DEC_FPW: 1 or 2 frames per wave in the decoder (0 = by batch size)
        bool ra_ref_sum = false;     // JRC_RA_REF_SUM: the estimator's noise sum always by the reference-order double chain (tests)
        bool rd_generic = false;     // JRC_RD_GENERIC: range-Doppler block by block (stock FFTs + transpose)
        bool rd_fold = false;        // JRC_RD_FOLD: range-Doppler with the fold kernel also where the pruned-FFT kernel applies
        int ra_pace = -1;            // JRC_RA_PACE: store pacing word of the fused range-angle kernel (chain.hip; -1 = derived, chain_pace)
        bool rd_two_step = false;    // JRC_RD_TWO_STEP: range-Doppler product and Doppler FFT as two kernels also where the one-kernel form applies
        int rd_chunk_mb = 160;       // JRC_RD_CHUNK_MB: range-Doppler frames per pass = this many MiB of the compact [pair][subcarrier][Doppler] array, which then stays in the 256 MiB Infinity Cache between its two kernels (0: all frames at once)
        int rd_exp = 0;              // JRC_RD_EXP: range-Doppler pruned-FFT kernel experiments, TIMING ONLY, WRONG RESULTS: 1: no stores; 2: first and last pass only
        bool chanest_u2 = false;     // JRC_CHANEST_U2: A1 with two symbols in flight per lane instead of four (<= 80 VGPRs: a wave fits a SIMD beside two waves of the detect-only kernel)
        int detect_exp = 0;          // JRC_DETECT_EXP: detect-only kernel experiments (chain.hip, MODE 1). 8: no pruning; 16: sum bound only. TIMING ONLY, WRONG RESULTS: 1: no angle stage; 2: no range-profile stores; 32: sampled rows never computed
        double ra_offered_tbps = 0;  // JRC_RA_OFFERED_TBPS: offered store rate the pacing word is derived from (0 = the kernel's measured optimum)
        int demod_spr = 0;           // JRC_DEMOD_SPR: symbols per round (2 or 4) of the A6+A7+A1 kernel (0 = by fft_len)
        bool eq_sig_full = false;    // JRC_EQ_SIG_FULL: the equalizer's SIG decoder always runs its trellis (comm.hip sig_viterbi_wave: no codeword shortcut)
        int eq_wpe = 0;              // JRC_EQ_WPE: waves per SIMD the equalizer kernel is compiled for (2, 4, 6, 8; 0 = by geometry)
        int eq_threads = 0;          // JRC_EQ_THREADS: equalizer workgroup size (64, 128, 256; 0 = by launch size, -1 = one lane per subcarrier)
    } tune;
};

int  jrc_fail(jrc_ctx* ctx, int status, const char* fmt, ...);
int  jrc_ensure_pinned(jrc_ctx* ctx, size_t bytes);
// staging copy between pageable and pinned host memory: one thread moves ~25 GB/s, half of what the PCIe link takes, so
// copies of 4 MiB and more are split over up to four threads
void jrc_host_copy(void* dst, const void* src, size_t bytes);
int  jrc_ensure_scratch(jrc_ctx* ctx, int slot, size_t bytes);
// dynamic LDS above 64 KiB must be opted into per kernel: raises the kernel's limit to `bytes` if it is not there yet
int  jrc_ensure_dyn_lds(jrc_ctx* ctx, const void* kernel, size_t bytes);
// full-circle table of n entries: tw[k] = exp(sign * j * 2*pi * k / n), computed in double
int  jrc_get_twiddles(jrc_ctx* ctx, int n, int sign, const float2** out);
// chirp c[k] = exp(sign*j*pi*k^2/n) (k < n) and bhat = FFT_M(conj(c) wrapped to M)/M, M = 2^k >= 2n-1, computed in double
int  jrc_get_bluestein(jrc_ctx* ctx, int n, int sign, jrc_ctx::bluestein_tab* out);

// Tracing hook: with JRC_ROCTX=1 in the environment every batched entry point brackets itself with a roctx range
// (librocprofiler-sdk-roctx / libroctx64, loaded lazily with dlopen so the library has no link-time dependency on a profiler);
// `rocprofv3 --marker-trace` then shows the C-ABI calls above the kernels they launch.  Off: one predictable branch.
void jrc_trace_push(const char* name);
void jrc_trace_pop();
struct jrc_trace_range {
    bool on;
    explicit jrc_trace_range(const char* name);
    ~jrc_trace_range() { if (on) jrc_trace_pop(); }
};
#define JRC_TRACE(name) jrc_trace_range _jrc_trace_range_(name)

#define JRC_HIP(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return jrc_fail((ctx), JRC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                            hipGetErrorString(_e), __FILE__, __LINE__);                         \
    } while (0)

// hipSetDevice is per host thread (GNU Radio runs one thread per block, and a process may hold contexts on several GPUs): every
// entry point that allocates, copies or launches binds the calling thread to its context's GPU first
#define JRC_BIND(ctx) JRC_HIP((ctx), hipSetDevice((ctx)->device))

#define JRC_TRY(expr)                       \
    do {                                    \
        int _s = (expr);                    \
        if (_s < 0) return _s;              \
    } while (0)

static inline bool jrc_is_pow2(long n) { return n > 0 && (n & (n - 1)) == 0; }
static inline int  jrc_ilog2(long n) { int l = 0; while ((1L << l) < n) l++; return l; }

// ---- device helpers -----------------------------------------------------------------------
#ifdef __HIPCC__

// The 64 lanes of a wavefront execute every instruction together, and a wave's LDS operations complete in program order: a kernel may let all
// lanes READ a wave-private LDS region and then WRITE it again with nothing in between.  JRC_LOCKSTEP() marks the places that rely on this.
// On the device it expands to nothing (the binary is the one without the marker).  The CPU emulation the tests run without a GPU
// (tests/hipcpu: one fiber per lane, run one after the other) makes it a rendezvous of the wave's lanes, which is what lockstep guarantees.
#ifdef HIPCPU_EMULATION
#define JRC_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#else
#define JRC_LOCKSTEP() ((void)0)
#endif

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// |z|^2 exactly as the reference computes it: std::pow(std::abs(z), 2) with abs = glibc hypotf
// (= (float)sqrt((double)x*x + (double)y*y)) and pow(float,int) evaluated in double
// (lib/range_angle_estimator_impl.cc:141).  Returns the double before the final float rounding.
__device__ __forceinline__ double ref_power_f64(float2 z)
{
    double d = (double)z.x * (double)z.x + (double)z.y * (double)z.y;
    float h = (float)sqrt(d);
    return (double)h * (double)h;
}
// sin and cos of a float angle within 1.5 ulp / 9.3e-8 absolute over |a| < 2^15 (numpy emulation against float64; ocml sincosf: 2 ulp), for the
// de-rotation loops that take one per sample: Cody-Waite reduction by pi/2 in two fused steps (exact for |a| < 2^15: a - n * float(pi/2) has
// at most 23 significant bits below 1), the single-precision minimax polynomials of Cephes' sinf / cosf on [-pi/4, pi/4], quadrant from n.
// About 25 instructions; ocml's sincosf is several times that.  Larger angles and non-numbers go to sincosf.
__device__ __forceinline__ void jrc_sincosf_fast(float a, float* sn, float* cs)
{
    if (!(fabsf(a) < 32768.f)) { sincosf(a, sn, cs); return; }
    const float n = rintf(a * 0.636619772367581343f);
    float r = fmaf(n, -1.57079637050628662109375f, a);
    r = fmaf(n, 4.37113882867379223e-8f, r);
    const float z = r * r;
    const float s = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(z, -0.5f, 1.0f));
    const int q = (int)n;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

__device__ __forceinline__ float ref_hypotf(float2 z)
{
    double d = (double)z.x * (double)z.x + (double)z.y * (double)z.y;
    return (float)sqrt(d);
}

// Running first-arg-max with the reference's strict '>' semantics (ties -> lowest flat index = first in the
// reference's scan order).  The exact f64 power above is only evaluated for candidates whose cheap f32 power
// is within 1e-5 of a WAVE-wide running maximum: f32 and exact powers differ by < 4e-7 relative, so the true
// arg-max always passes the filter, while after the first few elements almost no lane does, and the
// divergent f64 path stays off the streaming critical path.
__device__ __forceinline__ float fast_power(float2 z) { return fmaf(z.x, z.x, z.y * z.y); }

__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

struct PeakTracker {
    float run_max;    // wave-uniform running maximum of the f32 powers
    float best;       // exact power of the best candidate seen by this lane
    unsigned idx;     // its flat index
    __device__ __forceinline__ void init() { run_max = -1.0f; best = -1.0f; idx = 0xffffffffu; }
    // all lanes of the wave call this together with their local f32 maximum; returns the filter threshold
    __device__ __forceinline__ float raise(float lane_max)
    {
        run_max = fmaxf(run_max, wave_max_f32(lane_max));
        return run_max * (1.0f - 1e-5f);
    }
    __device__ __forceinline__ void exact(float2 z, unsigned flat)
    {
        float pe = (float)ref_power_f64(z);
        if (pe > best || (pe == best && flat < idx)) { best = pe; idx = flat; }
    }
    __device__ __forceinline__ void merge(float obest, unsigned oidx)
    {
        if (obest > best || (obest == best && oidx < idx)) { best = obest; idx = oidx; }
    }
};

struct PeakPartial { float best; unsigned idx; };

// block-wide (256 threads max 1024) reduction of PeakTracker; result valid in thread 0
__device__ __forceinline__ void block_reduce_peak(PeakTracker& t, PeakPartial* smem /* >= nwaves */)
{
    for (int off = 32; off > 0; off >>= 1) {
        float ob = __shfl_xor(t.best, off);
        unsigned oi = __shfl_xor(t.idx, off);
        t.merge(ob, oi);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { smem[wave].best = t.best; smem[wave].idx = t.idx; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < nw; w++) t.merge(smem[w].best, smem[w].idx);
}

#endif  // __HIPCC__
