// sync.hip — the sync front-end of the comm receive chain on the device:
//   moving_avg      lib/moving_avg_impl.cc:62-98
//   frame_detector  lib/frame_detector_impl.cc:70-205
//   frame_sync      lib/frame_sync_impl.cc:89-289
// These are sample-serial state machines in the reference.  Here the per-sample arithmetic is data-parallel and the
// state machines only walk what is sparse:
//   * moving averages are window sums, one output per lane (no running sum, hence none of its float drift);
//   * the detector's per-sample test (threshold < cor < 2) becomes a bit mask built by every lane at once; one wave then
//     steps through the mask 64 samples per word — words without a peak bit are skipped whole unless a peak run is open —
//     and emits copy segments; the carrier de-rotation of the copied samples is a separate parallel kernel;
//   * frame_sync's LTF matched filter is one lane per lag, the top-4 search a few wave reductions, and its COPY state
//     (drop the cyclic prefixes, de-rotate) is a closed-form index map evaluated per lane.
// Scheduler-visible behaviour (items consumed / produced per call, tags, state carried across calls) follows the
// reference call for call.
#include "jrc_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

// ---- moving_avg ---------------------------------------------------------------------------------------------------
__global__ void moving_avg_kernel(const float2* __restrict__ in, float2* __restrict__ out, int length, float scale, int n)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float2 sum = in[i];
    for (int k = 1; k < length; k++) { sum.x = sum.x + in[i + k].x; sum.y = sum.y + in[i + k].y; }
    if (length == 1) { sum.x = in[0].x + in[i].x; sum.y = in[0].y + in[i].y; }   // the reference seeds its running sum with in[0] (:79): length 1 counts it twice
    out[i] = make_float2(sum.x * scale, sum.y * scale);
}

extern "C" int jrc_moving_avg_dev(jrc_ctx* ctx, int length, float scale, int n_out, const jrc_cf32* d_in, jrc_cf32* d_out, void* stream)
{
    if (!ctx || length < 1 || n_out < 0 || (n_out > 0 && (!d_in || !d_out))) return JRC_ERR_INVALID_ARG;
    if (n_out == 0) return 0;
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    hipLaunchKernelGGL(moving_avg_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, (const float2*)d_in, (float2*)d_out, length, scale, n_out);
    JRC_HIP(ctx, hipGetLastError());
    return n_out;
}

extern "C" int jrc_moving_avg(jrc_ctx* ctx, int length, float scale, int max_iter, int noutput_items, const jrc_cf32* in, jrc_cf32* out)
{
    if (!ctx || length < 1 || noutput_items < 0 || (noutput_items > 0 && (!in || !out))) return JRC_ERR_INVALID_ARG;
    const int n = noutput_items > max_iter ? max_iter : noutput_items;                           // :78
    if (n <= 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ib = sizeof(float2) * ((size_t)n + length - 1), ob = sizeof(float2) * (size_t)n;
    JRC_TRY(jrc_ensure_pinned(ctx, ib + ob));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, ib));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, ob));
    memcpy(ctx->pinned, in, ib);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, ib, hipMemcpyHostToDevice, ctx->stream));
    JRC_TRY(jrc_moving_avg_dev(ctx, length, scale, n, (const jrc_cf32*)ctx->scratch[0], (jrc_cf32*)ctx->scratch[1], ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + ib, ctx->scratch[1], ob, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, (char*)ctx->pinned + ib, ob);
    return n;
}

// ---- detection metrics of the flowgraph's stock blocks (delay, conj, multiply, moving averages, mag, abs, divide) -----
// xd[i] = x[i - delay];  in_abs[i] = sum_{j = i-window+1..i} x[j] conj(x[j - delay]);  in_cor[i] = |in_abs[i]| / |pscale sum_{pwindow} |x|^2|
__global__ void sync_metrics_kernel(const float2* __restrict__ x, int n, int delay, int window, int pwindow, float pscale,
                                    float2* __restrict__ xd, float2* __restrict__ in_abs, float* __restrict__ in_cor)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    xd[i] = i >= delay ? x[i - delay] : make_float2(0.f, 0.f);
    float2 a = make_float2(0.f, 0.f);
    for (int j = i - window + 1; j <= i; j++) {
        if (j < delay) continue;                                          // the delayed stream starts with zeros
        const float2 u = x[j], v = x[j - delay];                          // conj(v) * u
        a.x = a.x + (v.x * u.x + v.y * u.y);
        a.y = a.y + (v.x * u.y - v.y * u.x);
    }
    float p = 0.f;
    for (int j = i - pwindow + 1; j <= i; j++) {
        if (j < 0) continue;
        const float2 u = x[j];
        p = p + (u.x * u.x + u.y * u.y);
    }
    in_abs[i] = a;
    in_cor[i] = ref_hypotf(a) / fabsf(p * pscale);
}

// The same metrics from an LDS tile: 1024 outputs per workgroup; the products x[j] conj(x[j-delay]) and the powers |x[j]|^2 are formed
// once per sample, then summed in a tree — runs of 4, runs of 16 (4 runs of 4), and the window as its runs of 16, of 4 and single
// samples, oldest first.  Additions only (no running sum, no prefix differences: a 60 dB power step inside the tile costs nothing),
// ~12 LDS accesses per output instead of window + power_window global loads (80 at fft_len 64, 1280 at fft_len 1024).
// MARKS = true is the run-to-completion front end's form (jrc_sync_frontend_dev): nobody downstream of the detector's peak test reads the three
// streams there — the test `threshold < cor < MAX_PEAK_VALUE` (lib/frame_detector_impl.cc:95) is taken on the value in the register and one
// ballot word per 64 samples is all that reaches HBM (1 bit per sample written instead of 20 bytes, and no second kernel reading `cor` back);
// the correlation the detector needs at a detection is re-formed by the scanning wave with the same tree (sm_abs_at), the delayed samples the
// synchroniser copies are read from x at an offset.
#define SM_TILE 1024
template <bool MARKS>
__global__ __launch_bounds__(256) void sync_metrics_tiled_kernel(const float2* __restrict__ x, int n, int delay, int window, int pwindow, float pscale,
                                                                 float2* __restrict__ xd, float2* __restrict__ in_abs, float* __restrict__ in_cor,
                                                                 unsigned long long* __restrict__ marks, double thr, double maxv)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float2 sm_lds[];
    const int halo = (window > pwindow ? window : pwindow) - 1 + delay;
    const int L = SM_TILE + halo;
    float2* A = sm_lds;                    // samples, then runs of 4 of the products
    float2* B = A + L;                     // products
    float2* E = B + L;                     // runs of 16 of the products
    float* Cw = (float*)(E + L);           // powers
    float* Dw = Cw + L;                    // runs of 4 of the powers
    float* Fw = Dw + L;                    // runs of 16 of the powers
    const long i0 = (long)blockIdx.x * SM_TILE;
    const long g0 = i0 - halo;
    const int tid = threadIdx.x;
    for (int k = tid; k < L; k += 256) {
        const long g = g0 + k;
        A[k] = (g >= 0 && g < n) ? x[g] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    for (int k = tid; k < L; k += 256) {
        const long g = g0 + k;
        const float2 u = A[k];
        float2 pr = make_float2(0.f, 0.f);
        if (k >= delay && g >= delay) {                                   // the delayed stream starts with zeros
            const float2 v = A[k - delay];                                // conj(v) * u
            pr = make_float2(v.x * u.x + v.y * u.y, v.x * u.y - v.y * u.x);
            if (!MARKS && k >= halo && g < n) xd[g] = v;
        } else if (!MARKS && k >= halo && g < n) xd[g] = make_float2(0.f, 0.f);
        B[k] = pr;
        Cw[k] = u.x * u.x + u.y * u.y;
    }
    __syncthreads();
    for (int k = tid; k + 3 < L; k += 256) {
        const float2 a = B[k], b = B[k + 1], c = B[k + 2], d = B[k + 3];
        A[k] = make_float2(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y);
        Dw[k] = ((Cw[k] + Cw[k + 1]) + Cw[k + 2]) + Cw[k + 3];
    }
    __syncthreads();
    for (int k = tid; k + 15 < L; k += 256) {
        const float2 a = A[k], b = A[k + 4], c = A[k + 8], d = A[k + 12];
        E[k] = make_float2(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y);
        Fw[k] = ((Dw[k] + Dw[k + 4]) + Dw[k + 8]) + Dw[k + 12];
    }
    __syncthreads();
    const int w16 = window >> 4, w4 = (window >> 2) & 3, w1 = window & 3;
    const int p16 = pwindow >> 4, p4 = (pwindow >> 2) & 3, p1 = pwindow & 3;
    for (int t = tid; t < SM_TILE; t += 256) {
        const long i = i0 + t;
        const bool valid = i < n;
        if (!MARKS && !valid) break;
        float2 a = make_float2(0.f, 0.f);
        float cor = 0.f;
        if (valid) {
            const int k = halo + t;
            int j = k - window + 1;
            for (int m = 0; m < w16; m++, j += 16) { a.x = a.x + E[j].x; a.y = a.y + E[j].y; }
            for (int m = 0; m < w4; m++, j += 4) { a.x = a.x + A[j].x; a.y = a.y + A[j].y; }
            for (int m = 0; m < w1; m++, j++) { a.x = a.x + B[j].x; a.y = a.y + B[j].y; }
            float p = 0.f;
            j = k - pwindow + 1;
            for (int m = 0; m < p16; m++, j += 16) p = p + Fw[j];
            for (int m = 0; m < p4; m++, j += 4) p = p + Dw[j];
            for (int m = 0; m < p1; m++, j++) p = p + Cw[j];
            cor = ref_hypotf(a) / fabsf(p * pscale);
        }
        if (MARKS) {                                                   // fd_marks_kernel's test and word layout (one word past the end when n % 64 != 0)
            const unsigned long long m = __ballot(valid && ((double)cor > thr) && ((double)cor < maxv));
            if ((tid & 63) == 0 && i < (long)n + 63) marks[i >> 6] = m;
        } else {
            in_abs[i] = a;
            in_cor[i] = cor;
        }
    }
}

// The marks-only metrics for windows that are whole runs of 16 (window, power_window multiples of 16, delay a multiple of 4: fft_len 64 has
// 48 / 64 / 16), four consecutive samples per lane: the products and powers of a lane's samples stay in registers, runs of 4 need the three
// products behind them (the next lane's, formed again here rather than fetched), runs of 16 the runs of 4 of the next three lanes (through LDS),
// the window sums the runs of 16 of earlier lanes (through LDS).  Every LDS access is 16 bytes wide, aligned and free of bank conflicts: lane t holds local samples
// 4t .. 4t+3 of a 4 * SMK_NT sample tile whose first `halo` samples (a number = 3 mod 4, so that k - (window - 1) is a multiple of 4 for the
// outputs k = halo + 4t' + i) only feed the sums.  Same products, same tree and the same test as sync_metrics_tiled_kernel<true> — the masks
// are equal bit for bit (tests/test_gpu_sync.py) — with 8 LDS instructions per sample instead of 32.
// The test itself: `threshold < cor < MAX_PEAK_VALUE` on cor = (float)sqrt((double)|a|^2) / |p * pscale| promoted to double.  A float
// estimate of cor (relative error < 1e-6) decides every sample further than 1e-5 from both bounds; only the others take the exact form.
#define SMK_NT 512
#define SMK_L (4 * SMK_NT)
__global__ __launch_bounds__(SMK_NT) void sync_marks_kernel(const float2* __restrict__ x, int n, int delay, int window, int pwindow, float pscale, int T, int halo,
                                                            unsigned long long* __restrict__ marks, double thr, double maxv, float thr_dn, float thr_up,
                                                            float max_dn, float max_up)
{
#pragma clang fp contract(off)
    // a lane's four samples as two 16-byte halves in two arrays (lo = samples 4t, 4t+1; hi = 4t+2, 4t+3): consecutive lanes are 16 bytes apart
    // in each, so every ds_read_b128 / ds_write_b128 lane group covers whole bank rows (32 bytes apart they would collide two-way)
    __shared__ float4 Xlo[SMK_NT + 4], Xhi[SMK_NT + 4];                  // samples, later the runs of 16 of the products
    __shared__ float4 Alo[SMK_NT + 4], Ahi[SMK_NT + 4];                  // runs of 4 of the products
    __shared__ float4 Ds[SMK_NT + 4];                                    // runs of 4 of the powers
    __shared__ float4 Fs[SMK_NT + 4];                                    // runs of 16 of the powers
    const int t = threadIdx.x, k0 = 4 * t;
    const long i0 = (long)blockIdx.x * T;
    const long g0 = i0 - halo;
    const long g = g0 + k0;
    float2 u[7], v[7];
    if (g >= 0 && g + 3 < n) {
#pragma unroll
        for (int m = 0; m < 4; m++) u[m] = x[g + m];
    } else {
#pragma unroll
        for (int m = 0; m < 4; m++) u[m] = (g + m >= 0 && g + m < n) ? x[g + m] : make_float2(0.f, 0.f);
    }
    Xlo[t] = make_float4(u[0].x, u[0].y, u[1].x, u[1].y);
    Xhi[t] = make_float4(u[2].x, u[2].y, u[3].x, u[3].y);
    if (t < 4) { Xlo[SMK_NT + t] = make_float4(0.f, 0.f, 0.f, 0.f); Xhi[SMK_NT + t] = make_float4(0.f, 0.f, 0.f, 0.f); }   // behind the tile: read, used by no output
    __syncthreads();
    {
        const float4 a = Xlo[t + 1], b = Xhi[t + 1];
        u[4] = make_float2(a.x, a.y); u[5] = make_float2(a.z, a.w); u[6] = make_float2(b.x, b.y);
    }
    if (k0 >= delay) {
        const int td = t - (delay >> 2);
        const float4 a = Xlo[td], b = Xhi[td], c = Xlo[td + 1], d = Xhi[td + 1];
        v[0] = make_float2(a.x, a.y); v[1] = make_float2(a.z, a.w); v[2] = make_float2(b.x, b.y); v[3] = make_float2(b.z, b.w);
        v[4] = make_float2(c.x, c.y); v[5] = make_float2(c.z, c.w); v[6] = make_float2(d.x, d.y);
    } else {
#pragma unroll
        for (int m = 0; m < 7; m++) v[m] = make_float2(0.f, 0.f);
    }
    float2 B[7];
    float C[7];
#pragma unroll
    for (int m = 0; m < 7; m++) {
        B[m] = make_float2(v[m].x * u[m].x + v[m].y * u[m].y, v[m].x * u[m].y - v[m].y * u[m].x);       // conj(v) * u
        C[m] = u[m].x * u[m].x + u[m].y * u[m].y;
    }
    if (g0 < delay) {                                                      // the first workgroup(s): the delayed stream starts with zeros
#pragma unroll
        for (int m = 0; m < 7; m++) if (g + m < delay) B[m] = make_float2(0.f, 0.f);
    }
    float2 A[4];
    float D[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        A[i] = make_float2(((B[i].x + B[i + 1].x) + B[i + 2].x) + B[i + 3].x, ((B[i].y + B[i + 1].y) + B[i + 2].y) + B[i + 3].y);
        D[i] = ((C[i] + C[i + 1]) + C[i + 2]) + C[i + 3];
    }
    Alo[t] = make_float4(A[0].x, A[0].y, A[1].x, A[1].y);
    Ahi[t] = make_float4(A[2].x, A[2].y, A[3].x, A[3].y);
    Ds[t] = make_float4(D[0], D[1], D[2], D[3]);
    __syncthreads();
    {
        float2 E[4];
        float F[4];
        float2 An[3][4];
        float Dn[3][4];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float4 a = Alo[t + r + 1], b = Ahi[t + r + 1];
            An[r][0] = make_float2(a.x, a.y); An[r][1] = make_float2(a.z, a.w); An[r][2] = make_float2(b.x, b.y); An[r][3] = make_float2(b.z, b.w);
            const float4 d = Ds[t + r + 1];
            Dn[r][0] = d.x; Dn[r][1] = d.y; Dn[r][2] = d.z; Dn[r][3] = d.w;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            E[i] = make_float2(((A[i].x + An[0][i].x) + An[1][i].x) + An[2][i].x, ((A[i].y + An[0][i].y) + An[1][i].y) + An[2][i].y);
            F[i] = ((D[i] + Dn[0][i]) + Dn[1][i]) + Dn[2][i];
        }
        Xlo[t] = make_float4(E[0].x, E[0].y, E[1].x, E[1].y);              // the samples are dead: every read of them was before the last barrier
        Xhi[t] = make_float4(E[2].x, E[2].y, E[3].x, E[3].y);
        Fs[t] = make_float4(F[0], F[1], F[2], F[3]);
    }
    __syncthreads();
    if (4 * t >= T) return;                                                // whole waves: T is a multiple of 256
    const int w16 = window >> 4, p16 = pwindow >> 4;
    float2 a[4];
    float pw[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = make_float2(0.f, 0.f); pw[i] = 0.f; }
    {
        int c = ((halo - (window - 1)) >> 2) + t;                          // the lane whose runs of 16 start this lane's windows
        for (int m = 0; m < w16; m++, c += 4) {
            const float4 e0 = Xlo[c], e1 = Xhi[c];
            a[0].x = a[0].x + e0.x; a[0].y = a[0].y + e0.y; a[1].x = a[1].x + e0.z; a[1].y = a[1].y + e0.w;
            a[2].x = a[2].x + e1.x; a[2].y = a[2].y + e1.y; a[3].x = a[3].x + e1.z; a[3].y = a[3].y + e1.w;
        }
        c = ((halo - (pwindow - 1)) >> 2) + t;
        for (int m = 0; m < p16; m++, c += 4) {
            const float4 f = Fs[c];
            pw[0] = pw[0] + f.x; pw[1] = pw[1] + f.y; pw[2] = pw[2] + f.z; pw[3] = pw[3] + f.w;
        }
    }
    unsigned long long bits[4];
    const int left = (int)min((long)T, (long)n - i0) - 4 * t;              // outputs of the capture from this lane's first on
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float den = fabsf(pw[i] * pscale);
        const float est = __builtin_amdgcn_sqrtf(fmaf(a[i].x, a[i].x, a[i].y * a[i].y)) * __builtin_amdgcn_rcpf(den);
        bool pk = est > thr_up && est < max_dn;
        if (!pk && !(est < thr_dn || est > max_up)) {                      // within 1e-5 of a bound (or not a number): the reference's own arithmetic
            const float cor = ref_hypotf(a[i]) / den;
            pk = ((double)cor > thr) && ((double)cor < maxv);
        }
        bits[i] = __ballot(pk && i < left);
    }
    // lane l holds samples 4l .. 4l+3 of the wave's 256: word m = lanes 16m .. 16m+15, bit 4q+i of it = bits[i] of lane 16m+q
    const int lane = t & 63;
    if (lane < 4) {
        unsigned long long wd = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned long long f = (bits[i] >> (16 * lane)) & 0xffffull;
            f = (f | (f << 24)) & 0x000000ff000000ffull;
            f = (f | (f << 12)) & 0x000f000f000f000full;
            f = (f | (f << 6)) & 0x0303030303030303ull;
            f = (f | (f << 3)) & 0x1111111111111111ull;
            wd |= f << i;
        }
        const long ws = i0 + 4 * (t - lane) + 64 * lane;                   // first sample of the word
        if (ws < (long)n + 63) marks[ws >> 6] = wd;
    }
}

// in_abs[i] as the tiled kernel forms it (same products, same tree: runs of 16 of runs of 4, then runs of 4, then single products, oldest
// first), for one sample, by one lane: the detector reads it once per detection (lib/frame_detector_impl.cc:112)
__device__ float2 sm_abs_at(const float2* __restrict__ x, long i, int delay, int window)
{
#pragma clang fp contract(off)
    auto B = [&](long g) -> float2 {
        if (g < delay) return make_float2(0.f, 0.f);
        const float2 u = x[g], v = x[g - delay];
        return make_float2(v.x * u.x + v.y * u.y, v.x * u.y - v.y * u.x);
    };
    auto A4 = [&](long g) -> float2 {
        const float2 a = B(g), b = B(g + 1), c = B(g + 2), d = B(g + 3);
        return make_float2(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y);
    };
    auto E16 = [&](long g) -> float2 {
        const float2 a = A4(g), b = A4(g + 4), c = A4(g + 8), d = A4(g + 12);
        return make_float2(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y);
    };
    const int w16 = window >> 4, w4 = (window >> 2) & 3, w1 = window & 3;
    float2 a = make_float2(0.f, 0.f);
    long j = i - window + 1;
    for (int m = 0; m < w16; m++, j += 16) { const float2 e = E16(j); a.x = a.x + e.x; a.y = a.y + e.y; }
    for (int m = 0; m < w4; m++, j += 4) { const float2 e = A4(j); a.x = a.x + e.x; a.y = a.y + e.y; }
    for (int m = 0; m < w1; m++, j++) { const float2 e = B(j); a.x = a.x + e.x; a.y = a.y + e.y; }
    return a;
}
// the same value formed by a whole wavefront (window <= 64): lane l takes product l of the window, the runs of 4 and of 16 come out of lane
// shuffles in the tree's order, the window's runs are then added oldest first — every lane returns the sum
__device__ float2 sm_abs_at_wave(const float2* __restrict__ x, long i, int delay, int window, int lane)
{
#pragma clang fp contract(off)
    const long g = i - window + 1 + lane;
    float2 B = make_float2(0.f, 0.f);
    if (lane < window && g >= delay) {
        const float2 u = x[g], v = x[g - delay];
        B = make_float2(v.x * u.x + v.y * u.y, v.x * u.y - v.y * u.x);
    }
    auto down = [&](float2 v, int k) { return make_float2(__shfl_down(v.x, k), __shfl_down(v.y, k)); };
    const float2 b1 = down(B, 1), b2 = down(B, 2), b3 = down(B, 3);
    const float2 A = make_float2(((B.x + b1.x) + b2.x) + b3.x, ((B.y + b1.y) + b2.y) + b3.y);
    const float2 a4 = down(A, 4), a8 = down(A, 8), a12 = down(A, 12);
    const float2 E = make_float2(((A.x + a4.x) + a8.x) + a12.x, ((A.y + a4.y) + a8.y) + a12.y);
    const int w16 = window >> 4, w4 = (window >> 2) & 3, w1 = window & 3;
    float2 a = make_float2(0.f, 0.f);
    int j = 0;
    for (int m = 0; m < w16; m++, j += 16) { a.x = a.x + __shfl(E.x, j); a.y = a.y + __shfl(E.y, j); }
    for (int m = 0; m < w4; m++, j += 4) { a.x = a.x + __shfl(A.x, j); a.y = a.y + __shfl(A.y, j); }
    for (int m = 0; m < w1; m++, j++) { a.x = a.x + __shfl(B.x, j); a.y = a.y + __shfl(B.y, j); }
    return a;
}
// where the scans take the correlation of a detection from: the stream (in_abs != NULL) or the capture itself
struct FdAbs {
    const float2* in_abs; const float2* x; int delay, window;
    __device__ float2 at(int i) const { return in_abs ? in_abs[i] : sm_abs_at(x, i, delay, window); }
};

extern "C" int jrc_sync_metrics_dev(jrc_ctx* ctx, int n, int delay, int window, int pwindow, float pscale, const jrc_cf32* d_x,
                                    jrc_cf32* d_xd, jrc_cf32* d_in_abs, float* d_in_cor, void* stream)
{
    if (!ctx || n < 0 || delay < 0 || window < 1 || pwindow < 1 || (n > 0 && (!d_x || !d_xd || !d_in_abs || !d_in_cor))) return JRC_ERR_INVALID_ARG;
    if (n == 0) return JRC_OK;
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const size_t L = (size_t)SM_TILE + (size_t)(window > pwindow ? window : pwindow) - 1 + (size_t)delay;
    const size_t lds = L * (3 * sizeof(float2) + 3 * sizeof(float));
    if (lds <= 150 * 1024 && !ctx->tune.sync_naive) {
        JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)sync_metrics_tiled_kernel<false>, lds));
        hipLaunchKernelGGL(sync_metrics_tiled_kernel<false>, dim3((n + SM_TILE - 1) / SM_TILE), dim3(256), lds, s, (const float2*)d_x, n, delay, window, pwindow,
                           pscale, (float2*)d_xd, (float2*)d_in_abs, d_in_cor, (unsigned long long*)nullptr, 0.0, 0.0);
    } else {        // windows too long for an LDS tile: one lane per output, window + power_window loads each
        hipLaunchKernelGGL(sync_metrics_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float2*)d_x, n, delay, window, pwindow, pscale,
                           (float2*)d_xd, (float2*)d_in_abs, d_in_cor);
    }
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// ---- frame_detector ---------------------------------------------------------------------------------------------
struct FdState {
    int state;                       // 0 SEARCH, 1 COPY
    unsigned n_peaks;
    unsigned long long first_peak_ind;
    int copied_samples;
    float coarse_cfo_est;
    unsigned long long nread, nwritten;
};
struct FdSeg { int off, len, copied0; float cfo; };     // out[off + k] = in[off + k] * exp(-j cfo (copied0 + k))
#define FD_MAX_TAGS 8
#define FD_MAX_SEGS 16
struct FdResult {
    int consumed, produced, n_tags, n_segs;
    unsigned long long tag_off[FD_MAX_TAGS];
    double tag_cfo[FD_MAX_TAGS];
    FdSeg seg[FD_MAX_SEGS];
};
struct FdParams { int fft_len, min_n_peaks, ignore_gap, max_peak_distance, max_samples; double threshold, max_peak_value; };

__global__ void fd_marks_kernel(const float* __restrict__ cor, unsigned long long* __restrict__ marks, int n, double thr, double maxv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int pk = (i < n) && ((double)cor[i] > thr) && ((double)cor[i] < maxv);       // :95 (float promoted to double)
    const unsigned long long m = __ballot(pk);
    if ((threadIdx.x & 63) == 0 && i < n + 63) marks[i >> 6] = m;
}

// one general_work call (:76-195) walked by a single lane over the peak-mask words; `calls` > 1 repeats the call on what is
// left (the batched pipeline: a scheduler that always offers everything)
__global__ void fd_scan_kernel(FdParams p, FdState* __restrict__ st_g, const unsigned long long* __restrict__ marks,
                               const float2* __restrict__ in_abs, int ninput, int noutput, FdResult* __restrict__ res)
{
    if (threadIdx.x != 0) return;
    FdState s = *st_g;
    FdResult r;
    r.consumed = 0; r.produced = 0; r.n_tags = 0; r.n_segs = 0;
    auto is_mark = [&](int i) -> bool { return (marks[i >> 6] >> (i & 63)) & 1ull; };
    auto cfo_at = [&](int i) -> float { const float2 a = in_abs[i]; return (float)((double)atan2f(a.y, a.x) / (p.fft_len / 4.0)); };   // :112
    if (s.state == 0) {                                                                  // SEARCH (:89-134)
        int n_in = 0;
        while (n_in < ninput) {
            if (s.n_peaks == 0 && (n_in & 63) == 0 && n_in + 64 <= ninput && marks[n_in >> 6] == 0ull) { n_in += 64; continue; }
            const unsigned long long pos = s.nread + (unsigned long long)n_in;
            if (is_mark(n_in)) {
                if (s.n_peaks < (unsigned)p.min_n_peaks) {
                    s.n_peaks++;
                    if (s.n_peaks == 1) s.first_peak_ind = pos;
                } else if ((pos - s.first_peak_ind) < (unsigned long long)p.max_peak_distance) {
                    s.state = 1;
                    s.copied_samples = 0;
                    s.coarse_cfo_est = cfo_at(n_in);
                    s.n_peaks = 0;
                    s.first_peak_ind = 0;
                    if (r.n_tags < FD_MAX_TAGS) { r.tag_off[r.n_tags] = s.nwritten; r.tag_cfo[r.n_tags] = s.coarse_cfo_est; r.n_tags++; }
                    break;
                } else {
                    s.n_peaks = 0;
                    s.first_peak_ind = 0;
                }
            } else if ((pos - s.first_peak_ind) > (unsigned long long)p.max_peak_distance) {
                s.n_peaks = 0;
                s.first_peak_ind = 0;
            }
            n_in++;
        }
        r.consumed = n_in;
        s.nread += (unsigned long long)n_in;
    } else {                                                                             // COPY (:136-191)
        int n_out = 0;
        FdSeg seg; seg.off = 0; seg.len = 0; seg.copied0 = s.copied_samples; seg.cfo = s.coarse_cfo_est;
        while (n_out < ninput && n_out < noutput && s.copied_samples < p.max_samples) {
            if (s.n_peaks == 0 && (n_out & 63) == 0 && marks[n_out >> 6] == 0ull) {     // no peak in this word: copy it whole
                int run = 64;
                if (run > ninput - n_out) run = ninput - n_out;
                if (run > noutput - n_out) run = noutput - n_out;
                if (run > p.max_samples - s.copied_samples) run = p.max_samples - s.copied_samples;
                if (run == 64) { n_out += 64; s.copied_samples += 64; seg.len += 64; continue; }
            }
            const unsigned long long pos = s.nread + (unsigned long long)n_out;
            if (is_mark(n_out)) {
                if (s.n_peaks < (unsigned)p.min_n_peaks) {
                    s.n_peaks++;
                    if (s.n_peaks == 1) s.first_peak_ind = pos;
                } else if ((pos - s.first_peak_ind) < (unsigned long long)p.max_peak_distance) {
                    if (s.copied_samples > p.ignore_gap) {                               // a new frame before MAX_SAMPLES (:153-165)
                        s.copied_samples = 0;
                        s.n_peaks = 0;
                        s.first_peak_ind = 0;
                        s.coarse_cfo_est = cfo_at(n_out);
                        if (r.n_tags < FD_MAX_TAGS) { r.tag_off[r.n_tags] = s.nwritten + (unsigned long long)n_out; r.tag_cfo[r.n_tags] = s.coarse_cfo_est; r.n_tags++; }
                        break;
                    }
                } else {
                    s.n_peaks = 0;
                    s.first_peak_ind = 0;
                }
            } else if ((pos - s.first_peak_ind) > (unsigned long long)p.max_peak_distance) {
                s.n_peaks = 0;
                s.first_peak_ind = 0;
            }
            n_out++; s.copied_samples++; seg.len++;                                      // out[n_out] = in[n_out] * exp(...) (:178)
        }
        if (seg.len) r.seg[r.n_segs++] = seg;
        if (s.copied_samples == p.max_samples) s.state = 0;
        r.consumed = n_out; r.produced = n_out;
        s.nread += (unsigned long long)n_out;
        s.nwritten += (unsigned long long)n_out;
    }
    *st_g = s;
    *res = r;
}

__global__ void fd_copy_kernel(const float2* __restrict__ in, float2* __restrict__ out, int off, int len, int copied0, float cfo)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= len) return;
    const float ang = -cfo * (float)(copied0 + k);                                       // -coarse_cfo_est * copied_samples: float * int
    float sn, cs;
    sincosf(ang, &sn, &cs);
    out[off + k] = cmul(in[off + k], make_float2(cs, sn));
}

struct jrc_frame_detector {
    jrc_ctx* ctx;
    FdParams p;
    FdState* d_state = nullptr;
    FdResult* d_res = nullptr;
    FdResult* h_res = nullptr;        // pinned
};

extern "C" jrc_frame_detector* jrc_frame_detector_create(jrc_ctx* ctx, int fft_len, int cp_len, double threshold, unsigned min_n_peaks,
                                                         unsigned ignore_gap)
{
    if (!ctx) return nullptr;
    if (fft_len < 4 || cp_len < 0) { jrc_fail(ctx, JRC_ERR_INVALID_ARG, "frame_detector: invalid sizes"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    jrc_frame_detector* d = new jrc_frame_detector();
    d->ctx = ctx;
    d->p.fft_len = fft_len; d->p.min_n_peaks = (int)min_n_peaks; d->p.ignore_gap = (int)ignore_gap;
    d->p.threshold = threshold; d->p.max_peak_value = 2.0;                               // :56
    d->p.max_peak_distance = 2 * (fft_len + cp_len); d->p.max_samples = 540 * (fft_len + cp_len);   // :57-58
    FdState z; memset(&z, 0, sizeof(z));
    if (hipMalloc((void**)&d->d_state, sizeof(FdState)) != hipSuccess || hipMalloc((void**)&d->d_res, sizeof(FdResult)) != hipSuccess ||
        hipHostMalloc((void**)&d->h_res, sizeof(FdResult), hipHostMallocDefault) != hipSuccess ||
        hipMemcpy(d->d_state, &z, sizeof(z), hipMemcpyHostToDevice) != hipSuccess) {
        jrc_fail(ctx, JRC_ERR_NOMEM, "frame_detector: allocation failed");
        delete d;
        return nullptr;
    }
    return d;
}

extern "C" void jrc_frame_detector_destroy(jrc_frame_detector* d)
{
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    (void)hipFree(d->d_state); (void)hipFree(d->d_res); (void)hipHostFree(d->h_res);
    delete d;
}

// one general_work call on device buffers; result (consumed / produced / tags) lands in the detector's pinned record
static int fd_work_dev(jrc_frame_detector* d, int noutput, int ninput, const float2* d_in, const float2* d_in_abs, const float* d_in_cor,
                       float2* d_out, unsigned long long* d_marks, hipStream_t s)
{
    jrc_ctx* ctx = d->ctx;
    const int nblk = (ninput + 63 + 255) / 256;
    hipLaunchKernelGGL(fd_marks_kernel, dim3(nblk > 0 ? nblk : 1), dim3(256), 0, s, d_in_cor, d_marks, ninput, d->p.threshold, d->p.max_peak_value);
    hipLaunchKernelGGL(fd_scan_kernel, dim3(1), dim3(64), 0, s, d->p, d->d_state, (const unsigned long long*)d_marks, d_in_abs, ninput, noutput, d->d_res);
    JRC_HIP(ctx, hipGetLastError());
    JRC_HIP(ctx, hipMemcpyAsync(d->h_res, d->d_res, sizeof(FdResult), hipMemcpyDeviceToHost, s));
    JRC_HIP(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < d->h_res->n_segs; i++) {
        const FdSeg& g = d->h_res->seg[i];
        hipLaunchKernelGGL(fd_copy_kernel, dim3((g.len + 255) / 256), dim3(256), 0, s, d_in, d_out, g.off, g.len, g.copied0, g.cfo);
    }
    JRC_HIP(ctx, hipGetLastError());
    return d->h_res->produced;
}

extern "C" int jrc_frame_detector_work(jrc_frame_detector* d, int noutput_items, int ninput_items, const jrc_cf32* in, const jrc_cf32* in_abs,
                                       const float* in_cor, jrc_cf32* out, int* n_consumed, uint64_t* tag_offsets, double* tag_cfo,
                                       int max_tags, int* n_tags)
{
    if (!d || !n_consumed || !n_tags) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = d->ctx;
    *n_consumed = 0; *n_tags = 0;
    if (ninput_items < 0 || noutput_items < 0 || (ninput_items > 0 && (!in || !in_abs || !in_cor)) || (noutput_items > 0 && !out))
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "frame_detector: invalid buffers");
    if (ninput_items == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cb = sizeof(float2) * (size_t)ninput_items, fb = sizeof(float) * (size_t)ninput_items;
    const size_t mb = sizeof(unsigned long long) * ((size_t)ninput_items / 64 + 2);
    JRC_TRY(jrc_ensure_pinned(ctx, 3 * cb + fb));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, 2 * cb + fb));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, cb));
    JRC_TRY(jrc_ensure_scratch(ctx, 2, mb));
    char* hp = (char*)ctx->pinned;
    memcpy(hp, in, cb); memcpy(hp + cb, in_abs, cb); memcpy(hp + 2 * cb, in_cor, fb);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], hp, 2 * cb + fb, hipMemcpyHostToDevice, ctx->stream));
    char* d0 = (char*)ctx->scratch[0];
    const int produced = fd_work_dev(d, noutput_items, ninput_items, (const float2*)d0, (const float2*)(d0 + cb), (const float*)(d0 + 2 * cb),
                                     (float2*)ctx->scratch[1], (unsigned long long*)ctx->scratch[2], ctx->stream);
    if (produced < 0) return produced;
    if (produced > 0) {
        JRC_HIP(ctx, hipMemcpyAsync(hp + 2 * cb + fb, ctx->scratch[1], sizeof(float2) * (size_t)produced, hipMemcpyDeviceToHost, ctx->stream));
        JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(out, hp + 2 * cb + fb, sizeof(float2) * (size_t)produced);
    }
    *n_consumed = d->h_res->consumed;
    const int nt = d->h_res->n_tags < max_tags ? d->h_res->n_tags : max_tags;
    for (int i = 0; i < nt; i++) { tag_offsets[i] = d->h_res->tag_off[i]; tag_cfo[i] = d->h_res->tag_cfo[i]; }
    *n_tags = nt;
    return produced;
}

// ---- frame_sync -------------------------------------------------------------------------------------------------
struct FsSearch { int frame_start; float freq_offset; int top_idx[4]; };

// LTF matched filter (gr::filter::kernel::fir_filter_ccc::filterN: y[i] = sum_k taps[k] x[i + ntaps-1-k]) over n_cor lags
// with sample offsets off0 + i, then search_frame_start() (:232-287) on the SYNC_LENGTH collected values
// body shared by the per-call kernel and the batched pipeline: s_corr[sync_length] holds the matched-filter output
__device__ __forceinline__ void fs_search_block(const float2* s_corr, int sync_length, int fft_len, FsSearch* res_out,
                                                float* s_best, int* s_bidx, int* top)
{
#pragma clang fp contract(off)
    // d_cor.sort(compare_abs2) is stable and descending in |value|: the first four are four rounds of "largest, lowest index"
    for (int round = 0; round < 4; round++) {
        float best = -1.f; int bidx = 0x7fffffff;
        for (int i = threadIdx.x; i < sync_length; i += blockDim.x) {
            bool taken = false;
            for (int q = 0; q < round; q++) taken |= (top[q] == i);
            if (taken) continue;
            const float m = ref_hypotf(s_corr[i]);     // std::abs(gr_complex)
            if (m > best || (m == best && i < bidx)) { best = m; bidx = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bidx, off);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if ((threadIdx.x & 63) == 0) { s_best[threadIdx.x >> 6] = best; s_bidx[threadIdx.x >> 6] = bidx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nw = (blockDim.x + 63) >> 6;
            for (int w = 1; w < nw; w++)
                if (s_best[w] > best || (s_best[w] == best && s_bidx[w] < bidx)) { best = s_best[w]; bidx = s_bidx[w]; }
            top[round] = bidx;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        FsSearch r;
        r.frame_start = sync_length;                                                     // :242
        r.freq_offset = 0.f;
        bool have_freq = false;
        float2 v[4]; int ix[4];
        for (int q = 0; q < 4; q++) { ix[q] = top[q]; v[q] = s_corr[top[q]]; r.top_idx[q] = top[q]; }
        bool done = false;
        for (int i = 0; i < 3 && !done; i++)
            for (int k = i + 1; k < 4 && !done; k++) {
                float2 first, second;
                if (ix[i] > ix[k]) { first = v[k]; second = v[i]; } else { first = v[i]; second = v[k]; }
                const int diff = abs(ix[i] - ix[k]);
                const int mn = min(ix[i], ix[k]);
                const float2 pr = cmul(first, make_float2(second.x, -second.y));
                const float ang = atan2f(pr.y, pr.x);
                if (diff == fft_len) { r.frame_start = mn; r.freq_offset = ang / fft_len; have_freq = true; done = true; }
                else if (diff == fft_len - 1) { r.frame_start = mn; r.freq_offset = ang / (fft_len - 1); have_freq = true; }
                else if (diff == fft_len + 1) { r.frame_start = mn; r.freq_offset = ang / (fft_len + 1); have_freq = true; }
            }
        if (!have_freq) r.freq_offset = __int_as_float(0x7fc00000);                       // "keep the previous d_freq_offset": resolved by the caller
        *res_out = r;
    }
}

__global__ __launch_bounds__(1024) void fs_search_kernel(const float2* __restrict__ in, const float2* __restrict__ taps, int ntaps,
                                                         int sync_length, int fft_len, FsSearch* __restrict__ res)
{
#pragma clang fp contract(off)
    extern __shared__ float2 s_corr[];                 // [sync_length]
    __shared__ float s_best[16];
    __shared__ int s_bidx[16];
    __shared__ int top[4];
    for (int i = threadIdx.x; i < sync_length; i += blockDim.x) {
        float2 acc = make_float2(0.f, 0.f);
        for (int k = 0; k < ntaps; k++) {
            const float2 p = cmul(taps[k], in[i + ntaps - 1 - k]);
            acc.x = acc.x + p.x; acc.y = acc.y + p.y;
        }
        s_corr[i] = acc;
    }
    __syncthreads();
    fs_search_block(s_corr, sync_length, fft_len, res, s_best, s_bidx, top);
}

// COPY (:175-202) as an index map: input sample n (sample offset so0 + n) is kept when rel >= 0 and it is not a cyclic
// prefix; kept samples are packed in order.  kept_before(rel) is the closed form of the loop's n_out.
__host__ __device__ static inline long fs_kept_before(long rel, int N, int cp)
{
    if (rel <= 0) return 0;
    if (rel <= 2L * N) return rel;
    const long q = (rel - 2L * N) / (N + cp), rem = (rel - 2L * N) % (N + cp);
    return 2L * N + q * N + (rem > cp ? rem - cp : 0);
}
__global__ void fs_copy_kernel(const float2* __restrict__ in_delayed, float2* __restrict__ out, int n_in, int so0, int frame_start, float freq_offset,
                               int N, int cp, long kept0, int noutput)
{
#pragma clang fp contract(off)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_in) return;
    const int so = so0 + n;
    const long rel = (long)so - frame_start;
    if (rel < 0) return;
    const bool keep = rel < 2L * N || ((rel - 2L * N) % (N + cp)) > cp - 1;
    if (!keep) return;
    const long o = fs_kept_before(rel, N, cp) - kept0;
    if (o < 0 || o >= noutput) return;
    float sn, cs;
    sincosf((float)so * freq_offset, &sn, &cs);                                          // sample_offset * d_freq_offset: int * float (:193)
    out[o] = cmul(in_delayed[n], make_float2(cs, sn));
}

struct jrc_frame_sync {
    jrc_ctx* ctx;
    int fft_len, cp_len, sync_length, ntaps;
    float2* d_taps = nullptr;
    FsSearch* d_res = nullptr;
    FsSearch* h_res = nullptr;
    // block state (:37-56)
    int state = 0;                    // 0 SYNC, 1 COPY, 2 RESET
    int sample_offset = 0, frame_start = 0, total_out_count = 0;
    float freq_offset = 0.f;
    double cfo_coarse_est = 0.0;
    uint64_t nread = 0, nwritten = 0;
    std::vector<float2> sync_buf;     // samples collected while in SYNC across calls
};

extern "C" jrc_frame_sync* jrc_frame_sync_create(jrc_ctx* ctx, int fft_len, int cp_len, unsigned sync_length, const jrc_cf32* ltf_seq_time, int ntaps)
{
    if (!ctx) return nullptr;
    if (fft_len < 4 || cp_len < 0 || sync_length < 4 || sync_length > 8192 || !ltf_seq_time || ntaps < 1) {
        jrc_fail(ctx, JRC_ERR_INVALID_ARG, "frame_sync: invalid configuration");
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    jrc_frame_sync* f = new jrc_frame_sync();
    f->ctx = ctx; f->fft_len = fft_len; f->cp_len = cp_len; f->sync_length = (int)sync_length; f->ntaps = ntaps;
    if (hipMalloc((void**)&f->d_taps, sizeof(float2) * (size_t)ntaps) != hipSuccess || hipMalloc((void**)&f->d_res, sizeof(FsSearch)) != hipSuccess ||
        hipHostMalloc((void**)&f->h_res, sizeof(FsSearch), hipHostMallocDefault) != hipSuccess ||
        hipMemcpy(f->d_taps, ltf_seq_time, sizeof(float2) * (size_t)ntaps, hipMemcpyHostToDevice) != hipSuccess) {
        jrc_fail(ctx, JRC_ERR_NOMEM, "frame_sync: allocation failed");
        delete f;
        return nullptr;
    }
    return f;
}

extern "C" void jrc_frame_sync_destroy(jrc_frame_sync* f)
{
    if (!f) return;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    (void)hipFree(f->d_taps); (void)hipFree(f->d_res); (void)hipHostFree(f->h_res);
    delete f;
}

extern "C" int jrc_frame_sync_state(const jrc_frame_sync* f, int* state, int* frame_start, float* freq_offset)
{
    if (!f) return JRC_ERR_INVALID_ARG;
    if (state) *state = f->state;
    if (frame_start) *frame_start = f->frame_start;
    if (freq_offset) *freq_offset = f->freq_offset;
    return JRC_OK;
}

// search on device samples: d_in must hold sync_length + ntaps - 1 samples
static int fs_search_dev(jrc_frame_sync* f, const float2* d_in, hipStream_t s)
{
    jrc_ctx* ctx = f->ctx;
    hipLaunchKernelGGL(fs_search_kernel, dim3(1), dim3(256), sizeof(float2) * (size_t)f->sync_length, s, d_in, (const float2*)f->d_taps, f->ntaps,
                       f->sync_length, f->fft_len, f->d_res);
    JRC_HIP(ctx, hipGetLastError());
    JRC_HIP(ctx, hipMemcpyAsync(f->h_res, f->d_res, sizeof(FsSearch), hipMemcpyDeviceToHost, s));
    JRC_HIP(ctx, hipStreamSynchronize(s));
    f->frame_start = f->h_res->frame_start;
    if (f->h_res->freq_offset == f->h_res->freq_offset) f->freq_offset = f->h_res->freq_offset;      // NaN = no pair matched: keep the old value
    return JRC_OK;
}

extern "C" int jrc_frame_sync_work(jrc_frame_sync* f, int noutput_items, int ninput0, int ninput1, const jrc_cf32* in, const jrc_cf32* in_delayed,
                                   const uint64_t* tag_offsets, const double* tag_values, int n_tags, jrc_cf32* out, int* n_consumed,
                                   uint64_t* tag_out_offset, double* tag_out_value, int* n_tag_out)
{
    if (!f || !n_consumed || !n_tag_out) return JRC_ERR_INVALID_ARG;
    jrc_ctx* ctx = f->ctx;
    *n_consumed = 0; *n_tag_out = 0;
    if (noutput_items < 0 || ninput0 < 0 || ninput1 < 0 || n_tags < 0) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "frame_sync: invalid sizes");
    int ninput = std::min(std::min(ninput0, ninput1), 8192);                             // :111
    if (ninput > 0 && (!in || !in_delayed)) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "frame_sync: null input");
    if (noutput_items > 0 && !out) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "frame_sync: null output");
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    // the first frame_start tag in [nread, nread + ninput) (:120-146)
    bool have = false; uint64_t first_off = 0; double first_val = 0;
    for (int i = 0; i < n_tags; i++)
        if (tag_offsets[i] >= f->nread && tag_offsets[i] < f->nread + (uint64_t)ninput && (!have || tag_offsets[i] < first_off)) {
            have = true; first_off = tag_offsets[i]; first_val = tag_values[i];
        }
    if (have) {
        if (first_off > f->nread) {
            ninput = (int)(first_off - f->nread);
        } else {
            if (f->sample_offset && f->state == 0) return jrc_fail(ctx, JRC_ERR_LENGTH_MISMATCH, "[FRAME SYNC] Something is wrong!");   // :135
            if (f->state == 1) f->state = 2;
            f->cfo_coarse_est = first_val;
        }
    }
    int n_in = 0, n_out = 0;
    if (f->state == 0) {                                                                 // SYNC (:153-173)
        // the reference correlates min(SYNC_LENGTH, ninput - fft_len - 1) lags per call and collects one per consumed sample;
        // here the samples of the SYNC window are gathered across calls and correlated once it is complete
        const int usable = ninput - (f->fft_len - 1);                                    // while (n_in + fft_len - 1 < ninput)
        int take = usable > 0 ? usable : 0;
        if (take > f->sync_length - f->sample_offset) take = f->sync_length - f->sample_offset;
        if (take > 0) {
            const int need_tail = (f->sample_offset + take == f->sync_length) ? f->ntaps - 1 : 0;
            const int avail_tail = std::min(need_tail, ninput - take);
            const float2* src = (const float2*)in;
            f->sync_buf.insert(f->sync_buf.end(), src, src + take + avail_tail);
            n_in = take;
            f->sample_offset += take;
            if (f->sample_offset == f->sync_length) {
                f->sync_buf.resize((size_t)f->sync_length + f->ntaps - 1, make_float2(0.f, 0.f));
                const size_t b = sizeof(float2) * f->sync_buf.size();
                JRC_TRY(jrc_ensure_pinned(ctx, b));
                JRC_TRY(jrc_ensure_scratch(ctx, 0, b));
                memcpy(ctx->pinned, f->sync_buf.data(), b);
                JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, b, hipMemcpyHostToDevice, ctx->stream));
                JRC_TRY(fs_search_dev(f, (const float2*)ctx->scratch[0], ctx->stream));
                f->sync_buf.clear();
                f->sample_offset = 0;
                f->total_out_count = 0;
                f->state = 1;
            }
        }
    } else if (f->state == 1) {                                                          // COPY (:175-202)
        const long kept0 = fs_kept_before((long)f->sample_offset - f->frame_start, f->fft_len, f->cp_len);
        // how far the loop gets: stops at n_in == ninput or as soon as n_out == noutput
        int lo = 0, hi = ninput;                                                          // largest n with kept(n) - kept0 <= noutput ... then exact stop rule
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (fs_kept_before((long)f->sample_offset + mid - f->frame_start, f->fft_len, f->cp_len) - kept0 < noutput_items) lo = mid; else hi = mid - 1;
        }
        // lo = number of samples consumed while n_out < noutput held before each of them; one more is consumed if it exists
        n_in = lo < ninput ? lo + 1 : lo;
        if (noutput_items == 0) n_in = 0;
        n_out = (int)(fs_kept_before((long)f->sample_offset + n_in - f->frame_start, f->fft_len, f->cp_len) - kept0);
        if (n_in > 0) {
            const long rel0 = (long)f->sample_offset - f->frame_start;
            if (rel0 <= 0 && rel0 + n_in > 0) {                                          // the sample with rel == 0 is in this call: tag (:180-186)
                tag_out_offset[0] = f->nwritten;                                         // nitems_written(0): nothing is produced before rel == 0
                tag_out_value[0] = f->cfo_coarse_est - f->freq_offset;
                *n_tag_out = 1;
            }
            const size_t ib = sizeof(float2) * (size_t)n_in, ob = sizeof(float2) * (size_t)(n_out > 0 ? n_out : 1);
            JRC_TRY(jrc_ensure_pinned(ctx, ib + ob));
            JRC_TRY(jrc_ensure_scratch(ctx, 0, ib));
            JRC_TRY(jrc_ensure_scratch(ctx, 1, ob));
            memcpy(ctx->pinned, in_delayed, ib);
            JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, ib, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(fs_copy_kernel, dim3((n_in + 255) / 256), dim3(256), 0, ctx->stream, (const float2*)ctx->scratch[0], (float2*)ctx->scratch[1],
                               n_in, f->sample_offset, f->frame_start, f->freq_offset, f->fft_len, f->cp_len, kept0, n_out);
            JRC_HIP(ctx, hipGetLastError());
            if (n_out > 0) {
                JRC_HIP(ctx, hipMemcpyAsync((char*)ctx->pinned + ib, ctx->scratch[1], sizeof(float2) * (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream));
                JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                memcpy(out, (char*)ctx->pinned + ib, sizeof(float2) * (size_t)n_out);
            } else {
                JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
            f->sample_offset += n_in;
        }
    } else {                                                                             // RESET (:204-223)
        while (n_out < noutput_items) {
            if (((f->total_out_count + n_out) % f->fft_len) == 0) { f->sample_offset = 0; f->state = 0; break; }
            ((float2*)out)[n_out] = make_float2(0.f, 0.f);
            n_out++;
        }
    }
    f->total_out_count += n_out;
    *n_consumed = n_in;
    f->nread += (uint64_t)n_in;
    f->nwritten += (uint64_t)n_out;
    return n_out;
}

// ---- batched, device-resident front end: capture in HBM -> per-frame symbol streams in HBM -------------------------------
// The three blocks run to completion on a whole capture (a scheduler that always offers everything): the detector scan
// lists the frames (start sample, coarse CFO, samples copied until the next detection / MAX_SAMPLES / end of capture);
// one workgroup per frame then does frame_sync's work on the de-rotated samples it would have received: matched filter +
// search on the first sync_length samples, then the CP-dropping copy with the fine de-rotation, zero-filled to a whole
// symbol like the RESET state does.  frame k of the capture lands in row k of d_frames.
struct SfFrame { int start, len; float coarse_cfo; int frame_start; float fine_cfo; double tag_value; int n_out; int pad_; };

// one wave: the lanes fetch 64 mask words (4096 samples) at a time, the state machine itself is wave-uniform and steps
// through them from registers; stretches without a peak bit are skipped a word — or, with no peak run open, a whole fetch —
// at a time
__global__ __launch_bounds__(64) void fd_scan_all_kernel(FdParams p, const unsigned long long* __restrict__ marks, FdAbs src,
                                                        int n, SfFrame* __restrict__ frames, int max_frames, int* __restrict__ n_frames)
{
    const int lane = threadIdx.x;
    int state = 0, copied = 0, nf = 0;
    unsigned n_peaks = 0;
    long first = 0;
    int cur_len = 0;                                             // samples copied into frame nf-1 so far (flushed on change)
    float cur_cfo = 0.f; int cur_start = 0;
    auto flush = [&]() {
        if (nf > 0 && lane == 0) {
            SfFrame f; f.start = cur_start; f.len = cur_len; f.coarse_cfo = cur_cfo; f.frame_start = 0; f.fine_cfo = 0.f; f.tag_value = 0; f.n_out = 0; f.pad_ = 0;
            frames[nf - 1] = f;
        }
    };
    const int n_words = (n + 63) >> 6;
    bool full = false;
    for (int wb = 0; wb < n_words && !full; wb += 64) {
        const int wi = wb + lane;
        const unsigned long long mine = wi < n_words ? marks[wi] : 0ull;
        const unsigned long long nz = __ballot(mine != 0ull);
        for (int j = 0; j < 64 && wb + j < n_words && !full; j++) {
            const int base = (wb + j) << 6;
            const int cnt = min(64, n - base);
            if (n_peaks == 0 && cnt == 64) {
                // no peak run open: every word up to the next one with a peak bit only advances the copy counter
                const unsigned long long rest = nz >> j;
                int skip = rest ? __ffsll((long long)rest) - 1 : 64 - j;                 // words without any peak bit from here
                if (wb + j + skip > n_words - 1) skip = max(0, n_words - 1 - (wb + j)); // keep the (possibly partial) last word for the slow path
                if (skip > 0) {
                    if (state == 1) {
                        const int room = (p.max_samples - copied) / 64;                   // whole words before MAX_SAMPLES
                        if (skip > room) skip = room;
                    }
                    if (skip > 0) {
                        if (state == 1) { copied += 64 * skip; cur_len += 64 * skip; if (copied == p.max_samples) state = 0; }
                        j += skip - 1;
                        continue;
                    }
                }
            }
            // the word into scalar registers: everything the state machine touches below is wave-uniform, so it runs on the
            // scalar unit instead of as 64-wide vector code
            const unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), j) << 32) |
                                            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine & 0xffffffffull), j);
            // event driven inside the word: unmarked stretches only move the copy counter and may close an open peak run
            auto advance = [&](int k) {
                if (state == 1 && k > 0) {
                    const int take = min(k, p.max_samples - copied);
                    copied += take; cur_len += take;
                    if (copied == p.max_samples) state = 0;
                }
            };
            int pos = 0;
            while (pos < cnt && !full) {
                const unsigned long long rem = word >> pos;
                const int nm = rem ? pos + __ffsll((long long)rem) - 1 : cnt;             // next peak bit, or the end of the word
                if (nm > pos) {
                    if (n_peaks > 0 && (long)(base + nm - 1) - first > p.max_peak_distance) { n_peaks = 0; first = 0; }   // :125-129 / :173-177
                    advance(nm - pos);
                    pos = nm;
                    if (pos == cnt) break;
                }
                const int i = base + pos;
                // a run of consecutive peak bits is taken in as few steps as the rules allow
                const unsigned long long inv = ~rem;
                int run = inv ? __ffsll((long long)inv) - 1 : 64;
                if (run > cnt - pos) run = cnt - pos;
                if (state == 1 && run > p.max_samples - copied) run = p.max_samples - copied;     // the state flips there
                bool detect = false;
                if (n_peaks < (unsigned)p.min_n_peaks) {                                          // counting up to min_n_peaks
                    const int k = min(run, p.min_n_peaks - (int)n_peaks);
                    if (n_peaks == 0) first = i;
                    n_peaks += (unsigned)k;
                    advance(k); pos += k;
                    continue;
                } else if ((i - first) < p.max_peak_distance) {
                    if (state == 0 || copied > p.ignore_gap) detect = true;
                    else {                                                                        // inside COPY, too early for a new frame: nothing happens
                        int k = min(run, (int)(first + p.max_peak_distance - i));
                        k = min(k, p.ignore_gap - copied + 1);
                        advance(k); pos += k;
                        continue;
                    }
                } else { n_peaks = 0; first = 0; }
                if (detect) {                                    // SEARCH -> COPY (:108-118) or a new frame inside COPY (:153-165)
                    if (nf == max_frames) { full = true; break; }
                    flush();
                    const float2 av = src.at(i);
                    state = 1; copied = 0;
                    nf++; cur_start = i; cur_len = 0;
                    cur_cfo = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)((double)atan2f(av.y, av.x) / (p.fft_len / 4.0)))));
                    n_peaks = 1; first = i;                      // the COPY call that follows examines this sample again with fresh counters
                }
                advance(1);
                pos++;
            }
        }
    }
    flush();
    if (lane == 0) *n_frames = nf;
}

// ---- the same scan, many waves --------------------------------------------------------------------------------------------
// The detector's decisions depend on the past only through the open peak run (n_peaks, first) and through "SEARCH, or more than
// ignore_gap samples copied since the last detection".  At a marked sample preceded by at least G = ignore_gap + max_peak_distance
// + 2 unmarked ones both are known without any history: the peak run is stale (it is reset before it is used, :125-129 / :173-177)
// and the last detection lies more than ignore_gap samples back (detections only happen at marked samples), so the next eligible
// sample is a detection whatever the state was.  Such samples are anchors.  The capture is cut into segments of 4096 samples; the
// wave of a segment starts the state machine at the segment's first anchor (wave 0: at sample 0, in SEARCH) and runs it — the very
// code of the single-wave scan — until the first anchor at or beyond its segment's end, where the next wave has started.  The
// pieces are disjoint, in order and together cover every detection, so: pass 1 counts the detections of every wave, pass 2 writes
// them at the prefix sums of the counts, and a last kernel fills in the copy lengths (next detection - start, capped by
// MAX_SAMPLES and the end of the capture).  A capture without quiet stretches has no anchors: wave 0 then walks it alone, as before.
#define FD_SEG_WORDS 64
#define FD_SEG_KEEP 8       // detections of a segment the counting pass keeps (start sample); more: the writing pass walks the segment again

template <bool WRITE>
__global__ __launch_bounds__(64) void fd_scan_seg_kernel(FdParams p, const unsigned long long* __restrict__ marks, FdAbs src,
                                                         int n, int G, int* __restrict__ counts, SfFrame* __restrict__ frames, int max_frames,
                                                         int* __restrict__ overflow_start, int* __restrict__ kept /* [n_seg][FD_SEG_KEEP] starts */,
                                                         const int* __restrict__ raw_counts)
{
    const int lane = threadIdx.x, w = blockIdx.x;
    if (WRITE && raw_counts[w] <= FD_SEG_KEEP) return;           // the counting pass kept this segment's detections: fd_scan_gather_kernel lists them
    const int n_words = (n + 63) >> 6;
    const long seg_start = (long)w * FD_SEG_WORDS * 64;
    const long seg_end = min((long)n, seg_start + (long)FD_SEG_WORDS * 64);
    int out_idx = 0;
    if (WRITE) {                                                 // detections of the waves before this one (fd_scan_prefix_kernel)
        out_idx = counts[w];
        if (out_idx > max_frames) return;                        // everything from here on is beyond the caller's list
    }
    bool running = (w == 0), done = false;
    long quiet = 0;                                              // unmarked samples immediately before the current position
    int state = 0, copied = 0, nd = 0;
    unsigned n_peaks = 0;
    long first = 0;
    int wb0 = 0;
    if (w > 0) { wb0 = w * FD_SEG_WORDS - (G + 63) / 64; if (wb0 < 0) wb0 = 0; }
    for (int wb = wb0; wb < n_words && !done; wb += 64) {
        const int wi = wb + lane;
        const unsigned long long mine = wi < n_words ? marks[wi] : 0ull;
        const unsigned long long nz = __ballot(mine != 0ull);
        for (int j = 0; j < 64 && wb + j < n_words && !done; j++) {
            const int base = (wb + j) << 6;
            const int cnt = min(64, n - base);
            if (!running && base >= seg_end) { done = true; break; }             // no anchor in this segment: nothing to do
            if (n_peaks == 0 && cnt == 64) {
                // no peak run open: every word up to the next one with a peak bit only advances the counters
                const unsigned long long rest = nz >> j;
                int skip = rest ? __ffsll((long long)rest) - 1 : 64 - j;
                if (wb + j + skip > n_words - 1) skip = max(0, n_words - 1 - (wb + j));
                if (skip > 0) {
                    if (state == 1) {
                        const int room = (p.max_samples - copied) / 64;
                        if (skip > room) skip = room;
                    }
                    if (skip > 0) {
                        if (state == 1) { copied += 64 * skip; if (copied == p.max_samples) state = 0; }
                        quiet += 64L * skip;
                        j += skip - 1;
                        continue;
                    }
                }
            }
            const unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), j) << 32) |
                                            (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine & 0xffffffffull), j);
            auto advance = [&](int k) {
                if (state == 1 && k > 0) {
                    const int take = min(k, p.max_samples - copied);
                    copied += take;
                    if (copied == p.max_samples) state = 0;
                }
            };
            int pos = 0;
            while (pos < cnt) {
                const unsigned long long rem = word >> pos;
                const int nm = rem ? pos + __ffsll((long long)rem) - 1 : cnt;
                if (nm > pos) {
                    if (n_peaks > 0 && (long)(base + nm - 1) - first > p.max_peak_distance) { n_peaks = 0; first = 0; }
                    advance(nm - pos);
                    quiet += nm - pos;
                    pos = nm;
                    if (pos == cnt) break;
                }
                const int i = base + pos;
                const unsigned long long inv = ~rem;
                int run = inv ? __ffsll((long long)inv) - 1 : 64;
                if (run > cnt - pos) run = cnt - pos;
                if (!running) {
                    if (i >= seg_end) { done = true; break; }
                    if (i >= seg_start && quiet >= G) { running = true; state = 0; copied = 0; n_peaks = 0; first = 0; }   // anchor: start here
                    else { pos += run; quiet = 0; continue; }                                                              // still hunting
                } else if (i >= seg_end && quiet >= G) { done = true; break; }                                             // the next piece starts here
                quiet = 0;
                if (state == 1 && run > p.max_samples - copied) run = p.max_samples - copied;
                bool detect = false;
                if (n_peaks < (unsigned)p.min_n_peaks) {
                    const int k = min(run, p.min_n_peaks - (int)n_peaks);
                    if (n_peaks == 0) first = i;
                    n_peaks += (unsigned)k;
                    advance(k); pos += k;
                    continue;
                } else if ((i - first) < p.max_peak_distance) {
                    if (state == 0 || copied > p.ignore_gap) detect = true;
                    else {
                        int k = min(run, (int)(first + p.max_peak_distance - i));
                        k = min(k, p.ignore_gap - copied + 1);
                        advance(k); pos += k;
                        continue;
                    }
                } else { n_peaks = 0; first = 0; }
                if (detect) {
                    if (WRITE) {
                        const int idx = out_idx + nd;
                        if (idx < max_frames) {
                            const float2 av = src.at(i);
                            if (lane == 0) {
                                SfFrame f; f.start = i; f.len = 0; f.coarse_cfo = (float)((double)atan2f(av.y, av.x) / (p.fft_len / 4.0));
                                f.frame_start = 0; f.fine_cfo = 0.f; f.tag_value = 0; f.n_out = 0; f.pad_ = 0;
                                frames[idx] = f;
                            }
                        } else {
                            if (idx == max_frames && lane == 0) *overflow_start = i;
                            done = true;
                            break;
                        }
                    }
                    if (!WRITE && nd < FD_SEG_KEEP && lane == 0) kept[w * FD_SEG_KEEP + nd] = i;
                    nd++;
                    state = 1; copied = 0;
                    n_peaks = 1; first = i;
                }
                advance(1);
                pos++;
            }
        }
    }
    if (!WRITE && lane == 0) counts[w] = nd;
}

// the detections the counting pass kept (segments with at most FD_SEG_KEEP of them — frames are ignore_gap samples apart, a 4096-sample segment
// rarely holds more) go to their places in the list, one wavefront per segment; the coarse CFO of each is formed here (:112), the window's
// products spread over the lanes
__global__ __launch_bounds__(256) void fd_scan_gather_kernel(FdParams p, FdAbs src, const int* __restrict__ counts, const int* __restrict__ prefix,
                                                             const int* __restrict__ kept, int n_seg, SfFrame* __restrict__ frames, int max_frames,
                                                             int* __restrict__ overflow_start)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_seg) return;
    const int c = counts[w];
    if (c > FD_SEG_KEEP) return;                                 // listed by fd_scan_seg_kernel<true>
    const int base = prefix[w];
    for (int d = 0; d < c; d++) {
        const int idx = base + d, i = kept[w * FD_SEG_KEEP + d];
        if (idx < max_frames) {
            const float2 av = (!src.in_abs && src.window <= 64) ? sm_abs_at_wave(src.x, i, src.delay, src.window, lane) : src.at(i);
            if (lane == 0) {
                SfFrame f; f.start = i; f.len = 0; f.coarse_cfo = (float)((double)atan2f(av.y, av.x) / (p.fft_len / 4.0));
                f.frame_start = 0; f.fine_cfo = 0.f; f.tag_value = 0; f.n_out = 0; f.pad_ = 0;
                frames[idx] = f;
            }
        } else {
            if (idx == max_frames && lane == 0) *overflow_start = i;
            break;
        }
    }
}

// exclusive prefix sums of the segments' detection counts, 1024 segments per workgroup: sums of the blocks first, then every block scans its
// own counts behind the sum of the blocks before it; the total lands behind the last prefix
__global__ __launch_bounds__(1024) void fd_scan_blocksum_kernel(const int* __restrict__ counts, int n_seg, int* __restrict__ blocksum)
{
    __shared__ int s_wave[16];
    const int t = threadIdx.x, i = blockIdx.x * 1024 + t;
    int v = i < n_seg ? counts[i] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((t & 63) == 0) s_wave[t >> 6] = v;
    __syncthreads();
    if (t == 0) { int sum = 0; for (int k = 0; k < 16; k++) sum += s_wave[k]; blocksum[blockIdx.x] = sum; }
}
__global__ __launch_bounds__(1024) void fd_scan_prefix_kernel(const int* __restrict__ counts, int n_seg, const int* __restrict__ blocksum, int* __restrict__ prefix)
{
    __shared__ int s_wave[16], s_base[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, i = blockIdx.x * 1024 + t;
    int before = 0;
    for (int k = t; k < (int)blockIdx.x; k += 1024) before += blocksum[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if (lane == 0) s_base[wv] = before;
    const int v = i < n_seg ? counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < 16; k++) base += s_base[k];
    for (int k = 0; k < wv; k++) base += s_wave[k];
    if (i < n_seg) prefix[i] = base + incl - v;
    if (i == n_seg - 1) prefix[n_seg] = base + incl;
}

// the frame count and the copy lengths (next detection - start, capped by MAX_SAMPLES and by the end of the capture / of the caller's list)
__global__ __launch_bounds__(256) void fd_scan_finish_kernel(FdParams p, const int* __restrict__ prefix, int n_seg, int n, SfFrame* __restrict__ frames,
                                                             int max_frames, const int* __restrict__ overflow_start, int* __restrict__ n_frames)
{
    const int total = prefix[n_seg];
    const int nf = total < max_frames ? total : max_frames;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) *n_frames = nf;
    if (k >= nf) return;
    const int start = frames[k].start;
    const int next = (k + 1 < nf) ? frames[k + 1].start : (total > max_frames ? *overflow_start : n);
    int len = next - start;
    if (len > p.max_samples) len = p.max_samples;
    frames[k].len = len;
}

// list the frames of a capture: segment-parallel scan (JRC_FD_SERIAL=1: the single-wave scan, kept for cross-checks)
static int launch_fd_scan(jrc_ctx* ctx, const FdParams& p, const unsigned long long* d_marks, FdAbs d_abs, int n_samples, SfFrame* d_info,
                          int max_frames, int* d_n_frames, hipStream_t s)
{
    const int n_seg = (n_samples + FD_SEG_WORDS * 64 - 1) / (FD_SEG_WORDS * 64);
    if (n_seg <= 1 || ctx->tune.fd_serial) {
        hipLaunchKernelGGL(fd_scan_all_kernel, dim3(1), dim3(64), 0, s, p, d_marks, d_abs, n_samples, d_info, max_frames, d_n_frames);
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    const int n_blk = (n_seg + 1023) / 1024;
    JRC_TRY(jrc_ensure_scratch(ctx, 3, sizeof(int) * ((2 + FD_SEG_KEEP) * (size_t)n_seg + n_blk + 16)));
    int* counts = (int*)ctx->scratch[3];
    int* prefix = counts + n_seg;                                   // [n_seg + 1]
    int* overflow = prefix + n_seg + 1;
    int* blocksum = overflow + 1;                                   // [n_blk]
    int* kept = blocksum + n_blk;                                   // [n_seg][FD_SEG_KEEP]
    const int G = p.ignore_gap + p.max_peak_distance + 2;
    hipLaunchKernelGGL(fd_scan_seg_kernel<false>, dim3(n_seg), dim3(64), 0, s, p, d_marks, d_abs, n_samples, G, counts, d_info, max_frames, overflow, kept,
                       (const int*)nullptr);
    hipLaunchKernelGGL(fd_scan_blocksum_kernel, dim3(n_blk), dim3(1024), 0, s, (const int*)counts, n_seg, blocksum);
    hipLaunchKernelGGL(fd_scan_prefix_kernel, dim3(n_blk), dim3(1024), 0, s, (const int*)counts, n_seg, (const int*)blocksum, prefix);
    hipLaunchKernelGGL(fd_scan_gather_kernel, dim3((n_seg + 3) / 4), dim3(256), 0, s, p, d_abs, (const int*)counts, (const int*)prefix, (const int*)kept, n_seg,
                       d_info, max_frames, overflow);
    hipLaunchKernelGGL(fd_scan_seg_kernel<true>, dim3(n_seg), dim3(64), 0, s, p, d_marks, d_abs, n_samples, G, prefix, d_info, max_frames, overflow, (int*)nullptr,
                       (const int*)counts);
    const int cap = max_frames < n_samples ? max_frames : n_samples;  // a frame holds at least one sample
    hipLaunchKernelGGL(fd_scan_finish_kernel, dim3((cap + 255) / 256 > 0 ? (cap + 255) / 256 : 1), dim3(256), 0, s, p, (const int*)prefix, n_seg, n_samples, d_info,
                       max_frames, (const int*)overflow, d_n_frames);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// Two launches: PHASE 0 searches every frame's long training field (frame_start, fine offset — not a number when no pair of peaks matched),
// PHASE 1 copies.  Between them lies the one thing frame_sync carries from frame to frame: d_freq_offset is only assigned when a pair matches
// (:262-284), so a frame whose search fails — a false detection inside a payload, typically — is copied with the offset of the last frame
// before it whose search succeeded (0 if there is none); PHASE 1 looks that up in the list PHASE 0 completed.
template <int PHASE>
__global__ __launch_bounds__(256) void sf_frames_kernel(const float2* __restrict__ xs, int xs_delay /* xd[g] = xs[g - xs_delay], 0 before the stream starts */, int n,
                                                        const float2* __restrict__ taps, int ntaps,
                                                        int sync_length, int N, int cp, SfFrame* __restrict__ frames, const int* __restrict__ n_frames,
                                                        float2* __restrict__ out, long out_stride /* samples per row */)
{
#pragma clang fp contract(off)
    extern __shared__ float2 sf_lds[];                           // [sync_length + ntaps - 1] de-rotated samples, then [sync_length] correlation
    __shared__ float s_best[16];
    __shared__ int s_bidx[16];
    __shared__ int top[4];
    __shared__ FsSearch sr;
    const int f = blockIdx.x;
    if (f >= *n_frames) return;
    SfFrame fr = frames[f];
    float2* s_in = sf_lds;
    float2* s_corr = sf_lds + sync_length + ntaps - 1;
    auto det_out = [&](int so) -> float2 {                      // what frame_detector hands over: xd de-rotated by the coarse CFO (:178)
        if (so >= fr.len || fr.start + so >= n) return make_float2(0.f, 0.f);
        float sn, cs;
        jrc_sincosf_fast(-fr.coarse_cfo * (float)so, &sn, &cs);
        const int g = fr.start + so - xs_delay;
        return g >= 0 ? cmul(xs[g], make_float2(cs, sn)) : make_float2(0.f, 0.f);
    };
    if (PHASE == 0) {
        for (int i = threadIdx.x; i < sync_length + ntaps - 1; i += blockDim.x) s_in[i] = det_out(i);
        __syncthreads();
        for (int i = threadIdx.x; i < sync_length; i += blockDim.x) {
            float2 acc = make_float2(0.f, 0.f);
            for (int k = 0; k < ntaps; k++) {
                const float2 p = cmul(taps[k], s_in[i + ntaps - 1 - k]);
                acc.x = acc.x + p.x; acc.y = acc.y + p.y;
            }
            s_corr[i] = acc;
        }
        __syncthreads();
        fs_search_block(s_corr, sync_length, N, &sr, s_best, s_bidx, top);
        __syncthreads();
        if (threadIdx.x == 0) { frames[f].frame_start = sr.frame_start; frames[f].fine_cfo = sr.freq_offset; }
        return;
    }
    const int fstart = fr.frame_start;
    if (threadIdx.x == 0) {
        float v = fr.fine_cfo;
        for (int k = f - 1; k >= 0 && v != v; k--) v = frames[k].fine_cfo;             // (a frame resolved meanwhile holds the same value its own walk would find)
        sr.freq_offset = (v == v) ? v : 0.f;                                            // no pair matched on a fresh synchroniser: 0
    }
    __syncthreads();
    const float fine = sr.freq_offset;
    // COPY: the synchroniser sees the segment again through the sync_length delay, sample offsets counted from the tag; the
    // next tag (or the end of the stream) arrives on the undelayed port sync_length samples before the delayed port has
    // delivered the segment's tail, so the last sync_length samples of a segment are never copied
    const long copy_len = (long)fr.len - sync_length;
    const long kept = fs_kept_before(copy_len - fstart, N, cp);
    const long n_out = ((kept + N - 1) / N) * N;                  // RESET completes the last symbol with zeros (:204-223)
    float2* o = out + (size_t)f * out_stride;
    // walked by OUTPUT index (the inverse of fs_kept_before): oi < 2N is sample rel = oi behind the frame start (the two long training symbols, no
    // prefix between them), oi = 2N + q N + m is rel = 2N + q (N + cp) + cp + m; (q, m) advance by a constant per step, no division in the loop
    const int lim = (int)(kept < out_stride ? kept : out_stride);
    auto put = [&](int oi, int rel) {
        const int so = fstart + rel;
        float sn, cs;
        jrc_sincosf_fast((float)so * fine, &sn, &cs);
        o[oi] = cmul(det_out(so), make_float2(cs, sn));
    };
    for (int oi = threadIdx.x; oi < lim && oi < 2 * N; oi += blockDim.x) put(oi, oi);
    {
        int q = (int)threadIdx.x / N, m = (int)threadIdx.x % N;
        const int dq = (int)blockDim.x / N, dm = (int)blockDim.x % N, bd = (int)blockDim.x;
        // every sample the loop reads lies inside the frame's segment (so < copy_len < fr.len); only a frame at the very start or end of the
        // capture can reach outside the capture
        const bool inside = fr.start - xs_delay >= 0 && (long)fr.start + fr.len <= (long)n;
        for (int oi0 = 2 * N + (int)threadIdx.x; oi0 < lim; oi0 += 4 * bd) {                 // four steps at a time: their loads are in flight together
            int so[4];
            float2 xv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                so[u] = fstart + 2 * N + q * (N + cp) + cp + m;
                q += dq; m += dm;
                if (m >= N) { m -= N; q++; }
                const int g = fr.start + so[u] - xs_delay;
                const bool live = oi0 + u * bd < lim && (inside || (so[u] < fr.len && fr.start + so[u] < n && g >= 0));
                xv[u] = live ? xs[g] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int oi = oi0 + u * bd;
                if (oi >= lim) break;
                float sn, cs, sn2, cs2;
                jrc_sincosf_fast(-fr.coarse_cfo * (float)so[u], &sn, &cs);                    // frame_detector's de-rotation (:178), then frame_sync's (:193)
                jrc_sincosf_fast((float)so[u] * fine, &sn2, &cs2);
                o[oi] = cmul(cmul(xv[u], make_float2(cs, sn)), make_float2(cs2, sn2));
            }
        }
    }
    for (long oi = kept + threadIdx.x; oi < n_out && oi < out_stride; oi += blockDim.x) o[oi] = make_float2(0.f, 0.f);
    if (threadIdx.x == 0) {
        fr.frame_start = fstart; fr.fine_cfo = fine; fr.tag_value = (double)fr.coarse_cfo - (double)fine;   // :186
        fr.n_out = (int)(n_out < out_stride ? n_out : (out_stride / N) * N);
        frames[f] = fr;
    }
}

extern "C" int jrc_sync_frontend_dev(jrc_ctx* ctx, const jrc_sync_cfg* c, int n_samples, const jrc_cf32* d_x, jrc_cf32* d_work /* 2*n cf32 + n float + n/8+64 bytes */,
                                     int max_frames, int max_symbols, jrc_cf32* d_frames, jrc_sync_frame* d_info, int* d_n_frames, void* stream)
{
    JRC_TRACE("jrc_sync_frontend_dev");
    if (!ctx || !c || n_samples < 0 || max_frames < 1 || max_symbols < 2 || !d_x || !d_work || !d_frames || !d_info || !d_n_frames || !c->d_ltf_taps)
        return JRC_ERR_INVALID_ARG;
    if (c->fft_len < 4 || c->cp_len < 0 || c->sync_length < 4 || c->sync_length > 4096 || c->n_taps < 1 || c->n_taps > 1024 || c->delay < 0 || c->window < 1 ||
        c->power_window < 1)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "sync front end: invalid configuration");
    static_assert(sizeof(jrc_sync_frame) == sizeof(SfFrame), "jrc_sync_frame layout");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const size_t n = (size_t)n_samples;
    float2* d_xd = (float2*)d_work;
    float2* d_abs = d_xd + n;
    float* d_cor = (float*)(d_abs + n);
    unsigned long long* d_marks = (unsigned long long*)(d_cor + ((n + 1) & ~(size_t)1));
    FdParams p;
    p.fft_len = c->fft_len; p.min_n_peaks = (int)c->min_n_peaks; p.ignore_gap = (int)c->ignore_gap; p.threshold = c->threshold; p.max_peak_value = 2.0;
    p.max_peak_distance = 2 * (c->fft_len + c->cp_len); p.max_samples = 540 * (c->fft_len + c->cp_len);
    const size_t tile = (size_t)SM_TILE + (size_t)(c->window > c->power_window ? c->window : c->power_window) - 1 + (size_t)c->delay;
    const size_t tile_lds = tile * (3 * sizeof(float2) + 3 * sizeof(float));
    const float2* d_delayed = nullptr;                            // the delayed stream as an array of its own (three-kernel form), else x at an offset
    FdAbs src;
    if (n_samples > 0 && tile_lds <= 150 * 1024 && !ctx->tune.sync_naive && !ctx->tune.sync_streams) {
        // the metric streams never reach HBM: peak mask straight from the tile kernel, correlation re-formed at the detections (see sync_metrics_tiled_kernel)
        const int halo = (c->window > c->power_window ? c->window : c->power_window) - 1 + c->delay;
        const int T = ((SMK_L - halo) / 256) * 256;
        if (c->window % 16 == 0 && c->power_window % 16 == 0 && c->delay % 4 == 0 && T >= 256 && !ctx->tune.sync_tile) {
            const double m = 1e-5;
            hipLaunchKernelGGL(sync_marks_kernel, dim3((n_samples + T - 1) / T), dim3(SMK_NT), 0, s, (const float2*)d_x, n_samples, c->delay, c->window,
                               c->power_window, c->power_scale, T, halo, d_marks, p.threshold, p.max_peak_value, (float)(p.threshold * (1 - m)),
                               (float)(p.threshold * (1 + m)), (float)(p.max_peak_value * (1 - m)), (float)(p.max_peak_value * (1 + m)));
        } else {
            JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)sync_metrics_tiled_kernel<true>, tile_lds));
            hipLaunchKernelGGL(sync_metrics_tiled_kernel<true>, dim3((n_samples + SM_TILE - 1) / SM_TILE), dim3(256), tile_lds, s, (const float2*)d_x,
                               n_samples, c->delay, c->window, c->power_window, c->power_scale, (float2*)nullptr, (float2*)nullptr, (float*)nullptr, d_marks,
                               p.threshold, p.max_peak_value);
        }
        src.in_abs = nullptr; src.x = (const float2*)d_x; src.delay = c->delay; src.window = c->window;
    } else {
        JRC_TRY(jrc_sync_metrics_dev(ctx, n_samples, c->delay, c->window, c->power_window, c->power_scale, d_x, (jrc_cf32*)d_xd, (jrc_cf32*)d_abs, d_cor, s));
        const int nblk = (n_samples + 63 + 255) / 256;
        hipLaunchKernelGGL(fd_marks_kernel, dim3(nblk > 0 ? nblk : 1), dim3(256), 0, s, (const float*)d_cor, d_marks, n_samples, p.threshold, p.max_peak_value);
        src.in_abs = d_abs; src.x = nullptr; src.delay = 0; src.window = 0;
        d_delayed = d_xd;
    }
    JRC_TRY(launch_fd_scan(ctx, p, (const unsigned long long*)d_marks, src, n_samples, (SfFrame*)d_info, max_frames, d_n_frames, s));
    const size_t lds = sizeof(float2) * ((size_t)2 * c->sync_length + c->n_taps - 1);       // up to 73 KB at sync_length 4096, n_taps 1024
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)sf_frames_kernel<0>, lds));
    hipLaunchKernelGGL(sf_frames_kernel<0>, dim3(max_frames), dim3(256), lds, s, d_delayed ? d_delayed : (const float2*)d_x, d_delayed ? 0 : c->delay, n_samples,
                       (const float2*)c->d_ltf_taps, c->n_taps, c->sync_length, c->fft_len, c->cp_len, (SfFrame*)d_info, (const int*)d_n_frames,
                       (float2*)d_frames, (long)max_symbols * c->fft_len);
    hipLaunchKernelGGL(sf_frames_kernel<1>, dim3(max_frames), dim3(256), 0, s, d_delayed ? d_delayed : (const float2*)d_x, d_delayed ? 0 : c->delay, n_samples,
                       (const float2*)c->d_ltf_taps, c->n_taps, c->sync_length, c->fft_len, c->cp_len, (SfFrame*)d_info, (const int*)d_n_frames,
                       (float2*)d_frames, (long)max_symbols * c->fft_len);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// the detector alone, run to completion on precomputed metric streams: the list of frames (start, samples copied, coarse CFO)
extern "C" int jrc_frame_detector_scan_dev(jrc_ctx* ctx, int fft_len, int cp_len, double threshold, unsigned min_n_peaks, unsigned ignore_gap,
                                           int n_samples, const jrc_cf32* d_in_abs, const float* d_in_cor, unsigned long long* d_marks,
                                           int max_frames, jrc_sync_frame* d_info, int* d_n_frames, void* stream)
{
    if (!ctx || n_samples < 0 || max_frames < 1 || !d_in_abs || !d_in_cor || !d_marks || !d_info || !d_n_frames) return JRC_ERR_INVALID_ARG;
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    FdParams p;
    p.fft_len = fft_len; p.min_n_peaks = (int)min_n_peaks; p.ignore_gap = (int)ignore_gap; p.threshold = threshold; p.max_peak_value = 2.0;
    p.max_peak_distance = 2 * (fft_len + cp_len); p.max_samples = 540 * (fft_len + cp_len);
    const int nblk = (n_samples + 63 + 255) / 256;
    hipLaunchKernelGGL(fd_marks_kernel, dim3(nblk > 0 ? nblk : 1), dim3(256), 0, s, d_in_cor, d_marks, n_samples, p.threshold, p.max_peak_value);
    FdAbs src; src.in_abs = (const float2*)d_in_abs; src.x = nullptr; src.delay = 0; src.window = 0;
    JRC_TRY(launch_fd_scan(ctx, p, (const unsigned long long*)d_marks, src, n_samples, (SfFrame*)d_info, max_frames, d_n_frames, s));
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

extern "C" size_t jrc_sync_frontend_work_bytes(int n_samples)
{
    const size_t n = (size_t)(n_samples > 0 ? n_samples : 0);
    return 2 * n * sizeof(float2) + ((n + 1) & ~(size_t)1) * sizeof(float) + (n / 64 + 2) * sizeof(unsigned long long) + 64;
}
