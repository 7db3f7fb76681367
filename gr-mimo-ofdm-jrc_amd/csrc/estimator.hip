// estimator.hip — A5 range_angle_estimator and B1 fft_peak_detect
//
//   A5 replaces range_angle_estimator_impl::work (reference lib/range_angle_estimator_impl.cc:121-284)
//   B1 replaces fft_peak_detect_impl::work       (reference lib/fft_peak_detect_impl.cc:77-111)
//
// A5 is a full-map arg-max (HBM-bound read of the map) followed by a tiny per-frame epilogue.  The
// arg-max keeps the reference's exact power arithmetic (hypotf in double, squared in double) but
// only evaluates it for candidates within 1e-5 of the running f32 maximum, so the scan itself runs
// at memory speed; ties resolve to the lowest flat index = first in the reference's scan order.
#include "radar_kernels.h"

#include <cmath>

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ra_partial_kernel(const float2* __restrict__ map, size_t total,
                                                         PeakPartial* __restrict__ partials)
{
    __shared__ PeakPartial red[4];
    PeakTracker t;
    t.init();
    // 4 cells per lane per trip (two 16-byte loads); trip count is uniform across the block so the wave-wide
    // running maximum in PeakTracker::raise() is well defined
    const size_t per_trip = (size_t)gridDim.x * blockDim.x * 4;
    const size_t trips = (total + per_trip - 1) / per_trip;
    const bool vec_ok = ((total & 1) == 0) && ((reinterpret_cast<size_t>(map) & 15) == 0);
    for (size_t trip = 0; trip < trips; trip++) {
        const size_t i0 = trip * per_trip + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
        float2 z[4];
        float m = -1.0f;
        if (vec_ok && i0 + 4 <= total) {
            const float4 v0 = *reinterpret_cast<const float4*>(map + i0);
            const float4 v1 = *reinterpret_cast<const float4*>(map + i0 + 2);
            z[0] = make_float2(v0.x, v0.y); z[1] = make_float2(v0.z, v0.w);
            z[2] = make_float2(v1.x, v1.y); z[3] = make_float2(v1.z, v1.w);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) z[j] = (i0 + j < total) ? map[i0 + j] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) if (i0 + j < total) m = fmaxf(m, fast_power(z[j]));
        const float thr = t.raise(m);
        if (m >= thr) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (i0 + j < total && fast_power(z[j]) >= thr) t.exact(z[j], (unsigned)(i0 + j));
        }
    }
    block_reduce_peak(t, red);
    if (threadIdx.x == 0) { partials[blockIdx.x].best = t.best; partials[blockIdx.x].idx = t.idx; }
}

int launch_ra_partial(jrc_ctx* ctx, const float2* d_map, size_t total, PeakPartial* d_partials, int n_blocks,
                      hipStream_t stream)
{
    hipLaunchKernelGGL(ra_partial_kernel, dim3(n_blocks), dim3(256), 0, stream, d_map, total, d_partials);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// per-frame epilogue: merge the partial maxima, locate the null-angle noise window and sum it in the
// reference's order (one lane adds sequentially; the other lanes only pre-compute |z| into LDS).
#define RA_CHUNK 2048

__global__ __launch_bounds__(256) void ra_finalize_kernel(const float2* __restrict__ maps, size_t map_stride,
                                                          const PeakPartial* __restrict__ partials, int ppf,
                                                          RaParams prm, const float* __restrict__ range_bins,
                                                          const float* __restrict__ angle_bins,
                                                          jrc_ra_result* __restrict__ results,
                                                          int force_ref_sum,   // tests: take the reference-order double sum for every chunk
                                                          int win_rows)        // > 0 (detect-only mode): `maps` holds only the noise-window rows of each
                                                                               // frame, [win_rows][vlen], row i = range bin start_range + i (chain.hip MODE 2)
{
    __shared__ PeakPartial red[4];
    __shared__ int s_win[5];     // start_range, end_range, start_angle, end_angle, valid
    __shared__ __attribute__((aligned(16))) float s_h[RA_CHUNK];
    __shared__ __attribute__((aligned(16))) float s_run[RA_CHUNK];   // running sums of the fast chain
    const int f = blockIdx.x;
    const float2* map = maps + (size_t)f * map_stride;
    const int vlen = prm.vlen, n_inputs = prm.n_inputs;

    if (prm.n_angle_bins <= RA_CHUNK)         // angle bins -> LDS (s_run is free until the noise sum); visible after the barrier in the reduction
        for (int i = threadIdx.x; i < prm.n_angle_bins; i += blockDim.x) s_run[i] = angle_bins[i];
    PeakTracker t;
    t.init();
    for (int i = threadIdx.x; i < ppf; i += blockDim.x) t.merge(partials[(size_t)f * ppf + i].best, partials[(size_t)f * ppf + i].idx);
    block_reduce_peak(t, red);

    jrc_ra_result r;
    if (threadIdx.x == 0) {
        const int nab = prm.n_angle_bins, nrb = prm.n_range_bins;
        r.peak_range_idx = (int)(t.idx / (unsigned)vlen);
        r.peak_angle_idx = (int)(t.idx % (unsigned)vlen);
        r.peak_power = t.best;
        // the angle bins were staged in LDS at kernel entry (more bins than the staging buffer holds: read in place)
        const float* ab = nab <= RA_CHUNK ? s_run : angle_bins;
        r.angle_val = ab[r.peak_angle_idx];
        r.range_val = range_bins[r.peak_range_idx];
        float angle_null = r.angle_val + 90;                 // :155-160
        if (angle_null >= 90) angle_null = angle_null - 180;
        int lo = 0, hi = nab;                                // std::lower_bound (:163-167): log2(n) dependent LDS reads instead of global loads
        while (lo < hi) { int mid = lo + (hi - lo) / 2; if (ab[mid] < angle_null) lo = mid + 1; else hi = mid; }
        int null_idx;
        if (lo == 0) null_idx = 0;                           // :172-173
        else if (lo == nab) null_idx = nab - 1;              // iter == end(): defined as size-1
        else {
            double a = ab[lo - 1], b = ab[lo];
            null_idx = (fabs(angle_null - a) < fabs(angle_null - b)) ? lo - 1 : lo;   // :175-180
        }
        if (null_idx == nab - 1) null_idx = nab - 2;         // :184-187
        r.angle_null_idx = null_idx;
        int dr = (int)(prm.noise_discard_range_m / (range_bins[1] - range_bins[0]));                          // :189
        int da = (int)(prm.noise_discard_angle_deg / (ab[(null_idx + 1) % nab] - ab[null_idx])); // :190
        if (da <= 0) da = 1;                                 // :192-195
        r.discard_range_idx = dr; r.discard_angle_idx = da;
        s_win[0] = r.peak_range_idx + nrb / 2 - dr;          // :197-201
        s_win[1] = r.peak_range_idx + nrb / 2 + dr;
        s_win[2] = null_idx - da;
        s_win[3] = null_idx + da;
    }
    __syncthreads();
    const int sr = s_win[0], er = s_win[1], sa = s_win[2], ea = s_win[3];
    const int wr = er > sr ? er - sr : 0, wa = ea > sa ? ea - sa : 0;
    const int ncells = wr * wa;
    if (win_rows > 0 && wr != win_rows) {       // the host sized the window buffer with another discard_range_idx than the device computes: refuse
        if (threadIdx.x == 0) { r.n_noise_samples = -1; r.noise_power = nanf(""); r.snr_est = 0.f; r.published = 0; results[f] = r; }
        return;
    }
    float noise = 0.f;
    for (int base = 0; base < ncells; base += RA_CHUNK) {
        const int cnt = min(RA_CHUNK, ncells - base);
        for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
            const int c = base + j;
            const int ir = sr + c / wa, ia = sa + c % wa;
            const int r_idx = ((ir % n_inputs) + n_inputs) % n_inputs;      // :211
            const int a_idx = ((ia % vlen) + vlen) % vlen;                  // :215
            s_h[j] = ref_hypotf(win_rows > 0 ? map[(size_t)a_idx + (size_t)vlen * (c / wa)] : map[(size_t)a_idx + (size_t)vlen * r_idx]);
        }
        __syncthreads();
        // The reference adds in order, `float += double` rounded at every step (:216): a chain of three dependent double-precision
        // operations per cell on one lane.  fmaf(h, h, noise) rounds the same exact sum once instead of twice and differs from it only
        // when the intermediate double lands on a float tie (~2^-29 per step), so: one lane runs the chain with fmaf and records the
        // running sums, all lanes then check their steps against the reference expression in parallel, and a chunk with any mismatch
        // is redone the slow way.  Bit-exact, ~2.5x shorter.
        if (threadIdx.x == 0) {
            float sacc = noise;
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {                  // eight values per trip: the LDS reads run ahead of the dependent chain
                const float4 a = *reinterpret_cast<const float4*>(s_h + j), b = *reinterpret_cast<const float4*>(s_h + j + 4);
                float4 ra, rb;
                sacc = fmaf(a.x, a.x, sacc); ra.x = sacc; sacc = fmaf(a.y, a.y, sacc); ra.y = sacc;
                sacc = fmaf(a.z, a.z, sacc); ra.z = sacc; sacc = fmaf(a.w, a.w, sacc); ra.w = sacc;
                sacc = fmaf(b.x, b.x, sacc); rb.x = sacc; sacc = fmaf(b.y, b.y, sacc); rb.y = sacc;
                sacc = fmaf(b.z, b.z, sacc); rb.z = sacc; sacc = fmaf(b.w, b.w, sacc); rb.w = sacc;
                *reinterpret_cast<float4*>(s_run + j) = ra; *reinterpret_cast<float4*>(s_run + j + 4) = rb;
            }
            for (; j < cnt; j++) { const float h = s_h[j]; sacc = fmaf(h, h, sacc); s_run[j] = sacc; }
        }
        __syncthreads();
        int bad = force_ref_sum;
        for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
            const float prev = j ? s_run[j - 1] : noise;
            const double h = (double)s_h[j];
            bad |= ((float)((double)prev + h * h) != s_run[j]);
        }
        if (__syncthreads_or(bad)) {
            if (threadIdx.x == 0) {
                for (int j = 0; j < cnt; j++) {
                    double h = (double)s_h[j];
                    noise = (float)((double)noise + h * h);
                }
                s_run[cnt - 1] = noise;
            }
            __syncthreads();
        }
        noise = s_run[cnt - 1];                             // every lane carries the sum: lane j = 0 of the next chunk checks against it
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        r.n_noise_samples = ncells;
        r.noise_power = noise / ncells;                      // :231
        r.snr_est = 0.f;                                     // completed on the host (libm log10f)
        r.published = 0;
        results[f] = r;
    }
}

int launch_ra_finalize(jrc_ctx* ctx, const float2* d_map, size_t map_stride, const PeakPartial* d_partials,
                       int partials_per_frame, const RaParams& prm, const float* d_range_bins,
                       const float* d_angle_bins, jrc_ra_result* d_results, int n_frames, int win_rows, hipStream_t stream)
{
    hipLaunchKernelGGL(ra_finalize_kernel, dim3(n_frames), dim3(256), 0, stream, d_map, map_stride, d_partials,
                       partials_per_frame, prm, d_range_bins, d_angle_bins, d_results, ctx->tune.ra_ref_sum ? 1 : 0, win_rows);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

void ra_finish_host(jrc_ra_result* r, float snr_threshold, float power_threshold)
{
    r->snr_est = 10 * log10f(r->peak_power / r->noise_power);                               // :232
    r->published = (r->snr_est >= snr_threshold && r->peak_power >= power_threshold);        // :234
}

extern "C" int jrc_ra_estimate(jrc_ctx* ctx, int vlen, int n_inputs, const jrc_cf32* in, const float* range_bins,
                               int n_range_bins, const float* angle_bins, int n_angle_bins,
                               float noise_discard_range_m, float noise_discard_angle_deg, float snr_threshold,
                               float power_threshold, jrc_ra_result* result)
{
    if (!ctx || !in || !range_bins || !angle_bins || !result) return JRC_ERR_INVALID_ARG;
    if (vlen <= 0 || n_inputs <= 0 || n_range_bins < 2 || n_angle_bins < 2 || n_inputs > n_range_bins ||
        vlen > n_angle_bins)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "range_angle_estimator: map %dx%d does not fit bins %d/%d", n_inputs,
                        vlen, n_range_bins, n_angle_bins);
    const size_t total = (size_t)vlen * n_inputs;
    if (total > 0xfffffff0ull) return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "range_angle_estimator: map too large");
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t map_bytes = sizeof(float2) * total;
    const size_t bins_bytes = sizeof(float) * (size_t)(n_range_bins + n_angle_bins);
    int n_blocks = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (n_blocks > 2048) n_blocks = 2048;
    if (n_blocks < 1) n_blocks = 1;
    JRC_TRY(jrc_ensure_pinned(ctx, map_bytes + bins_bytes + sizeof(jrc_ra_result)));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, map_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, bins_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 2, sizeof(PeakPartial) * n_blocks + sizeof(jrc_ra_result)));
    char* hp = (char*)ctx->pinned;
    jrc_host_copy(hp, in, map_bytes);
    memcpy(hp + map_bytes, range_bins, sizeof(float) * n_range_bins);
    memcpy(hp + map_bytes + sizeof(float) * n_range_bins, angle_bins, sizeof(float) * n_angle_bins);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], hp, map_bytes, hipMemcpyHostToDevice, ctx->stream));
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[1], hp + map_bytes, bins_bytes, hipMemcpyHostToDevice, ctx->stream));
    PeakPartial* d_part = (PeakPartial*)ctx->scratch[2];
    jrc_ra_result* d_res = (jrc_ra_result*)((char*)ctx->scratch[2] + sizeof(PeakPartial) * n_blocks);
    JRC_TRY(launch_ra_partial(ctx, (const float2*)ctx->scratch[0], total, d_part, n_blocks, ctx->stream));
    RaParams prm;
    prm.vlen = vlen; prm.n_inputs = n_inputs; prm.n_range_bins = n_range_bins; prm.n_angle_bins = n_angle_bins;
    prm.noise_discard_range_m = noise_discard_range_m; prm.noise_discard_angle_deg = noise_discard_angle_deg;
    const float* d_rb = (const float*)ctx->scratch[1];
    JRC_TRY(launch_ra_finalize(ctx, (const float2*)ctx->scratch[0], total, d_part, n_blocks, prm, d_rb,
                               d_rb + n_range_bins, d_res, 1, 0, ctx->stream));
    jrc_ra_result* h_res = (jrc_ra_result*)(hp + map_bytes + bins_bytes);
    JRC_HIP(ctx, hipMemcpyAsync(h_res, d_res, sizeof(jrc_ra_result), hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *result = *h_res;
    ra_finish_host(result, snr_threshold, power_threshold);
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// B1 fft_peak_detect: arg-max of |x| over [prot, n-prot) among samples with |x|^2 > 10^(thr/10)
struct MagPartial { float mag; int idx; };

__global__ __launch_bounds__(256) void peak_mag_partial_kernel(const float2* __restrict__ in, long lo, long hi,
                                                               double thr, MagPartial* __restrict__ partials)
{
    __shared__ MagPartial red[4];
    float best = -1.f;
    int idx = -1;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = lo + (long)blockIdx.x * blockDim.x + threadIdx.x; p < hi; p += stride) {
        float m = ref_hypotf(in[p]);                         // std::abs(in[p])
        if ((double)m * (double)m > thr) {                   // std::pow(abs,2) > std::pow(10, thr/10.0)  (:91)
            if (m > best || (m == best && (idx < 0 || p < idx))) { best = m; idx = (int)p; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        float ob = __shfl_xor(best, off);
        int oi = __shfl_xor(idx, off);
        if (oi >= 0 && (ob > best || (ob == best && (idx < 0 || oi < idx)))) { best = ob; idx = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave].mag = best; red[wave].idx = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++)
            if (red[w].idx >= 0 && (red[w].mag > best || (red[w].mag == best && (idx < 0 || red[w].idx < idx)))) {
                best = red[w].mag; idx = red[w].idx;
            }
        partials[blockIdx.x].mag = best; partials[blockIdx.x].idx = idx;
    }
}

extern "C" int jrc_fft_peak_detect(jrc_ctx* ctx, int samp_rate, float interp_factor, float threshold,
                                   int samp_protect, size_t ninput_items, const jrc_cf32* in, float* out_freq,
                                   float* out_phase, float* out_mag, int* k_out)
{
    if (!ctx || !in || !out_freq || !out_phase || !out_mag) return JRC_ERR_INVALID_ARG;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const long n = (long)ninput_items;
    const long lo = samp_protect, hi = n - samp_protect;     // :90
    int k = -1;
    if (hi > lo && lo >= 0) {
        const size_t bytes = sizeof(float2) * ninput_items;
        int n_blocks = (int)((hi - lo + 1023) / 1024);
        if (n_blocks > 512) n_blocks = 512;
        JRC_TRY(jrc_ensure_pinned(ctx, bytes + sizeof(MagPartial) * n_blocks));
        JRC_TRY(jrc_ensure_scratch(ctx, 0, bytes));
        JRC_TRY(jrc_ensure_scratch(ctx, 1, sizeof(MagPartial) * n_blocks));
        jrc_host_copy(ctx->pinned, in, bytes);
        JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
        const double thr = pow(10, threshold / 10.0);        // std::pow(10, d_threshold / 10.0)
        hipLaunchKernelGGL(peak_mag_partial_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream,
                           (const float2*)ctx->scratch[0], lo, hi, thr, (MagPartial*)ctx->scratch[1]);
        JRC_HIP(ctx, hipGetLastError());
        MagPartial* hp = (MagPartial*)((char*)ctx->pinned + bytes);
        JRC_HIP(ctx, hipMemcpyAsync(hp, ctx->scratch[1], sizeof(MagPartial) * n_blocks, hipMemcpyDeviceToHost, ctx->stream));
        JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float best = -1.f;
        for (int b = 0; b < n_blocks; b++)
            if (hp[b].idx >= 0 && (hp[b].mag > best || (hp[b].mag == best && (k < 0 || hp[b].idx < k)))) {
                best = hp[b].mag; k = hp[b].idx;
            }
    }
    if (k != -1) {                                           // :98-107 (scalar epilogue stays on the host: libm-exact)
        const int ni = (int)ninput_items;
        if (k <= ni / 2) out_freq[0] = k / (float)ni * (samp_rate * interp_factor);
        else out_freq[0] = -((float)samp_rate * interp_factor) + k * (samp_rate * interp_factor / (float)ni);
        out_phase[0] = atan2f(in[k].im, in[k].re);
        out_mag[0] = hypotf(in[k].re, in[k].im);
    }
    if (k_out) *k_out = k;
    return 1;                                                // :110
}
