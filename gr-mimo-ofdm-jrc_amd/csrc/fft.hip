// fft.hip — stock-block stages between the reference's hot-path blocks, plus the two copy blocks
//
//   A2/A4/A7  gr::fft::fft_vcc (FFTW3f inside GNU Radio 3.8; wired in
//             examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:877-1047)
//   A3        matrix_transpose_impl::work          (reference lib/matrix_transpose_impl.cc:69-110)
//   A6        ofdm_cyclic_prefix_remover_impl::work (reference lib/ofdm_cyclic_prefix_remover_impl.cc:69-99)
//
// The FFT here is the literal per-block drop-in (any power of two up to 16384, LDS resident, one
// transform per workgroup or several small ones packed into one).  The roofline-critical 2-D
// range-angle transform does NOT go through this kernel; it is the fused kernel in chain.hip.
#include "radar_kernels.h"
#include "fft_device.h"

// ------------------------------------------------------------------------------------------------
// in-place radix-2 decimation-in-frequency in LDS; input natural order, result bit-reversed in LDS,
// un-permuted (and fftshift-ed) on the way out.  tw[k] = exp(sign*j*2*pi*k/n), k < n.
__global__ __launch_bounds__(256) void fft_pow2_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                       const float2* __restrict__ tw,
                                                       const float* __restrict__ window, int n, int logn,
                                                       int forward, int shift, size_t batch, long in_stride,
                                                       int in_offset, int tp /* threads per transform */,
                                                       long out_stride, int cp_out /* cyclic prefix to prepend */)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int per_block = blockDim.x / tp;
    const int lt = threadIdx.x % tp;
    const int lb = threadIdx.x / tp;
    const size_t b = (size_t)blockIdx.x * per_block + lb;
    const bool live = b < batch;
    float2* x = lds + (size_t)lb * n;
    const int half_n = n >> 1;

    if (live) {
        const float2* src = in + b * (size_t)in_stride + in_offset;
        for (int i = lt; i < n; i += tp) {
            int si = (!forward && shift) ? ((i + half_n) & (n - 1)) : i;   // ifftshift on the way in
            float2 v = src[si];
            if (window) { float w = window[si]; v.x *= w; v.y *= w; }
            x[i] = v;
        }
    }
    for (int half = half_n; half >= 1; half >>= 1) {
        __syncthreads();
        if (live) {
            const int tstep = half_n / half;
            for (int j = lt; j < half_n; j += tp) {
                const int k = j & (half - 1);
                const int i0 = ((j - k) << 1) + k;
                const int i1 = i0 + half;
                float2 a = x[i0], c = x[i1];
                x[i0] = cadd(a, c);
                float2 d = csub(a, c);
                x[i1] = (k == 0) ? d : cmul(d, tw[k * tstep]);
            }
        }
    }
    __syncthreads();
    if (live) {
        float2* dst = out + b * (size_t)out_stride + cp_out;
        for (int pos = lt; pos < n; pos += tp) {
            int k = (forward && shift) ? ((pos + half_n) & (n - 1)) : pos;   // fftshift on the way out
            unsigned r = __brev((unsigned)k) >> (32 - logn);
            dst[pos] = x[logn ? r : 0];
        }
        for (int j = lt; j < cp_out; j += tp) {                              // cyclic prefix = last cp_out samples
            unsigned r = __brev((unsigned)(n - cp_out + j)) >> (32 - logn);
            dst[j - cp_out] = x[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Stockham autosort FFT, radix 4 (one leading radix-2 pass when log2 n is odd): natural order in and out, so no
// bit-reversal gather; the first pass reads global memory directly and the last pass writes it directly, with
// coalesced 512-byte wave accesses on both sides.  n <= 8192 (two LDS buffers of n points).
__global__ __launch_bounds__(256) void fft_stockham_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                           const float2* __restrict__ tw, const float* __restrict__ window,
                                                           int n, int logn, int forward, int shift, size_t batch,
                                                           long in_stride, int in_offset, int tp, long out_stride, int cp_out)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int per_block = blockDim.x / tp;
    const int lt = threadIdx.x % tp, lb = threadIdx.x / tp;
    const size_t b = (size_t)blockIdx.x * per_block + lb;
    const bool live = b < batch;
    float2* buf0 = lds + (size_t)lb * 2 * n;
    float2* buf1 = buf0 + n;
    const int sign = forward ? -1 : 1;
    const float2* src_g = in + b * (size_t)in_stride + in_offset;
    float2* dst_g = out + b * (size_t)out_stride + cp_out;
    const long wrap = (!forward && shift) ? n : 0;
    const int rot = (forward && shift) ? (n >> 1) : 0;
    const bool need_copy = cp_out > 0;            // the cyclic prefix needs the finished symbol: last pass goes through LDS

    int Ns = 1;
    const float2* cur = nullptr;                  // nullptr = still in global memory
    float2* nxt = buf0;
    bool first = true;
    while (Ns < n) {
        const int R = ((logn & 1) && first) ? 2 : 4;
        const bool last = Ns * R == n;
        float2* dl = (last && !need_copy) ? nullptr : nxt;
        float2* dg = (last && !need_copy) ? dst_g : nullptr;
        if (live) {
            if (R == 2) stockham_pass<2>(first ? src_g : nullptr, wrap, first ? window : nullptr, cur, dl, dg, rot, tw, n, Ns, sign, lt, tp);
            else stockham_pass<4>(first ? src_g : nullptr, wrap, first ? window : nullptr, cur, dl, dg, rot, tw, n, Ns, sign, lt, tp);
        }
        __syncthreads();
        cur = nxt; nxt = (nxt == buf0) ? buf1 : buf0;
        Ns *= R; first = false;
    }
    if (need_copy && live) {
        for (int pos = lt; pos < n; pos += tp) dst_g[pos] = cur[rot ? ((pos + rot) & (n - 1)) : pos];
        for (int jj = lt; jj < cp_out; jj += tp) { const int pos = n - cp_out + jj; dst_g[jj - cp_out] = cur[rot ? ((pos + rot) & (n - 1)) : pos]; }
    }
}

// ------------------------------------------------------------------------------------------------
#define JRC_FFT_MAX_ANY 4096     // largest non-power-of-two fft_size (chirp-z over M <= 8192 points in LDS)

__global__ void fft_copy1_kernel(const float2* __restrict__ in, float2* __restrict__ out, const float* __restrict__ window, size_t batch,
                                 long in_stride, int in_offset, long out_stride, int cp_out)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    float2 v = in[b * (size_t)in_stride + in_offset];
    if (window) { v.x *= window[0]; v.y *= window[0]; }
    float2* d = out + b * (size_t)out_stride + cp_out;
    d[0] = v;
    for (int j = 0; j < cp_out; j++) d[j - cp_out] = v;
}

// fft sizes that are not powers of two (3 TX x 2 RX -> 96 angle bins, fft_len 48, ...): chirp-z transform in LDS.
//   X[k] = c[k] sum_i (x[i] c[i]) conj(c)[k - i],  c[i] = exp(sign j pi i^2 / n)
// as one circular convolution of length M = 2^m >= 2n-1: load x.c zero-padded, Stockham FFT_M, multiply by the
// precomputed spectrum of the wrapped conj chirp (/M folded in), inverse Stockham FFT_M, multiply by c.  Same window /
// shift / stride / cyclic-prefix semantics as the power-of-two kernels (gr-fft 3.8 fft_vcc: forward shift rotates the
// output by ceil(n/2), reverse shift rotates the input by floor(n/2)).
__global__ __launch_bounds__(256) void fft_bluestein_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                            const float2* __restrict__ chirp, const float2* __restrict__ bhat,
                                                            const float2* __restrict__ tw_f, const float2* __restrict__ tw_i,
                                                            const float* __restrict__ window, int n, int M, int logM, int forward,
                                                            int shift, size_t batch, long in_stride, int in_offset, int tp,
                                                            long out_stride, int cp_out)
{
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int per_block = blockDim.x / tp;
    const int lt = threadIdx.x % tp, lb = threadIdx.x / tp;
    const size_t b = (size_t)blockIdx.x * per_block + lb;
    const bool live = b < batch;
    float2* buf0 = lds + (size_t)lb * 2 * M;
    float2* buf1 = buf0 + M;
    const float2* src_g = in + b * (size_t)in_stride + in_offset;
    float2* dst_g = out + b * (size_t)out_stride + cp_out;
    const int rot_in = (!forward && shift) ? n / 2 : 0;
    const int rot_out = (forward && shift) ? n / 2 : 0;           // X[k] lands at (k + floor(n/2)) % n  <=>  out[j] = X[(j + ceil(n/2)) % n]
    if (live)
        for (int i = lt; i < M; i += tp) {
            float2 v = make_float2(0.f, 0.f);
            if (i < n) {
                int si = i + rot_in; if (si >= n) si -= n;
                v = src_g[si];
                if (window) { const float w = window[si]; v.x *= w; v.y *= w; }
                v = cmul(v, chirp[i]);
            }
            buf0[i] = v;
        }
    __syncthreads();
    const float2* cur = buf0;
    float2* nxt = buf1;
    for (int dir = 0; dir < 2; dir++) {
        const float2* tw = dir ? tw_i : tw_f;
        const int sign = dir ? 1 : -1;
        int Ns = 1;
        bool first = true;
        while (Ns < M) {
            const int R = ((logM & 1) && first) ? 2 : 4;
            if (live) {
                if (R == 2) stockham_pass<2>(nullptr, 0, nullptr, cur, nxt, nullptr, 0, tw, M, Ns, sign, lt, tp);
                else stockham_pass<4>(nullptr, 0, nullptr, cur, nxt, nullptr, 0, tw, M, Ns, sign, lt, tp);
            }
            __syncthreads();
            const float2* t = cur; cur = nxt; nxt = const_cast<float2*>(t);
            Ns *= R; first = false;
        }
        if (dir == 0) {
            if (live) {
                float2* w = const_cast<float2*>(cur);
                for (int k = lt; k < M; k += tp) w[k] = cmul(w[k], bhat[k]);
            }
            __syncthreads();
        }
    }
    if (live) {
        for (int k = lt; k < n; k += tp) {                        // results to the other buffer in output order
            int j = k + rot_out; if (j >= n) j -= n;
            nxt[j] = cmul(cur[k], chirp[k]);
        }
    }
    __syncthreads();
    if (live) {
        for (int j = lt; j < n; j += tp) dst_g[j] = nxt[j];
        for (int jj = lt; jj < cp_out; jj += tp) dst_g[jj - cp_out] = nxt[n - cp_out + jj];
    }
}

static int launch_fft_vcc_ex(jrc_ctx* ctx, int n, int forward, int shift, const float* d_window, size_t batch,
                             const float2* d_in, float2* d_out, long in_stride, int in_offset, long out_stride, int cp_out,
                             hipStream_t stream);

int launch_fft_vcc(jrc_ctx* ctx, int n, int forward, int shift, const float* d_window, size_t batch,
                   const float2* d_in, float2* d_out, long in_stride, int in_offset, hipStream_t stream)
{
    return launch_fft_vcc_ex(ctx, n, forward, shift, d_window, batch, d_in, d_out, in_stride, in_offset, n, 0, stream);
}

static int launch_fft_vcc_ex(jrc_ctx* ctx, int n, int forward, int shift, const float* d_window, size_t batch,
                             const float2* d_in, float2* d_out, long in_stride, int in_offset, long out_stride, int cp_out,
                             hipStream_t stream)
{
    if (n < 1 || (jrc_is_pow2(n) ? n > 16384 : n > JRC_FFT_MAX_ANY))
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "fft_vcc: fft_size %d is outside [1, 16384] (powers of two) / [1, %d] (any size)", n, JRC_FFT_MAX_ANY);
    if (batch == 0) return JRC_OK;
    if (n == 1) {                        // the 1-point transform is a copy (window applied)
        hipLaunchKernelGGL(fft_copy1_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, stream, d_in, d_out, d_window, batch,
                           in_stride, in_offset, out_stride, cp_out);
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    if (!jrc_is_pow2(n)) {
        jrc_ctx::bluestein_tab bt;
        JRC_TRY(jrc_get_bluestein(ctx, n, forward ? -1 : +1, &bt));
        const float2 *twf = nullptr, *twi = nullptr;
        JRC_TRY(jrc_get_twiddles(ctx, bt.M, -1, &twf));
        JRC_TRY(jrc_get_twiddles(ctx, bt.M, +1, &twi));
        int tp = bt.M / 4; if (tp > 256) tp = 256; if (tp < 1) tp = 1;
        const int per_block = 256 / tp;
        const size_t blocks = (batch + per_block - 1) / per_block;
        const size_t lds_bytes = sizeof(float2) * 2 * (size_t)bt.M * per_block;
        JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)fft_bluestein_kernel, lds_bytes));
        hipLaunchKernelGGL(fft_bluestein_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, d_in, d_out, (const float2*)bt.chirp,
                           (const float2*)bt.bhat, twf, twi, d_window, n, bt.M, jrc_ilog2(bt.M), forward, shift, batch, in_stride,
                           in_offset, tp, out_stride, cp_out);
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    const float2* tw = nullptr;
    JRC_TRY(jrc_get_twiddles(ctx, n, forward ? -1 : +1, &tw));
    const int logn = jrc_ilog2(n);
    if (n >= 4 && n <= 8192) {            // Stockham radix-4, natural order (the default)
        int tp = n / 4; if (tp > 256) tp = 256; if (tp < 1) tp = 1;
        const int per_block = 256 / tp;
        const size_t blocks = (batch + per_block - 1) / per_block;
        const size_t lds_bytes = sizeof(float2) * 2 * (size_t)n * per_block;
        JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)fft_stockham_kernel, lds_bytes));
        hipLaunchKernelGGL(fft_stockham_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, d_in, d_out, tw,
                           d_window, n, logn, forward, shift, batch, in_stride, in_offset, tp, out_stride, cp_out);
        JRC_HIP(ctx, hipGetLastError());
        return JRC_OK;
    }
    int tp = n / 2; if (tp > 256) tp = 256; if (tp < 1) tp = 1;
    const int per_block = 256 / tp;
    const size_t blocks = (batch + per_block - 1) / per_block;
    const size_t lds_bytes = sizeof(float2) * (size_t)n * per_block;
    JRC_TRY(jrc_ensure_dyn_lds(ctx, (const void*)fft_pow2_kernel, lds_bytes));
    hipLaunchKernelGGL(fft_pow2_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, stream, d_in, d_out, tw,
                       d_window, n, logn, forward, shift, batch, in_stride, in_offset, tp, out_stride, cp_out);
    JRC_HIP(ctx, hipGetLastError());
    return JRC_OK;
}

extern "C" int jrc_fft_vcc_dev(jrc_ctx* ctx, int fft_size, int forward, int shift, const float* d_window,
                               size_t batch, const jrc_cf32* d_in, jrc_cf32* d_out, void* stream)
{
    if (!ctx || !d_in || !d_out) return JRC_ERR_INVALID_ARG;
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    return launch_fft_vcc(ctx, fft_size, forward, shift, d_window, batch, (const float2*)d_in, (float2*)d_out,
                          fft_size, 0, s);
}

extern "C" int jrc_fft_vcc(jrc_ctx* ctx, int fft_size, int forward, int shift, const float* window, size_t batch,
                           const jrc_cf32* in, jrc_cf32* out)
{
    if (!ctx || !in || !out) return JRC_ERR_INVALID_ARG;
    if (fft_size < 1 || (jrc_is_pow2(fft_size) ? fft_size > 16384 : fft_size > JRC_FFT_MAX_ANY))
        return jrc_fail(ctx, JRC_ERR_UNSUPPORTED, "fft_vcc: fft_size %d is outside [1, 16384] (powers of two) / [1, %d] (any size)", fft_size, JRC_FFT_MAX_ANY);
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = sizeof(float2) * (size_t)fft_size * batch;
    const size_t wbytes = window ? sizeof(float) * (size_t)fft_size : 0;
    JRC_TRY(jrc_ensure_pinned(ctx, bytes + wbytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 2, wbytes ? wbytes : 4));
    jrc_host_copy(ctx->pinned, in, bytes);
    if (window) memcpy((char*)ctx->pinned + bytes, window, wbytes);
    if (bytes) JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (window) JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[2], (char*)ctx->pinned + bytes, wbytes, hipMemcpyHostToDevice, ctx->stream));
    JRC_TRY(launch_fft_vcc(ctx, fft_size, forward, shift, window ? (const float*)ctx->scratch[2] : nullptr, batch,
                           (const float2*)ctx->scratch[0], (float2*)ctx->scratch[1], fft_size, 0, ctx->stream));
    if (bytes) JRC_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->scratch[1], bytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    jrc_host_copy(out, ctx->pinned, bytes);
    return JRC_OK;
}

// ------------------------------------------------------------------------------------------------
// A3 matrix_transpose: out[l][k] = in[k][l] (k < ninput_items), zero for ninput_items <= k < W
// 64x64 tiles through LDS so both the reads (along l) and the writes (along k) are coalesced.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                            int input_len, int ninput, int W)
{
    __shared__ float2 tile[64][65];
    const size_t b = blockIdx.z;
    const float2* src = in + b * (size_t)ninput * input_len;
    float2* dst = out + b * (size_t)input_len * W;
    const int l0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    const bool any = k0 < ninput;
    if (any) {
        for (int kk = ty; kk < 64; kk += 4) {
            int k = k0 + kk, l = l0 + tx;
            if (k < ninput && l < input_len) tile[kk][tx] = src[(size_t)k * input_len + l];
        }
    }
    __syncthreads();
    for (int ll = ty; ll < 64; ll += 4) {
        int l = l0 + ll, k = k0 + tx;
        if (l < input_len && k < W) {
            float2 v = make_float2(0.f, 0.f);
            if (any && k < ninput) v = tile[tx][ll];
            dst[(size_t)l * W + k] = v;
        }
    }
}

static int transpose_check(jrc_ctx* ctx, int input_len, int output_len, int interp_factor, int ninput_items)
{
    if (input_len <= 0 || output_len <= 0 || interp_factor <= 0 || ninput_items < 0)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "matrix_transpose: non-positive size");
    // lib/matrix_transpose_impl.cc:82  (float division vs integer division)
    if (ninput_items * float(input_len) / float(output_len) - ninput_items * input_len / output_len != 0)
        return jrc_fail(ctx, JRC_ERR_LENGTH_MISMATCH, "%s", jrc_strerror(JRC_ERR_LENGTH_MISMATCH));
    if (ninput_items > output_len * interp_factor)
        return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "matrix_transpose: %d input items do not fit an output row of %d",
                        ninput_items, output_len * interp_factor);
    return JRC_OK;
}

extern "C" int jrc_matrix_transpose_dev(jrc_ctx* ctx, int input_len, int output_len, int interp_factor,
                                        int ninput_items, size_t batch, const jrc_cf32* d_in, jrc_cf32* d_out,
                                        void* stream)
{
    if (!ctx || !d_in || !d_out) return JRC_ERR_INVALID_ARG;
    JRC_TRY(transpose_check(ctx, input_len, output_len, interp_factor, ninput_items));
    if (batch == 0) return input_len;
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int W = output_len * interp_factor;
    // gridDim.z carries the batch and is limited to 65535: larger batches go in chunks
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        const size_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        dim3 grid((input_len + 63) / 64, (W + 63) / 64, (unsigned)nb);
        hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, s, (const float2*)d_in + b0 * (size_t)ninput_items * input_len,
                           (float2*)d_out + b0 * (size_t)input_len * W, input_len, ninput_items, W);
    }
    JRC_HIP(ctx, hipGetLastError());
    return input_len;
}

extern "C" int jrc_matrix_transpose(jrc_ctx* ctx, int input_len, int output_len, int interp_factor,
                                    int ninput_items, const jrc_cf32* in, jrc_cf32* out)
{
    if (!ctx || !in || !out) return JRC_ERR_INVALID_ARG;
    JRC_TRY(transpose_check(ctx, input_len, output_len, interp_factor, ninput_items));
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = sizeof(float2) * (size_t)ninput_items * input_len;
    const size_t out_bytes = sizeof(float2) * (size_t)input_len * output_len * interp_factor;
    JRC_TRY(jrc_ensure_pinned(ctx, in_bytes > out_bytes ? in_bytes : out_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, in_bytes ? in_bytes : 8));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, out_bytes));
    jrc_host_copy(ctx->pinned, in, in_bytes);
    if (in_bytes) JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    int r = jrc_matrix_transpose_dev(ctx, input_len, output_len, interp_factor, ninput_items, 1,
                                     (const jrc_cf32*)ctx->scratch[0], (jrc_cf32*)ctx->scratch[1], nullptr);
    if (r < 0) return r;
    JRC_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->scratch[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    jrc_host_copy(out, ctx->pinned, out_bytes);
    return input_len;
}

// ------------------------------------------------------------------------------------------------
// A6 cyclic prefix removal: out[k][0..N) = in[k*(N+cp)+cp ...]
__global__ void cp_remove_kernel(const float2* __restrict__ in, float2* __restrict__ out, int n, int cp,
                                 size_t total /* nsym*n */)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        size_t k = i / n;
        int j = (int)(i - k * n);
        out[i] = in[k * (size_t)(n + cp) + cp + j];
    }
}

extern "C" int jrc_cp_remove(jrc_ctx* ctx, int fft_len, int cp_len, size_t ninput_items, const jrc_cf32* in,
                             jrc_cf32* out)
{
    if (!ctx || !in || !out) return JRC_ERR_INVALID_ARG;
    if (fft_len <= 0 || cp_len < 0) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "cp_remover: bad fft_len/cp_len");
    const size_t nsym = ninput_items / (size_t)(fft_len + cp_len);   // :86
    if (nsym == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = sizeof(float2) * nsym * (size_t)(fft_len + cp_len);
    const size_t out_bytes = sizeof(float2) * nsym * (size_t)fft_len;
    JRC_TRY(jrc_ensure_pinned(ctx, in_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, in_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, out_bytes));
    jrc_host_copy(ctx->pinned, in, in_bytes);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    const size_t total = nsym * (size_t)fft_len;
    unsigned blocks = (unsigned)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cp_remove_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const float2*)ctx->scratch[0],
                       (float2*)ctx->scratch[1], fft_len, cp_len, total);
    JRC_HIP(ctx, hipGetLastError());
    JRC_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->scratch[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    jrc_host_copy(out, ctx->pinned, out_bytes);
    return (int)nsym;
}

// A6 + A7 fused: strided load (skipping the prefix) straight into the LDS FFT, fftshift on the way out
extern "C" int jrc_cp_remove_fft_dev(jrc_ctx* ctx, int fft_len, int cp_len, size_t n_symbols, const jrc_cf32* d_in,
                                     jrc_cf32* d_out, void* stream)
{
    if (!ctx || !d_in || !d_out) return JRC_ERR_INVALID_ARG;
    if (cp_len < 0) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "cp_remover: bad cp_len");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    JRC_TRY(launch_fft_vcc(ctx, fft_len, 1, 1, nullptr, n_symbols, (const float2*)d_in, (float2*)d_out,
                           (long)fft_len + cp_len, cp_len, s));
    return (int)n_symbols;
}

extern "C" int jrc_cp_remove_fft(jrc_ctx* ctx, int fft_len, int cp_len, size_t ninput_items, const jrc_cf32* in,
                                 jrc_cf32* out)
{
    if (!ctx || !in || !out) return JRC_ERR_INVALID_ARG;
    if (fft_len <= 0 || cp_len < 0) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "cp_remover: bad fft_len/cp_len");
    const size_t nsym = ninput_items / (size_t)(fft_len + cp_len);
    if (nsym == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = sizeof(float2) * nsym * (size_t)(fft_len + cp_len);
    const size_t out_bytes = sizeof(float2) * nsym * (size_t)fft_len;
    JRC_TRY(jrc_ensure_pinned(ctx, in_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, in_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, out_bytes));
    jrc_host_copy(ctx->pinned, in, in_bytes);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    int r = jrc_cp_remove_fft_dev(ctx, fft_len, cp_len, nsym, (const jrc_cf32*)ctx->scratch[0],
                                  (jrc_cf32*)ctx->scratch[1], nullptr);
    if (r < 0) return r;
    JRC_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->scratch[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    jrc_host_copy(out, ctx->pinned, out_bytes);
    return (int)nsym;
}

// ------------------------------------------------------------------------------------------------
// TX OFDM modulator: the stock chain after mimo_precoder in the flowgraphs,
//   fft_vxx reverse + shift + window  ->  digital_ofdm_cyclic_prefixer(cp_len, rolloff 0)
// (examples/simulation/radar/mimo_ofdm_jrc_radar_sim.grc:801-897), fused: x = N*ifft(ifftshift(X .* w)), out = [x[N-cp:], x]
extern "C" int jrc_ofdm_mod_dev(jrc_ctx* ctx, int fft_len, int cp_len, const float* d_window, size_t n_symbols,
                                const jrc_cf32* d_in, jrc_cf32* d_out, void* stream)
{
    if (!ctx || !d_in || !d_out) return JRC_ERR_INVALID_ARG;
    if (cp_len < 0 || cp_len > fft_len) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "ofdm_mod: bad cp_len");
    JRC_BIND(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    JRC_TRY(launch_fft_vcc_ex(ctx, fft_len, 0, 1, d_window, n_symbols, (const float2*)d_in, (float2*)d_out, fft_len, 0,
                              (long)fft_len + cp_len, cp_len, s));
    return (int)n_symbols;
}

extern "C" int jrc_ofdm_mod(jrc_ctx* ctx, int fft_len, int cp_len, const float* window, size_t n_symbols,
                            const jrc_cf32* in, jrc_cf32* out)
{
    if (!ctx || !in || !out) return JRC_ERR_INVALID_ARG;
    if (fft_len <= 0 || cp_len < 0 || cp_len > fft_len) return jrc_fail(ctx, JRC_ERR_INVALID_ARG, "ofdm_mod: bad fft_len/cp_len");
    if (n_symbols == 0) return 0;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = sizeof(float2) * n_symbols * (size_t)fft_len, out_bytes = sizeof(float2) * n_symbols * (size_t)(fft_len + cp_len);
    const size_t wbytes = window ? sizeof(float) * (size_t)fft_len : 0;
    JRC_TRY(jrc_ensure_pinned(ctx, (in_bytes + wbytes > out_bytes) ? in_bytes + wbytes : out_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 0, in_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 1, out_bytes));
    JRC_TRY(jrc_ensure_scratch(ctx, 2, wbytes ? wbytes : 4));
    jrc_host_copy(ctx->pinned, in, in_bytes);
    if (window) memcpy((char*)ctx->pinned + in_bytes, window, wbytes);
    JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[0], ctx->pinned, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (window) JRC_HIP(ctx, hipMemcpyAsync(ctx->scratch[2], (char*)ctx->pinned + in_bytes, wbytes, hipMemcpyHostToDevice, ctx->stream));
    int r = jrc_ofdm_mod_dev(ctx, fft_len, cp_len, window ? (const float*)ctx->scratch[2] : nullptr, n_symbols,
                             (const jrc_cf32*)ctx->scratch[0], (jrc_cf32*)ctx->scratch[1], nullptr);
    if (r < 0) return r;
    JRC_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->scratch[1], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    jrc_host_copy(out, ctx->pinned, out_bytes);
    return (int)n_symbols;
}
