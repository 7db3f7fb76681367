// ctx.hip — context, error reporting, staging memory and twiddle tables for include/jrc.h
#include "jrc_internal.h"
#include <cstring>

#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <thread>

#include <cmath>

int jrc_fail(jrc_ctx* ctx, int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return status;
}

extern "C" int jrc_abi_version(void) { return JRC_ABI_VERSION; }

extern "C" int jrc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* jrc_strerror(int status)
{
    switch (status) {
        case JRC_OK: return "ok";
        case JRC_ERR_NO_DEVICE: return "no usable HIP device (the HIP path is mandatory; there is no CPU fallback)";
        case JRC_ERR_HIP: return "HIP runtime error";
        case JRC_ERR_INVALID_ARG: return "invalid argument";
        case JRC_ERR_UNSUPPORTED: return "size or shape not supported by the HIP kernels";
        case JRC_ERR_LENGTH_MISMATCH: return "[MATRIX TRANSPOSE] input_len and output_len do not match to packet length";
        case JRC_ERR_SHORT_INPUT: return "not enough input items for one frame";
        case JRC_ERR_NOMEM: return "out of memory";
        case JRC_ERR_SIG_FIELD: return "[MIMO PRECODER] something is wrong!! (OFDM symbol count mismatch)";
        case JRC_ERR_IO: return "could not open file";
        default: return "unknown jrc status";
    }
}

extern "C" int jrc_create(int device, jrc_ctx** out)
{
    if (!out) return JRC_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return JRC_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return JRC_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return JRC_ERR_NO_DEVICE;
    jrc_ctx* ctx = new jrc_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return JRC_ERR_NO_DEVICE;
    }
    if (hipDeviceGetAttribute(&ctx->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->n_cus <= 0) ctx->n_cus = 256;
    if (const char* e = getenv("JRC_NCUS")) { const int v = atoi(e); if (v >= 1 && v <= ctx->n_cus) ctx->n_cus = v; }   // experiments: size the resident grids for fewer CUs
    // XCDs = L2 domains consecutive workgroups are dealt over.  CUs per XCD is a property of the part, not of the CU count: 32 on gfx950
    // (MI355X / MI350X: 8 x 32; DPX / QPX / CPX partitions expose 4 / 2 / 1 of them), 38 on gfx942 (MI300X 8 x 38, MI300A 6 x 38).  An unknown
    // architecture gets 1: results never depend on it, and a wrong guess would split a frame's workgroups over several L2s for nothing.
    {
        hipDeviceProp_t prop;
        int per_xcd = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            if (strncmp(prop.gcnArchName, "gfx950", 6) == 0) per_xcd = 32;
            else if (strncmp(prop.gcnArchName, "gfx942", 6) == 0) per_xcd = 38;
        }
        ctx->n_xcd = (per_xcd && ctx->n_cus % per_xcd == 0) ? ctx->n_cus / per_xcd : 1;
        // LDS a workgroup may be granted (with the per-kernel dynamic-LDS opt-in): 160 KB on gfx950 whatever the runtime's attribute says — it may
        // report the 64 KB a kernel gets WITHOUT the opt-in, and the target simulator's direct route has run with up to 132 KB on this part since
        // round 5; elsewhere the attribute (read below) is believed
        if (per_xcd == 32) ctx->max_lds_per_block = 160 * 1024;
    }
    if (const char* e = getenv("JRC_XCDS")) { const int v = atoi(e); if (v >= 1 && v <= 64) ctx->n_xcd = v; }
    { int lds = 0; if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0 && (size_t)lds > ctx->max_lds_per_block) ctx->max_lds_per_block = (size_t)lds; }
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) ctx->wall_clock_khz = khz; }
    if (const char* e = getenv("JRC_CHANEST_CHUNK")) ctx->tune.chanest_chunk = atoi(e);
    ctx->tune.chanest_x1 = getenv("JRC_CHANEST_X1") != nullptr;
    ctx->tune.chanest_u2 = getenv("JRC_CHANEST_U2") != nullptr;
    ctx->tune.fd_serial = getenv("JRC_FD_SERIAL") != nullptr;
    ctx->tune.sync_naive = getenv("JRC_SYNC_NAIVE") != nullptr;
    ctx->tune.sync_streams = getenv("JRC_SYNC_STREAMS") != nullptr;
    ctx->tune.sync_tile = getenv("JRC_SYNC_TILE") != nullptr;
    ctx->tune.dec_single = getenv("JRC_DEC_SINGLE") != nullptr;
    if (const char* e = getenv("JRC_DEC_FPW")) ctx->tune.dec_frames_per_wave = atoi(e);
    ctx->tune.ra_ref_sum = getenv("JRC_RA_REF_SUM") != nullptr;
    ctx->tune.rd_generic = getenv("JRC_RD_GENERIC") != nullptr;
    ctx->tune.rd_fold = getenv("JRC_RD_FOLD") != nullptr;
    ctx->tune.rd_two_step = getenv("JRC_RD_TWO_STEP") != nullptr;
    if (const char* e = getenv("JRC_RA_PACE")) ctx->tune.ra_pace = (int)strtol(e, nullptr, 0);
    if (const char* e = getenv("JRC_DETECT_EXP")) ctx->tune.detect_exp = atoi(e);
    if (const char* e = getenv("JRC_RD_EXP")) ctx->tune.rd_exp = atoi(e);
    if (const char* e = getenv("JRC_RD_CHUNK_MB")) ctx->tune.rd_chunk_mb = atoi(e);
    if (const char* e = getenv("JRC_RA_OFFERED_TBPS")) ctx->tune.ra_offered_tbps = atof(e);
    if (const char* e = getenv("JRC_DEMOD_SPR")) ctx->tune.demod_spr = atoi(e);
    ctx->tune.eq_sig_full = getenv("JRC_EQ_SIG_FULL") != nullptr;
    if (const char* e = getenv("JRC_EQ_WPE")) ctx->tune.eq_wpe = atoi(e);
    if (const char* e = getenv("JRC_EQ_THREADS")) ctx->tune.eq_threads = atoi(e);
    // the timing experiments leave work out (wrong results): they exist only in a library built for them (-DJRC_TIMING_EXPERIMENTS on this file,
    // tools/ra_variants.py), and say so where nobody can miss it; any other build drops those bits
    if ((ctx->tune.detect_exp & (1 | 2 | 32)) || (ctx->tune.rd_exp & (1 | 2))) {
#ifdef JRC_TIMING_EXPERIMENTS
        fprintf(stderr, "libjrc_hip: JRC_DETECT_EXP=%d / JRC_RD_EXP=%d select TIMING-ONLY kernel experiments on this context: RESULTS ARE WRONG\n",
                ctx->tune.detect_exp, ctx->tune.rd_exp);
#else
        fprintf(stderr, "libjrc_hip: the work-skipping bits of JRC_DETECT_EXP=%d / JRC_RD_EXP=%d are ignored (not a -DJRC_TIMING_EXPERIMENTS build)\n",
                ctx->tune.detect_exp, ctx->tune.rd_exp);
        ctx->tune.detect_exp &= ~(1 | 2 | 32);
        ctx->tune.rd_exp &= ~(1 | 2);
#endif
    }
    *out = ctx;
    return JRC_OK;
}

extern "C" void jrc_destroy(jrc_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->twiddles) (void)hipFree(kv.second);
    for (auto& kv : ctx->bluestein) { (void)hipFree(kv.second.chirp); (void)hipFree(kv.second.bhat); }
    for (int i = 0; i < 4; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char* jrc_last_error(const jrc_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int jrc_device_name(const jrc_ctx* ctx, char* buf, size_t len)
{
    if (!ctx || !buf || len == 0) return JRC_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return JRC_ERR_HIP;
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return JRC_OK;
}

extern "C" int jrc_sync(jrc_ctx* ctx)
{
    if (!ctx) return JRC_ERR_INVALID_ARG;
    JRC_BIND(ctx);
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JRC_OK;
}

extern "C" void* jrc_stream(jrc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int jrc_dev_malloc(jrc_ctx* ctx, size_t bytes, void** dptr)
{
    if (!ctx || !dptr) return JRC_ERR_INVALID_ARG;
    JRC_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return jrc_fail(ctx, JRC_ERR_NOMEM, "hipMalloc(%zu) out of memory", bytes);
    JRC_HIP(ctx, e);
    return JRC_OK;
}

extern "C" int jrc_dev_free(jrc_ctx* ctx, void* dptr)
{
    if (!ctx) return JRC_ERR_INVALID_ARG;
    JRC_BIND(ctx);
    if (dptr) JRC_HIP(ctx, hipFree(dptr));
    return JRC_OK;
}

extern "C" int jrc_dev_memset(jrc_ctx* ctx, void* dptr, int value, size_t bytes)
{
    if (!ctx) return JRC_ERR_INVALID_ARG;
    JRC_BIND(ctx);
    JRC_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JRC_OK;
}

extern "C" int jrc_memcpy_h2d(jrc_ctx* ctx, void* dptr, const void* hptr, size_t bytes)
{
    if (!ctx) return JRC_ERR_INVALID_ARG;
    JRC_BIND(ctx);
    JRC_HIP(ctx, hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JRC_OK;
}

extern "C" int jrc_memcpy_d2h(jrc_ctx* ctx, void* hptr, const void* dptr, size_t bytes)
{
    if (!ctx) return JRC_ERR_INVALID_ARG;
    JRC_BIND(ctx);
    JRC_HIP(ctx, hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JRC_OK;
}

int jrc_ensure_dyn_lds(jrc_ctx* ctx, const void* kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return JRC_OK;
    auto it = ctx->dyn_lds.find(kernel);
    if (it != ctx->dyn_lds.end() && it->second >= bytes) return JRC_OK;
    JRC_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->dyn_lds[kernel] = bytes;
    return JRC_OK;
}

namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)();
struct roctx_api {
    roctx_push_fn push = nullptr;
    roctx_pop_fn pop = nullptr;
    roctx_api()
    {
        if (!getenv("JRC_ROCTX")) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
        pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
const roctx_api& roctx() { static const roctx_api api; return api; }
}  // namespace

void jrc_trace_push(const char* name) { if (roctx().push) roctx().push(name); }
void jrc_trace_pop() { if (roctx().pop) roctx().pop(); }
// JRC_LOG_CALLS=1: every batched entry point writes "[jrc <pid> rank <RANK>] <entry point>" to stderr (unbuffered) before it launches, so a
// process that dies under a device fault has named the call it was in (the 8-rank same-device run that hung on the driver box)
static bool jrc_log_calls()
{
    static const bool on = []() { const char* v = getenv("JRC_LOG_CALLS"); return v && *v && *v != '0'; }();
    return on;
}
jrc_trace_range::jrc_trace_range(const char* name) : on(roctx().push != nullptr)
{
    if (jrc_log_calls()) {
        const char* rk = getenv("RANK");
        fprintf(stderr, "[jrc %d rank %s] %s\n", (int)getpid(), rk ? rk : "-", name);
    }
    if (on) jrc_trace_push(name);
}

void jrc_host_copy(void* dst, const void* src, size_t bytes)
{
    const int n_thr = bytes >= ((size_t)16 << 20) ? 4 : (bytes >= ((size_t)4 << 20) ? 2 : 1);
    if (n_thr == 1) { if (bytes) memcpy(dst, src, bytes); return; }
    std::thread th[3];
    const size_t part = ((bytes / n_thr) + 4095) & ~(size_t)4095;
    for (int i = 1; i < n_thr; i++) {
        const size_t off = part * i, len = off < bytes ? (off + part < bytes ? part : bytes - off) : 0;
        th[i - 1] = std::thread([=]() { if (len) memcpy((char*)dst + off, (const char*)src + off, len); });
    }
    memcpy(dst, src, part < bytes ? part : bytes);
    for (int i = 1; i < n_thr; i++) th[i - 1].join();
}

int jrc_ensure_pinned(jrc_ctx* ctx, size_t bytes)
{
    if (ctx->pinned_bytes >= bytes) return JRC_OK;
    if (ctx->pinned) { (void)hipHostFree(ctx->pinned); ctx->pinned = nullptr; ctx->pinned_bytes = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    JRC_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
    return JRC_OK;
}

int jrc_ensure_scratch(jrc_ctx* ctx, int slot, size_t bytes)
{
    if (ctx->scratch_bytes[slot] >= bytes) return JRC_OK;
    if (ctx->scratch[slot]) {
        JRC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->scratch[slot]);
        ctx->scratch[slot] = nullptr;
        ctx->scratch_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&ctx->scratch[slot], want);
    if (e == hipErrorOutOfMemory) return jrc_fail(ctx, JRC_ERR_NOMEM, "hipMalloc(%zu) out of memory", want);
    JRC_HIP(ctx, e);
    ctx->scratch_bytes[slot] = want;
    return JRC_OK;
}

int jrc_get_twiddles(jrc_ctx* ctx, int n, int sign, const float2** out)
{
    long key = (long)sign * n;
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) { *out = it->second; return JRC_OK; }
    std::vector<float2> h((size_t)n);
    for (int k = 0; k < n; k++) {
        // exact octant symmetry is not needed; double cos/sin rounded once to float
        double ang = (double)sign * 2.0 * M_PI * (double)k / (double)n;
        h[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    float2* d = nullptr;
    JRC_HIP(ctx, hipMalloc((void**)&d, sizeof(float2) * (size_t)n));
    JRC_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    ctx->twiddles[key] = d;
    *out = d;
    return JRC_OK;
}

int jrc_get_bluestein(jrc_ctx* ctx, int n, int sign, jrc_ctx::bluestein_tab* out)
{
    const long key = (long)sign * n;
    auto it = ctx->bluestein.find(key);
    if (it != ctx->bluestein.end()) { *out = it->second; return JRC_OK; }
    int M = 1;
    while (M < 2 * n - 1) M <<= 1;
    std::vector<double> cr((size_t)n), ci((size_t)n), br((size_t)M, 0.0), bi((size_t)M, 0.0);
    for (long k = 0; k < n; k++) {
        const long q = (k * k) % (2L * n);                       // k^2 mod 2n keeps the angle exact
        const double a = M_PI * (double)q / (double)n;
        cr[k] = cos(a); ci[k] = (double)sign * sin(a);
        br[k] = cr[k]; bi[k] = -ci[k];                           // conj(c), wrapped
        if (k) { br[M - k] = cr[k]; bi[M - k] = -ci[k]; }
    }
    // forward FFT of length M in double (iterative radix 2)
    for (int i = 1, j = 0; i < M; i++) {
        int bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(br[i], br[j]); std::swap(bi[i], bi[j]); }
    }
    for (int len = 2; len <= M; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i0 = 0; i0 < M; i0 += len)
            for (int k = 0; k < len / 2; k++) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int a = i0 + k, b = a + len / 2;
                const double xr = br[b] * wr - bi[b] * wi, xi = br[b] * wi + bi[b] * wr;
                br[b] = br[a] - xr; bi[b] = bi[a] - xi;
                br[a] += xr; bi[a] += xi;
            }
    }
    std::vector<float2> hc((size_t)n), hb((size_t)M);
    for (int k = 0; k < n; k++) hc[k] = make_float2((float)cr[k], (float)ci[k]);
    for (int k = 0; k < M; k++) hb[k] = make_float2((float)(br[k] / M), (float)(bi[k] / M));
    jrc_ctx::bluestein_tab t{nullptr, nullptr, M};
    JRC_HIP(ctx, hipMalloc((void**)&t.chirp, sizeof(float2) * (size_t)n));
    JRC_HIP(ctx, hipMalloc((void**)&t.bhat, sizeof(float2) * (size_t)M));
    JRC_HIP(ctx, hipMemcpy(t.chirp, hc.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    JRC_HIP(ctx, hipMemcpy(t.bhat, hb.data(), sizeof(float2) * (size_t)M, hipMemcpyHostToDevice));
    ctx->bluestein[key] = t;
    *out = t;
    return JRC_OK;
}
