// fft_device.h — device-side FFT building blocks shared by fft.hip (fft_vcc), chain.hip (fused range-angle kernel)
// and tsim.hip (target simulator): an in-register power-of-two FFT up to 16 points and one Stockham radix-2/4 pass.
#pragma once

#include "jrc_internal.h"

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

// ---- the same transform with its arithmetic pinned instruction by instruction -------------------------------------------------------
// Under hipcc's default -ffp-contract=fast the compiler decides per call site which multiply of a complex product or of a twiddled
// butterfly it fuses into an FMA, so two kernels that inline the very same source may round differently in the last bit.  The range-angle
// kernel is instantiated in a map-writing and a detect-only variant whose results must be bit-identical (chain.hip): its complex
// products and butterflies are spelled out here with explicit fmaf under contraction off — the same instruction count the compiler
// reaches on its own (2 mul + 2 fma per complex product; one fma per twiddled butterfly output).
__device__ __forceinline__ float2 cmul_pin(float2 a, float2 b)
{
#pragma clang fp contract(off)
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

// one butterfly of the last stage: lo = a + w_P^K v, hi = a - w_P^K v
template <int P, int K>
__device__ __forceinline__ void bfly_pin_one(const float2 a, const float2 v, float2& lo, float2& hi)
{
#pragma clang fp contract(off)
    constexpr int idx = K * (16 / P);   // w_P^K in sixteenths of a turn, 0..7
    constexpr float R2 = 0.70710678118654752440f;
    constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
    if constexpr (idx == 0) {
        lo = make_float2(a.x + v.x, a.y + v.y); hi = make_float2(a.x - v.x, a.y - v.y);
    } else if constexpr (idx == 4) {    // w v = (v.y, -v.x)
        lo = make_float2(a.x + v.y, a.y - v.x); hi = make_float2(a.x - v.y, a.y + v.x);
    } else if constexpr (idx == 2) {    // w v = ((v.x + v.y) R2, (v.y - v.x) R2)
        const float s = v.x + v.y, d = v.y - v.x;
        lo = make_float2(fmaf(s, R2, a.x), fmaf(d, R2, a.y)); hi = make_float2(fmaf(-s, R2, a.x), fmaf(-d, R2, a.y));
    } else if constexpr (idx == 6) {    // w v = ((v.y - v.x) R2, -(v.x + v.y) R2)
        const float s = v.x + v.y, d = v.y - v.x;
        lo = make_float2(fmaf(d, R2, a.x), fmaf(-s, R2, a.y)); hi = make_float2(fmaf(-d, R2, a.x), fmaf(s, R2, a.y));
    } else {
        float tx, ty;
        if constexpr (idx == 1) { tx = fmaf(v.x, C1, v.y * S1); ty = fmaf(v.y, C1, -(v.x * S1)); }
        else if constexpr (idx == 3) { tx = fmaf(v.x, S1, v.y * C1); ty = fmaf(v.y, S1, -(v.x * C1)); }
        else if constexpr (idx == 5) { tx = fmaf(v.y, C1, -(v.x * S1)); ty = -fmaf(v.x, C1, v.y * S1); }
        else { tx = fmaf(v.y, S1, -(v.x * C1)); ty = -fmaf(v.x, S1, v.y * C1); }   // idx == 7
        lo = make_float2(a.x + tx, a.y + ty); hi = make_float2(a.x - tx, a.y - ty);
    }
}

template <int P, int K>
struct BflyPin {
    static __device__ __forceinline__ void run(float2* x, const float2* e, const float2* o)
    {
        bfly_pin_one<P, K>(e[K], o[K], x[K], x[K + P / 2]);
        if constexpr (K + 1 < P / 2) BflyPin<P, K + 1>::run(x, e, o);
    }
};

template <int P>
__device__ __forceinline__ void fft_fwd_small_pin(float2 (&x)[P])
{
    if constexpr (P > 1) {
        float2 e[P / 2], o[P / 2];
#pragma unroll
        for (int k = 0; k < P / 2; k++) { e[k] = x[2 * k]; o[k] = x[2 * k + 1]; }
        fft_fwd_small_pin<P / 2>(e);
        fft_fwd_small_pin<P / 2>(o);
        BflyPin<P, 0>::run(x, e, o);
    }
}

// ---- lane pairs without the LDS crossbar -----------------------------------------------------------------------------------------
// lo / hi = the values held by the lower / upper lane of the pair of lanes that differ in bit B of the lane index — what a radix-2
// butterfly across the wavefront (or an in-place trellis step, comm.hip) needs from `x`.  Bits 5 and 4 are gfx950's permlane swaps
// (one instruction yields both), bits 3..0 one DPP move per side; a __shfl_xor is a ds_bpermute round trip (~100+ cycles) instead.
template <int B>
__device__ __forceinline__ void lane_pair(int x, int& lo, int& hi)
{
    if constexpr (B == 5) { const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); lo = r[0]; hi = r[1]; }
    else if constexpr (B == 4) { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); lo = r[0]; hi = r[1]; }
    else if constexpr (B == 3) { lo = __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xc, false); hi = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xf, 0x3, false); }   // row_shr:8 into lanes 8-15, row_shl:8 into lanes 0-7
    else if constexpr (B == 2) { lo = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xa, false); hi = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false); }   // row_shr:4 into banks 1, 3; row_shl:4 into banks 0, 2
    else if constexpr (B == 1) { lo = __builtin_amdgcn_mov_dpp(x, 0x44, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(x, 0xee, 0xf, 0xf, false); }                  // quad_perm [0,1,0,1], [2,3,2,3]
    else { lo = __builtin_amdgcn_mov_dpp(x, 0xa0, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(x, 0xf5, 0xf, 0xf, false); }                                         // quad_perm [0,0,2,2], [1,1,3,3]
}
template <int B>
__device__ __forceinline__ void lane_pair(float2 v, float2& lo, float2& hi)
{
    int a, b, c, d;
    lane_pair<B>(__float_as_int(v.x), a, b);
    lane_pair<B>(__float_as_int(v.y), c, d);
    lo = make_float2(__int_as_float(a), __int_as_float(c));
    hi = make_float2(__int_as_float(b), __int_as_float(d));
}

// one radix-2 DIF stage of a 64-point transform held one point per lane: the lanes of a pair differ in bit B; `t` is this lane's twiddle
template <int B>
__device__ __forceinline__ float2 wave_dif_stage(float2 v, float2 t, int lane)
{
#pragma clang fp contract(off)
    float2 lo, hi;
    lane_pair<B>(v, lo, hi);
    const float2 s = make_float2(lo.x + hi.x, lo.y + hi.y), d = cmul_pin(make_float2(lo.x - hi.x, lo.y - hi.y), t);
    return (lane & (1 << B)) ? d : s;
}

// ---- one Stockham autosort pass (radix R in {2,4}) over n points; sources / destinations may be global or LDS ----
template <int R>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ src_g, long src_wrap /* n if ifftshift else 0 */,
                                              const float* __restrict__ window, const float2* src_l, float2* dst_l,
                                              float2* __restrict__ dst_g, int dst_rot /* n/2 if fftshift else 0 */,
                                              const float2* __restrict__ tw, int n, int Ns, int sign, int lt, int tp)
{
    const int nb = n / R;
    for (int j = lt; j < nb; j += tp) {
        const int k = j & (Ns - 1);
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int idx = j + r * nb;
            if (src_g) {
                const int si = src_wrap ? ((idx + (n >> 1)) & (n - 1)) : idx;          // ifftshift on the way in
                v[r] = src_g[si];
                if (window) { const float w = window[si]; v[r].x *= w; v[r].y *= w; }
            } else {
                v[r] = src_l[idx];
            }
            if (r && k) v[r] = cmul(v[r], tw[(k * r * (n / (Ns * R))) & (n - 1)]);
        }
        if (R == 2) {
            const float2 a = v[0], b = v[1];
            v[0] = cadd(a, b); v[1] = csub(a, b);
        } else {
            const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
            // forward: X1 = b - j d, X3 = b + j d ; inverse: X1 = b + j d, X3 = b - j d
            const float2 jd = sign < 0 ? make_float2(d.y, -d.x) : make_float2(-d.y, d.x);
            v[0] = cadd(a, c); v[2] = csub(a, c); v[1] = cadd(b, jd); v[3] = csub(b, jd);
        }
        const int j0 = ((j - k) * R) + k;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int o = j0 + r * Ns;
            if (dst_g) dst_g[dst_rot ? ((o + dst_rot) & (n - 1)) : o] = v[r];          // fftshift on the way out
            else dst_l[o] = v[r];
        }
    }
}

