// k2_proto.hip — experiment (tools only): how fast can the angle-FFT + store + arg-max half of the fused kernel run when the range
// profiles of a class are simply loaded (8 KiB per class) instead of being computed in the same workgroup?
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include tools/k2_proto.hip -o /tmp/k2 && /tmp/k2
#include "../gr-mimo-ofdm-jrc_amd/csrc/fft_device.h"
#include <cstdio>
#include <vector>
#define RA_L 64

template <int P, int NT, int WPS, int VAR>
__global__ __launch_bounds__(NT, WPS) void k2_kernel(const float2* __restrict__ Rc /* [F][C][P][64] */, float2* __restrict__ map,
                                                     PeakPartial* __restrict__ partials, const float2* __restrict__ twA, int NR, int Ia, int F, int WPF)
{
    __shared__ float2 s_g[2][P * RA_L];
    const int NA = P * Ia, C = NR / RA_L;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, f = (j / WPF) * 8 + xcd, slice = j % WPF;
    if (f >= F) return;
    const int tid = threadIdx.x;
    const int r = tid % Ia;
    float2 ta[P];
#pragma unroll
    for (int p = 1; p < P; p++) ta[p] = twA[(p * r) & (NA - 1)];
    PeakTracker trk; trk.init();
    const int items = RA_L * Ia, ahalf = NA >> 1, amask = NA - 1;
    float2* mapf = map + (size_t)f * NR * NA;
    const float2* Rf = Rc + (size_t)f * C * P * RA_L;
    // prefetch first class
    constexpr int PER = (P * RA_L) / NT;          // float2 per thread per class
    float2 pre[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) pre[q] = Rf[(size_t)slice * P * RA_L + tid + NT * q];
    int buf = 0;
#pragma unroll 1
    for (int c = slice; c < C; c += WPF) {
#pragma unroll
        for (int q = 0; q < PER; q++) s_g[buf][tid + NT * q] = pre[q];
        const int cn = c + WPF;
        if (cn < C) {
#pragma unroll
            for (int q = 0; q < PER; q++) pre[q] = Rf[(size_t)cn * P * RA_L + tid + NT * q];
        }
        __syncthreads();
        const float2* g = s_g[buf];
#pragma unroll 1
        for (int w0 = 0; w0 < items; w0 += NT) {
            const int w = w0 + tid;
            if (w >= items) break;
            const int ql = w / Ia;
            const int k = C * ql + c;
            float2 y[P];
            y[0] = g[ql];
#pragma unroll
            for (int p = 1; p < P; p++) y[p] = (VAR >= 2) ? g[p * RA_L + ql] : cmul(g[p * RA_L + ql], ta[p]);
            if (VAR < 3) fft_fwd_small<P>(y);
            float2* row = mapf + (size_t)k * NA;
            float m = -1.0f;
#pragma unroll
            for (int u = 0; u < P; u++) {
                const int a = (Ia * u + r + ahalf) & amask;
                row[a] = y[u];
                if (VAR == 0) m = fmaxf(m, fast_power(y[u]));
            }
            const float thr = (VAR == 0) ? trk.raise(m) : 1e30f;
            if (VAR == 0 && m >= thr) {
                const unsigned flat0 = (unsigned)k * (unsigned)NA;
#pragma unroll
                for (int u = 0; u < P; u++)
                    if (fast_power(y[u]) >= thr) trk.exact(y[u], flat0 + ((Ia * u + r + ahalf) & amask));
            }
        }
        buf ^= 1;
    }
    __shared__ PeakPartial red[16];
    __syncthreads();
    block_reduce_peak(trk, red);
    if (tid == 0) { partials[(size_t)f * WPF + slice].best = trk.best; partials[(size_t)f * WPF + slice].idx = trk.idx; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int NT, int WPS, int VAR>
static int run(const char* name, int F, int NR, int Ia, int WPF, const float2* Rc, float2* map, PeakPartial* part, const float2* twA)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int groups = (F + 7) / 8;
    dim3 grid(groups * 8 * WPF);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k2_kernel<16, NT, WPS, VAR>), grid, dim3(NT), 0, 0, Rc, map, part, twA, NR, Ia, F, WPF);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k2_kernel<16, NT, WPS, VAR>), grid, dim3(NT), 0, 0, Rc, map, part, twA, NR, Ia, F, WPF);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double bytes = (double)F * NR * 16 * Ia * 8;
    printf("%-34s F=%d NR=%d WPF=%d grid=%d: %.3f ms  %.0f GB/s (map bytes)\n", name, F, NR, WPF, grid.x, ms, bytes / ms / 1e6);
    return 0;
}

int main()
{
    const int P = 16, Ia = 16, NA = P * Ia;
    for (int cfg = 0; cfg < 2; cfg++) {
        const int NR = cfg ? 8192 : 2048, F = cfg ? 128 : 512, C = NR / 64;
        float2 *Rc, *map, *twA; PeakPartial* part;
        const size_t nrc = (size_t)F * C * P * 64;
        CK(hipMalloc(&Rc, nrc * 8)); CK(hipMalloc(&map, (size_t)F * NR * NA * 8)); CK(hipMalloc(&twA, NA * 8)); CK(hipMalloc(&part, (size_t)F * 64 * 8));
        std::vector<float2> h(nrc);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v.x = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; s = s * 1664525u + 1013904223u; v.y = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        CK(hipMemcpy(Rc, h.data(), nrc * 8, hipMemcpyHostToDevice));
        std::vector<float2> tw(NA);
        for (int i = 0; i < NA; i++) { tw[i].x = cosf(-2 * 3.14159265f * i / NA); tw[i].y = sinf(-2 * 3.14159265f * i / NA); }
        CK(hipMemcpy(twA, tw.data(), NA * 8, hipMemcpyHostToDevice));
        printf("---- %s\n", cfg ? "config D shape" : "config B shape");
        const int wpf = cfg ? 4 : 16;
        run<256, 2, 0>("full", F, NR, Ia, wpf, Rc, map, part, twA);
        run<256, 2, 1>("no arg-max", F, NR, Ia, wpf, Rc, map, part, twA);
        run<256, 2, 2>("no arg-max, no twiddle", F, NR, Ia, wpf, Rc, map, part, twA);
        run<256, 2, 3>("no arg-max, no twiddle, no FFT", F, NR, Ia, wpf, Rc, map, part, twA);
        hipFree(Rc); hipFree(map); hipFree(twA); hipFree(part);
    }
    return 0;
}
