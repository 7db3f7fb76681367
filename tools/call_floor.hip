// What one synchronous small call costs on this box, by how it is issued (tools only).  A stand-in for a per-block work() call of the size of
// mimo_ofdm_radar at the reference flowgraph's point: 12 KiB in, a trivial kernel, 32 KiB out.
//   a) memcpy H2D + kernel + memcpy D2H + hipStreamSynchronize          (what the per-block entry points do)
//   b) the same three recorded once as a hipGraph, replayed + synchronise
//   c) the kernel reads the pinned host input and writes the pinned host output itself (no copies) + synchronise
//   d) as a) with hipDeviceScheduleSpin
// build: hipcc --offload-arch=gfx950 -O2 tools/call_floor.hip -o tools/call_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void work(const float2* in, float2* out, int n_in, int n_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) { const float2 v = in[i % n_in]; out[i] = make_float2(v.x * 2.f, v.y + 1.f); }
}
static double p50(std::vector<double>& t) { std::sort(t.begin(), t.end()); return t[t.size() / 2] * 1e6; }
int main(int argc, char** argv)
{
    const bool spin = argc > 1 && atoi(argv[1]);
    if (spin) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
    const int n_in = 12288 / 8, n_out = 32768 / 8, iters = 2000;
    float2 *h_in, *h_out, *d_in, *d_out;
    CK(hipHostMalloc((void**)&h_in, n_in * 8, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h_out, n_out * 8, hipHostMallocDefault));
    CK(hipMalloc((void**)&d_in, n_in * 8)); CK(hipMalloc((void**)&d_out, n_out * 8));
    for (int i = 0; i < n_in; i++) h_in[i] = make_float2(1.f, 2.f);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<double> ta, tb, tc;
    for (int it = 0; it < iters + 200; it++) {
        const double t0 = now();
        CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(work, dim3((n_out + 255) / 256), dim3(256), 0, s, d_in, d_out, n_in, n_out);
        CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (it >= 200) ta.push_back(now() - t0);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(work, dim3((n_out + 255) / 256), dim3(256), 0, s, d_in, d_out, n_in, n_out);
    CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < iters + 200; it++) {
        const double t0 = now();
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        if (it >= 200) tb.push_back(now() - t0);
    }
    float2 *m_in, *m_out;
    CK(hipHostGetDevicePointer((void**)&m_in, h_in, 0)); CK(hipHostGetDevicePointer((void**)&m_out, h_out, 0));
    for (int it = 0; it < iters + 200; it++) {
        const double t0 = now();
        hipLaunchKernelGGL(work, dim3((n_out + 255) / 256), dim3(256), 0, s, m_in, m_out, n_in, n_out);
        CK(hipStreamSynchronize(s));
        if (it >= 200) tc.push_back(now() - t0);
    }
    printf("%s: copies + kernel + sync p50 %.1f us | hipGraph replay + sync %.1f us | kernel on host-mapped buffers + sync %.1f us (out[5] = %g)\n",
           spin ? "hipDeviceScheduleSpin" : "default scheduling", p50(ta), p50(tb), p50(tc), h_out[5].x);
    return 0;
}
