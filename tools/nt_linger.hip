// nt_linger.hip — do non-temporal stores leave dirty lines in the Infinity Cache?  (tools only)
// A streaming read of 537 MB (stand-in for A1) is timed behind a kernel that writes X MB cached, non-temporal 8 B / lane, non-temporal 16 B / lane.
//   hipcc --offload-arch=gfx950 -O3 -o tools/nt_linger tools/nt_linger.hip && tools/nt_linger
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k_read(const float4* __restrict__ in, float* out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float s = 0;
    for (; i < n4; i += st) { float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> __global__ void k_write(float* out, size_t n) {   // n floats
    size_t st = (size_t)gridDim.x * blockDim.x, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (KIND == 0) { for (size_t j = i; j < n / 4; j += st) reinterpret_cast<float4*>(out)[j] = make_float4(j, 1, 2, 3); }
    if (KIND == 1) { for (size_t j = i; j < n / 2; j += st) { v2f t = {(float)j, 1.f}; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(out) + j); } }
    if (KIND == 2) { for (size_t j = i; j < n / 4; j += st) { v4f t = {(float)j, 1.f, 2.f, 3.f}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(out) + j); } }
}
int main() {
    const size_t rd = 537ull << 20, wr = 2048ull << 20;
    float4* in; float* o; float* junk;
    hipMalloc(&in, rd); hipMalloc(&o, 64); hipMalloc(&junk, wr);
    hipMemset(in, 0, rd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"cached", "nt 8 B", "nt 16 B"};
    for (int kind = 0; kind < 3; kind++)
        for (size_t mb : {0ull, 128ull, 1024ull}) {
            float best = 1e9, sum = 0;
            for (int it = 0; it < 12; it++) {
                if (mb) {
                    if (kind == 0) hipLaunchKernelGGL(k_write<0>, dim3(4096), dim3(256), 0, 0, junk, (mb << 20) / 4);
                    if (kind == 1) hipLaunchKernelGGL(k_write<1>, dim3(4096), dim3(256), 0, 0, junk, (mb << 20) / 4);
                    if (kind == 2) hipLaunchKernelGGL(k_write<2>, dim3(4096), dim3(256), 0, 0, junk, (mb << 20) / 4);
                }
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, in, o, rd / 16);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            printf("%-8s %5zu MB written before: read of 537 MB takes %.4f ms (best %.4f)\n", names[kind], (size_t)mb, sum / 10, best);
        }
    return 0;
}
