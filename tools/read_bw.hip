// read_bw.hip — read-bandwidth ceiling on MI355X (tools only): grid-stride 16-byte loads, temporal and non-temporal, several grid sizes
//   hipcc --offload-arch=gfx950 -O3 tools/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT_LOAD, int U>
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ in, size_t n4, float* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * st < n4; i += U * st) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT_LOAD ? __builtin_nontemporal_load(in + i + u * st) : in[i + u * st];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
int main()
{
    const size_t bytes = (size_t)2 << 30;
    v4f* d; float* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "temporal U=4 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_read<0, 4>), dim3(g), dim3(256), 0, 0, d, bytes / 16, o); });
        snprintf(nm, 64, "non-temporal U=4 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_read<1, 4>), dim3(g), dim3(256), 0, 0, d, bytes / 16, o); });
        snprintf(nm, 64, "non-temporal U=8 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_read<1, 8>), dim3(g), dim3(256), 0, 0, d, bytes / 16, o); });
    }
    return 0;
}
