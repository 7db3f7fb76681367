// pmc_calib.hip — what FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count, per access width (tools only).
// One kernel per pattern, each moving exactly 1 GiB (>> the 256 MiB Infinity Cache) once:
//   read  4 / 8 / 16 B per lane, temporal and non-temporal, coalesced (a wave instruction covers 256 / 512 / 1024 contiguous bytes)
//   write 4 / 8 / 16 B per lane, temporal and non-temporal, coalesced
//   read 8 B per lane with a 2 KiB row stride between a lane's consecutive accesses (the equalizer's symbol rows)
// Run under rocprofv3 --kernel-trace --pmc FETCH_SIZE  and again with --pmc WRITE_SIZE (tools/pmc_calib.sh); the factor a counter has
// to be multiplied with is  bytes / (counter x 1024)  [the counters are in KiB].
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename V, int NT_, int U>
__global__ __launch_bounds__(256) void calib_read(const V* __restrict__ in, size_t n, float* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i + (U - 1) * st < n; i += U * st) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT_ ? __builtin_nontemporal_load(in + i + u * st) : in[i + u * st];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (sizeof(V) == 4) acc += v[u];
            else if constexpr (sizeof(V) == 8) acc += v[u].x + v[u].y;
            else acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 12345.678f) out[0] = 1.f;
}

template <typename V, int NT_, int U>
__global__ __launch_bounds__(256) void calib_write(V* __restrict__ outp, size_t n, float seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    V v;
    if constexpr (sizeof(V) == 4) v = seed;
    else if constexpr (sizeof(V) == 8) v = V{seed, seed};
    else v = V{seed, seed, seed, seed};
    for (; i + (U - 1) * st < n; i += U * st) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT_) __builtin_nontemporal_store(v, outp + i + u * st); else outp[i + u * st] = v;
        }
    }
}

// a workgroup of 64 lanes walks 64 consecutive rows of 256 x 8 B (2 KiB), four 8-byte elements per lane and row: the equalizer's reads
__global__ __launch_bounds__(64) void calib_read_rows8(const v2f* __restrict__ in, size_t n_rows, float* out)
{
    const size_t rows_per_wg = 64;
    const v2f* p = in + (size_t)blockIdx.x * rows_per_wg * 256;
    float acc = 0.f;
    for (size_t r = 0; r < rows_per_wg; r++) {
        v2f v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = p[r * 256 + threadIdx.x + e * 64];
#pragma unroll
        for (int e = 0; e < 4; e++) acc += v[e].x + v[e].y;
    }
    if (acc == 12345.678f) out[0] = 1.f;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void* d; float* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    hipDeviceSynchronize();
    const int g = 4096;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((calib_read<float, 0, 8>), dim3(g), dim3(256), 0, 0, (const float*)d, bytes / 4, o);
        hipLaunchKernelGGL((calib_read<v2f, 0, 4>), dim3(g), dim3(256), 0, 0, (const v2f*)d, bytes / 8, o);
        hipLaunchKernelGGL((calib_read<v4f, 0, 4>), dim3(g), dim3(256), 0, 0, (const v4f*)d, bytes / 16, o);
        hipLaunchKernelGGL((calib_read<float, 1, 8>), dim3(g), dim3(256), 0, 0, (const float*)d, bytes / 4, o);
        hipLaunchKernelGGL((calib_read<v2f, 1, 4>), dim3(g), dim3(256), 0, 0, (const v2f*)d, bytes / 8, o);
        hipLaunchKernelGGL((calib_read<v4f, 1, 4>), dim3(g), dim3(256), 0, 0, (const v4f*)d, bytes / 16, o);
        hipLaunchKernelGGL(calib_read_rows8, dim3((unsigned)(bytes / (64 * 2048))), dim3(64), 0, 0, (const v2f*)d, bytes / 2048, o);
        hipLaunchKernelGGL((calib_write<float, 0, 8>), dim3(g), dim3(256), 0, 0, (float*)d, bytes / 4, 1.f);
        hipLaunchKernelGGL((calib_write<v2f, 0, 4>), dim3(g), dim3(256), 0, 0, (v2f*)d, bytes / 8, 1.f);
        hipLaunchKernelGGL((calib_write<v4f, 0, 4>), dim3(g), dim3(256), 0, 0, (v4f*)d, bytes / 16, 1.f);
        hipLaunchKernelGGL((calib_write<float, 1, 8>), dim3(g), dim3(256), 0, 0, (float*)d, bytes / 4, 1.f);
        hipLaunchKernelGGL((calib_write<v2f, 1, 4>), dim3(g), dim3(256), 0, 0, (v2f*)d, bytes / 8, 1.f);
        hipLaunchKernelGGL((calib_write<v4f, 1, 4>), dim3(g), dim3(256), 0, 0, (v4f*)d, bytes / 16, 1.f);
        hipDeviceSynchronize();
    }
    printf("pmc_calib: every kernel moved %zu bytes\n", bytes);
    return 0;
}
