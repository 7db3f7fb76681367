// valu_rate.hip — issue rate of scalar vs packed f32 VALU instructions on gfx950 (per SIMD, one to eight waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate
// Each lane runs ITER trips over 16 independent accumulators; a trip is 16 v_fma_f32 (scalar) or 8 v_pk_fma_f32 / 8 v_pk_mul_f32 +
// 8 v_pk_add_f32 (packed: the same 16 results).  Reports lane-results per clock per SIMD from wall_clock vs a known instruction count.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096

__global__ void k_scalar(float* out, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pk_fma(float* out, float a, float b)
{
    v2f x[8];
    const v2f va = {a, a}, vb = {b, b};
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = v2f{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pk_add(float* out, float a, float b)
{
    v2f x[8];
    const v2f va = {a, b};
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = v2f{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(va));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add(float* out, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int waves_per_simd, float* d, double results_per_instr, double instrs_per_trip)
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int threads = 256, blocks = cus * waves_per_simd;     // 4 waves per block = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int mhz = 0;
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0);
    const double wave_instr = (double)blocks * 4 * ITER * instrs_per_trip;
    const double per_simd_per_s = wave_instr / (cus * 4.0) / (ms * 1e-3);
    printf("%-12s waves/SIMD %d: %.3f ms  %.2f G wave-instr/s/SIMD  (%.2f cycles per wave-instr at %.0f MHz)  %.1f T lane-results/s chip\n", name, waves_per_simd,
           ms, per_simd_per_s / 1e9, (mhz * 1e3) / per_simd_per_s, mhz / 1e3, wave_instr * 64 * results_per_instr / (ms * 1e-3) / 1e12);
}

int main()
{
    float* d;
    hipMalloc(&d, sizeof(float) * 256 * 256 * 16);
    for (int w : {1, 2, 4, 8}) {
        run("v_fma_f32", k_scalar, w, d, 1, 16);
        run("v_add_f32", k_add, w, d, 1, 16);
        run("v_pk_fma_f32", k_pk_fma, w, d, 2, 8);
        run("v_pk_add_f32", k_pk_add, w, d, 2, 8);
    }
    return 0;
}
