// store_persist.hip — pure-store emulation of the fused range-angle kernel's map stream (tools only; not part of the product):
// persistent workgroups, one per (frame, slice), looping over residue classes c and writing rows k = C*ql + c as the kernel does —
// a wave-instruction covers 128-byte segments of four rows — with the kernel's throttle (vmcnt(1) after every 8th store) and pacing.
//   hipcc --offload-arch=gfx950 -O3 tools/store_persist.hip -o /tmp/store_persist && /tmp/store_persist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// SHAPE 0: the kernel's (lane = residue r, 16 lanes per row segment, stores u = 0..15 walk the row)
// SHAPE 1: a wave-instruction writes 512 contiguous bytes of ONE row (lane = 8-byte cell), four instructions finish the row
// SHAPE 2: SHAPE 1 with 16-byte stores: 1 KiB of one row per instruction
template <int NT, int SHAPE>
__global__ __launch_bounds__(NT) void k_persist(float2* out, int C, int F, int pace_T, int pace_K, int th_every, int group_sleep)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, f = j * 8 + xcd;
    if (f >= F) return;
    float2* mapf = out + (size_t)f * C * 64 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long t_next = wall_clock64();
    for (int c = 0; c < C; c++) {
        if (group_sleep) { __syncthreads(); for (int s = 0; s < group_sleep; s++) __builtin_amdgcn_s_sleep(64); __syncthreads(); }
        if (SHAPE == 0) {
            for (int w0 = 0; w0 < 1024; w0 += NT) {
                const int w = w0 + tid, r = w & 15, ql = w >> 4;
                float2* row = mapf + (size_t)(C * ql + c) * 256;
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    if ((u & 7) == 0 && pace_T) {
                        long long now = wall_clock64();
                        t_next += pace_T;
                        if (now - t_next > (long long)pace_K * pace_T) t_next = now - (long long)pace_K * pace_T;
                        while (now < t_next) { __builtin_amdgcn_s_sleep(1); now = wall_clock64(); }
                    }
                    const v2f t = {(float)u, (float)w};
                    __builtin_nontemporal_store(t, (v2f*)(row + ((16 * u + r + 128) & 255)));
                    if (th_every && (u % th_every) == th_every - 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                }
            }
        } else {
            // the same 16 KiB per wave per trip: SHAPE 1: 4 rows x 4 instr x 512 B; SHAPE 2: 4 rows x 2 instr x 1 KiB
            for (int q0 = wave * 4; q0 < 64; q0 += (NT / 64) * 4) {
                int n = 0;
                for (int qq = 0; qq < 4; qq++) {
                    float2* row = mapf + (size_t)(C * (q0 + qq) + c) * 256;
                    for (int part = 0; part < (SHAPE == 1 ? 4 : 2); part++) {
                        if ((n % (SHAPE == 1 ? 8 : 4)) == 0 && pace_T) {
                            long long now = wall_clock64();
                            t_next += pace_T;
                            if (now - t_next > (long long)pace_K * pace_T) t_next = now - (long long)pace_K * pace_T;
                            while (now < t_next) { __builtin_amdgcn_s_sleep(1); now = wall_clock64(); }
                        }
                        if (SHAPE == 1) { const v2f t = {(float)part, (float)lane}; __builtin_nontemporal_store(t, (v2f*)(row + part * 64 + lane)); }
                        else { const v4f t = {(float)part, (float)lane, 0.f, 1.f}; __builtin_nontemporal_store(t, (v4f*)(row + part * 128 + lane * 2)); }
                        n++;
                        if (th_every && (n % (SHAPE == 1 ? th_every : th_every / 2)) == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    }
                }
            }
        }
    }
}

int main(int argc, char** argv)
{
    const int C = getenv("SB_C") ? atoi(getenv("SB_C")) : 32;               // 32: config B (NR 2048), 128: config D (NR 8192)
    const int NT = getenv("SB_NT") ? atoi(getenv("SB_NT")) : (C == 32 ? 256 : 512);
    const int F = getenv("SB_F") ? atoi(getenv("SB_F")) : (C == 32 ? 512 : 256);
    const int shape = getenv("SB_SHAPE") ? atoi(getenv("SB_SHAPE")) : 0;
    const int th = getenv("SB_TH") ? atoi(getenv("SB_TH")) : 8;
    const int gs = getenv("SB_SLEEP") ? atoi(getenv("SB_SLEEP")) : 0;
    const size_t bytes = (size_t)F * C * 64 * 2048;
    float2* d; CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int a = 1; a < argc; a++) {
        const int word = (int)strtol(argv[a], nullptr, 0), T = word & 0xfff, K = (word >> 12) & 0xf;
        auto launch = [&] {
#define L(NT_, S_) hipLaunchKernelGGL((k_persist<NT_, S_>), dim3(((F + 7) / 8) * 8), dim3(NT_), 0, 0, d, C, F, T, K, th, gs)
            if (NT == 256) { if (shape == 0) L(256, 0); else if (shape == 1) L(256, 1); else L(256, 2); }
            else { if (shape == 0) L(512, 0); else if (shape == 1) L(512, 1); else L(512, 2); }
        };
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("C=%d NT=%d F=%d shape=%d th=%d sleep=%d pace T=%d K=%d: %8.4f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", C, NT, F, shape, th, gs, T, K, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
    }
    hipFree(d);
    return 0;
}
