// single-wave latency of a dependent chain "fetch partner, add" for the cross-lane mechanisms a trellis step could use (tools only)
//   hipcc --offload-arch=gfx950 -O3 -o xlane_lat xlane_lat.hip && ./xlane_lat
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void chain(unsigned* out, int iters)
{
    unsigned v = threadIdx.x * 2654435761u;
    const int lane = threadIdx.x & 63;
    const int k = lane >> 1;
    for (int i = 0; i < iters; i++) {
        unsigned p;
        if (MODE == 0) { const unsigned a = __shfl(v, k), b = __shfl(v, k + 32); p = a ^ (b + 1); }                       // two ds_bpermute
        else if (MODE == 1) p = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false);                 // quad_perm [1,0,3,2]: xor 1
        else if (MODE == 2) p = (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x101F | (4 << 10) & 0x7C00 | 0x1F);          // swizzle (bit mode), xor 4
        else if (MODE == 3) { p = (unsigned)__shfl_xor(v, 32); }                                                             // one ds_bpermute
        else if (MODE == 4) {                                                                                                // two DPP moves: xor 4 inside a row
            unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0,2
            p = (unsigned)__builtin_amdgcn_update_dpp((int)t, (int)v, 0x114, 0xf, 0xA, false);            // row_shr:4 into banks 1,3
        } else {
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
            auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); p = r[0] ^ r[1];
#else
            p = v;
#endif
        }
        v = v * 3u + p;
    }
    out[threadIdx.x] = v;
}

template <int MODE>
static void run(const char* name, unsigned* d)
{
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64), 0, 0, d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain<MODE>, dim3(1), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.1f ns per link\n", name, ms * 1e6 / iters);
}

int main()
{
    unsigned* d; hipMalloc(&d, 256);
    run<0>("two ds_bpermute (k, k+32)", d);
    run<3>("one ds_bpermute (xor 32)", d);
    run<2>("ds_swizzle", d);
    run<1>("DPP quad_perm (xor 1)", d);
    run<4>("two DPP row shifts (xor 4)", d);
    run<5>("v_permlane32_swap (xor 32)", d);
    return 0;
}
