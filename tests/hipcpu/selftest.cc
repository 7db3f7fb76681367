// Self-test of the emulation's cross-lane semantics against values worked out by hand from the ISA's definitions (DPP control words with row / bank
// masks and bound_ctrl, v_permlane16/32_swap, shuffles with a width, ballot under divergence, readlane, workgroup barriers with early exits,
// dynamic LDS, refused launches).  Built and run by tests/test_emulated_kernels.py.
#include <hip/hip_runtime.h>
#include <vector>

#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)
static int fails = 0;

__global__ void k_cross_lane(int* out /* [12][64] */)
{
    const int l = threadIdx.x & 63;
    int* o = out;
    o[0 * 64 + l] = __shfl_xor(l, 5);
    o[1 * 64 + l] = __shfl(l * 10, 3, 16);                                   // lane 3 of this lane's group of 16
    o[2 * 64 + l] = __shfl_down(l, 4, 8);                                    // within groups of 8, own value past the end
    o[3 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x118, 0xf, 0xc, false);      // row_shr:8 into banks 2, 3: pos - 8; banks 0, 1 keep old (-1)
    o[4 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xf, 0xf, true);       // row_shr:1, bound_ctrl: position 0 of every row reads 0
    o[5 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x142, 0xa, 0xf, false);      // row_bcast15 into rows 1 and 3
    o[6 * 64 + l] = __builtin_amdgcn_mov_dpp(l, 0xB1, 0xf, 0xf, false);              // quad_perm [1, 0, 3, 2]
    o[7 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x141, 0xf, 0xf, false);      // row_half_mirror
    { const auto r = __builtin_amdgcn_permlane32_swap(l, l, false, false); o[8 * 64 + l] = (int)r[0]; o[9 * 64 + l] = (int)r[1]; }
    { const auto r = __builtin_amdgcn_permlane16_swap(l, l, false, false); o[10 * 64 + l] = (int)r[0]; o[11 * 64 + l] = (int)r[1]; }
}

__global__ void k_divergent(unsigned long long* masks, int* vals)
{
    const int l = threadIdx.x;
    if (l & 1) {
        masks[l] = __ballot(l >= 32);                                        // only the odd lanes are here
        vals[l] = __builtin_amdgcn_readfirstlane(l);
    } else {
        masks[l] = __ballot(1);
        vals[l] = __builtin_amdgcn_readlane(l * 2, 6);
    }
}

__global__ void k_barriers(int* out, int n)
{
    __shared__ int s[256];
    int* const dyn = (int*)::hipcpu::dyn_lds();
    const int t = threadIdx.x;
    if (t >= n) return;                                                      // early exits: the barrier counts the live work-items
    s[t] = t + 1;
    dyn[t] = 1000 * (int)blockIdx.x;
    __syncthreads();
    const int v = s[(t + 1) % n] + dyn[(t + 7) % n];
    const int any = __syncthreads_or(t == n - 1);
    out[blockIdx.x * 256 + t] = v + (any ? 100000 : 0);
}

int main()
{
    int* d = nullptr;
    hipMalloc((void**)&d, sizeof(int) * 12 * 64);
    hipLaunchKernelGGL(k_cross_lane, dim3(1), dim3(64), 0, 0, d);
    for (int l = 0; l < 64; l++) {
        const int pos = l & 15, row = l >> 4;
        CHECK(d[0 * 64 + l] == (l ^ 5));
        CHECK(d[1 * 64 + l] == 10 * ((l & ~15) | 3));
        CHECK(d[2 * 64 + l] == (((l & 7) + 4 < 8) ? l + 4 : l));
        CHECK(d[3 * 64 + l] == (pos >= 8 ? l - 8 : -1));
        CHECK(d[4 * 64 + l] == (pos >= 1 ? l - 1 : 0));
        CHECK(d[5 * 64 + l] == ((row == 1 || row == 3) ? row * 16 - 1 : -1));
        CHECK(d[6 * 64 + l] == (l ^ 1));
        CHECK(d[7 * 64 + l] == row * 16 + ((pos & 8) | (7 - (pos & 7))));
        CHECK(d[8 * 64 + l] == (l & 31));                                    // lo of the pair (l, l ^ 32)
        CHECK(d[9 * 64 + l] == (l | 32));                                    // hi
        CHECK(d[10 * 64 + l] == (l & ~16));
        CHECK(d[11 * 64 + l] == (l | 16));
    }
    unsigned long long* m = nullptr; int* v = nullptr;
    hipMalloc((void**)&m, 8 * 64); hipMalloc((void**)&v, 4 * 64);
    hipLaunchKernelGGL(k_divergent, dim3(1), dim3(64), 0, 0, m, v);
    for (int l = 0; l < 64; l++) {
        if (l & 1) { CHECK(m[l] == 0xAAAAAAAA00000000ull); CHECK(v[l] == 1); }
        else { CHECK(m[l] == 0x5555555555555555ull); CHECK(v[l] == 12); }
    }
    int* o = nullptr;
    hipMalloc((void**)&o, sizeof(int) * 3 * 256);
    hipLaunchKernelGGL(k_barriers, dim3(3), dim3(256), 256 * sizeof(int), 0, o, 200);
    for (int b = 0; b < 3; b++)
        for (int t = 0; t < 200; t++) CHECK(o[b * 256 + t] == ((t + 1) % 200) + 1 + 1000 * b + 100000);
    CHECK(hipGetLastError() == hipSuccess);
    hipLaunchKernelGGL(k_barriers, dim3(1), dim3(256), 80 * 1024, 0, o, 200);            // dynamic LDS above 64 KB without the opt-in: refused
    CHECK(hipGetLastError() == hipErrorInvalidValue);
    CHECK(hipFuncSetAttribute(k_barriers, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess);
    hipLaunchKernelGGL(k_barriers, dim3(1), dim3(256), 80 * 1024, 0, o, 200);
    CHECK(hipGetLastError() == hipSuccess);
    hipLaunchKernelGGL(k_barriers, dim3(1), dim3(2048), 0, 0, o, 200);                   // more than 1024 work-items: refused
    CHECK(hipGetLastError() == hipErrorInvalidValue);
    // a captured stream records, a graph launch replays
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int i = 0; i < 3 * 256; i++) o[i] = -7;
    hipGraph_t g; hipGraphExec_t gx;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess);
    hipLaunchKernelGGL(k_barriers, dim3(1), dim3(256), 256 * sizeof(int), s, o, 64);
    CHECK(o[0] == -7);                                                                   // nothing ran during capture
    CHECK(hipStreamEndCapture(s, &g) == hipSuccess && hipGraphInstantiate(&gx, g, nullptr, nullptr, 0) == hipSuccess);
    CHECK(hipGraphLaunch(gx, s) == hipSuccess && o[0] == 2 + 100000);
    hipGraphDestroy(g); hipGraphExecDestroy(gx); hipStreamDestroy(s);
    hipFree(d); hipFree(m); hipFree(v); hipFree(o);
    printf(fails ? "selftest FAILED (%d)\n" : "selftest ok\n", fails);
    return fails ? 1 : 0;
}
