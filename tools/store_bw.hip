// store_bw.hip — micro-benchmark of map-store patterns on MI355X (tools only; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(float4 v, float4* p) { v4f w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (v4f*)p); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) linear
__global__ void k_linear(float4* out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float4 v = make_float4(i, 1, 2, 3);
    for (; i < n4; i += st) out[i] = v;
}
// (b) fused-kernel pattern: WG (f,c) writes rows k = C*ql + c, 64 rows, NA=256 cf32 per row (=128 float4).
// thread item w: i = w%8, ql = w/8; for u<16: a' = 16u+2i -> float4 index (k*256 + a')/2
template <int NT>
__global__ void k_pattern(float4* out, int C, int F) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, f = (j / C) * 8 + xcd, c = j % C;
    if (f >= F) return;
    float4* mapf = out + (size_t)f * C * 64 * 128;
    for (int w = threadIdx.x; w < 512; w += 256) {
        int i = w & 7, ql = w >> 3, k = C * ql + c;
        float4* row = mapf + (size_t)k * 128;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            float4 v = make_float4(u, w, k, f);
            if (NT) nt_store(v, row + ((16 * u + 2 * i + 128) & 255) / 2);
            else row[((16 * u + 2 * i + 128) & 255) / 2] = v;
        }
    }
}
// (b2) same rows, but 8-byte stores: 16 lanes cover one 128-byte line (one residue per lane)
__global__ void k_pattern_x2(float2* out, int C, int F) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, f = (j / C) * 8 + xcd, c = j % C;
    if (f >= F) return;
    float2* mapf = out + (size_t)f * C * 64 * 256;
    for (int w = threadIdx.x; w < 1024; w += 256) {
        int r = w & 15, ql = w >> 4, k = C * ql + c;
        float2* row = mapf + (size_t)k * 256;
#pragma unroll
        for (int u = 0; u < 16; u++) row[(16 * u + r + 128) & 255] = make_float2(u, w);
    }
}
// (b3) (b2) with non-temporal 8-byte stores (what the fused kernel does since the map became write-around)
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k_pattern_x2_nt(float2* out, int C, int F) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, f = (j / C) * 8 + xcd, c = j % C;
    if (f >= F) return;
    float2* mapf = out + (size_t)f * C * 64 * 256;
    for (int w = threadIdx.x; w < 1024; w += 256) {
        int r = w & 15, ql = w >> 4, k = C * ql + c;
        float2* row = mapf + (size_t)k * 256;
#pragma unroll
        for (int u = 0; u < 16; u++) { v2f t = {(float)u, (float)w}; __builtin_nontemporal_store(t, (v2f*)(row + ((16 * u + r + 128) & 255))); }
    }
}
// (d) row-contiguous: each wave writes 1 KB contiguous per instruction; WG writes its 64 rows, 2 instr per row
template <int NT>
__global__ void k_rows(float4* out, int C, int F) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, f = (j / C) * 8 + xcd, c = j % C;
    if (f >= F) return;
    float4* mapf = out + (size_t)f * C * 64 * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ql = wave; ql < 64; ql += 4) {
        float4* row = mapf + (size_t)(C * ql + c) * 128;
        float4 v = make_float4(ql, lane, c, f);
        if (NT) { nt_store(v, row + lane); nt_store(v, row + 64 + lane); }
        else { row[lane] = v; row[64 + lane] = v; }
    }
}
// (e) like (d) but rows of a WG are CONTIGUOUS (k = 64*c + ql): 128 KB contiguous per WG
__global__ void k_rows_contig(float4* out, int C, int F) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, f = (j / C) * 8 + xcd, c = j % C;
    if (f >= F) return;
    float4* mapf = out + (size_t)f * C * 64 * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int ql = wave; ql < 64; ql += 4) {
        float4* row = mapf + (size_t)(64 * c + ql) * 128;
        float4 v = make_float4(ql, lane, c, f);
        row[lane] = v; row[64 + lane] = v;
    }
}

// (f) range-Doppler layout: unit (fp, tile) writes, per class c < Ir, the 128-byte line `tile` of rows Ir*k + c, k < N (row pitch ND*8 B);
// 8 lanes per line (16-byte non-temporal stores), rows in the scattered order of a digit-reversed FFT or in natural order
template <int NT_, bool SCATTER>
__global__ void k_rd(float4* out, int N, int Ir, int ND, long n_units) {
    const int xcd = blockIdx.x & 7; const long u = (long)(blockIdx.x >> 3) * 8 + xcd;
    if (u >= n_units) return;
    const int tiles = ND / 16; const long fp = u / tiles; const int tile = (int)(u % tiles);
    float4* base = out + ((size_t)fp * N * Ir * ND + tile * 16) / 2;
    const int seg = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    for (int c = 0; c < Ir; c++)
        for (int j = 0; j < 8; j++) {
            int row = r0 + j * (NT_ / 8);
            if (SCATTER) { unsigned k = __brev((unsigned)row) >> (32 - (NT_ == 1024 ? 10 : 8)); row = (int)k; }
            nt_store(make_float4(row, c, seg, 1.f), base + ((size_t)(Ir * row + c) * ND) / 2 + seg);
        }
}
// (g) like (f) with 8-bin tiles: 64-byte half lines, 4 lanes per row
template <int NT_>
__global__ void k_rd64(float4* out, int N, int Ir, int ND, long n_units) {
    const int xcd = blockIdx.x & 7; const long u = (long)(blockIdx.x >> 3) * 8 + xcd;
    if (u >= n_units) return;
    const int tiles = ND / 8; const long fp = u / tiles; const int tile = (int)(u % tiles);
    float4* base = out + ((size_t)fp * N * Ir * ND + tile * 8) / 2;
    const int seg = threadIdx.x & 3, r0 = threadIdx.x >> 2;
    for (int c = 0; c < Ir; c++)
        for (int j = 0; j < N / (NT_ / 4); j++) {
            int row = r0 + j * (NT_ / 4);
            unsigned k = __brev((unsigned)row) >> 22; row = (int)k;
            nt_store(make_float4(row, c, seg, 1.f), base + ((size_t)(Ir * row + c) * ND) / 2 + seg);
        }
}

// (h) like (f) with DT-bin tiles: DT*8-byte pieces of the rows (DT/2 lanes per piece), scattered row order
template <int NT_, int DT, bool ROT = false>
__global__ void k_rdw(float4* out, int N, int Ir, int ND, long n_units) {
    const int xcd = blockIdx.x & 7; const long u = (long)(blockIdx.x >> 3) * 8 + xcd;
    if (u >= n_units) return;
    constexpr int LPR = DT / 2, RPI = NT_ / LPR;
    const int tiles = ND / DT; const long fp = u / tiles; const int tile = (int)(u % tiles);
    float4* base = out + ((size_t)fp * N * Ir * ND + tile * DT) / 2;
    const int seg = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
    for (int ci = 0; ci < Ir; ci++) {
        const int c = ROT ? (int)((ci + u) % Ir) : ci;                      // ROT: the workgroups do not all work on the same class at the same time
        for (int j = 0; j < N / RPI; j++) {
            int row = r0 + j * RPI;
            unsigned k = __brev((unsigned)row) >> 22; row = (int)k;
            nt_store(make_float4(row, c, seg, 1.f), base + ((size_t)(Ir * row + c) * ND) / 2 + seg);
        }
    }
}

// (i) like (h), one class per workgroup: the Ir workgroups of a unit run side by side on one XCD, so the rows Ir*k + 0 .. Ir*k + Ir-1 are
// written at about the same time
template <int NT_, int DT>
__global__ void k_rdc(float4* out, int N, int Ir, int ND, long n_units) {
    const int xcd = blockIdx.x & 7; const long j = blockIdx.x >> 3; const int c = (int)(j % Ir); const long u = (j / Ir) * 8 + xcd;
    if (u >= n_units) return;
    constexpr int LPR = DT / 2, RPI = NT_ / LPR;
    const int tiles = ND / DT; const long fp = u / tiles; const int tile = (int)(u % tiles);
    float4* base = out + ((size_t)fp * N * Ir * ND + tile * DT) / 2;
    const int seg = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
    for (int jj = 0; jj < N / RPI; jj++) {
        int row = r0 + jj * RPI;
        unsigned k = __brev((unsigned)row) >> 22; row = (int)k;
        nt_store(make_float4(row, c, seg, 1.f), base + ((size_t)(Ir * row + c) * ND) / 2 + seg);
    }
}

int main() {
    const int F = getenv("SB_F") ? atoi(getenv("SB_F")) : 256, C = 32;   // config B: 256 frames x 2048 rows x 2 KB = 1 GiB
    const size_t bytes = (size_t)F * C * 64 * 2048;
    float4* d; CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
        return 0;
    };
    const int grid = F * C;
    run("linear grid=2048", [&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, d, bytes / 16); });
    run("linear grid=8192", [&] { hipLaunchKernelGGL(k_linear, dim3(8192), dim3(256), 0, 0, d, bytes / 16); });
    run("linear grid=65536", [&] { hipLaunchKernelGGL(k_linear, dim3(65536), dim3(256), 0, 0, d, bytes / 16); });
    run("pattern 128B segs", [&] { hipLaunchKernelGGL(k_pattern<0>, dim3(grid), dim3(256), 0, 0, d, C, F); });
    run("pattern 128B segs dwordx2", [&] { hipLaunchKernelGGL(k_pattern_x2, dim3(grid), dim3(256), 0, 0, (float2*)d, C, F); });
    run("pattern 128B segs dwordx2 nt", [&] { hipLaunchKernelGGL(k_pattern_x2_nt, dim3(grid), dim3(256), 0, 0, (float2*)d, C, F); });
    run("pattern 128B segs nt", [&] { hipLaunchKernelGGL(k_pattern<1>, dim3(grid), dim3(256), 0, 0, d, C, F); });
    run("rows 1KB/instr strided", [&] { hipLaunchKernelGGL(k_rows<0>, dim3(grid), dim3(256), 0, 0, d, C, F); });
    run("rows 1KB/instr strided nt", [&] { hipLaunchKernelGGL(k_rows<1>, dim3(grid), dim3(256), 0, 0, d, C, F); });
    run("rows contiguous per WG", [&] { hipLaunchKernelGGL(k_rows_contig, dim3(grid), dim3(256), 0, 0, d, C, F); });
    CK(hipMemset(d, 0, bytes));
    run("memset-like linear again", [&] { hipLaunchKernelGGL(k_linear, dim3(8192), dim3(256), 0, 0, d, bytes / 16); });
    {   // config D: 8 frames x 16 pairs, 1024 x 8 rows x 128 bins = 1 GiB; config B: 64 x 16, 256 x 8 x 64
        const long nuD = 8L * 16 * 8, nuB = 64L * 16 * 4;
        run("rd config D scattered rows", [&] { hipLaunchKernelGGL((k_rd<1024, true>), dim3((unsigned)nuD), dim3(1024), 0, 0, d, 1024, 8, 128, nuD); });
        run("rd config D natural rows", [&] { hipLaunchKernelGGL((k_rd<1024, false>), dim3((unsigned)nuD), dim3(1024), 0, 0, d, 1024, 8, 128, nuD); });
        run("rd config D 64-byte half lines, 512 thr", [&] { hipLaunchKernelGGL((k_rd64<512>), dim3((unsigned)(nuD * 2)), dim3(512), 0, 0, d, 1024, 8, 128, nuD * 2); });
        {   // wider tiles on 32 frames (4 GiB), so that the 128-bin case still has 512 workgroups
            float4* d4; const size_t b4 = (size_t)32 * 16 * 8192 * 128 * 8;
            CK(hipMalloc(&d4, b4));
            const long nu = 32L * 16 * 8;
            auto run4 = [&](const char* name, auto launch) {
                for (int i = 0; i < 2; i++) launch();
                hipEventRecord(e0);
                for (int i = 0; i < 5; i++) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, b4 / ms / 1e6);
            };
            run4("rd config D x32 frames, 16-bin tiles (128 B)", [&] { hipLaunchKernelGGL((k_rdw<1024, 16>), dim3((unsigned)nu), dim3(1024), 0, 0, d4, 1024, 8, 128, nu); });
            run4("rd config D x32 frames, 32-bin tiles (256 B)", [&] { hipLaunchKernelGGL((k_rdw<1024, 32>), dim3((unsigned)(nu / 2)), dim3(1024), 0, 0, d4, 1024, 8, 128, nu / 2); });
            run4("rd config D x32 frames, 64-bin tiles (512 B)", [&] { hipLaunchKernelGGL((k_rdw<1024, 64>), dim3((unsigned)(nu / 4)), dim3(1024), 0, 0, d4, 1024, 8, 128, nu / 4); });
            run4("rd config D x32 frames, 128-bin tiles (1 KB)", [&] { hipLaunchKernelGGL((k_rdw<1024, 128>), dim3((unsigned)(nu / 8)), dim3(1024), 0, 0, d4, 1024, 8, 128, nu / 8); });
            run4("rd config D x32 frames, 16-bin tiles, classes rotated", [&] { hipLaunchKernelGGL((k_rdw<1024, 16, true>), dim3((unsigned)nu), dim3(1024), 0, 0, d4, 1024, 8, 128, nu); });
            run4("rd config D x32 frames, 128-bin tiles, classes rotated", [&] { hipLaunchKernelGGL((k_rdw<1024, 128, true>), dim3((unsigned)(nu / 8)), dim3(1024), 0, 0, d4, 1024, 8, 128, nu / 8); });
            run4("rd config D x32 frames, 16-bin tiles, WG per class", [&] { hipLaunchKernelGGL((k_rdc<1024, 16>), dim3((unsigned)(nu * 8)), dim3(1024), 0, 0, d4, 1024, 8, 128, nu); });
            run4("rd config D x32 frames, 16-bin tiles, WG per class 256 thr", [&] { hipLaunchKernelGGL((k_rdc<256, 16>), dim3((unsigned)(nu * 8)), dim3(256), 0, 0, d4, 1024, 8, 128, nu); });
            run4("rd config D x32 frames, 128-bin tiles, WG per class", [&] { hipLaunchKernelGGL((k_rdc<1024, 128>), dim3((unsigned)(nu)), dim3(1024), 0, 0, d4, 1024, 8, 128, nu / 8); });
            run4("rd config D x32 frames, 32-bin tiles, 512 thr", [&] { hipLaunchKernelGGL((k_rdw<512, 32>), dim3((unsigned)(nu / 2)), dim3(512), 0, 0, d4, 1024, 8, 128, nu / 2); });
            run4("rd config D x32 frames, 32-bin tiles, 256 thr", [&] { hipLaunchKernelGGL((k_rdw<256, 32>), dim3((unsigned)(nu / 2)), dim3(256), 0, 0, d4, 1024, 8, 128, nu / 2); });
            hipFree(d4);
        }
        run("rd config B scattered rows", [&] { hipLaunchKernelGGL((k_rd<256, true>), dim3((unsigned)nuB), dim3(256), 0, 0, d, 256, 8, 64, nuB); });
        run("rd config B natural rows", [&] { hipLaunchKernelGGL((k_rd<256, false>), dim3((unsigned)nuB), dim3(256), 0, 0, d, 256, 8, 64, nuB); });
    }
    hipFree(d);
    return 0;
}
