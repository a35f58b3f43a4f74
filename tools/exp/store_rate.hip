// Per-CU global store rate probe (experiment; built and run by tools/exp/store_rate.sh).  Each workgroup of 512 threads
// writes its own contiguous region with 16-byte stores, 8 rows x 128 B per wave instruction (the GEMM epilogue's shape), from
// registers; grids of 8 .. 2048 workgroups tell a per-CU limit from a chip-wide one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ unsigned long long g_ticks[2];
__global__ __launch_bounds__(512) void store_probe(uint4* __restrict__ out, int iters, int row_stride16, int mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // region of this workgroup: iters * 8 waves * 1 KB
    uint4* base = out + (size_t)blockIdx.x * iters * 8 * 64;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    if (mode == 0) {   // 1 KB contiguous per wave instruction
        for (int i = 0; i < iters; i++) base[((size_t)i * 8 + wave) * 64 + lane] = v;
    } else if (mode == 1) {   // 8 rows x 128 B, rows row_stride16 * 16 B apart (a 256-column bf16 tile row of an N-wide matrix)
        uint4* b = out + (size_t)blockIdx.x * 8;   // tile column offset
        for (int i = 0; i < iters; i++) {
            const size_t row = (size_t)(i * 8 + wave) * 8 + (lane >> 3);
            b[row * row_stride16 + (lane & 7)] = v;
        }
    } else {   // mode 2: like 0 but only waves 0..3 store twice as much (fewer issuing waves)
        if (wave < 4)
            for (int i = 0; i < 2 * iters; i++) base[((size_t)i * 4 + wave) * 64 + lane] = v;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { atomicAdd(&g_ticks[0], t1 - t0); atomicAdd(&g_ticks[1], t2 - t0); }
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    uint4* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 32;   // 32 x 8 KB = 256 KB per workgroup (two bf16 tiles' worth)
    for (int mode = 0; mode < 3; mode++)
        for (int wgs : {8, 32, 64, 128, 256, 512, 2048}) {
            const int stride16 = mode == 1 ? 4224 * 2 / 16 : 0;
            if (mode == 1 && wgs > 528 / 8 * 8) {}
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(store_probe, dim3(wgs), dim3(512), 0, 0, d, iters, stride16, mode);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            unsigned long long tk[2], z[2] = {0, 0};
            hipMemcpyFromSymbol(tk, HIP_SYMBOL(g_ticks), sizeof(tk));
            hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), z, sizeof(z));
            const double nw = 5.0 * wgs * (mode == 2 ? 4 : 8);
            printf("   in-kernel per wave: issue %.0f ns, issue + completion %.0f ns (256 KB per workgroup) -> %.1f GB/s per workgroup\n", tk[0] / nw * 10, tk[1] / nw * 10,
                   256.0 * 1024 / (tk[1] / nw * 10));
            const double b = (double)wgs * iters * 8 * 1024;
            printf("mode %d wgs %4d: %8.1f us  %7.1f GB/s total  %6.1f GB/s per workgroup (%.1f B/clk at 2.4 GHz)\n", mode, wgs, best * 1e3, b / best / 1e6,
                   b / best / 1e6 / (wgs < 256 ? wgs : 256), b / best / 1e6 / (wgs < 256 ? wgs : 256) / 2.4);
        }
    return 0;
}
