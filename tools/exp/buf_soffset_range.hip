// Does the hardware range check of a raw buffer access (stride 0) include the SCALAR offset?  (round 3: the persistent kernel's epilogues put
// the row inside the tile into soffset and rely on the descriptor's num_records to drop rows beyond M.)
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/buf_soffset_range.hip -o tools/exp/_build/buf_soffset_range
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
__global__ void k(unsigned* buf, unsigned* res, int soff, int voff_extra) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1024, 0x00020000);    // 1024 bytes in range
    const unsigned vo = threadIdx.x * 16 + voff_extra;
    u32x4_t v = {0xAAAA0000u + threadIdx.x, 1, 2, 3};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, vo, soff, 0);
    u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0);
    res[threadIdx.x] = r[0];
}
int main() {
    unsigned *buf, *res;
    hipMalloc(&buf, 8192); hipMalloc(&res, 256);
    for (int t = 0; t < 4; t++) {
        const int soff = (t & 1) ? 1024 : 0, vex = (t & 2) ? 1024 : 0;
        hipMemset(buf, 0, 8192); hipMemset(res, 0xFF, 256);
        hipLaunchKernelGGL(k, dim3(1), dim3(16), 0, 0, buf, res, soff, vex);
        unsigned h[2048], r[16];
        hipMemcpy(h, buf, 8192, hipMemcpyDeviceToHost); hipMemcpy(r, res, 64, hipMemcpyDeviceToHost);
        int first = -1, n = 0;
        for (int i = 0; i < 2048; i++) if (h[i]) { if (first < 0) first = i * 4; n++; }
        printf("soffset %4d voffset+%4d: first nonzero byte %d, %d nonzero dwords; loaded[0] = %08x\n", soff, vex, first, n, r[0]);
    }
    return 0;
}
