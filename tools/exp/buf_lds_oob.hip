// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS to LDS, or nothing?  (experiment; gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, int nbytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; i++) lds[lane * 4 + i] = 0xABABABABu;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const uint32_t voff = (lane & 1) ? 0xFFFFFF00u : lane * 16;   // odd lanes: out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = 0; i < 4; i++) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
    uint32_t h[256], *d, *o;
    for (int i = 0; i < 256; i++) h[i] = 0x1000 + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, (int)sizeof(h));
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane 0: %08x %08x | lane 1 (OOB): %08x %08x %08x %08x | lane 2: %08x | lane 3 (OOB): %08x\n", h[0], h[1], h[4], h[5], h[6], h[7], h[8], h[12]);
    return 0;
}
