#include <hip/hip_runtime.h>
#include <stdio.h>
#include "pair_h8.h"
__global__ void k(const float* x, float* y, unsigned* raw) {
    int i = threadIdx.x;
    f32x4 v = {x[4*i], x[4*i+1], x[4*i+2], x[4*i+3]};
    h8_u32x2 hi, lo;
    h8_split4(v, hi, lo);
    f32x4 r = h8_join4(hi, lo);
    for (int e = 0; e < 4; e++) y[4*i+e] = r[e];
    raw[4*i] = hi[0]; raw[4*i+1] = hi[1]; raw[4*i+2] = lo[0]; raw[4*i+3] = lo[1];
}
int main() {
    float hx[256], hy[256]; unsigned hr[256];
    for (int i = 0; i < 256; i++) hx[i] = (float)(i - 100) * 0.0137f + 0.001f * i * i;
    float *dx, *dy; unsigned* dr;
    hipMalloc(&dx, 1024); hipMalloc(&dy, 1024); hipMalloc(&dr, 1024);
    hipMemcpy(dx, hx, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dy, dr);
    hipMemcpy(hy, dy, 1024, hipMemcpyDeviceToHost); hipMemcpy(hr, dr, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 256; i++) { double e = fabs(hy[i] - hx[i]) / (fabs(hx[i]) + 1e-9); if (e > worst) worst = e; }
    printf("join(split(x)) worst rel err %.3e\n", worst);
    for (int i = 0; i < 2; i++) printf("x %g %g %g %g -> y %g %g %g %g  raw %08x %08x %08x %08x\n", hx[4*i], hx[4*i+1], hx[4*i+2], hx[4*i+3], hy[4*i], hy[4*i+1], hy[4*i+2], hy[4*i+3], hr[4*i], hr[4*i+1], hr[4*i+2], hr[4*i+3]);
    return 0;
}
