// conv_unit.hip -- direct unit check of launch_conv_gemm's GRU epilogue (RAFT's q convolution) against a host reference, twice (determinism).
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include -I videotgb_amd/csrc tools/exp/conv_unit.hip -o tools/exp/_build/conv_unit -L videotgb_amd -lvtgb -Wl,-rpath,'$ORIGIN/../../../videotgb_amd'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static float bf(float x) { return (float)(__bf16)x; }
static unsigned rng = 12345;
static float rnd() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
int main(int argc, char** argv) {
    const int n_img = argc > 1 ? atoi(argv[1]) : 2, H = 16, W = 16, M = n_img * H * W, N = 128, KH = argc > 2 ? atoi(argv[2]) : 1, KW = KH == 1 ? 5 : 1, Cin = 256, K = KH * KW * Cin;
    std::vector<float> rh((size_t)M * 128), x((size_t)M * 256), w((size_t)N * K), h((size_t)M * 128), z((size_t)M * 256);
    for (auto& v : rh) v = bf(rnd()); for (auto& v : x) v = bf(rnd()); for (auto& v : w) v = bf(rnd() * 0.05f);
    for (auto& v : h) v = rnd(); for (auto& v : z) v = bf(0.5f + 0.4f * rnd());
    // K order: 64-channel chunk major, tap minor: k = (chunk * taps + tap) * 64 + c
    auto widx = [&](int n, int c, int tap) { return (size_t)n * K + ((size_t)(c / 64) * (KH * KW) + tap) * 64 + (c % 64); };
    std::vector<float> ref((size_t)M * 128);
    for (int m = 0; m < M; m++) {
        const int img = m / (H * W), y = (m / W) % H, xx = m % W;
        for (int n = 0; n < N; n++) {
            double acc = 0;
            for (int tap = 0; tap < KH * KW; tap++) {
                const int dy = KH == 1 ? 0 : tap - 2, dx = KH == 1 ? tap - 2 : 0, yy = y + dy, xc = xx + dx;
                if (yy < 0 || yy >= H || xc < 0 || xc >= W) continue;
                const size_t pm = (size_t)img * H * W + yy * W + xc;
                for (int c = 0; c < 128; c++) acc += (double)rh[pm * 128 + c] * w[widx(n, c, tap)];
                for (int c = 0; c < 128; c++) acc += (double)x[pm * 256 + 128 + c] * w[widx(n, 128 + c, tap)];
            }
            const double zz = z[(size_t)m * 256 + n], hh = h[(size_t)m * 128 + n];
            ref[(size_t)m * 128 + n] = (float)((1 - zz) * hh + zz * tanh(acc));
        }
    }
    auto up16 = [&](const std::vector<float>& v) { std::vector<__bf16> o(v.size()); for (size_t i = 0; i < v.size(); i++) o[i] = (__bf16)v[i]; void* d; CK(hipMalloc(&d, o.size() * 2)); CK(hipMemcpy(d, o.data(), o.size() * 2, hipMemcpyHostToDevice)); return d; };
    void *d_rh = up16(rh), *d_x = up16(x), *d_w = up16(w), *d_z = up16(z), *d_zero, *d_hb;
    float* d_h;
    CK(hipMalloc(&d_h, h.size() * 4)); CK(hipMalloc(&d_zero, 256)); CK(hipMemset(d_zero, 0, 256)); CK(hipMalloc(&d_hb, h.size() * 2));
    std::vector<float> out[3];
    for (int run = 0; run < 3; run++) {
        CK(hipMemcpy(d_h, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        GemmDesc d; memset(&d, 0, sizeof(d));
        d.dtype = VTGB_BF16; d.M = M; d.N = N; d.K = K; d.epi = VTGB_EPI_GRU;
        d.A = d_rh; d.lda = 128; d.A2 = (char*)d_x + 128 * 2; d.lda2 = 256; d.W = d_w; d.ldw = K; d.out = d_h; d.ldo = 128;
        d.conv_H = H; d.conv_W = W; d.conv_KH = KH; d.conv_KW = KW; d.conv_Cin = Cin; d.conv_split = 128; d.zero_page = d_zero;
        d.resid = d_h; d.ldr = 128; d.aux = d_z; d.ldaux = 256; d.out2 = d_hb; d.ldo2 = 128;
        const int rc = launch_conv_gemm(d, 0);
        if (rc) { printf("launch failed %d: %s\n", rc, vtgb_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        out[run].resize(h.size());
        CK(hipMemcpy(out[run].data(), d_h, h.size() * 4, hipMemcpyDeviceToHost));
        double e = 0; int bad = 0, first = -1;
        for (size_t i = 0; i < h.size(); i++) { const double dd = fabs(out[run][i] - ref[i]); if (dd > e) e = dd; if (dd > 2e-2) { bad++; if (first < 0) first = (int)i; } }
        size_t dif = 0; int fd = -1;
        if (run) for (size_t i = 0; i < h.size(); i++) if (out[run][i] != out[0][i]) { dif++; if (fd < 0) fd = (int)i; }
        printf("run %d: max|gpu - ref| %.3e, elements off by > 2e-2: %d (first at row %d col %d); differs from run 0 in %zu elements (first row %d col %d)\n", run, e, bad,
               first < 0 ? -1 : first / 128, first < 0 ? -1 : first % 128, dif, fd < 0 ? -1 : fd / 128, fd < 0 ? -1 : fd % 128);
        { std::vector<unsigned short> hb(h.size()); CK(hipMemcpy(hb.data(), d_hb, h.size() * 2, hipMemcpyDeviceToHost)); size_t mism = 0; for (size_t i = 0; i < h.size(); i++) { unsigned u = (unsigned)hb[i] << 16; float f; memcpy(&f, &u, 4); if (fabs(f - out[run][i]) > 0.01f * fabs(out[run][i]) + 1e-3f) mism++; } printf("   hb (bf16 copy) inconsistent with h32 in %zu elements\n", mism); }
        if (bad) { int shown = 0; for (size_t i = 0; i < h.size() && shown < 12; i++) if (fabs(out[run][i] - ref[i]) > 2e-2) { printf("   row %zu col %zu: gpu %.4f ref %.4f  h_in %.4f z %.4f (1-z)h %.4f\n", i / 128, i % 128, out[run][i], ref[i], h[i], z[(i / 128) * 256 + i % 128], (1 - z[(i / 128) * 256 + i % 128]) * h[i]); shown++; } }
    }
    return 0;
}
