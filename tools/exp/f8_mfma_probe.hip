// f8_mfma_probe.hip -- what v_mfma_scale_f32_16x16x128_f8f6f4 and the fp8 conversions do on gfx950, checked against a host decode, before
// the f16c8 RAFT mode relies on them (round 6):
//   (1) A = e4m3 (cbsz 0), B = e5m2 (blgp 1): D[i][j] = 2^(sa - 127) 2^(sb - 127) sum_k A[i][k] B[k][j], with the lane's 32 operand bytes loaded
//       as two 16-byte pieces (bytes [16 g, 16 g + 16) and [64 + 16 g, ...) of a 128-byte row, g = lane >> 4) -- the fragment reads of the
//       implicit-GEMM kernels.  Any k-permutation common to A and B leaves the sum unchanged, so the test passes iff A and B use the same map.
//   (2) fp32 -> e5m2 / e4m3 conversions (v_cvt_pk_bf8_f32 / v_cvt_pk_fp8_f32): rounding and what happens beyond the format's range.
//   build + run:  hipcc --offload-arch=gfx950 -O2 tools/exp/f8_mfma_probe.hip -o /tmp/f8probe && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static float dec_e4m3(uint8_t b) {      // OCP e4m3fn: bias 7, no infinities, 0x7F / 0xFF = NaN
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    if (e == 15 && m == 7) return NAN;
    const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
static float dec_e5m2(uint8_t b) {      // OCP e5m2: bias 15, IEEE-like infinities / NaNs
    const int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
    if (e == 31) return m ? NAN : (s ? -INFINITY : INFINITY);
    const float v = e == 0 ? ldexpf((float)m, -16) : ldexpf(1.0f + m / 4.0f, e - 15);
    return s ? -v : v;
}

__global__ void mfma_probe_c(const uint8_t* A, const uint8_t* B, const float* C, float* D, int sa, int sb) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + r * 128 + g * 16), a1 = *reinterpret_cast<const i32x4*>(A + r * 128 + 64 + g * 16);
    const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + r * 128 + g * 16), b1 = *reinterpret_cast<const i32x4*>(B + r * 128 + 64 + g * 16);
    const i32x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    f32x4 acc;
    for (int e = 0; e < 4; e++) acc[e] = C[(g * 4 + e) * 16 + r];
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 1, 0, sa, 0, sb);
    for (int e = 0; e < 4; e++) D[(g * 4 + e) * 16 + r] = acc[e];
}

__global__ void mfma_probe(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + r * 128 + g * 16), a1 = *reinterpret_cast<const i32x4*>(A + r * 128 + 64 + g * 16);
    const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + r * 128 + g * 16), b1 = *reinterpret_cast<const i32x4*>(B + r * 128 + 64 + g * 16);
    const i32x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 1, 0, sa, 0, sb);
    // C/D layout: col = lane & 15 (B row index here = output column), row = (lane >> 4) * 4 + reg
    for (int e = 0; e < 4; e++) D[(g * 4 + e) * 16 + r] = acc[e];
}

// (3) accumulation precision of the f16 / bf16 16x16x32 MFMA over a long contraction: K = 32 * steps, one accumulator
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__global__ void chain_probe(const uint16_t* A, const uint16_t* B, float* D, int steps) {      // A, B [16][32 * steps] row-major 16-bit
    const int l = threadIdx.x, r = l & 15, g = l >> 4, K = 32 * steps;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; s++) {
        const i32x4 a = *reinterpret_cast<const i32x4*>(A + r * K + s * 32 + g * 8), b = *reinterpret_cast<const i32x4*>(B + r * K + s * 32 + g * 8);
        if (BF) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    for (int e = 0; e < 4; e++) D[(g * 4 + e) * 16 + r] = acc[e];
}
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float b2f(uint16_t u) { uint32_t v = (uint32_t)u << 16; float f; memcpy(&f, &v, 4); return f; }

__global__ void cvt_probe(const float* x, uint8_t* o5, uint8_t* o4, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p5 = __builtin_amdgcn_cvt_pk_bf8_f32(x[i], 0.f, 0, false), p4 = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false);
    o5[i] = (uint8_t)(p5 & 255);
    o4[i] = (uint8_t)(p4 & 255);
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128];
    srand(1);
    for (int i = 0; i < 16 * 128; i++) {
        do { hA[i] = (uint8_t)rand(); } while (isnan(dec_e4m3(hA[i])) || fabsf(dec_e4m3(hA[i])) > 8.f);
        do { hB[i] = (uint8_t)rand(); } while (!isfinite(dec_e5m2(hB[i])) || fabsf(dec_e5m2(hB[i])) > 8.f);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    int bad = 0;
    const int scales[3][2] = {{127, 127}, {115, 127}, {120, 130}};      // E8M0 bytes: 2^(v - 127)
    for (int t = 0; t < 3; t++) {
        mfma_probe<<<1, 64>>>(dA, dB, dD, scales[t][0], scales[t][1]);
        float hD[256];
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0, mx = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double s = 0;
                for (int k = 0; k < 128; k++) s += (double)dec_e4m3(hA[i * 128 + k]) * (double)dec_e5m2(hB[j * 128 + k]);
                s *= ldexp(1.0, scales[t][0] - 127 + scales[t][1] - 127);
                worst = fmax(worst, fabs(s - hD[i * 16 + j])); mx = fmax(mx, fabs(s));
            }
        printf("mfma_scale 16x16x128 (A e4m3, B e5m2), scale bytes (%d, %d): max|diff| = %.3e of max|ref| = %.3e  %s\n", scales[t][0], scales[t][1], worst, mx,
               worst <= 2e-4 * mx ? "OK" : "MISMATCH");
        bad += worst > 2e-4 * mx;      // (the instruction's internal sum is not an fp32 chain: observed 8e-5 of the largest sum)
    }
    // (1b) a large accumulator input next to small products: is C kept to fp32 precision?  C ~ 1e3 (24 significant bits), products scaled by 2^-12
    {
        float hC[256], hD[256];
        for (int i = 0; i < 256; i++) hC[i] = (float)((rand() % 2000001 - 1000000) * 1e-3 * (1.0 + 1e-7 * (rand() % 1000)));
        float* dC; hipMalloc(&dC, sizeof hC); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        mfma_probe_c<<<1, 64>>>(dA, dB, dC, dD, 115, 127);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst_ulp = 0, worst_rel_sum = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double s = 0;
                for (int k = 0; k < 128; k++) s += (double)dec_e4m3(hA[i * 128 + k]) * (double)dec_e5m2(hB[j * 128 + k]);
                s *= ldexp(1.0, -12);
                const double ref = (double)hC[i * 16 + j] + s, err = fabs(ref - hD[i * 16 + j]);
                const double ulp = ldexp(1.0, (int)floor(log2(fabs(ref))) - 23);
                worst_ulp = fmax(worst_ulp, err / ulp);
                worst_rel_sum = fmax(worst_rel_sum, err / fmax(fabs(s), 1e-30));
            }
        printf("C + products (C ~ 1e3, products x 2^-12): worst error = %.2f ulp of the result, %.3e of the product sum  %s\n", worst_ulp, worst_rel_sum,
               worst_ulp <= 1.0 ? "OK (fp32 accumulate)" : "C IS TRUNCATED");
        bad += worst_ulp > 1.0;
    }
    // (1c) one large product next to many small ones: how far below the largest term does a term still count?
    for (int sh = 4; sh <= 28; sh += 4) {
        uint8_t tA[16 * 128], tB[16 * 128];
        for (int i = 0; i < 16 * 128; i++) { tA[i] = 0x38; tB[i] = 0x3c; }                 // 1.0 * 1.0 everywhere ...
        for (int r = 0; r < 16; r++) { tA[r * 128] = 0x38 + 8 * 7; tB[r * 128] = 0x3c + 4 * 14; }      // ... but k = 0: 2^7 * 2^14 = 2^21
        (void)sh;
        hipMemcpy(dA, tA, sizeof tA, hipMemcpyHostToDevice); hipMemcpy(dB, tB, sizeof tB, hipMemcpyHostToDevice);
        mfma_probe<<<1, 64>>>(dA, dB, dD, 127, 127);
        float hD[256];
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        printf("2^21 + 127 x 1.0: D = %.1f (exact %.1f)\n", hD[0], 2097152.0 + 127.0);
        break;
    }
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    // (3) f16 / bf16 chains: K = 2048, positive-mean data (post-ReLU activations x mixed-sign weights) -- error vs a double sum, in units of the result's RMS
    for (int bf = 0; bf < 2; bf++) {
        const int steps = 64, K = 32 * steps;
        uint16_t* hA2 = (uint16_t*)malloc(16 * K * 2); uint16_t* hB2 = (uint16_t*)malloc(16 * K * 2);
        for (int i = 0; i < 16 * K; i++) {
            const float a = (float)((rand() % 20001 - 10000) * 1e-4) * 0.1f, b = fabsf((float)((rand() % 20001 - 10000) * 1e-4)) * 3.f;
            hA2[i] = bf ? f2b(a) : f2h(a); hB2[i] = bf ? f2b(b) : f2h(b);
        }
        uint16_t *dA2, *dB2; hipMalloc(&dA2, 16 * K * 2); hipMalloc(&dB2, 16 * K * 2);
        hipMemcpy(dA2, hA2, 16 * K * 2, hipMemcpyHostToDevice); hipMemcpy(dB2, hB2, 16 * K * 2, hipMemcpyHostToDevice);
        if (bf) chain_probe<true><<<1, 64>>>(dA2, dB2, dD, steps); else chain_probe<false><<<1, 64>>>(dA2, dB2, dD, steps);
        float hD[256];
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double se = 0, sr = 0, sf = 0, bias = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double s = 0; float f = 0.f;
                for (int k = 0; k < K; k++) {
                    const float a = bf ? b2f(hA2[i * K + k]) : h2f(hA2[i * K + k]), b = bf ? b2f(hB2[j * K + k]) : h2f(hB2[j * K + k]);
                    s += (double)a * (double)b; f = fmaf(a, b, f);
                }
                se += (s - hD[i * 16 + j]) * (s - hD[i * 16 + j]); sr += s * s; sf += (s - f) * (s - f); bias += (hD[i * 16 + j] - s);
            }
        printf("%s 16x16x32 chain, K = %d: rms error / rms result = %.3e (an fp32 fmaf chain: %.3e), mean signed error / rms result = %+.3e\n", bf ? "bf16" : "f16 ", K,
               sqrt(se / sr), sqrt(sf / sr), bias / 256 / sqrt(sr / 256));
    }
    // conversions: every interesting magnitude
    const float xs[] = {0.f, 1.f, 1.0625f, 1.125f, 1.1875f, 1.25f, 1.3f, 1.375f, 3.3e-5f, 1.6e-5f, 7e-6f, 0.0019f, 0.001f, 440.f, 448.f, 460.f, 464.f, 480.f, 1000.f,
                        50000.f, 57344.f, 60000.f, 61440.f, 65536.f, 1e6f, -1e6f, INFINITY, NAN};
    const int n = sizeof xs / sizeof xs[0];
    float* dx; uint8_t *d5, *d4;
    hipMalloc(&dx, sizeof xs); hipMalloc(&d5, n); hipMalloc(&d4, n);
    hipMemcpy(dx, xs, sizeof xs, hipMemcpyHostToDevice);
    cvt_probe<<<1, 64>>>(dx, d5, d4, n);
    uint8_t h5[64], h4[64];
    hipMemcpy(h5, d5, n, hipMemcpyDeviceToHost); hipMemcpy(h4, d4, n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("x = %-12g  e5m2 0x%02x = %-12g  e4m3 0x%02x = %-12g\n", xs[i], h5[i], dec_e5m2(h5[i]), h4[i], dec_e4m3(h4[i]));
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad;
}
