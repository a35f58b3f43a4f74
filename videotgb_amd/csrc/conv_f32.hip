// conv_f32.hip -- exactness mode (VTGB_F32) of launch_conv_gemm: the implicit-GEMM convolution / plain GEMM
// that RAFT's encoders and update block run on, with fp32 operands, fp32 FMAs and k summed in order
// (the reference runs RAFT in fp32: src/models/components/xraft.py:118-119).  Same GemmDesc, same NHWC
// activation layout, same K order (64-channel chunk major, tap minor) and the same epilogues as the bf16
// MFMA kernel of gemm.hip, so raft.hip / raft_enc.hip drive both modes with one launch sequence; every
// buffer the bf16 mode keeps in bf16 is fp32 here (GemmDesc::aux, out2, resid_bf16 point at floats).
// 128 x 64 output tile, 16-deep k-slabs through LDS, fp32-input MFMA (r5; rounds 2-4: 4 x 4 outputs per thread, scalar FMAs).
#include <string.h>

#include "common.h"

enum { F_EPI_STORE = VTGB_EPI_STORE, F_EPI_STORE_F32 = VTGB_EPI_STORE_F32, F_EPI_GRU = VTGB_EPI_GRU };

__device__ __forceinline__ float f32_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}

// r5: the products run on the fp32-input matrix instruction v_mfma_f32_32x32x2_f32 -- bit for bit a k-ordered fmaf chain (one rounding per
// product, no wider accumulation: cdna_hip_programming.md, "FP32-input MFMA"), i.e. the SAME numbers as the scalar-FMA loop of rounds 2-4, at
// the packed-FMA rate instead of the plain-FMA rate that loop was pinned to (78 TFLOP/s: RAFT's exactness mode ran at 81).
// Workgroup tile 128 pixels x 64 channels, 16-deep k-slabs through LDS (k-major), four waves of 64 x 32: the weight slab is the A operand
// (rows = output channels), the activation slab the B operand, so a lane ends up with four runs of four consecutive channels of ONE pixel;
// the next slab's global loads are in flight while the current one is multiplied.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int CF_BM = 128, CF_BN = 64;

template <bool CONV, bool GRU>
__global__ __launch_bounds__(256) void conv_f32_kernel(const GemmDesc p) {
    __shared__ float As[16][CF_BM + 4];
    __shared__ float Ws[16][CF_BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;                      // wave tile: pixels [wm * 64, +64), channels [wn * 32, +32)
    const int m0 = blockIdx.x * CF_BM, n0 = blockIdx.y * CF_BN;
    const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
    const float* __restrict__ A2 = reinterpret_cast<const float*>(p.A2);
    const float* __restrict__ W = reinterpret_cast<const float*>(p.W);
    const float* __restrict__ pad = reinterpret_cast<const float*>(p.zero_page);
    const int lrow = tid >> 2, lk = (tid & 3) * 4;              // staged rows lrow, lrow + 64 (activations) / lrow (weights); k offset lk
    int wr = n0 + lrow; wr = wr < p.N ? wr : p.N - 1;
    const float* wp = W + (int64_t)wr * p.ldw;
    const float* ap[2] = {nullptr, nullptr};
    int img_px[2] = {0, 0}, oy[2] = {0, 0}, ox[2] = {0, 0};
    const int taps = CONV ? p.conv_KH * p.conv_KW : 1;
    const int cst = p.conv_stride ? p.conv_stride : 1;
    const int Hi = p.conv_Hi ? p.conv_Hi : p.conv_H, Wi = p.conv_Wi ? p.conv_Wi : p.conv_W;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        int am = m0 + lrow + h * 64; am = am < p.M ? am : p.M - 1;
        if constexpr (CONV) {
            const int hw = p.conv_H * p.conv_W, img = am / hw, rem = am - img * hw;
            img_px[h] = img * Hi * Wi;
            oy[h] = rem / p.conv_W;
            ox[h] = rem - oy[h] * p.conv_W;
        } else {
            ap[h] = A + map_row(p.a_map, am) * p.lda;
        }
    }
    auto fetch = [&](int k0, float4 (&av)[2], float4& wv) {
        const int k = k0 + lk;
        av[0] = av[1] = wv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < p.K) {
            wv = *reinterpret_cast<const float4*>(wp + k);
            if constexpr (CONV) {
                const int blk = k >> 6, c = k & 63, chunk = blk / taps, tap = blk - chunk * taps;
                const int dy = tap / p.conv_KW - (p.conv_KH >> 1), dx = tap % p.conv_KW - (p.conv_KW >> 1);
                int ch = chunk * 64 + c;
                const bool first = ch < p.conv_split;
                const float* base = first ? A : A2;
                const int64_t ld = first ? p.lda : p.lda2;
                if (!first) ch -= p.conv_split;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int y = oy[h] * cst + dy, x = ox[h] * cst + dx;
                    if ((unsigned)y < (unsigned)Hi && (unsigned)x < (unsigned)Wi)
                        av[h] = *reinterpret_cast<const float4*>(base + (int64_t)(img_px[h] + y * Wi + x) * ld + ch);
                    else
                        av[h] = *reinterpret_cast<const float4*>(pad + c);
                }
            } else {
                av[0] = *reinterpret_cast<const float4*>(ap[0] + k);
                av[1] = *reinterpret_cast<const float4*>(ap[1] + k);
            }
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[j][e] = 0.f;
    float4 av[2], wv;
    fetch(0, av, wv);
    const int l31 = lane & 31, kh = lane >> 5;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            As[lk + 0][lrow + h * 64] = av[h].x; As[lk + 1][lrow + h * 64] = av[h].y; As[lk + 2][lrow + h * 64] = av[h].z; As[lk + 3][lrow + h * 64] = av[h].w;
        }
        Ws[lk + 0][lrow] = wv.x; Ws[lk + 1][lrow] = wv.y; Ws[lk + 2][lrow] = wv.z; Ws[lk + 3][lrow] = wv.w;
        __syncthreads();
        if (k0 + 16 < p.K) fetch(k0 + 16, av, wv);                // the next slab's loads fly under this slab's products
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {                           // k = k0 + 2 kk + {0, 1}: ascending inside the instruction and across them
            const float wf = Ws[2 * kk + kh][wn * 32 + l31];
#pragma unroll
            for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, As[2 * kk + kh][wm * 64 + j * 32 + l31], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue.  D[i][j]: column j = lane & 31 -> pixel, rows i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> channel
    const float sc = p.out_scale != 0.f ? p.out_scale : 1.0f;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int m = m0 + wm * 64 + j * 32 + l31;
        if (m >= p.M) continue;
        const int64_t orow = map_row(p.o_map, m);
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int n = n0 + wn * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
            if (n >= p.N) continue;
            float v = acc[j][reg] + (p.bias ? p.bias[n] : 0.f);
            if constexpr (GRU) {
                // h' = (1 - z) h + z tanh(acc + bias) (update.py:57-58, :64-65)
                const float h = p.resid[map_row(p.r_map, m) * p.ldr + n];
                const float z = reinterpret_cast<const float*>(p.aux)[(int64_t)m * p.ldaux + n];
                const float hn = (1.0f - z) * h + z * tanhf(v);
                reinterpret_cast<float*>(p.out)[orow * p.ldo + n] = hn;
                reinterpret_cast<float*>(p.out2)[(int64_t)m * p.ldo2 + n] = hn;
            } else {
                v = f32_act(v, p.act) * sc;
                if (p.gate_from > 0 && n >= p.gate_from) {
                    const int ng = n - p.gate_from;
                    reinterpret_cast<float*>(p.out2)[(int64_t)m * p.ldo2 + ng] = v * reinterpret_cast<const float*>(p.aux)[(int64_t)m * p.ldaux + ng];
                } else {
                    if (p.resid_bf16) {
                        v += reinterpret_cast<const float*>(p.resid_bf16)[(int64_t)m * p.ldrb + n];
                        if (p.post_relu) v = fmaxf(v, 0.f);
                    }
                    reinterpret_cast<float*>(p.out)[orow * p.ldo + n] = v;
                }
            }
        }
    }
}

int launch_conv_f32(const GemmDesc& d, hipStream_t s) {
    const bool conv = d.conv_KH > 0;
    VTGB_REQUIRE((d.K % 4) == 0 && (d.lda % 4) == 0 && (d.ldw % 4) == 0 && (!d.A2 || (d.lda2 % 4) == 0), VTGB_EUNSUPPORTED,
                 "conv gemm fp32: K=%d and the row strides must be multiples of 4", d.K);
    VTGB_REQUIRE(!d.col_stats, VTGB_EUNSUPPORTED, "conv gemm fp32: column statistics are taken by the separate pass in this mode");
    const dim3 grid((unsigned)((d.M + CF_BM - 1) / CF_BM), (unsigned)((d.N + CF_BN - 1) / CF_BN));
    ProfScope prof(conv ? VTGB_PROF_CONV : VTGB_PROF_GEMM, 2.0 * d.M * d.N * d.K, s);
    if (d.epi == F_EPI_GRU) {
        VTGB_REQUIRE(conv && d.resid && d.aux && d.out2, VTGB_EINVAL, "conv gemm: GRU epilogue needs h, z and both outputs");
        hipLaunchKernelGGL((conv_f32_kernel<true, true>), grid, dim3(256), 0, s, d);
    } else if (d.epi == F_EPI_STORE || d.epi == F_EPI_STORE_F32) {
        if (conv)
            hipLaunchKernelGGL((conv_f32_kernel<true, false>), grid, dim3(256), 0, s, d);
        else
            hipLaunchKernelGGL((conv_f32_kernel<false, false>), grid, dim3(256), 0, s, d);
    } else {
        vtgb_set_error("conv gemm fp32: unsupported epilogue %d", d.epi);
        return VTGB_EINVAL;
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
