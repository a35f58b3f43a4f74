// conv_f32.hip -- exactness mode (VTGB_F32) of launch_conv_gemm: the implicit-GEMM convolution / plain GEMM
// that RAFT's encoders and update block run on, with fp32 operands, fp32 FMAs and k summed in order
// (the reference runs RAFT in fp32: src/models/components/xraft.py:118-119).  Same GemmDesc, same NHWC
// activation layout, same K order (64-channel chunk major, tap minor) and the same epilogues as the bf16
// MFMA kernel of gemm.hip, so raft.hip / raft_enc.hip drive both modes with one launch sequence; every
// buffer the bf16 mode keeps in bf16 is fp32 here (GemmDesc::aux, out2, resid_bf16 point at floats).
// 64 x 64 output tile, 16-deep k-slabs through LDS, 4 x 4 outputs per thread.
#include <string.h>

#include "common.h"

enum { F_EPI_STORE = VTGB_EPI_STORE, F_EPI_STORE_F32 = VTGB_EPI_STORE_F32, F_EPI_GRU = VTGB_EPI_GRU };

__device__ __forceinline__ float f32_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <bool CONV, bool GRU>
__global__ __launch_bounds__(256) void conv_f32_kernel(const GemmDesc p) {
    __shared__ float As[16][68];
    __shared__ float Ws[16][68];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
    const float* __restrict__ A2 = reinterpret_cast<const float*>(p.A2);
    const float* __restrict__ W = reinterpret_cast<const float*>(p.W);
    const float* __restrict__ pad = reinterpret_cast<const float*>(p.zero_page);
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    int am = m0 + lrow; am = am < p.M ? am : p.M - 1;
    int wr = n0 + lrow; wr = wr < p.N ? wr : p.N - 1;
    const float* wp = W + (int64_t)wr * p.ldw;
    // staged row: plain GEMM -> its base pointer; convolution -> image base pixel and output (y, x)
    const float* ap = nullptr;
    int img_px = 0, oy = 0, ox = 0;
    const int taps = CONV ? p.conv_KH * p.conv_KW : 1;
    const int cst = p.conv_stride ? p.conv_stride : 1;
    const int Hi = p.conv_Hi ? p.conv_Hi : p.conv_H, Wi = p.conv_Wi ? p.conv_Wi : p.conv_W;
    if constexpr (CONV) {
        const int hw = p.conv_H * p.conv_W, img = am / hw, rem = am - img * hw;
        img_px = img * Hi * Wi;
        oy = rem / p.conv_W;
        ox = rem - oy * p.conv_W;
    } else {
        ap = A + map_row(p.a_map, am) * p.lda;
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const int k = k0 + lk;
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), wv = av;
        if (k < p.K) {
            wv = *reinterpret_cast<const float4*>(wp + k);
            if constexpr (CONV) {
                const int blk = k >> 6, c = k & 63, chunk = blk / taps, tap = blk - chunk * taps;
                const int dy = tap / p.conv_KW - (p.conv_KH >> 1), dx = tap % p.conv_KW - (p.conv_KW >> 1);
                const int y = oy * cst + dy, x = ox * cst + dx;
                int ch = chunk * 64 + c;
                const bool first = ch < p.conv_split;
                const float* base = first ? A : A2;
                const int64_t ld = first ? p.lda : p.lda2;
                if (!first) ch -= p.conv_split;
                if ((unsigned)y < (unsigned)Hi && (unsigned)x < (unsigned)Wi)
                    av = *reinterpret_cast<const float4*>(base + (int64_t)(img_px + y * Wi + x) * ld + ch);
                else
                    av = *reinterpret_cast<const float4*>(pad + c);
            } else {
                av = *reinterpret_cast<const float4*>(ap + k);
            }
        }
        As[lk + 0][lrow] = av.x; As[lk + 1][lrow] = av.y; As[lk + 2][lrow] = av.z; As[lk + 3][lrow] = av.w;
        Ws[lk + 0][lrow] = wv.x; Ws[lk + 1][lrow] = wv.y; Ws[lk + 2][lrow] = wv.z; Ws[lk + 3][lrow] = wv.w;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const float4 a = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            const float4 w = *reinterpret_cast<const float4*>(&Ws[kk][tx * 4]);
            const float a4[4] = {a.x, a.y, a.z, a.w}, w4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a4[i], w4[j], acc[i][j]);
        }
        __syncthreads();
    }
    // ---- epilogue: 4 rows x 4 consecutive columns per thread
    const int nb = n0 + tx * 4;
    if (nb >= p.N) return;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (nb + j < p.N) bias[j] = p.bias[nb + j];
    }
    const float sc = p.out_scale != 0.f ? p.out_scale : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
        const int64_t orow = map_row(p.o_map, m);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = nb + j;
            if (n >= p.N) continue;
            float v = acc[i][j] + bias[j];
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
    const dim3 grid((unsigned)((d.M + 63) / 64), (unsigned)((d.N + 63) / 64));
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
