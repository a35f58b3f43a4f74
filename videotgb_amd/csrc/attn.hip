// attn.hip -- softmax(scale * Q K^T + key_mask) V for gfx950, token-major operands.
//
// One kernel serves the ViT (16 heads x 88, S = 257, no mask), the Q-Former (12 x 64; self
// S = 32+Lt with the -10000 pad mask, cross 32 x 257) and the TGB (12 x 64, S = L+2, rotary
// Q/K, self / cross masks).  Sequences are short (<= 512 keys), so the whole K and V of one
// (batch, head) live in LDS and the softmax is single pass (no online rescaling).
//
// bf16 path.  A workgroup (8 waves, two per SIMD so one wave's softmax VALU work overlaps the
// other's MFMAs; 4 waves for the 512-key instantiation whose score tile needs > 256 registers)
// owns one (batch, head [, q-split]); its waves walk 16-query tiles.
// The products are computed transposed with v_mfma_f32_16x16x32_bf16 so that no LDS round
// trip is needed between them:
//   S^T[key][q]  = K (A operand: rows = keys, LDS [key][d])  x  Q^T (B operand from global)
//   O^T[d][q]    = V^T (A operand: rows = d,  LDS [d][key])  x  P^T (B operand = S^T accum.)
// The S^T accumulator of two 16-key tiles (lane: column q, rows 4*(lane>>4)+r) is packed to
// bf16 and used directly as the B fragment of a 32-key step; the matching A fragment takes
// its keys in the same permuted order (two 8-byte LDS reads instead of one 16-byte read).
// K rows are padded to HD*2+16 bytes (the 16 rows of a fragment read land on 16 distinct 16-byte bank groups).  V stays
// ROW-MAJOR [key][d] (r3; rounds 1-2 transposed it while staging: eight 4-byte LDS writes per 16 bytes loaded, a third of the
// kernel) and the V^T fragments are read with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group fetches a
// 4-key x 16-d block and each lane receives one d column of it.  V rows are padded to HD*2+32 bytes: the 8 rows a 32-lane
// half touches start 8 banks apart modulo 64.
// Rotary embedding (interleaved pairs, sin|cos table rows) is applied while staging K and
// loading Q.  fp32 path: one wave per query row, plain FMAs (exactness mode).
#include "common.h"

#include <math.h>

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

template <int HD, int NKP>
struct AttnCfg {
    static constexpr int NK = NKP * 32;       // padded key count
    // K row stride (bytes) and piece placement.  HD > 64 (r4): 256-byte rows, 16-byte piece c of row r at slot c ^ (r & 15) -- a
    // ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} (rows 0-3, 12-15 at piece c, rows 4-11 at piece c + 1), not in runs of 16
    // lanes: with the padded stride HD * 2 + 16 of rounds 1-3 five row pairs of a group shared their banks (SQ_LDS_BANK_CONFLICT = a
    // third of the kernel's LDS cycles, profiles/r04_pmc_attn.txt); with the XOR every group touches 16 different slots.
    static constexpr bool KSWZ = HD > 64;
    static constexpr int KS = KSWZ ? 256 : HD * 2 + 16;
    static __device__ __forceinline__ int koff(int row, int piece) { return row * KS + ((KSWZ ? (piece ^ (row & 15)) : piece) << 4); }
    static constexpr int VS = HD * 2 + 32;    // V row stride (bytes): 8 consecutive rows x 32 bytes land on 64 distinct banks
    static constexpr int K_BYTES = NK * KS;
    static constexpr int V_BYTES = NK * VS;
    static constexpr int LDS = K_BYTES + V_BYTES + NK * 4;
};

// rotate 8 consecutive elements (4 interleaved pairs) starting at even d0: x*cos + rot(x)*sin
__device__ __forceinline__ bf16x8 rope8(bf16x8 v, const float* tab_row, int d0, int head_dim) {
    const int half = head_dim >> 1;
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float sn = tab_row[(d0 >> 1) + i], cs = tab_row[half + (d0 >> 1) + i];
        const float x0 = (float)v[2 * i], x1 = (float)v[2 * i + 1];
        o[2 * i] = (bf16_t)(x0 * cs - x1 * sn);
        o[2 * i + 1] = (bf16_t)(x1 * cs + x0 * sn);
    }
    return o;
}

// PLAIN: no key mask, not causal (the softmax does less per score).  TAILONLY (PLAIN only): s_kv > NK - 32, i.e. only the last key pair can be
// partial -- a launch-time fact, a template parameter so that the two forms of the padded-key selects do not meet in one function (as a
// run-time branch they cost 63 register copies per query tile at the join).
template <int HD, int NKP, bool PLAIN, bool TAILONLY = false>
__global__ __launch_bounds__(NKP <= 9 ? 576 : 256) void attn_bf16_kernel(const AttnDesc p, const int tiles_per_split) {
    using C = AttnCfg<HD, NKP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + C::K_BYTES;
    float* maskv = reinterpret_cast<float*>(smem + C::K_BYTES + C::V_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int head = blockIdx.y, b = blockIdx.z;
    const int hd = p.head_dim;
    const bf16_t* __restrict__ Q = reinterpret_cast<const bf16_t*>(p.q) + (int64_t)b * p.q_batch + head * hd;
    const bf16_t* __restrict__ K = reinterpret_cast<const bf16_t*>(p.k) + (int64_t)b * p.kv_batch + head * hd;
    const bf16_t* __restrict__ V = reinterpret_cast<const bf16_t*>(p.v) + (int64_t)b * p.kv_batch + head * hd;
    bf16_t* __restrict__ O = reinterpret_cast<bf16_t*>(p.out) + (int64_t)b * p.o_batch + head * hd;
    constexpr int CH = HD / 8;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- stage K [key][d] (rope applied, zero padded) and V^T [d][key].  All global loads of a thread are issued
    // before the first LDS write: the naive load -> write loop serialised ~14 dependent HBM round trips per thread
    // (about two thirds of the kernel at 257 keys); the whole K / V of a head is 90 KB, i.e. <= 8 + 8 loads per thread.
    constexpr int TMIN = NKP <= 9 ? 512 : 256;                                           // workgroup size of the full-size launches
    constexpr int KI = (C::NK * CH + TMIN - 1) / TMIN;
    if (nthreads >= TMIN) {
        bf16x8 kreg[KI], vreg[KI];
#pragma unroll
        for (int i = 0; i < KI; i++) {
            const int idx = tid + i * nthreads, key = idx / CH, c = idx - key * CH;
            kreg[i] = zero8; vreg[i] = zero8;
            if (idx < C::NK * CH && key < p.s_kv && c * 8 < hd) {
                kreg[i] = *reinterpret_cast<const bf16x8*>(K + (int64_t)key * p.kv_tok + c * 8);
                vreg[i] = *reinterpret_cast<const bf16x8*>(V + (int64_t)key * p.kv_tok + c * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < KI; i++) {
            const int idx = tid + i * nthreads, key = idx / CH, c = idx - key * CH;
            if (idx < C::NK * CH) {
                bf16x8 val = kreg[i];
                if (p.rope_k && key < p.s_kv && c * 8 < hd) val = rope8(val, p.rope_k + (int64_t)key * hd, c * 8, hd);
                *reinterpret_cast<bf16x8*>(Ks + C::koff(key, c)) = val;
                *reinterpret_cast<bf16x8*>(Vs + key * C::VS + c * 16) = vreg[i];
            }
        }
    } else {
        for (int idx = tid; idx < C::NK * CH; idx += nthreads) {
            const int key = idx / CH, c = idx - key * CH;
            bf16x8 val = zero8, vv = zero8;
            if (key < p.s_kv && c * 8 < hd) {
                val = *reinterpret_cast<const bf16x8*>(K + (int64_t)key * p.kv_tok + c * 8);
                vv = *reinterpret_cast<const bf16x8*>(V + (int64_t)key * p.kv_tok + c * 8);
                if (p.rope_k) val = rope8(val, p.rope_k + (int64_t)key * hd, c * 8, hd);
            }
            *reinterpret_cast<bf16x8*>(Ks + C::koff(key, c)) = val;
            *reinterpret_cast<bf16x8*>(Vs + key * C::VS + c * 16) = vv;
        }
    }
    for (int key = tid; key < C::NK; key += nthreads) {
        float m = -INFINITY;
        if (key < p.s_kv) m = p.key_mask ? p.key_mask[(int64_t)b * p.s_kv + key] : 0.f;
        maskv[key] = m;
    }
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    const int n_qt = (p.s_q + 15) >> 4;
    const int qt_begin = blockIdx.x * tiles_per_split;
    int qt_end = qt_begin + tiles_per_split;
    qt_end = qt_end < n_qt ? qt_end : n_qt;

    for (int qt = qt_begin + wave; qt < qt_end; qt += nwaves) {
        const int q = qt * 16 + fr;
        const bool qvalid = q < p.s_q;
        bf16x8 qf[HD / 32];
#pragma unroll
        for (int ks = 0; ks < HD / 32; ks++) {
            const int d0 = (ks * 4 + fg) * 8;
            bf16x8 val = zero8;
            if (qvalid && d0 < hd) {
                val = *reinterpret_cast<const bf16x8*>(Q + (int64_t)q * p.q_tok + d0);
                if (p.rope_q) val = rope8(val, p.rope_q + (int64_t)q * hd, d0, hd);
            }
            qf[ks] = val;
        }
        // ---- S^T tiles, two 16-key tiles (one 32-key pair) per step; the next pair's K fragments
        // are issued before the current pair's MFMAs, and sched_barrier keeps the compiler from
        // hoisting every LDS read of the unrolled loop to the top (which spills).
        f32x4 s[2 * NKP];
        bf16x8 kcur[2][HD / 32], knxt[2][HD / 32];
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int ks = 0; ks < HD / 32; ks++)
                kcur[tt][ks] = *reinterpret_cast<const bf16x8*>(Ks + C::koff(tt * 16 + fr, ks * 4 + fg));
#pragma unroll
        for (int u = 0; u < NKP; u++) {
            if (u + 1 < NKP) {
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int ks = 0; ks < HD / 32; ks++)
                        knxt[tt][ks] = *reinterpret_cast<const bf16x8*>(Ks + C::koff((2 * u + 2 + tt) * 16 + fr, ks * 4 + fg));
            }
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < HD / 32; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcur[tt][ks], qf[ks], acc, 0, 0, 0);
                s[2 * u + tt] = acc;
            }
            if (u + 1 < NKP) {
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int ks = 0; ks < HD / 32; ks++) kcur[tt][ks] = knxt[tt][ks];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- softmax over keys: registers (t, r) x lane groups fg
        float mx = -INFINITY;
        float sum = 0.f;
        if constexpr (PLAIN) {
            // no mask, not causal (the ViT: 7 936 heads x 17 tiles per call).  The loop is bound by the SIMD's VALU + MFMA issue, not by LDS
            // (profiles/r03_attention_experiments.md): per score one max, one FMA, one v_exp_f32 (2^x: scale * log2(e) folded into the
            // FMA) and one add instead of scale + mask, two causal selects, max, subtract, multiply, exp, add.  Only the tiles that
            // reach past s_kv (wave-uniform test) mask their padded keys.  (scale > 0: max and scale commute.)
            const float c = p.scale * 1.4426950408889634f;
            // r4: hipcc turned the "tile reaches past s_kv" test into selects for ALL 72 scores of a lane and kept the 72 lane masks in SGPR
            // pairs -- 131 of them spilled and came back through v_readlane: 144 + 131 of the ~800 vector-slot instructions of a query tile.
            // When only the last key pair can be partial (s_kv > NK - 32: the ViT's 257 of 288) the selects exist for those two tiles only.
            if constexpr (TAILONLY) {
#pragma unroll
                for (int t = 0; t < 2 * NKP; t++) {
                    if (t >= 2 * NKP - 2) {
#pragma unroll
                        for (int r = 0; r < 4; r++) s[t][r] = t * 16 + fg * 4 + r < p.s_kv ? s[t][r] : -INFINITY;
                    }
                    mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2 * NKP; t++) {
                    if (t * 16 + 16 > p.s_kv) {
#pragma unroll
                        for (int r = 0; r < 4; r++) s[t][r] = t * 16 + fg * 4 + r < p.s_kv ? s[t][r] : -INFINITY;
                    }
                    mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mc = -mx * c;
#pragma unroll
            for (int t = 0; t < 2 * NKP; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[t][r], c, mc));
                    s[t][r] = e;
                    sum += e;
                }
            }
        } else {
        const int klim = p.causal ? q + (p.s_kv - p.s_q) : 0x7fffffff;      // causal: query q sees keys <= q + (s_kv - s_q)
#pragma unroll
        for (int t = 0; t < 2 * NKP; t++) {
            const float4 mk = *reinterpret_cast<const float4*>(maskv + t * 16 + fg * 4);
            const int k0 = t * 16 + fg * 4;
            s[t][0] = k0 <= klim ? s[t][0] * p.scale + mk.x : -INFINITY;
            s[t][1] = k0 + 1 <= klim ? s[t][1] * p.scale + mk.y : -INFINITY;
            s[t][2] = k0 + 2 <= klim ? s[t][2] * p.scale + mk.z : -INFINITY;
            s[t][3] = k0 + 3 <= klim ? s[t][3] * p.scale + mk.w : -INFINITY;
            mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
        for (int t = 0; t < 2 * NKP; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float e = __expf(s[t][r] - mx);
                s[t][r] = e;
                sum += e;
            }
        }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        // ---- O^T = V^T P^T, one 32-key pair per step, V^T fragments prefetched one step ahead
        f32x4 o[HD / 16];
        bf16x8 vcur[HD / 16], vnxt[HD / 16];
        // A fragment of V^T (rows d, k = keys in the permuted order of pf below) straight from the row-major V image: lane 4 q + p of a
        // 16-lane group addresses row (key) q, columns (d) 4 p ... of a 4-key x 16-d block and receives the block's column fr
        const char* const vbase = Vs + (fg * 4 + (fr >> 2)) * C::VS + (fr & 3) * 8;
#define ATT_VFRAG(dst, dt_, u_)                                                                                          \
        {                                                                                                                \
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(vbase + ((u_) * 32) * C::VS + (dt_) * 32));        \
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(vbase + ((u_) * 32 + 16) * C::VS + (dt_) * 32));   \
            dst = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                        \
        }
#pragma unroll
        for (int dt = 0; dt < HD / 16; dt++) {
            o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            ATT_VFRAG(vcur[dt], dt, 0)
        }
#pragma unroll
        for (int u = 0; u < NKP; u++) {
            if (u + 1 < NKP) {
#pragma unroll
                for (int dt = 0; dt < HD / 16; dt++) ATT_VFRAG(vnxt[dt], dt, u + 1)
            }
            const bf16x8 pf = {(bf16_t)s[2 * u][0],     (bf16_t)s[2 * u][1],     (bf16_t)s[2 * u][2],     (bf16_t)s[2 * u][3],
                               (bf16_t)s[2 * u + 1][0], (bf16_t)s[2 * u + 1][1], (bf16_t)s[2 * u + 1][2], (bf16_t)s[2 * u + 1][3]};
#pragma unroll
            for (int dt = 0; dt < HD / 16; dt++) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vcur[dt], pf, o[dt], 0, 0, 0);
            if (u + 1 < NKP) {
#pragma unroll
                for (int dt = 0; dt < HD / 16; dt++) vcur[dt] = vnxt[dt];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef ATT_VFRAG
        const float inv = 1.0f / sum;
        if (qvalid) {
#pragma unroll
            for (int dt = 0; dt < HD / 16; dt++) {
                const int d = dt * 16 + fg * 4;
                if (d < hd) {
                    const bf16x4 pk = {(bf16_t)(o[dt][0] * inv), (bf16_t)(o[dt][1] * inv), (bf16_t)(o[dt][2] * inv),
                                       (bf16_t)(o[dt][3] * inv)};
                    *reinterpret_cast<bf16x4*>(O + (int64_t)q * p.o_tok + d) = pk;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// fp32: one wave per (batch, head, query)
// ---------------------------------------------------------------------------------------
constexpr int F32_MAX_KV = 1024, F32_MAX_HD = 128;

__device__ __forceinline__ float rope_elem(const float* x, int d, const float* tab_row, int head_dim) {
    const int half = head_dim >> 1, i = d >> 1;
    const float sn = tab_row[i], cs = tab_row[half + i];
    return (d & 1) ? (x[d] * cs + x[d - 1] * sn) : (x[d] * cs - x[d + 1] * sn);
}

__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnDesc p) {
    __shared__ float qs[4][F32_MAX_HD];
    __shared__ float sc[4][F32_MAX_KV];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t total = (int64_t)p.batch * p.heads * p.s_q;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    const bool active = row < total;
    if (!active) row = total - 1;   // keep every wave at the barriers; stores are predicated
    const int q = row % p.s_q, head = (row / p.s_q) % p.heads, b = row / ((int64_t)p.s_q * p.heads);
    const int hd = p.head_dim;
    const float* Q = reinterpret_cast<const float*>(p.q) + (int64_t)b * p.q_batch + (int64_t)q * p.q_tok + head * hd;
    const float* K = reinterpret_cast<const float*>(p.k) + (int64_t)b * p.kv_batch + head * hd;
    const float* V = reinterpret_cast<const float*>(p.v) + (int64_t)b * p.kv_batch + head * hd;
    for (int d = lane; d < hd; d += 64) qs[wave][d] = p.rope_q ? rope_elem(Q, d, p.rope_q + (int64_t)q * hd, hd) : Q[d];
    __syncthreads();
    float mx = -INFINITY;
    for (int key = lane; key < p.s_kv; key += 64) {
        const float* kr = K + (int64_t)key * p.kv_tok;
        float dot = 0.f;
        if (p.rope_k) {
            const float* tr = p.rope_k + (int64_t)key * hd;
            for (int d = 0; d < hd; d++) dot = fmaf(qs[wave][d], rope_elem(kr, d, tr, hd), dot);
        } else {
            for (int d = 0; d < hd; d++) dot = fmaf(qs[wave][d], kr[d], dot);
        }
        float v = dot * p.scale;
        if (p.key_mask) v += p.key_mask[(int64_t)b * p.s_kv + key];
        if (p.causal && key > q + (p.s_kv - p.s_q)) v = -INFINITY;
        sc[wave][key] = v;
        mx = fmaxf(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int key = lane; key < p.s_kv; key += 64) {
        const float e = expf(sc[wave][key] - mx);
        sc[wave][key] = e;
        sum += e;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    __syncthreads();
    const float inv = 1.0f / sum;
    float* O = reinterpret_cast<float*>(p.out) + (int64_t)b * p.o_batch + (int64_t)q * p.o_tok + head * hd;
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int key = 0; key < p.s_kv; key++) acc = fmaf(sc[wave][key] * inv, V[(int64_t)key * p.kv_tok + d], acc);
        if (active) O[d] = acc;
    }
}

template <int HD, int NKP, bool PLAIN, bool TAILONLY = false>
static int launch_bf16_v(const AttnDesc& d, hipStream_t s) {
    using C = AttnCfg<HD, NKP>;
    if constexpr (PLAIN && !TAILONLY) {
        if (d.s_kv > (2 * NKP - 2) * 16) return launch_bf16_v<HD, NKP, true, true>(d, s);
    }
    static DeviceOnce attr_set;
    VTGB_FUNC_LDS_ONCE(attr_set, (attn_bf16_kernel<HD, NKP, PLAIN, TAILONLY>), C::LDS);
    const int n_qt = (d.s_q + 15) / 16;
    int splits = 1;
    while ((int64_t)d.batch * d.heads * splits < 256 && splits * 2 <= n_qt && splits < 4) splits *= 2;
    const int tps = (n_qt + splits - 1) / splits;
    // waves per workgroup: the fewest rounds over the 16-query tiles with at most 9 (NKP <= 9) waves, then the fewest
    // waves for that many rounds (257 tokens = 17 tiles: 2 rounds of 9 waves instead of 3 rounds of 8)
    const int max_waves = NKP <= 9 ? 9 : 4;
    const int rounds = (tps + max_waves - 1) / max_waves;
    const int waves = (tps + rounds - 1) / rounds;
    ProfScope prof(VTGB_PROF_ATTN, 4.0 * d.batch * d.heads * (double)d.s_q * d.s_kv * d.head_dim, s);
    hipLaunchKernelGGL((attn_bf16_kernel<HD, NKP, PLAIN, TAILONLY>), dim3(splits, d.heads, d.batch), dim3(64 * waves), C::LDS, s, d, tps);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

template <int HD, int NKP>
static int launch_bf16(const AttnDesc& d, hipStream_t s) {
    return (!d.key_mask && !d.causal && d.scale > 0.f) ? launch_bf16_v<HD, NKP, true>(d, s) : launch_bf16_v<HD, NKP, false>(d, s);
}

int launch_attention(const AttnDesc& d, hipStream_t s) {
    VTGB_REQUIRE(d.q && d.k && d.v && d.out, VTGB_EINVAL, "attention: NULL operand");
    VTGB_REQUIRE(d.batch > 0 && d.heads > 0 && d.s_q > 0 && d.s_kv > 0, VTGB_EINVAL, "attention: empty problem");
    if (d.dtype == VTGB_F32) {
        VTGB_REQUIRE(d.s_kv <= F32_MAX_KV && d.head_dim <= F32_MAX_HD && (d.head_dim % 2) == 0, VTGB_EUNSUPPORTED,
                     "attention fp32: s_kv=%d head_dim=%d outside [<=%d, <=%d even]", d.s_kv, d.head_dim, F32_MAX_KV, F32_MAX_HD);
        const int64_t rows = (int64_t)d.batch * d.heads * d.s_q;
        hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, d);
        VTGB_HIP(hipGetLastError());
        return VTGB_OK;
    }
    VTGB_REQUIRE(d.dtype == VTGB_BF16, VTGB_EINVAL, "attention: bad dtype %d", d.dtype);
    VTGB_REQUIRE((d.head_dim % 8) == 0 && d.head_dim <= 128, VTGB_EUNSUPPORTED, "attention bf16: head_dim=%d must be a multiple of 8, <= 128", d.head_dim);
    VTGB_REQUIRE((d.q_tok % 8) == 0 && (d.kv_tok % 8) == 0 && (d.o_tok % 4) == 0 && (d.q_batch % 8) == 0 &&
                     (d.kv_batch % 8) == 0 && (d.o_batch % 4) == 0,
                 VTGB_EUNSUPPORTED, "attention bf16: strides must keep 16-byte alignment");
    const int kv = d.s_kv;
    if (d.head_dim <= 64) {
        if (kv <= 64) return launch_bf16<64, 2>(d, s);
        if (kv <= 128) return launch_bf16<64, 4>(d, s);
        if (kv <= 288) return launch_bf16<64, 9>(d, s);
        if (kv <= 512) return launch_bf16<64, 16>(d, s);
    } else if (d.head_dim <= 96) {
        if (kv <= 64) return launch_bf16<96, 2>(d, s);
        if (kv <= 128) return launch_bf16<96, 4>(d, s);
        if (kv <= 288) return launch_bf16<96, 9>(d, s);
    } else {      // 128: the language model's prefill (Llama heads; prefix + prompt <= 288 tokens)
        if (kv <= 64) return launch_bf16<128, 2>(d, s);
        if (kv <= 128) return launch_bf16<128, 4>(d, s);
        if (kv <= 288) return launch_bf16<128, 9>(d, s);
    }
    vtgb_set_error("attention bf16: s_kv=%d with head_dim=%d exceeds the single-pass LDS budget", kv, d.head_dim);
    return VTGB_EUNSUPPORTED;
}
