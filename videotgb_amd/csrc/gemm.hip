// gemm.hip -- out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias) for gfx950.
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out,in]), which
// is the natural MFMA layout: every lane's fragment is 16 contiguous bytes along K.
//
// bf16 path: 128x128x64 tile, 256 threads = 4 waves in 2(M) x 2(N), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 tiles (64 accumulator VGPRs).  W rows are the MFMA "A" operand
// and X rows the "B" operand, so a lane ends up with 4 consecutive n for one m: bias, residual
// and stores are 8/16-byte vectors.  Global -> register -> LDS staging with the next k-tile's
// loads issued before the current tile's MFMAs (one barrier per k-tile, two LDS buffers);
// LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 so the 16
// rows of a fragment read hit 16 distinct 16-byte slots of the 256-byte bank row.
//
// fp32 path (exactness mode): plain 64x64x16 register-tiled FMA kernel, k summed in order.
#include "common.h"

enum { EPI_STORE = VTGB_EPI_STORE, EPI_GELU = VTGB_EPI_GELU, EPI_RESID_F32 = VTGB_EPI_RESID_F32,
       EPI_STORE_F32 = VTGB_EPI_STORE_F32 };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Store 4 consecutive n (n0..n0+3) of logical row m.  TAct is the activation type of
// EPI_STORE / EPI_GELU outputs.
template <int EPI, typename TAct>
__device__ __forceinline__ void epilogue4(const GemmDesc& p, int m, int n0, float v0, float v1, float v2, float v3) {
    if (m >= p.M || n0 >= p.N) return;
    float v[4] = {v0, v1, v2, v3};
    const bool full = (n0 + 3 < p.N);
    if (p.bias) {
        if (full) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + n0);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) v[i] += p.bias[n0 + i];
        }
    }
    const int64_t orow = map_row(p.o_map, m);
    if (EPI == EPI_RESID_F32 || EPI == EPI_STORE_F32) {
        float* o = reinterpret_cast<float*>(p.out) + orow * p.ldo + n0;
        if (EPI == EPI_RESID_F32) {
            const float* r = p.resid + map_row(p.r_map, m) * p.ldr + n0;
            if (full && ((p.ldr & 3) == 0)) {
                const float4 rv = *reinterpret_cast<const float4*>(r);
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            } else {
    #pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) v[i] += r[i];
            }
        }
        if (full && ((p.ldo & 3) == 0)) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) o[i] = v[i];
        }
    } else {
        if (EPI == EPI_GELU) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = gelu_erf(v[i]);
        }
        TAct* o = reinterpret_cast<TAct*>(p.out) + orow * p.ldo + n0;
        if (full && ((p.ldo & 3) == 0)) {
            if constexpr (sizeof(TAct) == 2) {
                bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(o) = pk;
            } else {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < p.N) o[i] = (TAct)v[i];
        }
    }
}

// ---------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ---------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per buffer

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmDesc p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                   // X tile (m rows)   [2][TILE_BYTES]
    char* Ws = smem + 2 * TILE_BYTES;  // W tile (n rows)   [2][TILE_BYTES]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(p.W);

    // staging assignment: 1024 16-byte chunks per operand tile, 4 per thread (named, so nothing
    // is runtime-indexed and everything stays in registers)
#define STAGE_SETUP(i)                                                              \
    const int q##i = tid + 256 * i, row##i = q##i >> 3, c##i = q##i & 7;            \
    const int am##i = (m0 + row##i) < p.M ? (m0 + row##i) : p.M - 1;                \
    const int wr##i = (n0 + row##i) < p.N ? (n0 + row##i) : p.N - 1;                \
    const bf16_t* a_ptr##i = A + map_row(p.a_map, am##i) * p.lda + c##i * 8;        \
    const bf16_t* w_ptr##i = W + (int64_t)wr##i * p.ldw + c##i * 8;                 \
    const int lds_off##i = swz(row##i, c##i);                                       \
    const int kc##i = c##i * 8;
    STAGE_SETUP(0) STAGE_SETUP(1) STAGE_SETUP(2) STAGE_SETUP(3)
#undef STAGE_SETUP
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
#define STAGE_LOAD1(i, RA, RW, k0)                                                        \
    {                                                                                      \
        const bool ok = ((k0) + kc##i) < p.K;   /* out-of-range chunk: read k=0, then zero */ \
        const int kk = ok ? (k0) : -kc##i;                                                 \
        RA = *reinterpret_cast<const uint4*>(a_ptr##i + kk);                               \
        RW = *reinterpret_cast<const uint4*>(w_ptr##i + kk);                               \
        if (!ok) { RA = zero4; RW = zero4; }                                               \
    }
#define STAGE_LOAD(k0) STAGE_LOAD1(0, ra0, rw0, k0) STAGE_LOAD1(1, ra1, rw1, k0) STAGE_LOAD1(2, ra2, rw2, k0) STAGE_LOAD1(3, ra3, rw3, k0)
#define STAGE_WRITE1(i, RA, RW, buf)                                                      \
    *reinterpret_cast<uint4*>(As + (buf) * TILE_BYTES + lds_off##i) = RA;                  \
    *reinterpret_cast<uint4*>(Ws + (buf) * TILE_BYTES + lds_off##i) = RW;
#define STAGE_WRITE(buf) STAGE_WRITE1(0, ra0, rw0, buf) STAGE_WRITE1(1, ra1, rw1, buf) STAGE_WRITE1(2, ra2, rw2, buf) STAGE_WRITE1(3, ra3, rw3, buf)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    STAGE_LOAD(0)
    STAGE_WRITE(0)
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) { STAGE_LOAD((kt + 1) * BK) }
        const char* as = As + buf * TILE_BYTES;
        const char* ws = Ws + buf * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int rw_ = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const bf16x8*>(ws + swz(rw_, ks * 4 + fg));
                const int rx = wm * 64 + i * 16 + fr;
                xf[i] = *reinterpret_cast<const bf16x8*>(as + swz(rx, ks * 4 + fg));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) { STAGE_WRITE(buf ^ 1) }
        __syncthreads();
    }
#undef STAGE_LOAD
#undef STAGE_LOAD1
#undef STAGE_WRITE
#undef STAGE_WRITE1
    // D layout: column (lane & 15) <- X row (m), rows (lane >> 4) * 4 + reg <- W row (n)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = m0 + wm * 64 + j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            epilogue4<EPI, bf16_t>(p, m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// fp32 kernel (exactness mode; also the on-GPU cross-check for the MFMA kernel)
// ---------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmDesc p) {
    __shared__ float As[16][68];
    __shared__ float Ws[16][68];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
    const float* __restrict__ W = reinterpret_cast<const float*>(p.W);
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    int am = m0 + lrow; am = am < p.M ? am : p.M - 1;
    int wr = n0 + lrow; wr = wr < p.N ? wr : p.N - 1;
    const float* ap = A + map_row(p.a_map, am) * p.lda;
    const float* wp = W + (int64_t)wr * p.ldw;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = k0 + lk + i;
            As[lk + i][lrow] = k < p.K ? ap[k] : 0.f;
            Ws[lk + i][lrow] = k < p.K ? wp[k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 w = *reinterpret_cast<const float4*>(&Ws[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        epilogue4<EPI, float>(p, m0 + ty * 4 + i, n0 + tx * 4, acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
}

template <int EPI>
static int launch_epi(const GemmDesc& d, hipStream_t s) {
    if (d.dtype == VTGB_BF16) {
        dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM);
        const size_t lds = 4 * TILE_BYTES;
        static bool attr_set = false;
        if (!attr_set) {
            VTGB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        ProfScope prof(VTGB_PROF_GEMM, 2.0 * d.M * d.N * d.K, s);
        hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, grid, dim3(256), lds, s, d);
    } else {
        dim3 grid((d.N + 63) / 64, (d.M + 63) / 64);
        hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, dim3(256), 0, s, d);
    }
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

int launch_gemm(const GemmDesc& d, hipStream_t s) {
    VTGB_REQUIRE(d.A && d.W && d.out, VTGB_EINVAL, "gemm: NULL operand");
    VTGB_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, VTGB_EINVAL, "gemm: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
    VTGB_REQUIRE(d.dtype == VTGB_BF16 || d.dtype == VTGB_F32, VTGB_EINVAL, "gemm: bad dtype %d", d.dtype);
    if (d.dtype == VTGB_BF16) {
        VTGB_REQUIRE((d.K % 8) == 0 && (d.lda % 8) == 0 && (d.ldw % 8) == 0, VTGB_EUNSUPPORTED,
                     "gemm bf16: K=%d lda=%lld ldw=%lld must be multiples of 8", d.K, (long long)d.lda, (long long)d.ldw);
        VTGB_REQUIRE(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.W % 16) == 0, VTGB_EINVAL, "gemm bf16: operands must be 16-byte aligned");
    }
    if (d.epi == EPI_RESID_F32) VTGB_REQUIRE(d.resid != nullptr, VTGB_EINVAL, "gemm: residual epilogue without resid");
    switch (d.epi) {
        case EPI_STORE: return launch_epi<EPI_STORE>(d, s);
        case EPI_GELU: return launch_epi<EPI_GELU>(d, s);
        case EPI_RESID_F32: return launch_epi<EPI_RESID_F32>(d, s);
        case EPI_STORE_F32: return launch_epi<EPI_STORE_F32>(d, s);
    }
    vtgb_set_error("gemm: bad epilogue %d", d.epi);
    return VTGB_EINVAL;
}
