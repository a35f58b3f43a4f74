// select.hip -- K16/K17/K18: Gumbel top-k span selection, span -> frame index map, frame gather.
// Integer / index work: results are bit-exact against the oracle.  All three stay on the device
// (the reference does ~10 host round trips here, eval/utils/model.py:124-151).
#include "common.h"

// ---------------------------------------------------------------------------------------
// K16.  One wave per (draw, row): first argmax over L of (logit + noise) / tau.
// softmax is monotone non-decreasing, so argmax(softmax(y)) == argmax(y) unless two candidates
// for the maximum collapse to one float after exp/normalise (probability ~1e-6 per draw for
// continuous noise); see DESIGN.md.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void span_select_kernel(const vtgb_span_select_args a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // d * 2B + r
    const int rows = a.draws * 2 * a.B;
    if (row >= rows) return;
    const int r = row % (2 * a.B);
    const int b = r < a.B ? r : r - a.B, which = r < a.B ? 0 : 1;
    const float* lg = a.logits + (int64_t)b * a.L * 2 + which;
    const float* nz = a.noise + row * a.L;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < a.L; i += 64) {
        const float y = __fdiv_rn(__fadd_rn(lg[(int64_t)i * 2], nz[i]), a.tau);
        if (y > best || bi == 0x7fffffff) { best = y; bi = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) a.idx[row] = bi;
}

// ---------------------------------------------------------------------------------------
// K17.  One thread per clip; exact restatement of the reference's host arithmetic
// (float32 / float64 roundings: SURVEY.md Appendix B).
// ---------------------------------------------------------------------------------------
constexpr int MAP_MAX = 512;

__device__ static inline int map_endpoint(int64_t k, int V, int N, int variant, bool python_int) {
    if (python_int) {   // operands are Python ints -> float64
        if (variant == VTGB_MAP_A) return (int)(((double)k / (double)V) * (double)N);
        return (int)((double)(k * (int64_t)(N - 1)) / (double)(V - 1));
    }
    if (variant == VTGB_MAP_A) return (int)__fmul_rn(__fdiv_rn((float)k, (float)V), (float)N);
    return (int)__fdiv_rn((float)(k * (int64_t)(N - 1)), (float)(V - 1));
}

__global__ void span_to_frames_kernel(const vtgb_span_to_frames_args a) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.B) return;
    const int V = a.V ? a.V[j] : a.V_all;
    const int N = a.N, nframe = a.nframe;
    unsigned char in[MAP_MAX];
    int cand[MAP_MAX];
    for (int i = 0; i < N; i++) in[i] = 0;
    for (int ii = 0; ii < a.draws; ii++) {
        int64_t s = a.sel[(int64_t)ii * 2 * a.B + j], e = a.sel[(int64_t)ii * 2 * a.B + a.B + j];
        bool py = false;
        if (s >= V || e >= V || (s == 0 && e == 0)) { s = 0; e = V - 1; py = true; }
        int lo = map_endpoint(s, V, N, a.variant, py), hi = map_endpoint(e, V, N, a.variant, py);
        lo = lo < 0 ? 0 : lo;
        hi = hi > N ? N : hi;
        for (int x = lo; x < hi; x++) in[x] = 1;
    }
    int len = 0;
    for (int i = 0; i < N; i++) if (in[i]) cand[len++] = i;
    if (len == 0) { for (int i = 0; i < N; i++) cand[i] = i; len = N; }
    while (len < nframe) {   // duplicate every element, in place from the back
        for (int i = len - 1; i >= 0; i--) { cand[2 * i] = cand[i]; cand[2 * i + 1] = cand[i]; }
        len *= 2;
    }
    int64_t* out = a.frame_idx + (int64_t)j * nframe;
    if (len > nframe) {
        // np.linspace(0, len, nframe+1).astype(int): i * (len/nframe) in float64, last element = len
        const double step = (double)len / (double)nframe;
        for (int x = 0; x < nframe; x++) {
            const int lo = (int)((double)x * step);
            const int hi = (x + 1 == nframe) ? len : (int)((double)(x + 1) * step);
            out[x] = cand[(lo + hi - 1) / 2];
        }
    } else {
        for (int x = 0; x < nframe; x++) out[x] = cand[x];
    }
}

// ---------------------------------------------------------------------------------------
// K18.  out[b, i, :] = pixel_values[b, frame_idx[b, i], :], 16 bytes per lane.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_frames_kernel(const vtgb_gather_frames_args a) {
    const int i = blockIdx.y, b = blockIdx.z;
    const int64_t src_frame = a.frame_idx[(int64_t)b * a.nframe + i];
    const float4* src = reinterpret_cast<const float4*>(a.pixel_values + ((int64_t)b * a.N + src_frame) * a.frame_elems);
    float4* dst = reinterpret_cast<float4*>(a.out + ((int64_t)b * a.nframe + i) * a.frame_elems);
    const int64_t n4 = a.frame_elems >> 2;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += (int64_t)gridDim.x * blockDim.x) dst[c] = src[c];
}

extern "C" int vtgb_span_select(const vtgb_span_select_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->logits && a->noise && a->idx, VTGB_EINVAL, "span_select: NULL argument");
    VTGB_REQUIRE(a->B > 0 && a->L > 0 && a->draws > 0 && a->tau > 0.f, VTGB_EINVAL, "span_select: B=%d L=%d draws=%d tau=%g", a->B, a->L, a->draws, a->tau);
    const int rows = a->draws * 2 * a->B;
    hipLaunchKernelGGL(span_select_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, *a);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_span_to_frames(const vtgb_span_to_frames_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->sel && a->frame_idx, VTGB_EINVAL, "span_to_frames: NULL argument");
    VTGB_REQUIRE(a->B > 0 && a->draws > 0 && a->N > 0 && a->nframe > 0, VTGB_EINVAL, "span_to_frames: B=%d draws=%d N=%d nframe=%d", a->B, a->draws, a->N, a->nframe);
    VTGB_REQUIRE(a->variant == VTGB_MAP_A || a->variant == VTGB_MAP_B, VTGB_EINVAL, "span_to_frames: bad variant %d", a->variant);
    VTGB_REQUIRE(a->N <= MAP_MAX && 2 * a->nframe <= MAP_MAX, VTGB_EUNSUPPORTED, "span_to_frames: N=%d nframe=%d exceed %d", a->N, a->nframe, MAP_MAX);
    VTGB_REQUIRE(a->V || a->V_all > 1, VTGB_EINVAL, "span_to_frames: video length must be > 1");
    hipLaunchKernelGGL(span_to_frames_kernel, dim3((a->B + 63) / 64), dim3(64), 0, stream, *a);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

extern "C" int vtgb_gather_frames(const vtgb_gather_frames_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->pixel_values && a->frame_idx && a->out, VTGB_EINVAL, "gather_frames: NULL argument");
    VTGB_REQUIRE(a->B > 0 && a->N > 0 && a->nframe > 0 && a->frame_elems > 0 && (a->frame_elems % 4) == 0, VTGB_EINVAL,
                 "gather_frames: B=%d N=%d nframe=%d frame_elems=%lld", a->B, a->N, a->nframe, (long long)a->frame_elems);
    int64_t bx = (a->frame_elems / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(gather_frames_kernel, dim3((unsigned)bx, a->nframe, a->B), dim3(256), 0, stream, *a);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}

// ---------------------------------------------------------------------------------------
// f3: frame preprocessing (builder_utils.py:117-128).  HBM-bound: 3 B read per source pixel (gathered four
// at a time), 12 B written per output pixel.  The arithmetic follows ATen's CPU bilinear kernel (source
// index = scale * (dst + 0.5) - 0.5 clamped at 0, lambda clamped to [0, 1], rows interpolated first), the
// result is truncated to an integer exactly as `.to(torch.uint8)` does, then /255 and normalised.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(const vtgb_preprocess_args a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = a.size;
    if (i >= (int64_t)a.n_out * S * S) return;
    const int x = (int)(i % S), y = (int)((i / S) % S);
    const int64_t f = i / ((int64_t)S * S);
    const int64_t t = a.frame_idx ? a.frame_idx[f] : f;
    const float sy = (float)a.H0 / (float)S, sx = (float)a.W0 / (float)S;
    const float ry = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f), rx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)floorf(ry), a.H0 - 1), x0 = min((int)floorf(rx), a.W0 - 1);
    const int y1 = min(y0 + 1, a.H0 - 1), x1 = min(x0 + 1, a.W0 - 1);
    const float ly = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
    const float wy0 = 1.f - ly, wx0 = 1.f - lx;
    const uint8_t* base = a.raw + t * (int64_t)a.H0 * a.W0 * 3;
    const uint8_t* p00 = base + ((int64_t)y0 * a.W0 + x0) * 3;
    const uint8_t* p01 = base + ((int64_t)y0 * a.W0 + x1) * 3;
    const uint8_t* p10 = base + ((int64_t)y1 * a.W0 + x0) * 3;
    const uint8_t* p11 = base + ((int64_t)y1 * a.W0 + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float r0 = (float)p00[c] * wx0 + (float)p01[c] * lx;
        const float r1 = (float)p10[c] * wx0 + (float)p11[c] * lx;
        const float v = r0 * wy0 + r1 * ly;
        const float q = (float)(int)v;                                  // .to(torch.uint8): truncation (0 <= v <= 255)
        a.out[((f * 3 + c) * S + y) * S + x] = (q / 255.0f - a.mean[c]) / a.std[c];
    }
}

extern "C" int vtgb_preprocess_frames(const vtgb_preprocess_args* a, vtgb_stream_t stream) {
    VTGB_REQUIRE(a && a->raw && a->out, VTGB_EINVAL, "preprocess_frames: NULL argument");
    VTGB_REQUIRE(a->T > 0 && a->H0 > 0 && a->W0 > 0 && a->n_out > 0 && a->size > 0, VTGB_EINVAL, "preprocess_frames: T=%d H0=%d W0=%d n_out=%d size=%d",
                 a->T, a->H0, a->W0, a->n_out, a->size);
    VTGB_REQUIRE(a->std[0] != 0.f && a->std[1] != 0.f && a->std[2] != 0.f, VTGB_EINVAL, "preprocess_frames: zero std");
    const int64_t n = (int64_t)a->n_out * a->size * a->size;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *a);
    VTGB_HIP(hipGetLastError());
    return VTGB_OK;
}
