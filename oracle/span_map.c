/* oracle/span_map.c -- TEST INFRASTRUCTURE ONLY (see oracle/vtgb_oracle.py).
 *
 * Plain-C restatement of the integer stages of the path, compiled by `make -C oracle` into
 * oracle/libvtgb_oracle.so and used by tests/ as a second, independent checker of the HIP
 * kernels (and of the Python oracle):
 *   vo_span_select     eval/utils/model.py:101-113  (first argmax of (logit + noise) / tau)
 *   vo_span_to_frames  eval/utils/model.py:124-150 (variant A) / :337-366 (variant B)
 * Rounding rules: SURVEY.md Appendix B.  Build needs -ffp-contract=off (no fused a*b+c).
 */
#include <math.h>
#include <stdint.h>

void vo_span_select(const float* logits, const float* noise, int64_t* idx, int B, int L, int draws, float tau) {
    for (int d = 0; d < draws; d++)
        for (int r = 0; r < 2 * B; r++) {
            const int b = r < B ? r : r - B, which = r < B ? 0 : 1;
            const float* nz = noise + ((int64_t)d * 2 * B + r) * L;
            float best = -INFINITY;
            int bi = 0;
            for (int i = 0; i < L; i++) {
                volatile float s = logits[((int64_t)b * L + i) * 2 + which] + nz[i];
                const float y = s / tau;
                if (i == 0 || y > best) { best = y; bi = i; }
            }
            idx[(int64_t)d * 2 * B + r] = bi;
        }
}

static int endpoint(int64_t k, int V, int N, int variant, int python_int) {
    if (python_int) {
        if (variant == 0) { volatile double q = (double)k / (double)V; return (int)(q * (double)N); }
        return (int)((double)(k * (int64_t)(N - 1)) / (double)(V - 1));
    }
    if (variant == 0) { volatile float q = (float)k / (float)V; volatile float p = q * (float)N; return (int)p; }
    { volatile float q = (float)(k * (int64_t)(N - 1)) / (float)(V - 1); return (int)q; }
}

/* sel [draws, 2B]; V per clip (or NULL -> V_all); out [B, nframe]; returns 0, or -1 on bad sizes */
int vo_span_to_frames(const int64_t* sel, const int32_t* V, int64_t* out, int B, int draws, int V_all, int N, int nframe, int variant) {
    enum { MAXN = 4096 };
    static unsigned char in[MAXN];
    static int cand[MAXN];
    if (N > MAXN || 2 * nframe > MAXN) return -1;
    for (int j = 0; j < B; j++) {
        const int v = V ? V[j] : V_all;
        for (int i = 0; i < N; i++) in[i] = 0;
        for (int ii = 0; ii < draws; ii++) {
            int64_t s = sel[(int64_t)ii * 2 * B + j], e = sel[(int64_t)ii * 2 * B + B + j];
            int py = 0;
            if (s >= v || e >= v || (s == 0 && e == 0)) { s = 0; e = v - 1; py = 1; }
            int lo = endpoint(s, v, N, variant, py), hi = endpoint(e, v, N, variant, py);
            if (lo < 0) lo = 0;
            if (hi > N) hi = N;
            for (int x = lo; x < hi; x++) in[x] = 1;
        }
        int len = 0;
        for (int i = 0; i < N; i++) if (in[i]) cand[len++] = i;
        if (len == 0) { for (int i = 0; i < N; i++) cand[i] = i; len = N; }
        while (len < nframe) {
            for (int i = len - 1; i >= 0; i--) { cand[2 * i] = cand[i]; cand[2 * i + 1] = cand[i]; }
            len *= 2;
        }
        if (len > nframe) {
            const double step = (double)len / (double)nframe;
            for (int x = 0; x < nframe; x++) {
                const int lo = (int)((double)x * step);
                const int hi = (x + 1 == nframe) ? len : (int)((double)(x + 1) * step);
                out[(int64_t)j * nframe + x] = cand[(lo + hi - 1) / 2];
            }
        } else {
            for (int x = 0; x < nframe; x++) out[(int64_t)j * nframe + x] = cand[x];
        }
    }
    return 0;
}
