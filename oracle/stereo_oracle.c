/*
 * stereo_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's cost-volume / disparity-regression
 * hot path (Magicboomliu/StereoFormer).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped package
 * (stereoformer_amd/) never does.
 *
 * Pinning: every function here is checked against golden tensors produced by
 * the real reference (imported on CPU in the build container by
 * tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
 * The one exception is oracle_cross_attention (A7): the reference file
 * models/CrossAttentionCostVolume/cross_attention_cost_volume.py is 0 bytes,
 * so that function is "parity unpinned" by the reference and pinned only by
 * identities against A1/A4/A8.
 *
 * All tensors are fp32, contiguous, NCHW.  Citations are relative to the
 * reference root.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* bench.py's cpu_baseline leg times the oracle at a stated thread count. */
void oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#define I4(b, c, h, w, C, H, W) \
    ((((size_t)(b) * (size_t)(C) + (size_t)(c)) * (size_t)(H) + (size_t)(h)) * (size_t)(W) + (size_t)(w))

/* ------------------------------------------------------------------ A1 --
 * CostVolume.forward, 'correlation' branch
 * models/utils/build_cost_volume.py:39-47 (twin models/utils/cost.py:38-46)
 *   vol[b,d,h,w] = mean_c L[b,c,h,w] * R[b,c,h,w-d]   (w >= d), else 0.0
 */
void oracle_corr_volume(const float *L, const float *R, float *vol,
                        int B, int C, int H, int W, int D)
{
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)(W > 0 ? W : 1));
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h) {
                    float *o = vol + I4(b, d, h, 0, D, H, W);
                    for (int w = 0; w < W; ++w) acc[w] = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        const float *l = L + I4(b, c, h, 0, C, H, W);
                        const float *r = R + I4(b, c, h, 0, C, H, W);
                        for (int w = d; w < W; ++w) acc[w] += l[w] * r[w - d];
                    }
                    for (int w = 0; w < W; ++w) o[w] = (w < d) ? 0.0f : acc[w] / (float)C;
                }
        free(acc);
    }
}

/* ------------------------------------------------------------------ A2 --
 * 'difference' branch, build_cost_volume.py:21-28  -> [B,C,D,H,W]
 */
void oracle_diff_volume(const float *L, const float *R, float *vol,
                        int B, int C, int H, int W, int D)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h) {
                    const float *l = L + I4(b, c, h, 0, C, H, W);
                    const float *r = R + I4(b, c, h, 0, C, H, W);
                    float *o = vol + ((((size_t)b * C + c) * D + d) * H + h) * (size_t)W;
                    for (int w = 0; w < W; ++w) o[w] = (w < d) ? 0.0f : l[w] - r[w - d];
                }
}

/* 'concat' branch, build_cost_volume.py:30-37 (and build_concat_volume,
 * models/utils/submodule.py:30-41)  -> [B,2C,D,H,W]; left in channels
 * [0,C), shifted right in [C,2C). */
void oracle_concat_volume(const float *L, const float *R, float *vol,
                          int B, int C, int H, int W, int D)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < 2 * C; ++c)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h) {
                    float *o = vol + ((((size_t)b * 2 * C + c) * D + d) * H + h) * (size_t)W;
                    if (c < C) {
                        const float *l = L + I4(b, c, h, 0, C, H, W);
                        for (int w = 0; w < W; ++w) o[w] = (w < d) ? 0.0f : l[w];
                    } else {
                        const float *r = R + I4(b, c - C, h, 0, C, H, W);
                        for (int w = 0; w < W; ++w) o[w] = (w < d) ? 0.0f : r[w - d];
                    }
                }
}

/* ------------------------------------------------------------------ A3 --
 * groupwise_correlation + build_gwc_volume
 * models/CrossAttentionCostVolume/GWC_Cost_Volume.py:5-11,14-24
 *   vol[b,g,d,h,w] = mean_{c in group g} L[b,c,h,w]*R[b,c,h,w-d], 0 for w<d
 * Channels are grouped contiguously: group g = channels [g*C/G,(g+1)*C/G).
 * Returns -1 when C % G != 0 (the reference asserts).
 */
int oracle_gwc_volume(const float *L, const float *R, float *vol,
                      int B, int C, int H, int W, int D, int G)
{
    if (G <= 0 || C % G != 0) return -1;
    const int cpg = C / G;
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)(W > 0 ? W : 1));
#pragma omp for collapse(3) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int g = 0; g < G; ++g)
                for (int d = 0; d < D; ++d)
                    for (int h = 0; h < H; ++h) {
                        float *o = vol + ((((size_t)b * G + g) * D + d) * H + h) * (size_t)W;
                        for (int w = 0; w < W; ++w) acc[w] = 0.0f;
                        for (int cc = 0; cc < cpg; ++cc) {
                            const int c = g * cpg + cc;
                            const float *l = L + I4(b, c, h, 0, C, H, W);
                            const float *r = R + I4(b, c, h, 0, C, H, W);
                            for (int w = d; w < W; ++w) acc[w] += l[w] * r[w - d];
                        }
                        for (int w = 0; w < W; ++w) o[w] = (w < d) ? 0.0f : acc[w] / (float)cpg;
                    }
        free(acc);
    }
    return 0;
}

/* mean over the group axis: [B,G,D,H,W] -> [B,D,H,W] (torch .mean(1); used by
 * identity I1 and by the cfg-2 chain gwc -> group-mean -> soft-argmin). */
void oracle_group_mean(const float *volg, float *vol, int B, int G, int D, int H, int W)
{
    const size_t plane = (size_t)D * H * W;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < plane; ++i) {
            float s = 0.0f;
            for (int g = 0; g < G; ++g) s += volg[((size_t)b * G + g) * plane + i];
            vol[(size_t)b * plane + i] = s / (float)G;
        }
}

/* ------------------------------------------------------------------ A4 --
 * DisparityEstimation.forward / DisparityEstimationWithProb.forward
 * models/utils/estimation.py:14-33, 44-62 (and disparity_regression,
 * models/utils/submodule.py:23-27)
 *   p = softmax(+-vol, dim=1); disp = sum_d p[d]*d.   prob may be NULL.
 */
void oracle_soft_argmin(const float *vol, float *disp, float *prob,
                        int B, int D, int H, int W, int match_similarity)
{
    const float sgn = match_similarity ? 1.0f : -1.0f;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
        const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
        const float *v = vol + b * D * HW + p;
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) {
            const float x = sgn * v[(size_t)d * HW];
            m = x > m ? x : m;
        }
        float s = 0.0f;
        for (int d = 0; d < D; ++d) s += expf(sgn * v[(size_t)d * HW] - m);
        float e = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float pd = expf(sgn * v[(size_t)d * HW] - m) / s;
            if (prob) prob[b * D * HW + (size_t)d * HW + p] = pd;
            e += pd * (float)d;
        }
        disp[b * HW + p] = e;
    }
}

/* ------------------------------------------------------------------ A5 --
 * ste_ceil / ste_floor forward values,
 * models/LocalCostVolume/Attempts/fixed_local_cost_volume.py:5-9:
 *   ceil(x) - x + x evaluated left-to-right in fp32 (kept verbatim: the
 *   round trip can differ from ceil(x) by one ulp, and .long() truncates).
 */
static inline float ste_ceil_f(float x)
{
    volatile float t = ceilf(x) - x;
    return t + x;
}
static inline float ste_floor_f(float x)
{
    volatile float t = floorf(x) - x;
    return t + x;
}
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* one interpolated tap: build_cost_volume_from_volume,
 * fixed_local_cost_volume.py:61-80.  v points at vol[b,0,h,w]; stride = H*W. */
static inline float sample_tap(const float *v, size_t stride, int D, float c)
{
    const float cc = clampf(ste_ceil_f(c), 0.0f, (float)(D - 1));
    const float cf = clampf(ste_floor_f(c), 0.0f, (float)(D - 1));
    const long ic = (long)cc, ifl = (long)cf; /* .long() truncates toward zero */
    const float floor_rate = (float)ic - c;
    const float ceil_rate = 1.0f - floor_rate;
    return v[(size_t)ifl * stride] * floor_rate + v[(size_t)ic * stride] * ceil_rate;
}

/* build_cost_volume_from_volume(old_volume, sampling_candidates)
 * fixed_local_cost_volume.py:61-80: cand [B,K,H,W] -> out [B,K,H,W] */
void oracle_sample_volume(const float *vol, const float *cand, float *out,
                          int B, int D, int H, int W, int K)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
        const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
        const float *v = vol + b * D * HW + p;
        for (int k = 0; k < K; ++k) {
            const size_t o = b * K * HW + (size_t)k * HW + p;
            out[o] = sample_tap(v, HW, D, cand[o]);
        }
    }
}

/* candidates -> taps -> softmax -> expectation for one pixel; shared by A5,
 * A6 and the A11 sampler (all three reference bodies are textually the same:
 * fixed_local_cost_volume.py:33-56, variance_local_cost_volume.py:56-72,
 * Attempts/update.py:297-313). */
static float lookup_pixel(const float *v, size_t HW, int D, float cur, float lower, float upper,
                          int S, int invalid, float *scratch /* 2*(S+1) */)
{
    float *cand = scratch, *val = scratch + (S + 1);
    const float interval = (upper - lower) * 1.0f / (float)S;
    const float inv = invalid ? 1.0f : 0.0f;
    for (int k = 0; k <= S; ++k) {
        float c = lower + (float)k * interval;
        if (invalid >= 0) /* consider_valid: c*(1-m) + m*cur, update.py:304 */
            c = c * (1.0f - inv) + inv * cur;
        cand[k] = c;
        val[k] = sample_tap(v, HW, D, c);
    }
    float m = -INFINITY;
    for (int k = 0; k <= S; ++k) m = val[k] > m ? val[k] : m;
    float s = 0.0f;
    for (int k = 0; k <= S; ++k) s += expf(val[k] - m);
    float e = 0.0f;
    for (int k = 0; k <= S; ++k) e += (expf(val[k] - m) / s) * cand[k];
    return e;
}

/* LocalCostVolume.forward, fixed_local_cost_volume.py:18-58.
 * disp [B,1,H,W] -> out [B,1,H,W]. */
void oracle_local_lookup(const float *vol, const float *disp, float *out,
                         int B, int D, int H, int W, float radius, int S, int consider_valid)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel
    {
        float *scratch = (float *)malloc(sizeof(float) * 2 * (size_t)(S + 1));
#pragma omp for schedule(static)
        for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
            const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
            const float cur = disp[b * HW + p];
            float lower = cur - radius, upper = cur + radius;
            int invalid = -1;
            if (consider_valid) {
                invalid = (lower < 0.0f) || (upper >= (float)(D - 1)); /* :23-27 */
            } else {
                lower = lower < 0.0f ? 0.0f : lower;        /* :29 */
                upper = clampf(upper, 0.0f, (float)D);      /* :30 (max = D, not D-1) */
            }
            out[b * HW + p] = lookup_pixel(vol + b * D * HW + p, HW, D, cur, lower, upper, S, invalid, scratch);
        }
        free(scratch);
    }
}

/* ------------------------------------------------------------------ A6 --
 * GetVarince + Variance_Based_Local_Cost_Volume.forward
 * models/LocalCostVolume/Attempts/variance_local_cost_volume.py:12-25,34-78
 */
void oracle_variance_lookup(const float *vol, const float *disp, float *out,
                            int B, int D, int H, int W, float gamma, int S, int consider_valid)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel
    {
        float *scratch = (float *)malloc(sizeof(float) * 2 * (size_t)(S + 1));
#pragma omp for schedule(static)
        for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
            const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
            const int x = (int)(p % (size_t)W);
            const float *v = vol + b * D * HW + p;
            const float cur = disp[b * HW + p];
            float m = -INFINITY;
            for (int d = 0; d < D; ++d) m = v[(size_t)d * HW] > m ? v[(size_t)d * HW] : m;
            float s = 0.0f;
            for (int d = 0; d < D; ++d) s += expf(v[(size_t)d * HW] - m);
            float var = 0.0f;
            for (int d = 0; d < D; ++d) {
                const float pd = expf(v[(size_t)d * HW] - m) / s;
                const float e = (float)d - cur;
                var += pd * (e * e);
            }
            const float sigma = sqrtf(var);
            float lower = cur - gamma * sigma, upper = cur + gamma * sigma;
            int invalid = -1;
            if (consider_valid) { /* :43-50 */
                invalid = (lower < 0.0f) || (upper >= (float)(D - 1)) || (upper > (float)x);
            } else {              /* :52-53 */
                lower = clampf(lower, 0.0f, (float)(D - 1));
                upper = clampf(upper, 0.0f, (float)(D - 1));
            }
            out[b * HW + p] = lookup_pixel(v, HW, D, cur, lower, upper, S, invalid, scratch);
        }
        free(scratch);
    }
}

/* A11 sampler: learned non-negative offsets (lower_off, upper_off [B,1,H,W])
 * -> bounds cur-lower_off, cur+upper_off -> same sampler.
 * models/LocalCostVolume/Attempts/update.py:279-313 */
void oracle_bounds_lookup(const float *vol, const float *disp, const float *lower_off,
                          const float *upper_off, float *out,
                          int B, int D, int H, int W, int S, int consider_valid)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel
    {
        float *scratch = (float *)malloc(sizeof(float) * 2 * (size_t)(S + 1));
#pragma omp for schedule(static)
        for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
            const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
            const float cur = disp[b * HW + p];
            float lower = cur - lower_off[b * HW + p], upper = cur + upper_off[b * HW + p];
            int invalid = -1;
            if (consider_valid) {
                invalid = (lower < 0.0f) || (upper >= (float)(D - 1));
            } else {
                lower = lower < 0.0f ? 0.0f : lower;
                upper = clampf(upper, 0.0f, (float)D);
            }
            out[b * HW + p] = lookup_pixel(vol + b * D * HW + p, HW, D, cur, lower, upper, S, invalid, scratch);
        }
        free(scratch);
    }
}

/* A11 uncertainty volume, Attempts/update.py:71-72,86:
 *   U[b,d,h,w] = softmax_d(vol)[d] * d * (d - disp)^2 */
void oracle_uncertainty_volume(const float *vol, const float *disp, float *out,
                               int B, int D, int H, int W)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
        const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
        const float *v = vol + b * D * HW + p;
        const float cur = disp[b * HW + p];
        float m = -INFINITY;
        for (int d = 0; d < D; ++d) m = v[(size_t)d * HW] > m ? v[(size_t)d * HW] : m;
        float s = 0.0f;
        for (int d = 0; d < D; ++d) s += expf(v[(size_t)d * HW] - m);
        for (int d = 0; d < D; ++d) {
            const float pd = expf(v[(size_t)d * HW] - m) / s;
            const float e = (float)d - cur;
            out[b * D * HW + (size_t)d * HW + p] = pd * (float)d * e * e;
        }
    }
}

/* ------------------------------------------------------------------ A8 --
 * CorrBlock1D.corr, models/IterativeLocalCostVolume/corr/corr.py:103-111
 *   out[b,h,w1,w2] = sum_c f1[b,c,h,w1]*f2[b,c,h,w2] / sqrt(C)
 * (the reference's [B,H,W1,1,W2] with the unit axis dropped) */
void oracle_allpairs_corr(const float *f1, const float *f2, float *out,
                          int B, int C, int H, int W1, int W2)
{
    const float scale = sqrtf((float)C);
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)(W2 > 0 ? W2 : 1));
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h)
                for (int w1 = 0; w1 < W1; ++w1) {
                    for (int w2 = 0; w2 < W2; ++w2) acc[w2] = 0.0f;
                    for (int c = 0; c < C; ++c) {
                        const float a = f1[I4(b, c, h, w1, C, H, W1)];
                        const float *r = f2 + I4(b, c, h, 0, C, H, W2);
                        for (int w2 = 0; w2 < W2; ++w2) acc[w2] += a * r[w2];
                    }
                    float *o = out + (((size_t)b * H + h) * W1 + w1) * (size_t)W2;
                    for (int w2 = 0; w2 < W2; ++w2) o[w2] = acc[w2] / scale;
                }
        free(acc);
    }
}

/* pyramid step: F.avg_pool2d(corr,[1,2],stride=[1,2]) on rows of length Win
 * -> Win/2 (floor: last odd element dropped), corr.py:70-76 */
void oracle_avgpool_w2(const float *in, float *out, long long rows, int Win)
{
    const int Wout = Win / 2;
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < rows; ++r)
        for (int x = 0; x < Wout; ++x)
            out[(size_t)r * Wout + x] = (in[(size_t)r * Win + 2 * x] + in[(size_t)r * Win + 2 * x + 1]) * 0.5f;
}

/* 1-D linear tap with zeros padding in pixel coordinates that went through
 * bilinear_sampler's normalise (utils/utils.py:64-66) and grid_sample's
 * align_corners=True un-normalise. */
static inline float lerp_zero_pad_ac(const float *row, int Wl, float x)
{
    const float xg = 2.0f * x / (float)(Wl - 1) - 1.0f;
    const float ix = ((xg + 1.0f) / 2.0f) * (float)(Wl - 1);
    const float fx = floorf(ix);
    const float w1 = ix - fx, w0 = 1.0f - w1;
    float acc = 0.0f;
    if (fx >= 0.0f && fx <= (float)(Wl - 1)) acc += row[(long)fx] * w0;
    if (fx + 1.0f >= 0.0f && fx + 1.0f <= (float)(Wl - 1)) acc += row[(long)fx + 1] * w1;
    return acc;
}

/* ------------------------------------------------------------------ A9 --
 * CorrBlock1D.__call__ + bilinear_sampler
 * corr.py:78-101, models/IterativeLocalCostVolume/utils/utils.py:59-76
 * pyr[i]: [B*H*W1, Wl_i]; coords [B,2,H,W1] (channel 0 = x);
 * out [B, L*(2r+1), H, W1], level-major then tap. */
void oracle_corr_lookup(const float *const *pyr, const int *widths, int L, int r,
                        const float *coords, float *out, int B, int H, int W1)
{
    const int T = 2 * r + 1;
    const size_t HW = (size_t)H * W1;
#pragma omp parallel for schedule(static)
    for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
        const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
        const float x = coords[b * 2 * HW + p];
        for (int i = 0; i < L; ++i) {
            const float *row = pyr[i] + (size_t)bp * widths[i];
            const float xl = x / (float)(1 << i);
            for (int t = 0; t < T; ++t) {
                const float dx = (float)(t - r);
                out[b * (size_t)(L * T) * HW + (size_t)(i * T + t) * HW + p] =
                    lerp_zero_pad_ac(row, widths[i], dx + xl);
            }
        }
    }
}

/* ----------------------------------------------------------------- A10 --
 * PytorchAlternateCorrBlock1D.__call__, corr.py:8-51: on-the-fly variant.
 * Pools fmap2 along W per level, samples it bilinearly in (x,y) with
 * align_corners=True / zeros padding, dots with fmap1, / sqrt(C).
 * coords [B,2,H,W]; the y channel is used as given (corr.py:37-44). */
void oracle_alt_corr_lookup(const float *f1, const float *f2, const float *coords, float *out,
                            int B, int C, int H, int W, int L, int r)
{
    const int T = 2 * r + 1;
    const size_t HW = (size_t)H * W;
    const float scale = sqrtf((float)C);
    /* pooled copies of fmap2 */
    float **lv = (float **)malloc(sizeof(float *) * (size_t)L);
    int *wl = (int *)malloc(sizeof(int) * (size_t)L);
    lv[0] = (float *)f2;
    wl[0] = W;
    for (int i = 1; i < L; ++i) {
        wl[i] = wl[i - 1] / 2;
        lv[i] = (float *)malloc(sizeof(float) * (size_t)B * C * H * (size_t)(wl[i] > 0 ? wl[i] : 1));
        oracle_avgpool_w2(lv[i - 1], lv[i], (long long)B * C * H, wl[i - 1]);
    }
#pragma omp parallel for schedule(static)
    for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
        const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
        const int h = (int)(p / (size_t)W), w = (int)(p % (size_t)W);
        const float x = coords[b * 2 * HW + p], y = coords[b * 2 * HW + HW + p];
        for (int i = 0; i < L; ++i) {
            const int Wl = wl[i];
            for (int t = 0; t < T; ++t) {
                const float px = x / (float)(1 << i) + (float)(t - r);
                const float xg = 2.0f * px / (float)(Wl - 1) - 1.0f;
                const float yg = 2.0f * y / (float)(H - 1) - 1.0f;
                const float ix = ((xg + 1.0f) / 2.0f) * (float)(Wl - 1);
                const float iy = ((yg + 1.0f) / 2.0f) * (float)(H - 1);
                const float fx = floorf(ix), fy = floorf(iy);
                const float ax = ix - fx, ay = iy - fy;
                float dot = 0.0f;
                for (int c = 0; c < C; ++c) {
                    const float *img = lv[i] + ((size_t)b * C + c) * H * (size_t)Wl;
                    float s = 0.0f;
                    for (int jy = 0; jy < 2; ++jy)
                        for (int jx = 0; jx < 2; ++jx) {
                            const float yy = fy + (float)jy, xx = fx + (float)jx;
                            if (yy < 0.0f || yy > (float)(H - 1) || xx < 0.0f || xx > (float)(Wl - 1)) continue;
                            const float wgt = (jx ? ax : 1.0f - ax) * (jy ? ay : 1.0f - ay);
                            s += img[(size_t)(long)yy * Wl + (size_t)(long)xx] * wgt;
                        }
                    dot += s * f1[I4(b, c, h, w, C, H, W)];
                }
                out[b * (size_t)(L * T) * HW + (size_t)(i * T + t) * HW + p] = dot / scale;
            }
        }
    }
    for (int i = 1; i < L; ++i) free(lv[i]);
    free(lv);
    free(wl);
}

/* ----------------------------------------------------------------- A12 --
 * disp_warp, utils/disparity_warper.py:83-106 (+ meshgrid :60-80,
 * normalize_coords :48-58).  Grid normalised with (w-1),(h-1) but sampled by
 * F.grid_sample with its default align_corners=False; padding 'border' for
 * the image and 'zeros' for the ones-mask, thresholded at 0.9999.
 * img [B,Ci,H,W], disp [B,1,H,W] -> warped, mask [B,Ci,H,W] (mask may be NULL)
 * padding_border: 1 = 'border', 0 = 'zeros' for the image sample. */
void oracle_disp_warp(const float *img, const float *disp, float *warped, float *mask,
                      int B, int Ci, int H, int W, int padding_border)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (long long bp = 0; bp < (long long)B * (long long)HW; ++bp) {
        const size_t b = (size_t)(bp / (long long)HW), p = (size_t)(bp % (long long)HW);
        const int y = (int)(p / (size_t)W), x = (int)(p % (size_t)W);
        const float gx = (float)x + (-disp[b * HW + p]);
        const float gy = (float)y + 0.0f;
        const float nx = 2.0f * (gx / (float)(W - 1)) - 1.0f;
        const float ny = 2.0f * (gy / (float)(H - 1)) - 1.0f;
        const float ux = ((nx + 1.0f) * (float)W - 1.0f) / 2.0f; /* align_corners=False */
        const float uy = ((ny + 1.0f) * (float)H - 1.0f) / 2.0f;
        /* zeros-padding weights (mask) */
        {
            const float fx = floorf(ux), fy = floorf(uy);
            const float ax = ux - fx, ay = uy - fy;
            float msum = 0.0f;
            for (int jy = 0; jy < 2; ++jy)
                for (int jx = 0; jx < 2; ++jx) {
                    const float yy = fy + (float)jy, xx = fx + (float)jx;
                    if (yy < 0.0f || yy > (float)(H - 1) || xx < 0.0f || xx > (float)(W - 1)) continue;
                    msum += (jx ? ax : 1.0f - ax) * (jy ? ay : 1.0f - ay);
                }
            if (mask) {
                float mv = msum;
                if (mv < 0.9999f) mv = 0.0f;
                if (mv > 0.0f) mv = 1.0f;
                for (int c = 0; c < Ci; ++c) mask[(b * Ci + c) * HW + p] = mv;
            }
        }
        float sx = ux, sy = uy;
        if (padding_border) {
            sx = clampf(sx, 0.0f, (float)(W - 1));
            sy = clampf(sy, 0.0f, (float)(H - 1));
        }
        const float fx = floorf(sx), fy = floorf(sy);
        const float ax = sx - fx, ay = sy - fy;
        for (int c = 0; c < Ci; ++c) {
            const float *im = img + (b * Ci + c) * HW;
            float s = 0.0f;
            for (int jy = 0; jy < 2; ++jy)
                for (int jx = 0; jx < 2; ++jx) {
                    const float yy = fy + (float)jy, xx = fx + (float)jx;
                    if (yy < 0.0f || yy > (float)(H - 1) || xx < 0.0f || xx > (float)(W - 1)) continue;
                    s += im[(size_t)(long)yy * W + (size_t)(long)xx] * ((jx ? ax : 1.0f - ax) * (jy ? ay : 1.0f - ay));
                }
            warped[(b * Ci + c) * HW + p] = s;
        }
    }
}

/* ----------------------------------------------------------------- A13 --
 * upsample_convex8, models/utils/upsample.py:22-38 (f=8) and
 * RAFT_Stereo.upsample_flow, models/IterativeLocalCostVolume/raft_stereo.py:110-122
 * (factor 2^downsample, Cd flow channels).
 * disp [N,Cd,H,W], mask [N,9*f*f,H,W] -> out [N,Cd,f*H,f*W] */
void oracle_upsample_convex(const float *disp, const float *mask, float *out,
                            int N, int Cd, int H, int W, int f)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int i = 0; i < f; ++i)
                    for (int j = 0; j < f; ++j) {
                        float mk[9], mx = -INFINITY, s = 0.0f;
                        for (int k = 0; k < 9; ++k) {
                            mk[k] = mask[((size_t)n * 9 * f * f + (size_t)(k * f * f + i * f + j)) * HW + (size_t)y * W + x];
                            mx = mk[k] > mx ? mk[k] : mx;
                        }
                        for (int k = 0; k < 9; ++k) { mk[k] = expf(mk[k] - mx); s += mk[k]; }
                        for (int c = 0; c < Cd; ++c) {
                            float acc = 0.0f;
                            for (int k = 0; k < 9; ++k) {
                                const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
                                float nb = 0.0f; /* unfold zero padding */
                                if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                                    nb = (float)f * disp[((size_t)n * Cd + c) * HW + (size_t)yy * W + xx];
                                acc += (mk[k] / s) * nb;
                            }
                            out[(((size_t)n * Cd + c) * (size_t)(f * H) + (size_t)(f * y + i)) * (size_t)(f * W) + (size_t)(f * x + j)] = acc;
                        }
                    }
}

/* F.interpolate(mode='bilinear') to an explicit size, with align_corners
 * 0/1.  upsample_simple8 (upsample.py:41-43, align_corners=True, x8 applied
 * by the caller), interp (update/update.py:125-127, True) and the 1/8 image
 * resize inside BasicGuidanceNet (Attempts/update.py:73-77, False). */
void oracle_resize_bilinear(const float *in, float *out, int NC, int Hi, int Wi, int Ho, int Wo,
                            int align_corners)
{
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < NC; ++nc)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float sy, sx;
                if (align_corners) {
                    sy = Ho > 1 ? (float)y * ((float)(Hi - 1) / (float)(Ho - 1)) : 0.0f;
                    sx = Wo > 1 ? (float)x * ((float)(Wi - 1) / (float)(Wo - 1)) : 0.0f;
                } else {
                    sy = ((float)y + 0.5f) * ((float)Hi / (float)Ho) - 0.5f;
                    sx = ((float)x + 0.5f) * ((float)Wi / (float)Wo) - 0.5f;
                    if (sy < 0.0f) sy = 0.0f;
                    if (sx < 0.0f) sx = 0.0f;
                }
                int y0 = (int)sy, x0 = (int)sx;
                if (y0 > Hi - 1) y0 = Hi - 1;
                if (x0 > Wi - 1) x0 = Wi - 1;
                const int y1 = y0 < Hi - 1 ? y0 + 1 : y0, x1 = x0 < Wi - 1 ? x0 + 1 : x0;
                const float ay = sy - (float)y0, ax = sx - (float)x0;
                const float *im = in + (size_t)nc * Hi * Wi;
                const float top = im[(size_t)y0 * Wi + x0] * (1.0f - ax) + im[(size_t)y0 * Wi + x1] * ax;
                const float bot = im[(size_t)y1 * Wi + x0] * (1.0f - ax) + im[(size_t)y1 * Wi + x1] * ax;
                out[((size_t)nc * Ho + y) * Wo + x] = top * (1.0f - ay) + bot * ay;
            }
}

/* ------------------------------------------------------------------ A7 --
 * Epipolar cross-attention.  BUILD-DEFINED: the reference body is a 0-byte
 * file (models/CrossAttentionCostVolume/cross_attention_cost_volume.py), so
 * this function is PARITY-UNPINNED by the reference and pinned by identities:
 *   S[w,v] = <L[:,h,w], R[:,h,v]> * scale, band 0 <= w-v < D
 *   scale_mode 0: 1/sqrt(C)  (CorrBlock1D.corr, corr.py:111)
 *   scale_mode 1: 1/C        (CostVolume 'correlation' mean, build_cost_volume.py:44-45)
 *   mask_mode 0 ('zero'): band entries with v<0 keep logit 0.0 and take part
 *       in the softmax (what A1+A4 do: zero-filled volume, estimation.py:20)
 *   mask_mode 1 ('neginf'): those entries are excluded
 *   P = softmax over the band; disp[w] = sum_d P[d]*d; out[:,w] = sum_d P[d]*V[:,w-d]
 *   (V = value features [B,Cv,H,W]; out-of-image taps contribute 0)
 */
void oracle_cross_attention(const float *L, const float *R, const float *V, float *out, float *disp,
                            int B, int C, int Cv, int H, int W, int D, int mask_mode, int scale_mode)
{
    const float scale = scale_mode == 1 ? 1.0f / (float)C : 1.0f / sqrtf((float)C);
#pragma omp parallel
    {
        float *logit = (float *)malloc(sizeof(float) * (size_t)(D > 0 ? D : 1));
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    float m = -INFINITY;
                    for (int d = 0; d < D; ++d) {
                        if (w - d >= 0) {
                            float s = 0.0f;
                            for (int c = 0; c < C; ++c)
                                s += L[I4(b, c, h, w, C, H, W)] * R[I4(b, c, h, w - d, C, H, W)];
                            logit[d] = s * scale;
                        } else {
                            logit[d] = mask_mode == 0 ? 0.0f : -INFINITY;
                        }
                        m = logit[d] > m ? logit[d] : m;
                    }
                    float s = 0.0f;
                    for (int d = 0; d < D; ++d) s += expf(logit[d] - m);
                    float e = 0.0f;
                    for (int d = 0; d < D; ++d) {
                        logit[d] = expf(logit[d] - m) / s;
                        e += logit[d] * (float)d;
                    }
                    if (disp) disp[((size_t)b * H + h) * W + w] = e;
                    if (out)
                        for (int c = 0; c < Cv; ++c) {
                            float acc = 0.0f;
                            for (int d = 0; d < D && w - d >= 0; ++d)
                                acc += logit[d] * V[I4(b, c, h, w - d, Cv, H, W)];
                            out[I4(b, c, h, w, Cv, H, W)] = acc;
                        }
                }
        free(logit);
    }
}

/* ----------------------------------------------------------------- A11/A14
 * ConvGRU gate math (the convolutions themselves stay in the host framework).
 * Attempts/update.py:41-48:   h' = (1-z)*h + z*g,  z=sigmoid(zc), g=tanh(gc)
 * update/update.py:28-32:     z=sigmoid(zc+cz), q=tanh(qc+cq); same blend.
 * bias pointers may be NULL. */
void oracle_gru_blend(const float *h, const float *zc, const float *qc, const float *cz,
                      const float *cq, float *out, long long n, int softsign)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; ++i) {
        const float z = 1.0f / (1.0f + expf(-(zc[i] + (cz ? cz[i] : 0.0f))));
        const float qx = qc[i] + (cq ? cq[i] : 0.0f);
        const float q = softsign ? qx / (1.0f + fabsf(qx)) : tanhf(qx); /* nn.Softsign, Attempts/update.py:34 */
        out[i] = (1.0f - z) * h[i] + z * q;
    }
}
/* r*h (update/update.py:30) / b*h (Attempts/update.py:44-45) */
void oracle_gru_reset(const float *h, const float *rc, const float *cr, float *out, long long n)
{
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; ++i) {
        const float r = 1.0f / (1.0f + expf(-(rc[i] + (cr ? cr[i] : 0.0f))));
        out[i] = r * h[i];
    }
}

/* ------------------------------------------------------------------ N2 --
 * nn.InstanceNorm2d(C, eps, affine) followed by ReLU, as the reference's extractors apply them
 * (models/IterativeLocalCostVolume/extractor/extractor.py:135 norm1 = nn.InstanceNorm2d(64), :187-188
 * x = self.norm1(x); x = self.relu1(x); ResidualBlock :38-41,63-64).  Per (b, c) plane: biased variance over
 * H*W, y = (x - mean) / sqrt(var + eps) * gamma + beta, then max(y, 0) when relu.  gamma / beta may be NULL.
 * Statistics in double (torch accumulates its float plane sums in a cascade; both agree to ~1e-6). */
void oracle_instance_norm_relu(const float *x, const float *gamma, const float *beta, float *out,
                               int B, int C, int H, int W, float eps, int relu)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (long long p = 0; p < (long long)B * C; ++p) {
        const float *xp = x + (size_t)p * HW;
        float *op = out + (size_t)p * HW;
        double s = 0.0, ss = 0.0;
        for (size_t i = 0; i < HW; ++i) s += xp[i];
        const double mean = s / (double)HW;
        for (size_t i = 0; i < HW; ++i) {
            const double d = (double)xp[i] - mean;
            ss += d * d;
        }
        const double var = ss / (double)HW;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const int c = (int)(p % C);
        const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
        for (size_t i = 0; i < HW; ++i) {
            float y = ((xp[i] - (float)mean) * rstd) * g + b;
            if (relu && y < 0.0f) y = 0.0f;
            op[i] = y;
        }
    }
}
