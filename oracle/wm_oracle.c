/*
 * wm_oracle.c -- CPU restatement of the reference Watermark hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (watermarking-gpu_amd/,
 * include/) may call, link or load this file.  Allowed users: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker.
 *
 * PARITY UNPINNED: the reference (kar-dim/Watermarking-GPU @ 2025-05-23) holds no
 * tests, golden vectors or expected values for this path, and its own
 * implementation cannot be built here (ArrayFire + OpenCL + MSVC).
 * This file restates the published algorithm from the reference sources cited
 * per function below; it is cross-checked against an independent numpy
 * restatement (tests/np_restatement.py) on the reference's sample image/W pair.
 *
 * Conventions: planes are row-major f32, x(r,c) = buf[r*cols + c];
 * W(r,c) = file[r*cols + c] (Watermark.cpp:62-75 loads (cols,rows) column-major and
 * transposes); borders are replicate / clamp-to-edge (nvf.hpp:9, me_p3.hpp:45,
 * scaled_neighbors_p3.hpp:14).
 *
 * Precision policy of the oracle ("exact-sum" policy): element-wise maths in IEEE f32
 * in the reference's op order with explicit fmaf where the reference builds with
 * -cl-mad-enable (main.cpp:106-108); every global sum (Gram matrix, norms, dot) is
 * accumulated in f64 from exact f32xf32 products.  Two switches bracket the
 * reference's own numeric noise: accum_f32 (f32 running sums, as me_p3.hpp:65-66,80-81
 * and ArrayFire's f32 reductions) and fp16_products (products rounded to half before
 * summation, me_p3.hpp:8-21).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off is REQUIRED: fused ops appear only where fmaf() is written.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WMO_OK 0
#define WMO_UNSOLVABLE 1
#define WMO_BAD_ARG (-1)

enum { WMO_MASK_ME = 0, WMO_MASK_NVF = 1 }; /* Watermark.hpp:10-14 */

typedef struct {
    int accum_f32;     /* 1: f32 running sums (reference behaviour), 0: f64 (oracle policy) */
    int fp16_products; /* 1: round Gram products to IEEE half first (me_p3.hpp:10,16-20) */
    int ref_arith;     /* 1: the reference's own arithmetic for the prediction system, as far as its sources fix it:
                        *    products rounded to half (me_p3.hpp:8-21), 64-lane work-group sums in f32 in lane order
                        *    (me_p3.hpp:61-82; a work group = 64 consecutive columns of one row), f32 sum of the work-group
                        *    partials (af::sum, Watermark.cpp:148-149 -- ArrayFire's order is not published: a pairwise
                        *    tree is used here), f32 LU with partial pivoting (af::solve, Watermark.cpp:203).
                        *    Brackets what the reference itself would produce; norms and dot products stay f64 (their f32
                        *    noise is ~1e-6 relative, two orders below the bracket).  Overrides the two switches above. */
} wmo_opts;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- f32 -> f16 -> f32 round trip, round-to-nearest-even (vstore_half8 default rounding) ---- */
static float round_to_half(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    float out;
    if (a >= 0x7f800000u) return f;                       /* inf / nan */
    if (a >= 0x477ff000u) {                               /* >= 65520 rounds to inf */
        uint32_t inf = sign | 0x7f800000u;
        memcpy(&out, &inf, 4);
        return out;
    }
    if (a < 0x38800000u) {                                /* below half min normal 2^-14: subnormal grid 2^-24 */
        float af = fabsf(f);
        float q = af * 16777216.0f;                       /* exact scaling by 2^24 */
        q = nearbyintf(q);                                /* RNE under default rounding mode */
        out = q / 16777216.0f;
        return sign ? -out : out;
    }
    /* normal half: keep 10 mantissa bits, RNE on the 13 dropped bits */
    uint32_t lsb = (a >> 13) & 1u;
    a += 0x0fffu + lsb;
    a &= ~0x1fffu;
    a |= sign;
    memcpy(&out, &a, 4);
    return out;
}

/* Watermark.cpp:22  strengthFactor = 255 / sqrt(10^(psnr/10)), all in float */
float wmo_strength_factor(float psnr)
{
    return 255.0f / sqrtf(powf(10.0f, psnr / 10.0f));
}

static int gram_ref_arith(const float* x, int rows, int cols, double Rx[64], double rx[8]);

/*
 * Gram matrix of the 8 neighbours and cross-correlation with the centre pixel.
 * Follows me_p3.hpp:43-82 (neighbour order x0..x8 without the centre, the 8 rx products
 * and the 36 upper-triangle Rx products, zero contribution of padded lanes) and
 * Watermark.cpp:140-151 + Watermark.hpp:29-39 (partials folded into the full symmetric
 * 8x8 through RxMappings).  Rx is returned row-major 8x8 (symmetric), rx as 8 values.
 * Summation order: per image row left-to-right, then rows top-to-bottom (deterministic
 * for any thread count).
 */
int wmo_gram(const float* x, int rows, int cols, double Rx[64], double rx[8], const wmo_opts* opt)
{
    if (!x || rows < 1 || cols < 1) return WMO_BAD_ARG;
    if (opt && opt->ref_arith) return gram_ref_arith(x, rows, cols, Rx, rx);
    const int accum_f32 = opt ? opt->accum_f32 : 0;
    const int fp16 = opt ? opt->fp16_products : 0;
    double* rowacc = (double*)malloc((size_t)rows * 44 * sizeof(double));
    if (!rowacc) return WMO_BAD_ARG;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        const float* up = x + (size_t)clampi(r - 1, 0, rows - 1) * cols;
        const float* mid = x + (size_t)r * cols;
        const float* dn = x + (size_t)clampi(r + 1, 0, rows - 1) * cols;
        double acc[44];
        float accf[44];
        for (int k = 0; k < 44; k++) { acc[k] = 0.0; accf[k] = 0.0f; }
        for (int c = 0; c < cols; c++) {
            const int cm = c > 0 ? c - 1 : 0;
            const int cp = c < cols - 1 ? c + 1 : cols - 1;
            float n[8];
            n[0] = up[cm];  n[1] = up[c];  n[2] = up[cp];
            n[3] = mid[cm];                n[4] = mid[cp];
            n[5] = dn[cm];  n[6] = dn[c];  n[7] = dn[cp];
            const float ctr = mid[c];
            int k = 0;
            for (int i = 0; i < 8; i++) {
                for (int j = i; j < 8; j++, k++) {
                    if (fp16 || accum_f32) {
                        float pr = n[i] * n[j];
                        if (fp16) pr = round_to_half(pr);
                        if (accum_f32) accf[k] += pr; else acc[k] += (double)pr;
                    } else {
                        acc[k] += (double)n[i] * (double)n[j];
                    }
                }
            }
            for (int i = 0; i < 8; i++) {
                if (fp16 || accum_f32) {
                    float pr = n[i] * ctr;
                    if (fp16) pr = round_to_half(pr);
                    if (accum_f32) accf[36 + i] += pr; else acc[36 + i] += (double)pr;
                } else {
                    acc[36 + i] += (double)n[i] * (double)ctr;
                }
            }
        }
        for (int k = 0; k < 44; k++) rowacc[(size_t)r * 44 + k] = accum_f32 ? (double)accf[k] : acc[k];
    }
    double tot[44];
    float totf[44];
    for (int k = 0; k < 44; k++) { tot[k] = 0.0; totf[k] = 0.0f; }
    for (int r = 0; r < rows; r++)
        for (int k = 0; k < 44; k++) {
            if (accum_f32) totf[k] += (float)rowacc[(size_t)r * 44 + k];
            else tot[k] += rowacc[(size_t)r * 44 + k];
        }
    free(rowacc);
    int k = 0;
    for (int i = 0; i < 8; i++)
        for (int j = i; j < 8; j++, k++) {
            const double v = accum_f32 ? (double)totf[k] : tot[k];
            Rx[i * 8 + j] = v;
            Rx[j * 8 + i] = v;
        }
    for (int i = 0; i < 8; i++) rx[i] = accum_f32 ? (double)totf[36 + i] : tot[36 + i];
    return WMO_OK;
}

/* pairwise (tree) sum of n f32 values with stride `stride` */
static float tree_sum_f32(const float* v, size_t n, size_t stride)
{
    if (n == 1) return v[0];
    if (n == 2) return v[0] + v[stride];
    const size_t h = n / 2;
    return tree_sum_f32(v, h, stride) + tree_sum_f32(v + h * stride, n - h, stride);
}

/*
 * The Gram sums in the reference's arithmetic (wmo_opts.ref_arith): me_p3.hpp:23-83 launches work groups of 64 threads
 * along a row (cl::NDRange(64, 1), Watermark.cpp:190); every thread rounds its 36 + 8 products to half (vstore_half8),
 * thread `localId` then sums column RxMappings[localId] of the 64 x 36 table in f32 in thread order i = 0..63
 * (me_p3.hpp:61-67,76-82; threads beyond the image width contribute the zeros of the table's initialisation), and
 * af::sum folds the per-work-group partials (Watermark.cpp:148-149).
 */
static int gram_ref_arith(const float* x, int rows, int cols, double Rx[64], double rx[8])
{
    const int ng = (cols + 63) / 64;
    const size_t np = (size_t)rows * ng;
    float* part = (float*)malloc(np * 44 * sizeof(float));
    if (!part) return WMO_BAD_ARG;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        const float* up = x + (size_t)clampi(r - 1, 0, rows - 1) * cols;
        const float* mid = x + (size_t)r * cols;
        const float* dn = x + (size_t)clampi(r + 1, 0, rows - 1) * cols;
        for (int g = 0; g < ng; g++) {
            float acc[44];
            for (int k = 0; k < 44; k++) acc[k] = 0.0f;
            for (int i = 0; i < 64; i++) {
                const int c = 64 * g + i;
                if (c >= cols) break;
                const int cm = c > 0 ? c - 1 : 0;
                const int cp = c < cols - 1 ? c + 1 : cols - 1;
                float n[8];
                n[0] = up[cm];  n[1] = up[c];  n[2] = up[cp];
                n[3] = mid[cm];                n[4] = mid[cp];
                n[5] = dn[cm];  n[6] = dn[c];  n[7] = dn[cp];
                int k = 0;
                for (int a = 0; a < 8; a++)
                    for (int b = a; b < 8; b++, k++) acc[k] += round_to_half(n[a] * n[b]);
                for (int a = 0; a < 8; a++) acc[36 + a] += round_to_half(n[a] * mid[c]);
            }
            memcpy(part + ((size_t)r * ng + g) * 44, acc, sizeof(acc));
        }
    }
    int k = 0;
    for (int i = 0; i < 8; i++)
        for (int j = i; j < 8; j++, k++) {
            const double v = (double)tree_sum_f32(part + k, np, 44);
            Rx[i * 8 + j] = v;
            Rx[j * 8 + i] = v;
        }
    for (int i = 0; i < 8; i++) rx[i] = (double)tree_sum_f32(part + 36 + i, np, 44);
    free(part);
    return WMO_OK;
}

/* af::solve (Watermark.cpp:203) in f32: LU with partial pivoting, same unsolvable rule as wmo_solve */
int wmo_solve_f32(const double Rx[64], const double rx[8], float c[8])
{
    float A[8][9];
    float amax = 0.0f;
    for (int i = 0; i < 8; i++) {
        for (int j = 0; j < 8; j++) {
            A[i][j] = (float)Rx[i * 8 + j];
            if (fabsf(A[i][j]) > amax) amax = fabsf(A[i][j]);
        }
        A[i][8] = (float)rx[i];
    }
    for (int i = 0; i < 8; i++) c[i] = 0.0f;
    if (!(amax > 0.0f) || !isfinite(amax)) return WMO_UNSOLVABLE;
    const float tiny = 1e-12f * amax;
    for (int k = 0; k < 8; k++) {
        int piv = k;
        float pmax = fabsf(A[k][k]);
        for (int i = k + 1; i < 8; i++)
            if (fabsf(A[i][k]) > pmax) { pmax = fabsf(A[i][k]); piv = i; }
        if (!(pmax > tiny)) return WMO_UNSOLVABLE;
        if (piv != k)
            for (int j = 0; j < 9; j++) { float t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
        for (int i = k + 1; i < 8; i++) {
            const float f = A[i][k] / A[k][k];
            for (int j = k; j < 9; j++) A[i][j] -= f * A[k][j];
        }
    }
    float sol[8];
    for (int i = 7; i >= 0; i--) {
        float sacc = A[i][8];
        for (int j = i + 1; j < 8; j++) sacc -= A[i][j] * sol[j];
        sol[i] = sacc / A[i][i];
    }
    for (int i = 0; i < 8; i++) {
        if (!isfinite(sol[i])) { for (int j = 0; j < 8; j++) c[j] = 0.0f; return WMO_UNSOLVABLE; }
        c[i] = sol[i];
    }
    return WMO_OK;
}

/*
 * coefficients = solve(Rx, rx)   (Watermark.cpp:203, af::solve = LU with partial pivoting).
 * The reference relies on af::solve throwing for an unsolvable system
 * (Watermark.cpp:201-208); the build DEFINES unsolvable as: a pivot magnitude below
 * 1e-12 * max|Rx|, or any non-finite coefficient (SURVEY.md section 7 "hard parts").
 * f64 LU; coefficients returned as f32 (the reference's coefficient array is f32).
 */
int wmo_solve(const double Rx[64], const double rx[8], float c[8])
{
    double A[8][9];
    double amax = 0.0;
    for (int i = 0; i < 8; i++) {
        for (int j = 0; j < 8; j++) {
            A[i][j] = Rx[i * 8 + j];
            if (fabs(A[i][j]) > amax) amax = fabs(A[i][j]);
        }
        A[i][8] = rx[i];
    }
    for (int i = 0; i < 8; i++) c[i] = 0.0f;
    if (!(amax > 0.0) || !isfinite(amax)) return WMO_UNSOLVABLE;
    const double tiny = 1e-12 * amax;
    for (int k = 0; k < 8; k++) {
        int piv = k;
        double pmax = fabs(A[k][k]);
        for (int i = k + 1; i < 8; i++)
            if (fabs(A[i][k]) > pmax) { pmax = fabs(A[i][k]); piv = i; }
        if (!(pmax > tiny)) return WMO_UNSOLVABLE;
        if (piv != k)
            for (int j = 0; j < 9; j++) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
        for (int i = k + 1; i < 8; i++) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j < 9; j++) A[i][j] -= f * A[k][j];
        }
    }
    double sol[8];
    for (int i = 7; i >= 0; i--) {
        double s = A[i][8];
        for (int j = i + 1; j < 8; j++) s -= A[i][j] * sol[j];
        sol[i] = s / A[i][i];
    }
    for (int i = 0; i < 8; i++) {
        if (!isfinite(sol[i])) { for (int j = 0; j < 8; j++) c[j] = 0.0f; return WMO_UNSOLVABLE; }
        c[i] = (float)sol[i];
    }
    return WMO_OK;
}

/*
 * dot(r,c) = sum_k coeffs[k] * neighbour_k, sequential f32 accumulation in tap order
 * (scaled_neighbors_p3.hpp:35-42; built with -cl-mad-enable, main.cpp:108 => fmaf).
 * Output row-major (the reference's column-major store is AF-internal layout only).
 */
void wmo_scaled_neighbors(const float* x, int rows, int cols, const float c[8], float* out)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        const float* up = x + (size_t)clampi(r - 1, 0, rows - 1) * cols;
        const float* mid = x + (size_t)r * cols;
        const float* dn = x + (size_t)clampi(r + 1, 0, rows - 1) * cols;
        for (int cc = 0; cc < cols; cc++) {
            const int cm = cc > 0 ? cc - 1 : 0;
            const int cp = cc < cols - 1 ? cc + 1 : cols - 1;
            float dot = 0.0f;
            dot = fmaf(c[0], up[cm], dot);
            dot = fmaf(c[1], up[cc], dot);
            dot = fmaf(c[2], up[cp], dot);
            dot = fmaf(c[3], mid[cm], dot);
            dot = fmaf(c[4], mid[cp], dot);
            dot = fmaf(c[5], dn[cm], dot);
            dot = fmaf(c[6], dn[cc], dot);
            dot = fmaf(c[7], dn[cp], dot);
            out[(size_t)r * cols + cc] = dot;
        }
    }
}

/* e = x - scaled_neighbors(x; c)     (Watermark.cpp:210,224) */
void wmo_error_sequence(const float* x, int rows, int cols, const float c[8], float* e)
{
    wmo_scaled_neighbors(x, rows, cols, c, e);
    const size_t n = (size_t)rows * cols;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) e[i] = x[i] - e[i];
}

/*
 * NVF mask (nvf.hpp:37-50): p x p replicate-padded window, row-major tap order,
 * sum += v; sumSq = fma(v, v, sumSq) (-cl-mad-enable, main.cpp:106); mean = sum / p^2;
 * variance = sumSq / p^2 - mean * mean (two roundings, not fused); out = variance / (1 + variance).
 */
int wmo_nvf_mask(const float* x, int rows, int cols, int p, float* m)
{
    if (p != 3 && p != 5 && p != 7 && p != 9) return WMO_BAD_ARG; /* Watermark.cpp:24-25 */
    const int pad = p / 2;
    const float psq = (float)(p * p);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < cols; c++) {
            float sum = 0.0f, sumsq = 0.0f;
            for (int i = -pad; i <= pad; i++) {
                const float* row = x + (size_t)clampi(r + i, 0, rows - 1) * cols;
                for (int j = -pad; j <= pad; j++) {
                    const float v = row[clampi(c + j, 0, cols - 1)];
                    sum += v;
                    sumsq = fmaf(v, v, sumsq);
                }
            }
            const float mean = sum / psq;
            const float var = (sumsq / psq) - (mean * mean);
            m[(size_t)r * cols + c] = var / (1.0f + var);
        }
    }
    return WMO_OK;
}

/*
 * computePredictionErrorMask (Watermark.cpp:176-218): coefficients, error sequence and
 * (optionally) m = |e| / max|e|.  e, m may be NULL.  Returns WMO_UNSOLVABLE for a
 * singular system (reference: empty coefficients, Watermark.cpp:205-208).
 */
int wmo_me_mask(const float* x, int rows, int cols, float c[8], float* e, float* m, float* maxabs_out,
                const wmo_opts* opt)
{
    double Rx[64], rx[8];
    int st = wmo_gram(x, rows, cols, Rx, rx, opt);
    if (st != WMO_OK) return st;
    st = (opt && opt->ref_arith) ? wmo_solve_f32(Rx, rx, c) : wmo_solve(Rx, rx, c);
    if (st != WMO_OK) return st;
    const size_t n = (size_t)rows * cols;
    float* etmp = e ? e : (float*)malloc(n * sizeof(float));
    if (!etmp) return WMO_BAD_ARG;
    wmo_error_sequence(x, rows, cols, c, etmp);
    if (m || maxabs_out) {
        float mx = 0.0f; /* af::max<float>(abs(e)), Watermark.cpp:213-214 */
        for (size_t i = 0; i < n; i++) { const float a = fabsf(etmp[i]); if (a > mx) mx = a; }
        if (maxabs_out) *maxabs_out = mx;
        if (m) {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) m[i] = fabsf(etmp[i]) / mx;
        }
    }
    if (!e) free(etmp);
    return WMO_OK;
}

/* ||v||_2 with f64 accumulation in row order (af::norm, Watermark.cpp:170,230) */
static double norm2_rows(const float* v, int rows, int cols)
{
    double* rowacc = (double*)malloc((size_t)rows * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        double s = 0.0;
        const float* p = v + (size_t)r * cols;
        for (int c = 0; c < cols; c++) s += (double)p[c] * (double)p[c];
        rowacc[r] = s;
    }
    double tot = 0.0;
    for (int r = 0; r < rows; r++) tot += rowacc[r];
    free(rowacc);
    return sqrt(tot);
}

static double dot_rows(const float* a, const float* b, int rows, int cols)
{
    double* rowacc = (double*)malloc((size_t)rows * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        double s = 0.0;
        const float* p = a + (size_t)r * cols;
        const float* q = b + (size_t)r * cols;
        for (int c = 0; c < cols; c++) s += (double)p[c] * (double)q[c];
        rowacc[r] = s;
    }
    double tot = 0.0;
    for (int r = 0; r < rows; r++) tot += rowacc[r];
    free(rowacc);
    return tot;
}

/*
 * makeWatermark (Watermark.cpp:156-172).  gray: [rows,cols] f32 (the mask source);
 * base: channels planar planes [channels][rows][cols] f32 (grey or RGB, main.cpp:169-190);
 * out: same shape as base.  out may alias base.  On WMO_UNSOLVABLE out = base bit-exact and
 * *a is left untouched (Watermark.cpp:164-165).
 * u = mask * W; a = sF / (float)(norm(u) / sqrt(N)); out = clamp(base + u * a, 0, 255) with
 * the multiply-add fused (pinned choice, see header).
 */
int wmo_embed(const float* gray, const float* base, int channels, const float* W, int rows, int cols, int p,
              float psnr, int mask, float* out, float* a_out, float* mask_out, const wmo_opts* opt)
{
    if (!gray || !base || !W || !out || channels < 1) return WMO_BAD_ARG;
    if (p != 3 && p != 5 && p != 7 && p != 9) return WMO_BAD_ARG;
    if (mask == WMO_MASK_ME && p != 3) return WMO_BAD_ARG; /* main.cpp:89 */
    const size_t n = (size_t)rows * cols;
    float* m = (float*)malloc(n * sizeof(float));
    if (!m) return WMO_BAD_ARG;
    int st;
    if (mask == WMO_MASK_ME) {
        float c[8];
        st = wmo_me_mask(gray, rows, cols, c, NULL, m, NULL, opt);
    } else {
        st = wmo_nvf_mask(gray, rows, cols, p, m);
    }
    if (st != WMO_OK) {
        if (st == WMO_UNSOLVABLE && out != base) memcpy(out, base, n * channels * sizeof(float));
        free(m);
        return st;
    }
    if (mask_out) memcpy(mask_out, m, n * sizeof(float));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) m[i] = m[i] * W[i]; /* u */
    const double nrm = norm2_rows(m, rows, cols);
    const float a = wmo_strength_factor(psnr) / (float)(nrm / sqrt((double)n));
    if (a_out) *a_out = a;
    for (int ch = 0; ch < channels; ch++) {
        const float* b = base + (size_t)ch * n;
        float* o = out + (size_t)ch * n;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) {
            float y = fmaf(m[i], a, b[i]);
            y = y < 0.0f ? 0.0f : y;
            y = y > 255.0f ? 255.0f : y;
            o[i] = y;
        }
    }
    free(m);
    return WMO_OK;
}

/*
 * detectWatermark (Watermark.cpp:234-250) + computeErrorSequence (:221-225) +
 * computeCorrelation (:228-231).  Returns corr = 0.0f with WMO_UNSOLVABLE for a singular system.
 */
int wmo_detect(const float* img, const float* W, int rows, int cols, int p, int mask, float* corr_out,
               const wmo_opts* opt)
{
    if (!img || !W || !corr_out) return WMO_BAD_ARG;
    if (p != 3 && p != 5 && p != 7 && p != 9) return WMO_BAD_ARG;
    const size_t n = (size_t)rows * cols;
    float c[8];
    float* ew = (float*)malloc(n * sizeof(float));
    float* m = (float*)malloc(n * sizeof(float));
    float* eu = (float*)malloc(n * sizeof(float));
    if (!ew || !m || !eu) { free(ew); free(m); free(eu); return WMO_BAD_ARG; }
    int st;
    if (mask == WMO_MASK_NVF) {
        st = wmo_me_mask(img, rows, cols, c, ew, NULL, NULL, opt);
        if (st == WMO_OK) st = wmo_nvf_mask(img, rows, cols, p, m);
    } else {
        st = wmo_me_mask(img, rows, cols, c, ew, m, NULL, opt);
    }
    if (st != WMO_OK) {
        *corr_out = 0.0f; /* Watermark.cpp:246-247 */
        free(ew); free(m); free(eu);
        return st;
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) m[i] = m[i] * W[i]; /* u (Watermark.cpp:248) */
    wmo_error_sequence(m, rows, cols, c, eu);          /* e_u, same coefficients (Watermark.cpp:249) */
    const double d = dot_rows(eu, ew, rows, cols);
    const double nz = norm2_rows(ew, rows, cols);
    const double nu = norm2_rows(eu, rows, cols);
    *corr_out = (float)d / (float)(nz * nu);
    free(ew); free(m); free(eu);
    return WMO_OK;
}

/*
 * Video-frame contract (main.cpp:355-357,379-381,405): Y plane u8 row-major -> f32,
 * makeWatermark(frame, frame, ME), result cast to u8 by truncation.  out may alias in.
 */
int wmo_embed_u8(const uint8_t* in, const float* W, int rows, int cols, int p, float psnr, int mask,
                 uint8_t* out, float* a_out, const wmo_opts* opt)
{
    const size_t n = (size_t)rows * cols;
    float* f = (float*)malloc(n * sizeof(float));
    float* o = (float*)malloc(n * sizeof(float));
    if (!f || !o) { free(f); free(o); return WMO_BAD_ARG; }
    for (size_t i = 0; i < n; i++) f[i] = (float)in[i];
    int st = wmo_embed(f, f, 1, W, rows, cols, p, psnr, mask, o, a_out, NULL, opt);
    if (st == WMO_OK) for (size_t i = 0; i < n; i++) out[i] = (uint8_t)o[i];
    else if (st == WMO_UNSOLVABLE && out != in) memcpy(out, in, n);
    free(f); free(o);
    return st;
}

int wmo_detect_u8(const uint8_t* img, const float* W, int rows, int cols, int p, int mask, float* corr_out,
                  const wmo_opts* opt)
{
    const size_t n = (size_t)rows * cols;
    float* f = (float*)malloc(n * sizeof(float));
    if (!f) return WMO_BAD_ARG;
    for (size_t i = 0; i < n; i++) f[i] = (float)img[i];
    int st = wmo_detect(f, W, rows, cols, p, mask, corr_out, opt);
    free(f);
    return st;
}

/* rgb2gray with the harness weights (main.cpp:142-144,154), f32 planar RGB in [0,255] */
void wmo_rgb2gray(const float* rgb_planar, int rows, int cols, float* gray)
{
    const size_t n = (size_t)rows * cols;
    for (size_t i = 0; i < n; i++)
        gray[i] = 0.299f * rgb_planar[i] + 0.587f * rgb_planar[n + i] + 0.114f * rgb_planar[2 * n + i];
}
