// wm_kernels.hip -- hand-written gfx950 kernels of the watermark hot path + their launchers.
//
// Kernel map (reference function -> kernel), see DESIGN.md for bytes/roofline per kernel:
//   me kernel + af::sum partial folding (me_p3.hpp:23-83, Watermark.cpp:140-151)  -> k_gram
//   af::solve (Watermark.cpp:203)                                                 -> k_solve
//   scaled_neighbors + sub + abs + max + mask*W + norm (Watermark.cpp:210-214,169-170) -> k_me_stats, k_embed_scalars
//   u*a + base, clamp (Watermark.cpp:171)                                          -> k_embed_me / k_embed_nvf
//   nvf kernel (nvf.hpp:5-51)                                                      -> k_nvf_stats / k_embed_nvf / k_mask_nvf
//   detect: 2x scaled_neighbors, mask*W, dot, 2x norm (Watermark.cpp:221-250)      -> k_detect, k_corr_finalize
//
// All kernels share the strip-march execution shape of wm_device.hpp.  Every global sum is a
// fixed-order two-stage reduction (per-thread f32 over <= rps*4 pixels -> f64 per wave -> f64 per
// block -> f64 in the finalising kernel): no atomics, bitwise deterministic run to run.
// Compiled with -ffp-contract=off: fused multiply-adds appear only where fmaf() is written, which
// pins the same operation order as oracle/wm_oracle.c.
#include "wm_march.hpp"

namespace wmk {

// =================================================================================================
// k_gram: Gram matrix of the 3x3 neighbourhood in exact arithmetic, lag-product formulation.
//
//   T(u,v) = sum_{p in I} X(p+u) X(p+v)        X = replicate-padded image, u,v in {-1,0,1}^2
//          = sum_{q in I+u} X(q) X(q+d)        d = v-u, made lexicographically >= 0 by swapping u,v
//          = M[lag(d)] + B[t]
//   M[l] = sum_{q in Core} x(q) x(q+d_l)       13 lags, Core = {1<=r<=R-3, 2<=c<=C-3}: inside every
//                                              shifted rectangle I+u and free of clamping
//   B[t] = sum_{q in (I+u_t) \ Core} X(q) X(q+d_t)   a frame of <= 5 rows and 6 columns, per term t
//
// The 36 unique Rx entries and the 8 rx entries (me_p3.hpp:8-21, Watermark.hpp:29-39) are the 44 terms.
// Main blocks march the strips accumulating the 13 lag products per pixel with f64 FMAs (exact
// products of f32/u8 pixels, 13 instead of 44 multiply-adds per pixel); the `nbb` extra blocks of the
// same launch (placed first in the grid) evaluate the border frame.  tests/lag_gram_model.py is the numpy model of this split.
// =================================================================================================
__host__ __device__ constexpr int nb_dr(int i) { return i < 3 ? -1 : (i < 5 ? 0 : 1); }
__host__ __device__ constexpr int nb_dc(int i) { return i == 0 || i == 3 || i == 5 ? -1 : (i == 1 || i == 6 ? 0 : 1); }
struct GramTerm { int ur, uc, lag; };
__host__ __device__ constexpr GramTerm gram_term(int t)
{
    int i = 0, j = 0, ur = 0, uc = 0, vr = 0, vc = 0;
    if (t < 36) {
        int k = t;
        i = 0;
        while (k >= 8 - i) { k -= 8 - i; ++i; }
        j = i + k;
        ur = nb_dr(i); uc = nb_dc(i); vr = nb_dr(j); vc = nb_dc(j);
    } else {
        i = t - 36;
        ur = nb_dr(i); uc = nb_dc(i); vr = 0; vc = 0;
    }
    int dr = vr - ur, dc = vc - uc;
    if (dr < 0 || (dr == 0 && dc < 0)) { ur = vr; uc = vc; dr = -dr; dc = -dc; }
    const int lag = dr == 0 ? dc : (dr == 1 ? 3 + dc + 2 : 8 + dc + 2);
    return GramTerm{ur, uc, lag};
}
__host__ __device__ constexpr int lag_dr(int l) { return l < 3 ? 0 : (l < 8 ? 1 : 2); }
__host__ __device__ constexpr int lag_dc(int l) { return l < 3 ? l : (l < 8 ? l - 3 - 2 : l - 8 - 2); }

template <typename T>
__device__ __forceinline__ double padded(const T* __restrict__ x, long long pitch, int R, int C, int r, int c)
{
    return (double)x[(long long)clampi(r, 0, R - 1) * pitch + clampi(c, 0, C - 1)];
}

// q rows [rs, re) of one strip: stream rows rs .. re+1; f64 window of rows q, q+1, q+2 and columns
// c0-2 .. c0+5 in rotating slots (slot of stream row i = i % 3)
template <typename T, bool VEC>
__device__ __forceinline__ void gram_march(const T* __restrict__ xf, long long pitch, const Geom& g, const WaveJob& j,
                                           float* lds, double (&acc)[13])
{
    const int R = g.rows, C = g.cols;
    XMarch<T, 1, 2, 1, VEC, PFX> xm;
    const int n = j.re - j.rs + 2;
    xm.start(xf, pitch, g, j, lds, j.rs, n);
    const int c0 = j.c0s + 4 * j.lane;
    double w[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) w[a][b] = 0.0;
    // column validity is row-invariant: pixels outside the core contribute with a zero factor (no branch)
    bool cv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cv[k] = c0 + k >= 2 && c0 + k <= C - 3;
    march<2>(n, [&](int i, auto qc, auto emit) {
        constexpr int Q = decltype(qc)::value;
        xm.template step<Q>(i);
#pragma unroll
        for (int b = 0; b < 8; ++b) w[Q % 3][b] = (double)xm.win[0][2 + b];
        if (decltype(emit)::value) {
            const int r = j.rs + i - 2;  // q row: its window rows are slots (Q+1)%3, (Q+2)%3, Q%3
            // rows outside the core contribute with a zero factor too: the accumulation stays branch-free
            // (a branch here makes the compiler copy all 13 f64 accumulators at every step)
            const bool rowok = r >= 1 && r <= R - 3;
            const double* w0 = w[(Q + 1) % 3];
            const double* w1 = w[(Q + 2) % 3];
            const double* w2 = w[Q % 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double xq = (cv[k] && rowok) ? w0[2 + k] : 0.0;
                acc[0] = fma(xq, w0[2 + k], acc[0]);
                acc[1] = fma(xq, w0[3 + k], acc[1]);
                acc[2] = fma(xq, w0[4 + k], acc[2]);
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    acc[3 + b] = fma(xq, w1[k + b], acc[3 + b]);
                    acc[8 + b] = fma(xq, w2[k + b], acc[8 + b]);
                }
            }
        }
    });
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_gram(const T* __restrict__ x, long long pitch, long long fstride, Geom g,
                                                int nblk, int nbb, int aligned, double* __restrict__ pmain,
                                                double* __restrict__ pborder)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    __shared__ double s_red[WPB][NGRAM];
    const int frame = blockIdx.y;
    const int R = g.rows, C = g.cols;
    const bool core_empty = R < 4 || C < 5;
    const T* xf = x + (long long)frame * fstride;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;

    if ((int)blockIdx.x < nbb) {
        // ---------------- border frame (first blocks of the grid: few, latency-bound, overlap the march) ----------------
        const int bb = blockIdx.x;
        double acc[NGRAM];
#pragma unroll
        for (int t = 0; t < NGRAM; ++t) acc[t] = 0.0;
        const int nfull = core_empty ? R + 2 : 5;
        const int nel = nfull * (C + 2) + (core_empty ? 0 : 6 * (R - 3));
        for (int e = bb * BLOCK + (int)threadIdx.x; e < nel; e += nbb * BLOCK) {
            int r, c;
            if (e < nfull * (C + 2)) {
                const int k = e / (C + 2);
                c = e - k * (C + 2) - 1;
                r = core_empty ? k - 1 : (k == 0 ? -1 : (k == 1 ? 0 : R - 2 + (k - 2)));
            } else {
                const int e2 = e - nfull * (C + 2);
                const int rr = e2 / 6;
                r = 1 + rr;
                const int sidx = e2 - rr * 6;
                c = sidx < 3 ? sidx - 1 : C - 2 + (sidx - 3);
            }
            const double xq = padded(xf, pitch, R, C, r, c);
            double prod[13];
#pragma unroll
            for (int l = 0; l < 13; ++l) prod[l] = xq * padded(xf, pitch, R, C, r + lag_dr(l), c + lag_dc(l));
            // q belongs to the shifted rectangle I+u iff ur <= r <= R-1+ur and uc <= c <= C-1+uc; u in {-1,0,1}^2
            const bool rin[3] = {r <= R - 2, r >= 0 && r <= R - 1, r >= 1};
            const bool cin[3] = {c <= C - 2, c >= 0 && c <= C - 1, c >= 1};
#pragma unroll
            for (int t = 0; t < NGRAM; ++t) {
                const GramTerm gt = gram_term(t);
                acc[t] += (rin[gt.ur + 1] && cin[gt.uc + 1]) ? prod[gt.lag] : 0.0;
            }
        }
#pragma unroll
        for (int t = 0; t < NGRAM; ++t) {
            const double s = wave_sum(acc[t]);
            if (lane == 0) s_red[wave][t] = s;
        }
        __syncthreads();
        if (threadIdx.x < NGRAM)
            pborder[((long long)frame * nbb + bb) * NGRAM + threadIdx.x] =
                ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
        return;
    }

    // ---------------- main: 13 lag sums over the core ----------------
    const int mb = blockIdx.x - nbb;  // march block id
    const WaveJob j = make_job(g, nblk, mb);
    double acc[13];
#pragma unroll
    for (int l = 0; l < 13; ++l) acc[l] = 0.0;
    if (j.valid && !core_empty) {
        if (aligned && j.full) gram_march<T, true>(xf, pitch, g, j, s_row[j.wave], acc);
        else gram_march<T, false>(xf, pitch, g, j, s_row[j.wave], acc);
    }
#pragma unroll
    for (int l = 0; l < 13; ++l) {
        const double s = wave_sum(acc[l]);
        if (j.lane == 0) s_red[j.wave][l] = s;
    }
    __syncthreads();
    if (threadIdx.x < 13)
        pmain[((long long)frame * nblk + mb) * 13 + threadIdx.x] =
            ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
}

// =================================================================================================
// k_solve: fold the block partials (f64), 8x8 LU with partial pivoting in f64, coefficients as f32
// =================================================================================================
constexpr int SOLVE_THREADS = 1024;
constexpr int SOLVE_GM = SOLVE_THREADS / 13;     // 78 groups for the 13 lag sums
constexpr int SOLVE_GB = SOLVE_THREADS / NGRAM;  // 23 groups for the 44 border terms

__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(const double* __restrict__ pmain, int nblk,
                                                         const double* __restrict__ pborder, int nbb,
                                                         float* __restrict__ coef, int* __restrict__ status,
                                                         double* __restrict__ gram_tot)
{
    __shared__ double s_pm[SOLVE_GM][13];
    __shared__ double s_pb[SOLVE_GB][NGRAM];
    __shared__ double s_m[13];
    __shared__ double s_tot[NGRAM];
    __shared__ double A[8][9];
    const int frame = blockIdx.x;
    const int t = threadIdx.x;
    if (t < SOLVE_GM * 13) {
        const int k = t % 13, gq = t / 13;
        const double* p = pmain + (long long)frame * nblk * 13 + k;
        double s = 0.0;
        for (int b = gq; b < nblk; b += SOLVE_GM) s += p[(long long)b * 13];
        s_pm[gq][k] = s;
    }
    if (t < SOLVE_GB * NGRAM) {
        const int k = t % NGRAM, gq = t / NGRAM;
        const double* p = pborder + (long long)frame * nbb * NGRAM + k;
        double s = 0.0;
        for (int b = gq; b < nbb; b += SOLVE_GB) s += p[(long long)b * NGRAM];
        s_pb[gq][k] = s;
    }
    __syncthreads();
    if (t < 13) {
        double s = 0.0;
        for (int q = 0; q < SOLVE_GM; ++q) s += s_pm[q][t];
        s_m[t] = s;
    }
    __syncthreads();
    if (t < NGRAM) {
        double s = 0.0;
        for (int q = 0; q < SOLVE_GB; ++q) s += s_pb[q][t];
        int lag = 0;
#pragma unroll
        for (int tt = 0; tt < NGRAM; ++tt)
            if (tt == t) lag = gram_term(tt).lag;
        s += s_m[lag];
        s_tot[t] = s;
        gram_tot[(long long)frame * NGRAM + t] = s;
    }
    __syncthreads();
    if (t >= WAVE) return;  // one wave does the LU; LDS traffic below is ordered by wave_lds_fence
    {
        // unpack the 36 upper-triangle sums into the symmetric 8x8 (Watermark.hpp:29-39) + rhs
        const int i = t >> 3, jj = t & 7;
        const int a = i < jj ? i : jj, b = i < jj ? jj : i;
        const int idx = a * 8 - (a * (a - 1)) / 2 + (b - a);
        A[i][jj] = s_tot[idx];
        if (jj == 0) A[i][8] = s_tot[36 + i];
    }
    wave_lds_fence();
    double amax = 0.0;
    for (int i = 0; i < 8; ++i)
        for (int jj = 0; jj < 8; ++jj) amax = fmax(amax, fabs(A[i][jj]));
    bool singular = !(amax > 0.0) || !isfinite(amax);
    const double tiny = 1e-12 * amax;
    for (int k = 0; k < 8 && !singular; ++k) {
        int piv = k;
        double pmax = fabs(A[k][k]);
        for (int i = k + 1; i < 8; ++i) {
            const double v = fabs(A[i][k]);
            if (v > pmax) { pmax = v; piv = i; }
        }
        if (!(pmax > tiny)) { singular = true; break; }
        wave_lds_fence();
        if (piv != k && t < 9) {
            const double tmp = A[k][t];
            A[k][t] = A[piv][t];
            A[piv][t] = tmp;
        }
        wave_lds_fence();
        const int i = k + 1 + t / 9, jj = t % 9;
        double f = 0.0, akj = 0.0, aij = 0.0;
        const bool act = i < 8 && jj >= k;
        if (act) {
            f = A[i][k] / A[k][k];
            akj = A[k][jj];
            aij = A[i][jj];
        }
        wave_lds_fence();
        if (act) A[i][jj] = aij - f * akj;
        wave_lds_fence();
    }
    float c[8];
    if (!singular) {
        double sol[8];
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            double s = A[i][8];
#pragma unroll
            for (int jj = i + 1; jj < 8; ++jj) s -= A[i][jj] * sol[jj];
            sol[i] = s / A[i][i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!isfinite(sol[i])) singular = true;
            c[i] = (float)sol[i];
        }
    }
    if (t == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) coef[frame * 8 + i] = singular ? 0.0f : c[i];
        status[frame] = singular ? 1 : 0;
    }
}

// launchers
void launch_gram(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, double* pmain, double* pborder)
{
    WM_DISPATCH_T(x.dtype, hipLaunchKernelGGL(k_gram<T>, dim3((unsigned)(lg.nblk + lg.nbb), (unsigned)frames, 1), dim3(BLOCK), 0,
                                               s, (const T*)x.p, x.pitch, x.fstride, geom_of(lg), lg.nblk, lg.nbb, x.aligned,
                                               pmain, pborder));
}

void launch_solve(hipStream_t s, const LaunchGeom& lg, int frames, const double* pmain, const double* pborder, float* coef,
                  int* status, double* gram_tot)
{
    hipLaunchKernelGGL(k_solve, dim3(frames), dim3(SOLVE_THREADS), 0, s, pmain, lg.nblk, pborder, lg.nbb, coef, status,
                       gram_tot);
}

}  // namespace wmk
