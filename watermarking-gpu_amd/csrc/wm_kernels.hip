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
#include "wm_kernels.hpp"
#include "wm_device.hpp"

namespace wmk {

// rolling window over the x row stream: NR rows of (4 + 8*HC) columns per lane.
// NR == 3 (and 1): rows live in rotating slots (row of stream index i sits in slot i % 3); with the 6x
// unrolled march every slot index is a compile-time constant, so the window never moves registers.
// Other NR (NVF p > 3): rows are kept in order and shifted.
template <typename T, int HC, int NR, bool VEC>
struct XMarch {
    static constexpr int WN = 4 + 8 * HC;
    static constexpr bool ROT = (NR == 3 || NR == 1);
    XStream<T, HC, VEC> xs;
    typename XStream<T, HC, VEC>::Raw pre[PF];
    float win[NR][WN];
    float* buf;  // this wave's LDS row buffers: 2 x RowBuf<HC>::N floats
    int s0, n;

    __device__ __forceinline__ void start(const T* base, long long pitch, const Geom& g, const WaveJob& j, float* lds,
                                          int first_row, int count)
    {
        xs.init(base, pitch, g.rows, g.cols, j);
        buf = lds; s0 = first_row; n = count;
#pragma unroll
        for (int a = 0; a < NR; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) win[a][b] = 0.0f;
#pragma unroll
        for (int q = 0; q < PF; ++q)
            if (q < n) pre[q] = xs.issue(s0 + q);
    }
    // consume stream row i; q = i % UNROLL must be a compile-time constant at the call site
    __device__ __forceinline__ void step(int i, int q)
    {
        const typename XStream<T, HC, VEC>::Raw raw = pre[q % PF];
        if (i + PF < n) pre[q % PF] = xs.issue(s0 + i + PF);
        if (!ROT) {
#pragma unroll
            for (int a = 0; a + 1 < NR; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) win[a][b] = win[a + 1][b];
        }
        xs.consume(raw, buf + (q & 1) * RowBuf<HC>::N, win[ROT ? q % NR : NR - 1]);
    }
    // window row a (0 = oldest .. NR-1 = newest) after step(i, q)
    __device__ __forceinline__ const float* row(int a, int q) const { return win[ROT ? (q + 1 + a) % NR : a]; }
};

// PF-deep prefetch ring for a pointwise operand
template <typename T, bool VEC>
struct PMarch {
    PStream<T, VEC> ps;
    typename Elem<T>::vec4 pre[PF];
    int r0, n;
    __device__ __forceinline__ void start(const T* base, long long pitch, int cols, const WaveJob& j, int first_row, int count)
    {
        ps.init(base, pitch, cols, j);
        r0 = first_row; n = count;
#pragma unroll
        for (int q = 0; q < PF; ++q)
            if (q < n) pre[q] = ps.issue(r0 + q);
    }
    // value of row r0 + o (slot = o % PF must be a compile-time constant); refills the slot
    __device__ __forceinline__ float4 take(int o, int slot)
    {
        const typename Elem<T>::vec4 v = pre[slot];
        if (o + PF < n) pre[slot] = ps.issue(r0 + o + PF);
        return Elem<T>::cvt4(v);
    }
};

__device__ __forceinline__ float f4get(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }

// the march skeleton: `n` stream rows; inside, `i` is the stream row index and q = i % UNROLL a
// compile-time constant
#define WM_MARCH_BEGIN(n_)                                   \
    for (int ib_ = 0; ib_ < (n_); ib_ += UNROLL) {           \
        _Pragma("unroll") for (int q = 0; q < UNROLL; ++q) { \
            const int i = ib_ + q;                           \
            if (i < (n_)) {
#define WM_MARCH_END \
    }                \
    }                \
    }

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
    XMarch<T, 1, 1, VEC> xm;
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
    WM_MARCH_BEGIN(n)
        xm.step(i, q);
#pragma unroll
        for (int b = 0; b < 8; ++b) w[q % 3][b] = (double)xm.win[0][2 + b];
        const int r = j.rs + i - 2;  // q row: its window rows are slots (q+1)%3, (q+2)%3, q%3
        if (i >= 2 && r >= 1 && r <= R - 3) {
            const double* w0 = w[(q + 1) % 3];
            const double* w1 = w[(q + 2) % 3];
            const double* w2 = w[q % 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double xq = cv[k] ? w0[2 + k] : 0.0;
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
    WM_MARCH_END
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
        const long long nfull = core_empty ? (long long)(R + 2) : 5;
        const long long nel = nfull * (C + 2) + (core_empty ? 0 : 6LL * (R - 3));
        for (long long e = (long long)bb * BLOCK + threadIdx.x; e < nel; e += (long long)nbb * BLOCK) {
            int r, c;
            if (e < nfull * (C + 2)) {
                const int k = (int)(e / (C + 2));
                c = (int)(e % (C + 2)) - 1;
                r = core_empty ? k - 1 : (k == 0 ? -1 : (k == 1 ? 0 : R - 2 + (k - 2)));
            } else {
                const long long e2 = e - nfull * (C + 2);
                r = 1 + (int)(e2 / 6);
                const int sidx = (int)(e2 % 6);
                c = sidx < 3 ? sidx - 1 : C - 2 + (sidx - 3);
            }
            const double xq = padded(xf, pitch, R, C, r, c);
            double prod[13];
#pragma unroll
            for (int l = 0; l < 13; ++l) prod[l] = xq * padded(xf, pitch, R, C, r + lag_dr(l), c + lag_dc(l));
#pragma unroll
            for (int t = 0; t < NGRAM; ++t) {
                const GramTerm gt = gram_term(t);
                if (r >= gt.ur && r <= R - 1 + gt.ur && c >= gt.uc && c <= C - 1 + gt.uc) acc[t] += prod[gt.lag];
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

// =================================================================================================
// k_me_stats: e = x - c.nbrs;  per block: max|e| and sum (|e| W)^2
// =================================================================================================
template <typename T, bool VEC>
__device__ __forceinline__ void me_stats_march(const T* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                               const Geom& g, const WaveJob& j, float* lds, const float (&c)[8], float& mx,
                                               float& ss)
{
    XMarch<T, 1, 3, VEC> xm;
    PMarch<float, VEC> wm_;
    const int nout = j.re - j.rs, n = nout + 2;
    xm.start(xf, pitch, g, j, lds, j.rs - 1, n);
    wm_.start(W, g.cols, g.cols, j, j.rs, nout);
    const int c0 = j.c0s + 4 * j.lane;
    WM_MARCH_BEGIN(n)
        xm.step(i, q);
        if (i >= 2) {
            const float4 w = wm_.take(i - 2, (q + UNROLL - 2) % PF);
            const float* up = xm.row(0, q);
            const float* mid = xm.row(1, q);
            const float* dn = xm.row(2, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (VEC || c0 + k < g.cols) {
                    const float e = mid[4 + k] - predict<4>(up, mid, dn, k, c);
                    const float ae = fabsf(e);
                    mx = fmaxf(mx, ae);
                    const float t = ae * f4get(w, k);
                    ss = fmaf(t, t, ss);
                }
            }
        }
    WM_MARCH_END
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_me_stats(const T* __restrict__ x, long long pitch, long long fstride,
                                                    const float* __restrict__ W, Geom g, int nblk, int aligned,
                                                    const float* __restrict__ coef, const int* __restrict__ status,
                                                    float* __restrict__ pmax, double* __restrict__ pss)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    __shared__ float s_mx[WPB];
    __shared__ double s_ss[WPB];
    const int frame = blockIdx.y;
    const WaveJob j = make_job(g, nblk);
    float mx = 0.0f, ss = 0.0f;
    if (j.valid && status[frame] == 0) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = coef[frame * 8 + k];
        const T* xf = x + (long long)frame * fstride;
        if (aligned && j.full) me_stats_march<T, true>(xf, pitch, W, g, j, s_row[j.wave], c, mx, ss);
        else me_stats_march<T, false>(xf, pitch, W, g, j, s_row[j.wave], c, mx, ss);
    }
    mx = wave_max(mx);
    const double ssd = wave_sum((double)ss);
    if (j.lane == 0) { s_mx[j.wave] = mx; s_ss[j.wave] = ssd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        pmax[(long long)frame * nblk + blockIdx.x] = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
        pss[(long long)frame * nblk + blockIdx.x] = ((s_ss[0] + s_ss[1]) + s_ss[2]) + s_ss[3];
    }
}

// =================================================================================================
// NVF value of pixel k from a window of 2*PAD+1 rows (nvf.hpp:37-50): row-major taps,
// sum += v; sumSq = fma(v, v, sumSq); mean = sum / p^2; var = sumSq / p^2 - mean*mean; var / (1 + var)
// =================================================================================================
template <int PAD, int O, typename XM>
__device__ __forceinline__ float nvf_value(const XM& xm, int q, int k)
{
    float sum = 0.0f, sumsq = 0.0f;
#pragma unroll
    for (int a = 0; a < 2 * PAD + 1; ++a) {
        const float* rowp = xm.row(a, q);
#pragma unroll
        for (int b = -PAD; b <= PAD; ++b) {
            const float v = rowp[O + k + b];
            sum += v;
            sumsq = fmaf(v, v, sumsq);
        }
    }
    constexpr float psq = (float)((2 * PAD + 1) * (2 * PAD + 1));
    const float mean = sum / psq;
    const float var = (sumsq / psq) - (mean * mean);
    return var / (1.0f + var);
}

// =================================================================================================
// k_nvf_stats: per block sum (m_nvf W)^2          (p = 2*PAD+1)
// =================================================================================================
template <typename T, int PAD, bool VEC>
__device__ __forceinline__ void nvf_stats_march(const T* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                                const Geom& g, const WaveJob& j, float* lds, float& ss)
{
    constexpr int NR = 2 * PAD + 1;
    XMarch<T, 1, NR, VEC> xm;
    PMarch<float, VEC> wm_;
    const int nout = j.re - j.rs, n = nout + 2 * PAD;
    xm.start(xf, pitch, g, j, lds, j.rs - PAD, n);
    wm_.start(W, g.cols, g.cols, j, j.rs, nout);
    const int c0 = j.c0s + 4 * j.lane;
    WM_MARCH_BEGIN(n)
        xm.step(i, q);
        if (i >= 2 * PAD) {
            const float4 w = wm_.take(i - 2 * PAD, (q + 2 * UNROLL - 2 * PAD) % PF);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (VEC || c0 + k < g.cols) {
                    const float t = nvf_value<PAD, 4>(xm, q, k) * f4get(w, k);
                    ss = fmaf(t, t, ss);
                }
            }
        }
    WM_MARCH_END
}

template <typename T, int PAD>
__global__ __launch_bounds__(BLOCK) void k_nvf_stats(const T* __restrict__ x, long long pitch, long long fstride,
                                                     const float* __restrict__ W, Geom g, int nblk, int aligned,
                                                     double* __restrict__ pss)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    __shared__ double s_ss[WPB];
    const int frame = blockIdx.y;
    const WaveJob j = make_job(g, nblk);
    float ss = 0.0f;
    if (j.valid) {
        const T* xf = x + (long long)frame * fstride;
        if (aligned && j.full) nvf_stats_march<T, PAD, true>(xf, pitch, W, g, j, s_row[j.wave], ss);
        else nvf_stats_march<T, PAD, false>(xf, pitch, W, g, j, s_row[j.wave], ss);
    }
    const double ssd = wave_sum((double)ss);
    if (j.lane == 0) s_ss[j.wave] = ssd;
    __syncthreads();
    if (threadIdx.x == 0) pss[(long long)frame * nblk + blockIdx.x] = ((s_ss[0] + s_ss[1]) + s_ss[2]) + s_ss[3];
}

// =================================================================================================
// k_embed_scalars: fold stats partials -> a = sF / (float)(||u|| / sqrt(N))   (Watermark.cpp:170)
//   ME : ||u|| = sqrt(sum (|e| W)^2) / max|e|     NVF: ||u|| = sqrt(sum (m W)^2)
// =================================================================================================
__global__ __launch_bounds__(BLOCK) void k_embed_scalars(const float* __restrict__ pmax, const double* __restrict__ pss,
                                                         int nblk, const int* __restrict__ status, float sF,
                                                         double sqrt_n, EmbedScalars* __restrict__ scal,
                                                         OpResult* __restrict__ res)
{
    __shared__ float s_mx[BLOCK];
    __shared__ double s_ss[BLOCK];
    const int frame = blockIdx.x, t = threadIdx.x;
    float mx = 0.0f;
    double ss = 0.0;
    for (int b = t; b < nblk; b += BLOCK) {
        if (pmax) mx = fmaxf(mx, pmax[(long long)frame * nblk + b]);
        ss += pss[(long long)frame * nblk + b];
    }
    s_mx[t] = mx; s_ss[t] = ss;
    __syncthreads();
    for (int o = BLOCK / 2; o > 0; o >>= 1) {
        if (t < o) { s_mx[t] = fmaxf(s_mx[t], s_mx[t + o]); s_ss[t] += s_ss[t + o]; }
        __syncthreads();
    }
    if (t == 0) {
        const int st = status ? status[frame] : 0;
        EmbedScalars s;
        s.maxe = pmax ? s_mx[0] : 1.0f;
        const double nrm = pmax ? sqrt(s_ss[0]) / (double)s.maxe : sqrt(s_ss[0]);
        s.a = sF / (float)(nrm / sqrt_n);
        scal[frame] = s;
        res[frame].status = st;
        res[frame].value = s.a;
    }
}

// =================================================================================================
// k_embed: y = clamp(base + a * m * W, 0, 255) with the mask recomputed on the fly
//   MASK 0 (ME): m = |e| / max|e|;  MASK 1 (NVF): m = nvf(x)
// =================================================================================================
template <typename TX, typename TB, int NCH, int MASK, int PAD, bool VEC>
__device__ __forceinline__ void embed_march(const TX* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                            const TB* __restrict__ bptr, TB* __restrict__ optr, const PlaneDesc& base,
                                            const PlaneDesc& out, const Geom& g, const WaveJob& j, float* lds,
                                            const float (&c)[8], float a, float maxe)
{
    constexpr int NR = MASK == 0 ? 3 : 2 * PAD + 1;
    constexpr int HR = MASK == 0 ? 1 : PAD;  // halo rows above/below
    XMarch<TX, 1, NR, VEC> xm;
    PMarch<float, VEC> wm_;
    PMarch<TB, VEC> bm[NCH];
    const int nout = j.re - j.rs, n = nout + 2 * HR;
    const int c0 = j.c0s + 4 * j.lane;
    xm.start(xf, pitch, g, j, lds, j.rs - HR, n);
    wm_.start(W, g.cols, g.cols, j, j.rs, nout);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) bm[ch].start(bptr + (long long)ch * base.cstride, base.pitch, g.cols, j, j.rs, nout);
    WM_MARCH_BEGIN(n)
        xm.step(i, q);
        if (i >= 2 * HR) {
            const int o = i - 2 * HR;
            const int slot = (q + 2 * UNROLL - 2 * HR) % PF;
            const float4 w = wm_.take(o, slot);
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float m;
                if (MASK == 0) {
                    const float* mid = xm.row(1, q);
                    const float e = mid[4 + k] - predict<4>(xm.row(0, q), mid, xm.row(2, q), k, c);
                    m = fabsf(e) / maxe;  // Watermark.cpp:213-214
                } else {
                    m = nvf_value<PAD, 4>(xm, q, k);
                }
                u[k] = m * f4get(w, k);  // Watermark.cpp:169
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const float4 b = bm[ch].take(o, slot);
                float4 y;
                y.x = fminf(fmaxf(fmaf(u[0], a, b.x), 0.0f), 255.0f);
                y.y = fminf(fmaxf(fmaf(u[1], a, b.y), 0.0f), 255.0f);
                y.z = fminf(fmaxf(fmaf(u[2], a, b.z), 0.0f), 255.0f);
                y.w = fminf(fmaxf(fmaf(u[3], a, b.w), 0.0f), 255.0f);
                store4<TB, VEC>(optr + (long long)ch * out.cstride, out.pitch, j.rs + o, c0, g.cols, y);
            }
        }
    WM_MARCH_END
}

template <typename TX, typename TB, int NCH, int MASK, int PAD>
__global__ __launch_bounds__(BLOCK) void k_embed(const TX* __restrict__ x, long long pitch, long long fstride,
                                                 const float* __restrict__ W, PlaneDesc base, PlaneDesc out, Geom g,
                                                 int nblk, int aligned, const float* __restrict__ coef,
                                                 const int* __restrict__ status, const EmbedScalars* __restrict__ scal)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    const int frame = blockIdx.y;
    const WaveJob j = make_job(g, nblk);
    if (!j.valid) return;
    const TB* bptr = static_cast<const TB*>(base.p) + (long long)frame * base.fstride;
    TB* optr = static_cast<TB*>(const_cast<void*>(out.p)) + (long long)frame * out.fstride;
    const int st = MASK == 0 ? status[frame] : 0;
    if (st != 0) {
        // unsolvable: out = base bit-exact (Watermark.cpp:164-165)
        if (bptr != optr) {
            const int c0 = j.c0s + 4 * j.lane;
            for (int ch = 0; ch < NCH; ++ch)
                for (int r = j.rs; r < j.re; ++r) {
                    const TB* rb = bptr + (long long)ch * base.cstride + (long long)r * base.pitch;
                    TB* ro = optr + (long long)ch * out.cstride + (long long)r * out.pitch;
                    for (int k = 0; k < 4; ++k)
                        if (c0 + k < g.cols) ro[c0 + k] = rb[c0 + k];
                }
        }
        return;
    }
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MASK == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = coef[frame * 8 + k];
    }
    const float a = scal[frame].a;
    const float maxe = scal[frame].maxe;
    const TX* xf = x + (long long)frame * fstride;
    if (aligned && j.full) embed_march<TX, TB, NCH, MASK, PAD, true>(xf, pitch, W, bptr, optr, base, out, g, j, s_row[j.wave], c, a, maxe);
    else embed_march<TX, TB, NCH, MASK, PAD, false>(xf, pitch, W, bptr, optr, base, out, g, j, s_row[j.wave], c, a, maxe);
}

// =================================================================================================
// k_mask: materialise the mask (and the error sequence) -- parity-test building block
// =================================================================================================
template <typename T, int MASK, int PAD>
__global__ __launch_bounds__(BLOCK) void k_mask(const T* __restrict__ x, long long pitch, long long fstride, Geom g,
                                                int nblk, const float* __restrict__ coef, const int* __restrict__ status,
                                                const EmbedScalars* __restrict__ scal, PlaneDesc mo, PlaneDesc eo)
{
    constexpr int NR = MASK == 0 ? 3 : 2 * PAD + 1;
    constexpr int HR = MASK == 0 ? 1 : PAD;
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    const int frame = blockIdx.y;
    const WaveJob j = make_job(g, nblk);
    if (!j.valid) return;
    if (MASK == 0 && status[frame] != 0) return;
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float maxe = 1.0f;
    if (MASK == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = coef[frame * 8 + k];
        maxe = scal[frame].maxe;
    }
    float* mptr = static_cast<float*>(const_cast<void*>(mo.p)) + (long long)frame * mo.fstride;
    float* eptr = eo.p ? static_cast<float*>(const_cast<void*>(eo.p)) + (long long)frame * eo.fstride : nullptr;
    XMarch<T, 1, NR, false> xm;  // test helper: always the generic path
    const int nout = j.re - j.rs, n = nout + 2 * HR;
    const int c0 = j.c0s + 4 * j.lane;
    xm.start(x + (long long)frame * fstride, pitch, g, j, s_row[j.wave], j.rs - HR, n);
    WM_MARCH_BEGIN(n)
        xm.step(i, q);
        if (i >= 2 * HR) {
            float mv[4], ev[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (MASK == 0) {
                    const float* mid = xm.row(1, q);
                    ev[k] = mid[4 + k] - predict<4>(xm.row(0, q), mid, xm.row(2, q), k, c);
                    mv[k] = fabsf(ev[k]) / maxe;
                } else {
                    ev[k] = 0.0f;
                    mv[k] = nvf_value<PAD, 4>(xm, q, k);
                }
            }
            store4<float, false>(mptr, mo.pitch, j.rs + i - 2 * HR, c0, g.cols, make_float4(mv[0], mv[1], mv[2], mv[3]));
            if (MASK == 0 && eptr)
                store4<float, false>(eptr, eo.pitch, j.rs + i - 2 * HR, c0, g.cols, make_float4(ev[0], ev[1], ev[2], ev[3]));
        }
    WM_MARCH_END
}

// =================================================================================================
// k_detect: one fused sweep over the test image and W:
//   e_w = x - c.nbrs(x);  u = m W  (ME: m ~ |e_w|, the max|e_w| normalisation cancels in the
//   correlation; NVF: m = nvf(x));  e_u = u - c.nbrs(u)  with u replicate-padded;
//   per block: <e_u,e_w>, ||e_u||^2, ||e_w||^2          (Watermark.cpp:221-250)
// =================================================================================================
template <typename T, int MASK, int PAD, int HC, bool VEC>
__device__ __forceinline__ void detect_march(const T* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                             const Geom& g, const WaveJob& j, float* lds_x, float* lds_u,
                                             const float (&c)[8], float& dot, float& nu, float& nw)
{
    constexpr int HRX = MASK == 0 ? 1 : PAD;  // x rows needed above/below a u row
    constexpr int NR = 2 * HRX + 1;
    constexpr int O = 4 * HC;                 // own chunk offset in window rows
    constexpr int MID = HRX;                  // window row of the u row being produced
    const int R = g.rows, C = g.cols;
    // u rows t0..t1 are computed; x rows t0-HRX .. t1+HRX are streamed (clamped at load)
    const int t0 = j.rs > 0 ? j.rs - 1 : 0;
    const int t1 = j.re < R ? j.re : R - 1;
    const int nu_rows = t1 - t0 + 1;
    const int n = nu_rows + 2 * HRX;
    XMarch<T, HC, NR, VEC> xm;
    PMarch<float, VEC> wm_;
    xm.start(xf, pitch, g, j, lds_x, t0 - HRX, n);
    wm_.start(W, C, C, j, t0, nu_rows);
    const int c0 = j.c0s + 4 * j.lane;
    const bool left_edge = j.c0s == 0;
    const bool has_right = j.c0s + STRIP <= C - 1;  // column c0s+STRIP exists in the image
    // W at the strip's halo columns c0s-1 (lanes != 63) and c0s+STRIP (lane 63): loaded by every lane, no branch
    const int wh_col = j.lane == WAVE - 1 ? (j.c0s + STRIP < C ? j.c0s + STRIP : C - 1) : (j.c0s > 0 ? j.c0s - 1 : 0);
    float whpre[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        whpre[s] = 0.0f;
        if (s < nu_rows) whpre[s] = W[(long long)(t0 + s) * C + wh_col];
    }
    // rolling window of u rows (left neighbour, 4 own, right neighbour) in rotating slots, e_w of two rows
    float uw[3][6];
    float eww[2][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) uw[a][b] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) eww[a][b] = 0.f;
    const int last_col_local = C - 1 - j.c0s;  // strip-local index of the image's last column
    WM_MARCH_BEGIN(n)
        xm.step(i, q);
        if (i >= 2 * HRX) {
            const int o = i - 2 * HRX;  // u row index t = t0 + o; its slots: uw[q % 3], eww[q % 2]
            const int t = t0 + o;
            const int slot = (q + 2 * UNROLL - 2 * HRX) % PF;
            const float4 w = wm_.take(o, slot);
            const float wh = whpre[slot];
            if (o + PF < nu_rows) whpre[slot] = W[(long long)(t + PF) * C + wh_col];
            const float* xup = xm.row(MID - 1, q);
            const float* xmid = xm.row(MID, q);
            const float* xdn = xm.row(MID + 1, q);
            // ---- e_w and u of row t for the 4 own pixels
            float uu[4];
            float* ew = eww[q % 2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ew[k] = xmid[O + k] - predict<O>(xup, xmid, xdn, k, c);
                const float m = MASK == 0 ? fabsf(ew[k]) : nvf_value<PAD, O>(xm, q, k);
                uu[k] = m * f4get(w, k);
            }
            if (!VEC) {
                // replicate border inside the own chunk: u(c) := u(C-1) for c >= C
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (c0 + k >= C) uu[k] = uu[k - 1];
            }
            // ---- publish the u row: own chunk, strip halo columns, replicate border
            float* urow = lds_u + (q & 1) * RowBuf<1>::N;
            reinterpret_cast<float4*>(urow)[1 + j.lane] = make_float4(uu[0], uu[1], uu[2], uu[3]);
            if (j.lane == 0) {
                float uh;
                if (left_edge) uh = uu[0];
                else {
                    const float eh = xmid[O - 1] - predict<O>(xup, xmid, xdn, -1, c);
                    const float m = MASK == 0 ? fabsf(eh) : nvf_value<PAD, O>(xm, q, -1);
                    uh = m * wh;
                }
                urow[3] = uh;
            }
            if (j.lane == WAVE - 1 && has_right) {
                const float eh = xmid[O + 4] - predict<O>(xup, xmid, xdn, 4, c);
                const float m = MASK == 0 ? fabsf(eh) : nvf_value<PAD, O>(xm, q, 4);
                urow[4 + STRIP] = m * wh;
            }
            if (!has_right) {
                // image's last column lies in this strip: u(C) := u(C-1)
                const int lk = last_col_local - 4 * j.lane;
                if (lk >= 0 && lk < 4) urow[4 + last_col_local + 1] = uu[lk];
            }
            wave_lds_fence();
            float* un = uw[q % 3];
            un[0] = urow[3 + 4 * j.lane];
            un[1] = uu[0]; un[2] = uu[1]; un[3] = uu[2]; un[4] = uu[3];
            un[5] = urow[8 + 4 * j.lane];
            if (o == 0 && j.rs == 0) {
                // u(-1) := u(0): the first computed row is image row 0; seed the slot the next step reads as "um"
#pragma unroll
                for (int b = 0; b < 6; ++b) uw[(q + 2) % 3][b] = un[b];
            }
            // ---- emit e_u for row r = t-1: u rows r-1, r, r+1 are slots (q+1)%3, (q+2)%3, q%3
            const int r = t - 1;
            if (r >= j.rs && r < j.re) {
                const float* um = uw[(q + 1) % 3];
                const float* u0 = uw[(q + 2) % 3];
                const float* ewp = eww[(q + 1) % 2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (VEC || c0 + k < C) {
                        const float eu = u0[1 + k] - predict<1>(um, u0, un, k, c);
                        dot = fmaf(eu, ewp[k], dot);
                        nu = fmaf(eu, eu, nu);
                        nw = fmaf(ewp[k], ewp[k], nw);
                    }
                }
            }
            if (j.re == R && t == R - 1) {
                // last image row: u(R) := u(R-1); window (u(R-2), u(R-1), u(R-1))
                const float* u0 = uw[(q + 2) % 3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (VEC || c0 + k < C) {
                        const float eu = un[1 + k] - predict<1>(u0, un, un, k, c);
                        dot = fmaf(eu, ew[k], dot);
                        nu = fmaf(eu, eu, nu);
                        nw = fmaf(ew[k], ew[k], nw);
                    }
                }
            }
        }
    WM_MARCH_END
}

template <typename T, int MASK, int PAD, int HC>
__global__ __launch_bounds__(BLOCK) void k_detect(const T* __restrict__ x, long long pitch, long long fstride,
                                                  const float* __restrict__ W, Geom g, int nblk, int aligned,
                                                  const float* __restrict__ coef, const int* __restrict__ status,
                                                  double* __restrict__ pcorr)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<HC>::N];
    __shared__ __attribute__((aligned(16))) float s_u[WPB][2 * RowBuf<1>::N];
    __shared__ double s_red[WPB][3];
    const int frame = blockIdx.y;
    const WaveJob j = make_job(g, nblk);
    float dot = 0.0f, nu = 0.0f, nw = 0.0f;
    if (j.valid && status[frame] == 0) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = coef[frame * 8 + k];
        const T* xf = x + (long long)frame * fstride;
        if (aligned && j.full) detect_march<T, MASK, PAD, HC, true>(xf, pitch, W, g, j, s_row[j.wave], s_u[j.wave], c, dot, nu, nw);
        else detect_march<T, MASK, PAD, HC, false>(xf, pitch, W, g, j, s_row[j.wave], s_u[j.wave], c, dot, nu, nw);
    }
    const double d0 = wave_sum((double)dot), d1 = wave_sum((double)nu), d2 = wave_sum((double)nw);
    if (j.lane == 0) { s_red[j.wave][0] = d0; s_red[j.wave][1] = d1; s_red[j.wave][2] = d2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        pcorr[((long long)frame * nblk + blockIdx.x) * 3 + k] = ((s_red[0][k] + s_red[1][k]) + s_red[2][k]) + s_red[3][k];
    }
}

// corr = (float)dot / (float)(||e_w|| * ||e_u||)   (Watermark.cpp:230); unsolvable => 0.0f (:246-247)
__global__ __launch_bounds__(BLOCK) void k_corr_finalize(const double* __restrict__ pcorr, int nblk,
                                                         const int* __restrict__ status, OpResult* __restrict__ res)
{
    __shared__ double s[3][BLOCK];
    const int frame = blockIdx.x, t = threadIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int b = t; b < nblk; b += BLOCK) {
        const double* p = pcorr + ((long long)frame * nblk + b) * 3;
        a0 += p[0]; a1 += p[1]; a2 += p[2];
    }
    s[0][t] = a0; s[1][t] = a1; s[2][t] = a2;
    __syncthreads();
    for (int o = BLOCK / 2; o > 0; o >>= 1) {
        if (t < o) { s[0][t] += s[0][t + o]; s[1][t] += s[1][t + o]; s[2][t] += s[2][t + o]; }
        __syncthreads();
    }
    if (t == 0) {
        const int st = status[frame];
        float corr = 0.0f;
        if (st == 0) corr = (float)s[0][0] / (float)(sqrt(s[2][0]) * sqrt(s[1][0]));
        res[frame].status = st;
        res[frame].value = corr;
    }
}

// results of a mask-only op: status + coefficients
__global__ void k_mask_result(const int* __restrict__ status, const float* __restrict__ coef, OpResult* __restrict__ res,
                              float* __restrict__ coef_out)
{
    const int frame = blockIdx.x, t = threadIdx.x;
    if (t == 0) { res[frame].status = status ? status[frame] : 0; res[frame].value = 0.0f; }
    if (t < 8) coef_out[frame * 8 + t] = coef ? coef[frame * 8 + t] : 0.0f;
}

// =================================================================================================
// launchers
// =================================================================================================
static inline dim3 grid_of(const LaunchGeom& lg, int frames) { return dim3((unsigned)lg.nblk, (unsigned)frames, 1); }
static inline Geom geom_of(const LaunchGeom& lg) { Geom g; g.rows = lg.rows; g.cols = lg.cols; g.nstrips = lg.nstrips; g.nsegs = lg.nsegs; g.rps = lg.rps; return g; }

#define WM_DISPATCH_T(dtype, ...)                   \
    do {                                            \
        if ((dtype) == 0) { using T = float; __VA_ARGS__; } \
        else { using T = uint8_t; __VA_ARGS__; }    \
    } while (0)

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

void launch_me_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                     const float* coef, const int* status, float* pmax, double* pss)
{
    WM_DISPATCH_T(x.dtype, hipLaunchKernelGGL(k_me_stats<T>, grid_of(lg, frames), dim3(BLOCK), 0, s, (const T*)x.p, x.pitch,
                                               x.fstride, W, geom_of(lg), lg.nblk, (x.aligned && aligned_w) ? 1 : 0, coef, status, pmax,
                                               pss));
}

template <typename T>
static void launch_nvf_stats_t(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W,
                               int aligned_w, int pad, double* pss)
{
#define NVF_CASE(P)                                                                                                    \
    case P:                                                                                                            \
        hipLaunchKernelGGL((k_nvf_stats<T, P>), grid_of(lg, frames), dim3(BLOCK), 0, s, (const T*)x.p, x.pitch, x.fstride, \
                           W, geom_of(lg), lg.nblk, (x.aligned && aligned_w) ? 1 : 0, pss);                                     \
        break;
    switch (pad) { NVF_CASE(1) NVF_CASE(2) NVF_CASE(3) NVF_CASE(4) }
#undef NVF_CASE
}
void launch_nvf_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                      int pad, double* pss)
{
    WM_DISPATCH_T(x.dtype, launch_nvf_stats_t<T>(s, lg, frames, x, W, aligned_w, pad, pss));
}

void launch_embed_scalars(hipStream_t s, const LaunchGeom& lg, int frames, const float* pmax, const double* pss,
                          const int* status, float sF, EmbedScalars* scal, OpResult* res)
{
    hipLaunchKernelGGL(k_embed_scalars, dim3(frames), dim3(BLOCK), 0, s, pmax, pss, lg.nblk, status, sF,
                       sqrt((double)lg.rows * (double)lg.cols), scal, res);
}

template <typename TX, typename TB, int NCH>
static void launch_embed_tt(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                            const float* W, int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef,
                            const int* status, const EmbedScalars* scal)
{
#define EMB(MASK, P)                                                                                                     \
    hipLaunchKernelGGL((k_embed<TX, TB, NCH, MASK, P>), grid_of(lg, frames), dim3(BLOCK), 0, s, (const TX*)x.p, x.pitch,  \
                       x.fstride, W, base, out, geom_of(lg), lg.nblk,                                                  \
                       (x.aligned && aligned_w && base.aligned && out.aligned) ? 1 : 0, coef, status, scal)
    if (mask == 0) { EMB(0, 1); return; }
    switch (pad) {
        case 1: EMB(1, 1); break;
        case 2: EMB(1, 2); break;
        case 3: EMB(1, 3); break;
        case 4: EMB(1, 4); break;
    }
#undef EMB
}
template <typename TX, typename TB>
static void launch_embed_t(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                           const float* W, int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef,
                           const int* status, const EmbedScalars* scal)
{
    if (base.channels == 3) launch_embed_tt<TX, TB, 3>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal);
    else launch_embed_tt<TX, TB, 1>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal);
}
void launch_embed(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                  int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef, const int* status,
                  const EmbedScalars* scal)
{
    if (x.dtype == 0 && base.dtype == 0) launch_embed_t<float, float>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal);
    else if (x.dtype == 1 && base.dtype == 1) launch_embed_t<uint8_t, uint8_t>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal);
    else if (x.dtype == 0 && base.dtype == 1) launch_embed_t<float, uint8_t>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal);
    else launch_embed_t<uint8_t, float>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal);
}

template <typename T>
static void launch_mask_t(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                          const float* coef, const int* status, const EmbedScalars* scal, const PlaneDesc& mo,
                          const PlaneDesc& eo)
{
#define MSK(MASK, P)                                                                                                  \
    hipLaunchKernelGGL((k_mask<T, MASK, P>), grid_of(lg, frames), dim3(BLOCK), 0, s, (const T*)x.p, x.pitch, x.fstride, \
                       geom_of(lg), lg.nblk, coef, status, scal, mo, eo)
    if (mask == 0) { MSK(0, 1); return; }
    switch (pad) {
        case 1: MSK(1, 1); break;
        case 2: MSK(1, 2); break;
        case 3: MSK(1, 3); break;
        case 4: MSK(1, 4); break;
    }
#undef MSK
}
void launch_mask(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* coef,
                 const int* status, const EmbedScalars* scal, const PlaneDesc& mo, const PlaneDesc& eo)
{
    WM_DISPATCH_T(x.dtype, launch_mask_t<T>(s, lg, frames, mask, pad, x, coef, status, scal, mo, eo));
}

template <typename T>
static void launch_detect_t(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                            const float* W, int aligned_w, const float* coef, const int* status, double* pcorr)
{
#define DET(MASK, P, HC)                                                                                                \
    hipLaunchKernelGGL((k_detect<T, MASK, P, HC>), grid_of(lg, frames), dim3(BLOCK), 0, s, (const T*)x.p, x.pitch,       \
                       x.fstride, W, geom_of(lg), lg.nblk, (x.aligned && aligned_w) ? 1 : 0, coef, status, pcorr)
    if (mask == 0) { DET(0, 1, 1); return; }
    switch (pad) {
        case 1: DET(1, 1, 1); break;
        case 2: DET(1, 2, 1); break;
        case 3: DET(1, 3, 1); break;
        case 4: DET(1, 4, 2); break;
    }
#undef DET
}
void launch_detect(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                   int aligned_w, const float* coef, const int* status, double* pcorr)
{
    WM_DISPATCH_T(x.dtype, launch_detect_t<T>(s, lg, frames, mask, pad, x, W, aligned_w, coef, status, pcorr));
}

void launch_corr_finalize(hipStream_t s, const LaunchGeom& lg, int frames, const double* pcorr, const int* status,
                          OpResult* res)
{
    hipLaunchKernelGGL(k_corr_finalize, dim3(frames), dim3(BLOCK), 0, s, pcorr, lg.nblk, status, res);
}

void launch_mask_result(hipStream_t s, int frames, const int* status, const float* coef, OpResult* res, float* coef_out)
{
    hipLaunchKernelGGL(k_mask_result, dim3(frames), dim3(64), 0, s, status, coef, res, coef_out);
}

}  // namespace wmk
