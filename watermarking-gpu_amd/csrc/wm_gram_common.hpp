// wm_gram_common.hpp -- pieces of the Gram / solve step shared by the streaming kernel (wm_k_gram.hip) and the fused
// single-frame kernels (wm_k_fused.hip): the 44-term table of the lag-product formulation and the one-wave 8x8 solve
#pragma once
#include "wm_march.hpp"

namespace wmk {

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
struct GramTab { int ur[44], uc[44], lag[44]; };
__host__ __device__ constexpr GramTab make_gram_tab()
{
    GramTab g{};
    for (int t = 0; t < 44; ++t) {
        const GramTerm x = gram_term(t);
        g.ur[t] = x.ur; g.uc[t] = x.uc; g.lag[t] = x.lag;
    }
    return g;
}
__host__ __device__ constexpr int lag_dr(int l) { return l < 3 ? 0 : (l < 8 ? 1 : 2); }
__host__ __device__ constexpr int lag_dc(int l) { return l < 3 ? l : (l < 8 ? l - 3 - 2 : l - 8 - 2); }

// 8x8 solve from the 44 folded sums (in LDS) by ONE wave: LU with partial pivoting in f64, coefficients as f32.
// "Unsolvable" (status 1, zero coefficients): a pivot below 1e-12 max|Rx|, or a non-finite value.
__device__ __forceinline__ void lu_solve_wave(const double* s_tot, double (*A)[9], int t, int frame, float* coef, int* status)
{
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
        // agent-scope stores: the fused kernels hand the result to blocks on other XCDs inside the launch
        for (int i = 0; i < 8; ++i) st_agent(coef + frame * 8 + i, singular ? 0.0f : c[i]);
        st_agent(status + frame, singular ? 1 : 0);
    }
}

}  // namespace wmk
