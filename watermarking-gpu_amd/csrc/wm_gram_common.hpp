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

// ---- 8x8 solve from the 44 folded sums by ONE wave, in registers --------------------------------------------------
// LU with partial pivoting in f64 (the oracle's wmo_solve with the divisions as products with the pivots' reciprocals,
// recip_d), coefficients as f32.  Lane 8i + j
// holds A[i][j] and (replicated along the row) the right-hand side b[i]; what is uniform over the wave -- pivot search,
// pivot row, back substitution -- is computed from readlane values, row / column broadcasts are ds_bpermute (the LDS
// crossbar, no LDS memory).  The critical path is one f64 division per elimination step and per unknown: ~3 us instead of
// the ~6 us of the earlier LDS-resident version, on the exposed tail of every Gram sweep.
// "Unsolvable" (status 1, zero coefficients): a pivot below 1e-12 max|Rx|, or a non-finite value.
__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double bperm_d(double v, int src_lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max_d(double v)  // values >= 0 (or NaN)
{
    v = fmax(v, dpp_mov0<0x111, 0xF>(v));
    v = fmax(v, dpp_mov0<0x112, 0xF>(v));
    v = fmax(v, dpp_mov0<0x114, 0xF>(v));
    v = fmax(v, dpp_mov0<0x118, 0xF>(v));
    v = fmax(v, dpp_mov0<0x142, 0xA>(v));
    v = fmax(v, dpp_mov0<0x143, 0xC>(v));
    return readlane_d(v, 63);
}

// 1 / d for a pivot: v_rcp_f64 and two Newton steps (relative error of an ulp or two of f64).  The solve sits on the
// exposed tail of every Gram sweep and is a chain of 16 dependent quotients; the full IEEE division sequence costs ~4x
// this per link.  Products with the reciprocal differ from the oracle's quotients in the last bit of f64, which is
// 1e-16 x cond(Rx) <= 1e-10 in the coefficients -- far inside the f32 ulp they are rounded to.
__device__ __forceinline__ double recip_d(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// s_tot: the 36 upper-triangle sums (Watermark.hpp:29-39 order) then the 8 right-hand sides, readable by the whole wave
// (LDS or global).  All 64 lanes of the calling wave take part; c[] and the return value are uniform.
__device__ __forceinline__ int lu_solve_lanes(const double* s_tot, int lane, float (&c)[8])
{
    const int row = lane >> 3, col = lane & 7;
    double a, b;
    {
        const int lo = row < col ? row : col, hi = row < col ? col : row;
        a = s_tot[lo * 8 - (lo * (lo - 1)) / 2 + (hi - lo)];
        b = s_tot[36 + row];
    }
    // fmax drops NaNs: look for non-finite entries separately
    const bool finite_all = __all(isfinite(a) && isfinite(b)) != 0;
    const double amax = wave_max_d(fabs(a));
    bool singular = !(amax > 0.0) || !isfinite(amax) || !finite_all;
    const double tiny = 1e-12 * amax;
    double rinv[8];  // reciprocals of the pivots (uniform)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // pivot: the largest |A[i][k]|, i >= k, first one wins (uniform)
        int piv = k;
        double pmax = fabs(readlane_d(a, 9 * k));
#pragma unroll
        for (int i = k + 1; i < 8; ++i) {
            const double v = fabs(readlane_d(a, 8 * i + k));
            if (v > pmax) { pmax = v; piv = i; }
        }
        if (!(pmax > tiny)) singular = true;
        piv = __builtin_amdgcn_readfirstlane(piv);
        if (piv != k) {
            const int src = row == k ? piv * 8 + col : (row == piv ? k * 8 + col : lane);
            a = bperm_d(a, src);
            b = bperm_d(b, src);
        }
        rinv[k] = recip_d(readlane_d(a, 9 * k));
        const double bk = readlane_d(b, 8 * k);
        const double aik = bperm_d(a, (lane & ~7) + k);
        const double akj = bperm_d(a, 8 * k + col);
        const double f = aik * rinv[k];
        if (row > k) {
            if (col >= k) a = a - f * akj;
            b = b - f * bk;
        }
    }
    double sol[8];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        double s = readlane_d(b, 8 * i);
#pragma unroll
        for (int jj = i + 1; jj < 8; ++jj) s -= readlane_d(a, 8 * i + jj) * sol[jj];
        sol[i] = s * rinv[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (!isfinite(sol[i])) singular = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = singular ? 0.0f : (float)sol[i];
    return singular ? 1 : 0;
}

// the same, delivering to memory (streaming kernels): agent-scope stores by lane 0
__device__ __forceinline__ void lu_solve_wave(const double* s_tot, int t, int frame, float* coef, int* status)
{
    float c[8];
    const int st = lu_solve_lanes(s_tot, t, c);
    if (t == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) st_agent(coef + frame * 8 + i, c[i]);
        st_agent(status + frame, st);
    }
}

}  // namespace wmk
