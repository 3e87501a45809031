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

// ---- the border frame of the Gram matrix in 64-element chunks ----------------------------------------------------------
// The frame: the full rows -1, 0, R-2, R-1, R of the replicate-padded image (every row when the image is too small to
// have a core) and the 6 side columns -1, 0, 1, C-2, C-1, C over the rows 1..R-3.  Along a full row a chunk's lanes are
// consecutive columns (-1 + 64 k + lane), along a side column consecutive rows (1 + 64 k + lane).  The 44 terms of a chunk:
//     term_t = sum over lanes of [q in I + u_t] X(q) X(q + d_t),     [q in I + u] = rin_ur(r) and cin_uc(c),
// and along a chunk one of the two conditions is the same in every lane (a row chunk has one r; a side-column chunk one c
// and rows inside 1..R-3, where every rin holds): term_t = F_t * S[uc_t][lag_t] with 13 lane sums S (39 in the few row
// chunks that hold an image corner column, where cin differs between lanes) and a wave-uniform factor F_t: one 13-value
// recursive-halving reduction (wave_sum_multi) instead of 44 accumulators and 44 wave reductions; lane t < 44 ends with
// term t of the chunk.
struct BorderGeom {
    int R, C;
    int nfull, cpr, rpc;  // full rows (5, or R + 2 without a core); 64-column chunks per full row; 64-row chunks per side column
    int row_lo, row_hi;   // rows owned (a row band of a sharded image owns [row_lo, row_hi); else 0, R)
    bool core_empty;      // R < 4 or C < 5: no core, every padded row is a "full row"
    unsigned inv_cpr = 0, inv_rpc = 0;  // div_magic(cpr), div_magic(rpc), or 0: chunk_pos divides
    bool aligned = false;  // C % 4 == 0 and every row starts on a 16-byte (f32) / 4-byte (u8) boundary: side-column chunks by row loads
};
// ch / d == (ch * div_magic(d)) >> 32 for 0 <= ch < 65536 and 2 <= d < 65536 (0: no reciprocal, divide): a uniform integer
// division costs a wave ~25 vector instructions, a multiply-high is one scalar instruction
__host__ __device__ inline unsigned div_magic(int d) { return d >= 2 && d < 65536 ? (unsigned)(0x100000000ull / (unsigned)d) + 1u : 0u; }
__device__ __forceinline__ int div_by(int ch, int d, unsigned magic) { return magic ? (int)__umulhi((unsigned)ch, magic) : ch / d; }
__host__ __device__ inline BorderGeom border_geom(int R, int C, int row_lo, int row_hi)
{
    BorderGeom g;
    g.R = R; g.C = C; g.row_lo = row_lo; g.row_hi = row_hi;
    g.core_empty = R < 4 || C < 5;
    g.nfull = g.core_empty ? R + 2 : 5;
    g.cpr = (C + 2 + WAVE - 1) / WAVE;
    g.rpc = g.core_empty ? 0 : (R - 3 + WAVE - 1) / WAVE;
    return g;
}
__host__ __device__ inline int border_chunks(const BorderGeom& g) { return g.nfull * g.cpr + 6 * g.rpc; }

struct ChunkPos { int r, c; bool valid, rowchunk; int sidx; };
__device__ __forceinline__ ChunkPos chunk_pos(const BorderGeom& g, int ch, int lane)
{
    ChunkPos p;
    p.sidx = 0;
    p.rowchunk = ch < g.nfull * g.cpr;
    if (p.rowchunk) {
        const int k = div_by(ch, g.cpr, g.inv_cpr);  // scalar
        p.r = g.core_empty ? k - 1 : (k == 0 ? -1 : (k == 1 ? 0 : g.R - 2 + (k - 2)));
        p.c = (ch - k * g.cpr) * WAVE + lane - 1;
        // (row band of a sharded image: the rows above / below the image belong to the band that holds that border)
        p.valid = p.c <= g.C && (g.core_empty || (k < 2 ? g.row_lo == 0 : g.row_hi == g.R));
    } else {
        const int ch2 = ch - g.nfull * g.cpr;
        const int sidx = div_by(ch2, g.rpc, g.inv_rpc);  // scalar: which of the 6 side columns
        p.sidx = sidx;
        p.c = sidx < 3 ? sidx - 1 : g.C - 2 + (sidx - 3);
        p.r = 1 + (ch2 - sidx * g.rpc) * WAVE + lane;
        p.valid = p.r <= g.R - 3 && p.r >= (g.row_lo == 0 ? 1 : g.row_lo) && p.r < (g.row_hi == g.R ? g.R - 2 : g.row_hi);
    }
    return p;
}
template <typename T>
struct BorderVals { T v[15]; };  // slot = 5 * row offset + column offset + 2 (rows r..r+2, columns c-2..c+2); slots 0, 1 unused
// A side-column chunk (64 rows of one of the 6 border columns) on the aligned layout.  Every clamped neighbour column of
// the three left columns lies in columns 0..3, of the three right columns in C-4..C-1: ONE 4-pixel load per lane (its own
// row) holds all five, rows r+1 / r+2 are the next lanes' loads (DPP), the two rows below the chunk come from a second
// load by lanes 0 and 1.  66 cache lines per chunk instead of the 13 x 64 of one-element gathers per (row, column).
template <typename T, int SIDX>
__device__ __forceinline__ void border_column_fill(BorderVals<T>& bv, const uint32_t (&w)[3][sizeof(T) == 4 ? 4 : 1])
{
#pragma unroll
    for (int q = 2; q < 15; ++q) {
        const int dc = q % 5;
        const int k = SIDX < 3 ? (SIDX + dc - 3 > 0 ? SIDX + dc - 3 : 0) : (SIDX - 3 + dc < 3 ? SIDX - 3 + dc : 3);
        if constexpr (sizeof(T) == 4) bv.v[q] = __uint_as_float(w[q / 5][k]);
        else bv.v[q] = (T)((w[q / 5][0] >> (8 * k)) & 0xffu);
    }
}
// COH: loads past the caches that are not coherent across the chip (agent-scope loads, as the hand-off records are read) -- for
// a plane that another workgroup of the SAME launch has rewritten since this one may have cached it (k_fused_pair)
template <typename T, bool COH = false>
__device__ __forceinline__ BorderVals<T> border_column_chunk_issue(const T* xf, long long pitch, const BorderGeom& g, const ChunkPos& p, int lane)
{
    constexpr int NW = sizeof(T) == 4 ? 4 : 1;
    using V = typename Elem<T>::vec4;
    const T* colp = xf + (p.sidx >= 3 ? g.C - 4 : 0);
    union { V v; uint32_t u[NW]; } own, ext;
    auto ldv = [&](const T* q, uint32_t (&u)[NW], V& v) {
        if constexpr (COH) {
#pragma unroll
            for (int k = 0; k < NW; ++k) u[k] = ld_agent(reinterpret_cast<const uint32_t*>(q) + k);
        } else v = *reinterpret_cast<const V*>(q);
    };
    ldv(colp + (long long)clampi(p.r, 0, g.R - 1) * pitch, own.u, own.v);
#pragma unroll
    for (int k = 0; k < NW; ++k) ext.u[k] = 0u;
    if (lane < 2) ldv(colp + (long long)clampi(p.r + WAVE, 0, g.R - 1) * pitch, ext.u, ext.v);
    uint32_t w[3][NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int e0 = __builtin_amdgcn_readlane((int)ext.u[k], 0), e1 = __builtin_amdgcn_readlane((int)ext.u[k], 1);
        w[0][k] = own.u[k];
        w[1][k] = (uint32_t)__builtin_amdgcn_update_dpp(e0, (int)w[0][k], 0x130 /*wave_shl:1*/, 0xF, 0xF, false);  // lane l <- lane l + 1
        w[2][k] = (uint32_t)__builtin_amdgcn_update_dpp(e1, (int)w[1][k], 0x130, 0xF, 0xF, false);
    }
    BorderVals<T> bv;
    switch (p.sidx) {  // wave-uniform
    case 0: border_column_fill<T, 0>(bv, w); break;
    case 1: border_column_fill<T, 1>(bv, w); break;
    case 2: border_column_fill<T, 2>(bv, w); break;
    case 3: border_column_fill<T, 3>(bv, w); break;
    case 4: border_column_fill<T, 4>(bv, w); break;
    default: border_column_fill<T, 5>(bv, w); break;
    }
    return bv;
}
template <typename T, bool COH = false>
__device__ __forceinline__ BorderVals<T> border_chunk_issue(const T* xf, long long pitch, const BorderGeom& g, int ch, int lane)
{
    const ChunkPos p = chunk_pos(g, ch, lane);
    if (g.aligned && !p.rowchunk) return border_column_chunk_issue<T, COH>(xf, pitch, g, p, lane);
    long long roff[3];
    int coff[5];
#pragma unroll
    for (int a2 = 0; a2 < 3; ++a2) roff[a2] = (long long)clampi(p.r + a2, 0, g.R - 1) * pitch;
#pragma unroll
    for (int b2 = 0; b2 < 5; ++b2) coff[b2] = clampi(p.c + b2 - 2, 0, g.C - 1);
    BorderVals<T> bv;
#pragma unroll
    for (int q = 2; q < 15; ++q) bv.v[q] = COH ? ld_agent(xf + roff[q / 5] + coff[q % 5]) : xf[roff[q / 5] + coff[q % 5]];
    return bv;
}
// sc: 39 doubles of LDS private to the calling wave.  Returns term `lane` of the chunk in the lanes < 44.
template <typename T>
__device__ __forceinline__ double border_chunk_terms(const BorderVals<T>& bv, const BorderGeom& g, int ch, int lane, double* sc)
{
    const ChunkPos p = chunk_pos(g, ch, lane);
    const double xq = p.valid ? (double)bv.v[2] : 0.0;
    double prod[13];
#pragma unroll
    for (int l = 0; l < 13; ++l) prod[l] = xq * (double)bv.v[2 + l];  // lag l <-> slot 2 + l (row 0: columns 2..4, rows 1, 2: 0..4)
    // column conditions of u_c = -1, 0, +1 (uc <= c <= C-1+uc); a side-column chunk applies them as uniform factors below
    const bool cin[3] = {p.c <= g.C - 2, p.c >= 0 && p.c <= g.C - 1, p.c >= 1};
    const bool three = p.rowchunk && !__all((cin[0] == cin[1] && cin[1] == cin[2]) || !p.valid);  // wave-uniform
    int idx;
    if (!three) {
        double v[13];
#pragma unroll
        for (int l = 0; l < 13; ++l) v[l] = (!p.rowchunk || cin[1]) ? prod[l] : 0.0;
        const double s = wave_sum_multi<13>(v, lane, idx);
        if (idx < 13) { sc[idx] = s; sc[13 + idx] = s; sc[26 + idx] = s; }
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double v[13];
#pragma unroll
            for (int l = 0; l < 13; ++l) v[l] = cin[q] ? prod[l] : 0.0;
            const double s = wave_sum_multi<13>(v, lane, idx);
            if (idx < 13) sc[13 * q + idx] = s;
        }
    }
    wave_lds_fence();
    double term = 0.0;
    if (lane < NGRAM) {
        constexpr GramTab tab = make_gram_tab();
        const int ur = tab.ur[lane], uc = tab.uc[lane], lag = tab.lag[lane];
        // the uniform factor: the row condition of a row chunk, the column condition of a side column
        const int r = p.r, c = p.c;  // (uniform in the dimension that matters)
        const bool f = p.rowchunk ? (ur < 0 ? r <= g.R - 2 : (ur == 0 ? (r >= 0 && r <= g.R - 1) : r >= 1))
                                  : (uc < 0 ? c <= g.C - 2 : (uc == 0 ? (c >= 0 && c <= g.C - 1) : c >= 1));
        term = f ? sc[13 * (uc + 1) + lag] : 0.0;
    }
    wave_lds_fence();
    return term;
}

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

// The same system by Gauss-Jordan elimination WITHOUT pivoting, for the fused single-frame kernels, where the solve sits on
// the critical path of every call: Rx is a Gram matrix (symmetric positive semi-definite), for which elimination in the
// natural order is backward stable, so the pivot search, the row swap and the back substitution of lu_solve_lanes (two
// thirds of its dependent chain) can go: 8 steps of {reciprocal of the pivot, one row and one column broadcast, one
// multiply-add}.  The solutions of the two routines differ by a few 1e-16 x cond(Rx) <= 1e-10, far inside the f32 ulp the
// coefficients are rounded to.  "Unsolvable" as above: a pivot (here: of the Schur complement) below 1e-12 max|Rx|, or
// a non-finite value.
__device__ __forceinline__ int spd_solve_lanes(const double* s_tot, int lane, float (&c)[8])
{
    const int row = lane >> 3, col = lane & 7;
    double a, b;
    {
        const int lo = row < col ? row : col, hi = row < col ? col : row;
        a = s_tot[lo * 8 - (lo * (lo - 1)) / 2 + (hi - lo)];
        b = s_tot[36 + row];
    }
    const bool finite_all = __all(isfinite(a) && isfinite(b)) != 0;
    const double amax = wave_max_d(fabs(a));
    bool singular = !(amax > 0.0) || !isfinite(amax) || !finite_all;
    const double tiny = 1e-12 * amax;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double akk = readlane_d(a, 9 * k);
        if (!(akk > tiny)) singular = true;
        const double r = recip_d(akk);
        const double aik = bperm_d(a, (lane & ~7) + k);  // column k at this lane's row
        const double akj = bperm_d(a, 8 * k + col) * r;  // row k at this lane's column, scaled
        const double bk = readlane_d(b, 8 * k) * r;
        if (row == k) { a = akj; b = bk; }
        else { a = fma(-aik, akj, a); b = fma(-aik, bk, b); }
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double s = readlane_d(b, 8 * i);
        if (!isfinite(s)) bad = true;
        c[i] = (float)s;
    }
    if (singular || bad) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = 0.0f;
    }
    return singular || bad ? 1 : 0;
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
