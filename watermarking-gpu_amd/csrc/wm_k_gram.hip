// wm_k_gram.hip -- Gram-matrix side of the watermark hot path on gfx950: k_gram (march blocks + border blocks) and its
// fold tail solve_frame.
//
// Kernel map of the whole path (reference function -> kernel; DESIGN.md has bytes and rooflines):
//   me kernel + af::sum partial folding (me_p3.hpp:23-83, Watermark.cpp:140-151)       -> k_gram                        (this file)
//   af::solve (Watermark.cpp:203)                                                      -> solve_frame, tail of k_gram   (this file)
//   scaled_neighbors + sub + abs + max + mask*W + norm (Watermark.cpp:210-214,169-170) -> k_me_stats + embed_scalars_frame tail (wm_k_embed.hip)
//   nvf kernel + mask*W + norm (nvf.hpp:5-51)                                          -> k_nvf_stats + the same tail   (wm_k_embed.hip)
//   u*a + base, clamp (Watermark.cpp:171), mask recomputed                             -> k_embed                       (wm_k_embed.hip)
//   detect: 2x scaled_neighbors, mask*W, dot, 2x norm (Watermark.cpp:221-250)          -> k_detect + corr_finalize_frame tail (wm_k_detect.hip)
//
// All marching kernels share the strip-march execution shape of wm_device.hpp / wm_march.hpp.  Every global sum is
// a fixed-order multi-stage reduction (per thread -> DPP per wave -> LDS per block -> f64 over the block partials, in
// partial-index order, by the frame's last block): values are never accumulated with atomics, results are bitwise
// deterministic run to run.  Compiled with -ffp-contract=off: fused multiply-adds appear only where fma()/fmaf() is
// written, which pins the operation order the CPU oracle uses.
#include "wm_march.hpp"
#include "wm_gram_common.hpp"

#ifndef WM_GRAM_WAVES
#define WM_GRAM_WAVES 1
#endif
#ifndef WM_GRAM_PF
#define WM_GRAM_PF 6     // rows of x in flight per wave
#endif

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
// products of f32/u8 pixels, 13 instead of 44 multiply-adds per pixel); a separate
// k_gram_border kernel evaluates the border frame.  tests/lag_gram_model.py is the numpy model of this split.
// =================================================================================================
template <typename T>
__device__ __forceinline__ double padded(const T* __restrict__ x, long long pitch, int R, int C, int r, int c)
{
    return (double)x[(long long)clampi(r, 0, R - 1) * pitch + clampi(c, 0, C - 1)];
}

// q rows [rs, re) of one strip: stream rows rs .. re+1; f64 window of rows q, q+1, q+2 and columns
// c0-2 .. c0+5 in rotating slots (slot of stream row i = i % 3)
template <typename T, bool VEC, bool EDGE>
__device__ __forceinline__ void gram_march_impl(const T* __restrict__ xf, long long pitch, const Geom& g, const WaveJob& j,
                                                float* lds, double (&acc)[13])
{
    const int R = g.rows, C = g.cols;
    XMarch<T, 1, 2, 1, VEC, WM_GRAM_PF, EDGE> xm;
    // q rows of this segment that lie in the core (1 <= r <= R-3): the march covers exactly those, so no row needs
    // a validity factor
    // (a row band of a sharded image clips at true image borders only: g.row_lo == 0 / g.row_hi == R mark them)
    const int lo = g.row_lo == 0 ? 1 : g.row_lo, hi = g.row_hi == R ? R - 2 : g.row_hi;
    const int rs = j.rs > lo ? j.rs : lo;
    const int re = j.re < hi ? j.re : hi;
    if (re <= rs) return;
    const int n = re - rs + 2;
    xm.start(xf, pitch, g, j, lds, rs, n);
    const int c0 = j.c0s + 4 * j.lane;
    double w[3][8];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) w[a][b] = 0.0;
    // column validity is row-invariant: pixels outside the core contribute with a zero factor (no branch);
    // strips that hold none of the image's two first / two last columns run the instance without the factor
    bool cv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cv[k] = !EDGE || (c0 + k >= 2 && c0 + k <= C - 3 && 4 * j.lane >= j.dup);
    march<2>(n, [&](int i, auto qc, auto emit) {
        constexpr int Q = decltype(qc)::value;
        xm.template step<Q>(i);
#pragma unroll
        for (int b = 0; b < 8; ++b) w[Q % 3][b] = (double)xm.win[0][2 + b];
        if (decltype(emit)::value) {
            // q row r = rs + i - 2: its window rows are slots (Q+1)%3, (Q+2)%3, Q%3
            const double* w0 = w[(Q + 1) % 3];
            const double* w1 = w[(Q + 2) % 3];
            const double* w2 = w[Q % 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double xq = (!EDGE || cv[k]) ? w0[2 + k] : 0.0;
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

template <typename T, bool VEC>
__device__ __forceinline__ void gram_march(const T* __restrict__ xf, long long pitch, const Geom& g, const WaveJob& j,
                                           float* lds, double (&acc)[13])
{
    // wave-uniform: does this strip hold one of the columns 0, 1, C-2, C-1 (which are never q pixels)?
    const bool edge = !VEC || j.c0s == 0 || j.c0s + STRIP > g.cols - 2;
    if (edge) gram_march_impl<T, VEC, true>(xf, pitch, g, j, lds, acc);
    else gram_march_impl<T, VEC, false>(xf, pitch, g, j, lds, acc);
}
// Column seam S (the first column a strip owns) inside one segment [rs, re): the products of the core pixels in columns S-2,
// S-1 with partners in columns S, S+1 (the left strip's last lane has no right neighbour), and -- when the right strip loads
// from S on (`left_too`; a shifted last strip holds its left neighbours itself) -- of columns S, S+1 with partners in S-2, S-1.
// q rows [rs, re) with their partner rows up to re + 1 (k_embed's tiles reach two rows behind their segment).  A lane takes a
// row; rows r+1, r+2 come from the next lanes, so a wave covers 62 q rows per round.
// `sm`: the boundary's rows of the seam array the embed left ([rows][4]: columns S-2, S-1, S, S+1).
__device__ __forceinline__ void gram_colseam(const float4* __restrict__ sm, int R, int rs, int re, bool left_too, int lane, double (&acc)[13])
{
    const int rl = re + 2 < R ? re + 2 : R;  // rows that exist behind the q rows
    for (int base = rs; base < re; base += 62) {
        const int r = base + lane;
        const bool in = r < rl;
        const float4 v = sm[in ? r : rl - 1];
        const float a0[4] = {in ? v.x : 0.0f, in ? v.y : 0.0f, in ? v.z : 0.0f, in ? v.w : 0.0f};
        float a1[4], a2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1[k] = dpp_from_next(a0[k], 0.0f); a2[k] = dpp_from_next(a1[k], 0.0f); }
        const bool q = lane < 62 && r < re && r >= 1 && r <= R - 3;
        const double m2 = q ? (double)a0[0] : 0.0, m1 = q ? (double)a0[1] : 0.0, p0 = q ? (double)a0[2] : 0.0, p1 = q ? (double)a0[3] : 0.0;
        // lag index: 0..2 = (0, 0..2); 3 + b = (1, b - 2); 8 + b = (2, b - 2)
        acc[2] = fma(m2, (double)a0[2], acc[2]);
        acc[7] = fma(m2, (double)a1[2], acc[7]);
        acc[12] = fma(m2, (double)a2[2], acc[12]);
        acc[1] = fma(m1, (double)a0[2], acc[1]);
        acc[6] = fma(m1, (double)a1[2], acc[6]);
        acc[11] = fma(m1, (double)a2[2], acc[11]);
        acc[2] = fma(m1, (double)a0[3], acc[2]);
        acc[7] = fma(m1, (double)a1[3], acc[7]);
        acc[12] = fma(m1, (double)a2[3], acc[12]);
        if (left_too) {
            acc[4] = fma(p0, (double)a1[1], acc[4]);
            acc[9] = fma(p0, (double)a2[1], acc[9]);
            acc[3] = fma(p0, (double)a1[0], acc[3]);
            acc[8] = fma(p0, (double)a2[0], acc[8]);
            acc[3] = fma(p1, (double)a1[1], acc[3]);
            acc[8] = fma(p1, (double)a2[1], acc[8]);
        }
    }
}

// u8 frames on the aligned path: the 13 lag sums in EXACT INTEGER arithmetic.  A lane's 4 pixels of a row are one
// packed dword; the partner pixels x(q + (a,b)) of the 4 own pixels are the byte-shifted dwords
// v_alignbyte(neighbour, own, b), and one v_dot4_u32_u8 accumulates 4 products: 13 dot4 + 8 alignbyte per row
// per lane instead of 52 f64 FMAs + conversions.  u32 accumulators cannot overflow inside a segment
// (4 * 255^2 per step, rps <= 4096 rows); they are widened to f64 once, at the end.
template <bool VEC>
__device__ __forceinline__ void gram_march_u8(const uint8_t* __restrict__ xf, long long pitch, const Geom& g, const WaveJob& j,
                                              double (&acc)[13])
{
    static_assert(VEC, "integer path needs the aligned strip layout");
    const int R = g.rows, C = g.cols;
    XStream<uint8_t, 1, 4, true> xs;
    xs.init(xf, pitch, R, C, j);
    const int n = j.re - j.rs + 2, s0 = j.rs, last = j.rs + n - 1;
    typename XStream<uint8_t, 1, 4, true>::Raw pre[WM_GRAM_PF];
#pragma unroll
    for (int q = 0; q < WM_GRAM_PF; ++q) pre[q] = xs.issue(min(s0 + q, last));
    const int core_lo = g.row_lo == 0 ? 1 : g.row_lo, core_hi = g.row_hi == R ? R - 2 : g.row_hi;  // q rows of the core owned here
    const int c0 = j.c0s + 4 * j.lane;
    // byte mask of the own pixels that lie in the core columns 2 .. C-3
    uint32_t cmask = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c0 + k >= 2 && c0 + k <= C - 3 && 4 * j.lane >= j.dup) cmask |= 0xffu << (8 * k);
    uint32_t sh[3][5];  // per window row: dwords of columns c0+b .. c0+b+3, b = -2..2 (rotating slots, slot = row index % 3)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) sh[a][b] = 0;
    uint32_t iacc[13];
#pragma unroll
    for (int l = 0; l < 13; ++l) iacc[l] = 0;
    march<2>(n, [&](int i, auto qc, auto emit) {
        constexpr int Q = decltype(qc)::value;
        const uint32_t own = opaque(pre[Q % WM_GRAM_PF].v);
        const uint32_t halo = opaque(pre[Q % WM_GRAM_PF].h);
        __builtin_amdgcn_sched_barrier(0);
        pre[Q % WM_GRAM_PF] = xs.issue(min(s0 + i + WM_GRAM_PF, last));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // neighbour chunks by DPP; lane 0 / lane 63 keep the strip's halo dword (values at the image border never reach a
        // core product, so no replicate fix-up is needed here)
        const uint32_t L = (uint32_t)__builtin_amdgcn_update_dpp((int)halo, (int)own, 0x138, 0xF, 0xF, false);
        const uint32_t Rr = (uint32_t)__builtin_amdgcn_update_dpp((int)halo, (int)own, 0x130, 0xF, 0xF, false);
        uint32_t* s2 = sh[Q % 3];
        s2[0] = __builtin_amdgcn_alignbyte(own, L, 2);
        s2[1] = __builtin_amdgcn_alignbyte(own, L, 3);
        s2[2] = own;
        s2[3] = __builtin_amdgcn_alignbyte(Rr, own, 1);
        s2[4] = __builtin_amdgcn_alignbyte(Rr, own, 2);
        if (decltype(emit)::value) {
            const int r = j.rs + i - 2;
            const uint32_t* w0 = sh[(Q + 1) % 3];
            const uint32_t* w1 = sh[(Q + 2) % 3];
            const uint32_t A = (r >= core_lo && r < core_hi) ? (w0[2] & cmask) : 0u;
            iacc[0] = __builtin_amdgcn_udot4(A, w0[2], iacc[0], false);
            iacc[1] = __builtin_amdgcn_udot4(A, w0[3], iacc[1], false);
            iacc[2] = __builtin_amdgcn_udot4(A, w0[4], iacc[2], false);
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                iacc[3 + b] = __builtin_amdgcn_udot4(A, w1[b], iacc[3 + b], false);
                iacc[8 + b] = __builtin_amdgcn_udot4(A, s2[b], iacc[8 + b], false);
            }
        }
    });
#pragma unroll
    for (int l = 0; l < 13; ++l) acc[l] = (double)iacc[l];
}

// Border frame: the <= 5 full rows and 6 side columns outside the core (or everything when the image is too small to
// have a core), all 44 terms (wm_gram_common.hpp: chunks of 64 elements, 13 lane sums per chunk by recursive halving, lane
// t < 44 accumulates term t over the wave's chunks -- one f64 register instead of 44 accumulators).  The next chunk's
// 13 loads are in flight while the current chunk is reduced.
template <typename T, bool VEC>
__device__ __forceinline__ void gram_border_block(const T* __restrict__ x, long long pitch, long long fstride, int R, int C,
                                                  int nbb, int bb, int frame, double* pborder, int row_lo, int row_hi)
{
    __shared__ double s_red[WPB][NGRAM];
    __shared__ double s_sc[WPB][40];

    BorderGeom bg = border_geom(R, C, row_lo, row_hi);
    bg.aligned = VEC && C % 4 == 0 && !bg.core_empty;  // side-column chunks by row loads (wm_gram_common.hpp)
    bg.inv_cpr = div_magic(bg.cpr); bg.inv_rpc = div_magic(bg.rpc);
    const T* xf = x + (long long)frame * fstride;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunks = border_chunks(bg);
    const int step = nbb * WPB;
    double acc = 0.0;
    int ch = bb * WPB + wave;
    if (ch < nchunks) {
        BorderVals<T> cur = border_chunk_issue<T>(xf, pitch, bg, ch, lane);
        for (; ch < nchunks; ch += step) {
            const int nx = ch + step < nchunks ? ch + step : ch;  // (the last round re-reads its own chunk: no branch around the loads)
            const BorderVals<T> nxt = border_chunk_issue<T>(xf, pitch, bg, nx, lane);
            acc += border_chunk_terms<T>(cur, bg, ch, lane, s_sc[wave]);
            cur = nxt;
        }
    }
    if (lane < NGRAM) s_red[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < NGRAM)
        st_agent(pborder + ((long long)frame * nbb + bb) * NGRAM + threadIdx.x,
                 ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x]);
}

// =================================================================================================
// solve_frame (tail of k_gram): fold the block partials (f64), 8x8 LU with partial pivoting in f64, coefficients as f32
// =================================================================================================
constexpr int SOLVE_GM = BLOCK / 13;     // 19 thread groups for the 13 lag sums
constexpr int SOLVE_GB = BLOCK / NGRAM;  // 5 thread groups for the 44 border terms

// run by the 256 threads of the block that finished the frame's Gram sweep last
// (stride: records per frame of the pmain array when it holds more than the nblk records folded here; 0 = nblk)
__device__ __forceinline__ void solve_frame(int frame, const double* pmain, int nblk, const double* pborder, int nbb,
                                            float* __restrict__ coef, int* __restrict__ status,
                                            double* __restrict__ gram_tot, int stride = 0)
{
    __shared__ double s_pm[SOLVE_GM][13];
    __shared__ double s_pb[SOLVE_GB][NGRAM];
    __shared__ double s_m[13];
    __shared__ double s_tot[NGRAM];
    const int t = threadIdx.x;
    if (t < SOLVE_GM * 13) {
        const int k = t % 13, gq = t / 13;
        const double* p = pmain + (long long)frame * (stride ? stride : nblk) * 13 + k;
        // 8 partials in flight per thread (row index clamped, surplus terms dropped): a dependent load per term
        // would cost a memory latency each; the order of the sum is still the partial index
        double s = 0.0;
        for (int b0 = gq; b0 < nblk; b0 += 8 * SOLVE_GM) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld_agent(p + (long long)min(b0 + u * SOLVE_GM, nblk - 1) * 13);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += b0 + u * SOLVE_GM < nblk ? v[u] : 0.0;
        }
        s_pm[gq][k] = s;
    }
    if (t < SOLVE_GB * NGRAM) {
        const int k = t % NGRAM, gq = t / NGRAM;
        const double* p = pborder + (long long)frame * nbb * NGRAM + k;
        double s = 0.0;
        for (int b0 = gq; b0 < nbb; b0 += 8 * SOLVE_GB) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld_agent(p + (long long)min(b0 + u * SOLVE_GB, nbb - 1) * NGRAM);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += b0 + u * SOLVE_GB < nbb ? v[u] : 0.0;
        }
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
        constexpr GramTab tab = make_gram_tab();
        int lag = 0;
#pragma unroll
        for (int tt = 0; tt < NGRAM; ++tt)
            if (tt == t) lag = tab.lag[tt];
        s += s_m[lag];
        s_tot[t] = s;
        gram_tot[(long long)frame * NGRAM + t] = s;
    }
    __syncthreads();
    if (t < WAVE) lu_solve_wave(s_tot, t, frame, coef, status);  // one wave, in registers
}

// march blocks: 13 lag sums over the core, one partial record per block
// With nbb > 0 the first nbb blocks of the grid evaluate the border frame (they are few and latency-bound, so they
// should start first and overlap the march instead of costing a launch of their own); the march blocks follow.
template <typename T, bool VEC>
__global__ __launch_bounds__(BLOCK, WM_GRAM_WAVES) void k_gram(const T* __restrict__ x, long long pitch, long long fstride, Geom g, int nbb,
                                                double* pmain, double* pborder, SolveTail tail)
{
    const int nlead = nbb * g.frames;  // leading border blocks: nbb per frame
    if ((int)blockIdx.x < nlead) {
        const int bfr = (int)blockIdx.x / nbb;
        gram_border_block<T, VEC>(x, pitch, fstride, g.rows, g.cols, nbb, (int)blockIdx.x - bfr * nbb, bfr, pborder, g.row_lo, g.row_hi);
        if (last_block_of_frame(tail.ticket + bfr * TKS, (unsigned)tail.expected))
            solve_frame(bfr, pmain, g.nblk_total, pborder, tail.nbb_total, tail.coef, tail.status, tail.gram_tot);
        return;
    }
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    __shared__ double s_red[WPB][13];
    const int R = g.rows, C = g.cols;
    const bool core_empty = R < 4 || C < 5;
    const WaveJob j = make_job(g, (int)blockIdx.x - nlead);
    const int frame = j.frame;
    const T* xf = x + (long long)frame * fstride;
    double acc[13];
#pragma unroll
    for (int l = 0; l < 13; ++l) acc[l] = 0.0;
    if (j.valid && !core_empty) {
        if constexpr (VEC && std::is_same<T, uint8_t>::value) gram_march_u8<true>(xf, pitch, g, j, acc);
        else gram_march<T, VEC>(xf, pitch, g, j, s_row[j.wave], acc);
    }
    {
        int idx;
        const double s = wave_sum_multi<13>(acc, j.lane, idx);  // 13 sums in one recursive-halving pass
        if (idx < 13) s_red[j.wave][idx] = s;
    }
    __syncthreads();
    if (threadIdx.x < 13)
        st_agent(pmain + ((long long)frame * g.nblk_total + g.pb0 + j.tile) * 13 + threadIdx.x,
                 ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x]);
    if (last_block_of_frame(tail.ticket + frame * TKS, (unsigned)tail.expected))
        solve_frame(frame, pmain, g.nblk_total, pborder, tail.nbb_total, tail.coef, tail.status, tail.gram_tot);
}

// =================================================================================================
// k_gram_ho: the detector's Gram matrix of a plane y whose tile-internal lag sums the embed left behind (HandOver).  Blocks
// per frame: the border blocks (as in k_gram) and the column-seam blocks (4 waves = 4 (strip boundary, segment) pairs); every
// seam block leaves a 13-sum record behind the wave records, the frame's last block folds everything and solves (solve_frame).
// =================================================================================================
__global__ __launch_bounds__(BLOCK) void k_gram_ho(const float* __restrict__ y, long long pitch, long long fstride, Geom g, int nbb,
                                                   int nsb, HandOver ho, double* pborder, SolveTail tail)
{
    const int nlead = nbb * g.frames;
    if ((int)blockIdx.x < nlead) {
        const int bfr = (int)blockIdx.x / nbb;
        gram_border_block<float, true>(y, pitch, fstride, g.rows, g.cols, nbb, (int)blockIdx.x - bfr * nbb, bfr, pborder, g.row_lo, g.row_hi);
        if (last_block_of_frame(tail.ticket + bfr * TKS, (unsigned)tail.expected))
            solve_frame(bfr, ho.rec + (long long)g.nrec * 13, nsb, pborder, tail.nbb_total, tail.coef, tail.status, tail.gram_tot, ho.stride);
        return;
    }
    __shared__ double s_red[WPB][13];
    const int idx = (int)blockIdx.x - nlead;
    const int frame = idx / nsb, t = idx - frame * nsb;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double acc[13];
#pragma unroll
    for (int l = 0; l < 13; ++l) acc[l] = 0.0;
    {
        const int task = t * WPB + wave;
        if (task < (g.nstrips - 1) * g.nsegs) {
            const int seg = task / (g.nstrips - 1), k = task - seg * (g.nstrips - 1) + 1;  // the boundary in front of strip k
            const int rs = g.row_lo + seg * g.rps, re = rs + g.rps < g.row_hi ? rs + g.rps : g.row_hi;
            const int S = k * STRIP;
            const bool left_too = !(g.shift_last && S + STRIP > g.cols);  // strip k is not a shifted last strip
            const float4* sm = reinterpret_cast<const float4*>(ho.seam) + ((long long)frame * (g.nstrips - 1) + (k - 1)) * g.rows;
            gram_colseam(sm, g.rows, rs, re, left_too, lane, acc);
        }
    }
    {
        int ix;
        const double s = wave_sum_multi<13>(acc, lane, ix);
        if (ix < 13) s_red[wave][ix] = s;
    }
    // the embed's wave records are complete before this launch starts: every seam block folds its share of them (records t,
    // t + nsb, ... in that order) into its own record, so that the frame's last block folds nsb records instead of nrec + nsb
    double wsum = 0.0;
    if (threadIdx.x < 13) {
        const double* wr = ho.rec + (long long)frame * ho.stride * 13 + threadIdx.x;
        for (int r0 = t; r0 < g.nrec; r0 += 4 * nsb) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = wr[(long long)min(r0 + u * nsb, g.nrec - 1) * 13];
#pragma unroll
            for (int u = 0; u < 4; ++u) wsum += r0 + u * nsb < g.nrec ? v[u] : 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x < 13)
        st_agent(ho.rec + ((long long)frame * ho.stride + g.nrec + t) * 13 + threadIdx.x,
                 (((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x]) + wsum);
    if (last_block_of_frame(tail.ticket + frame * TKS, (unsigned)tail.expected))
        solve_frame(frame, ho.rec + (long long)g.nrec * 13, nsb, pborder, tail.nbb_total, tail.coef, tail.status, tail.gram_tot, ho.stride);
}

// band mode (intra-frame sharding): the Gram totals of a frame were all-reduced over the ranks; solve from them
__global__ __launch_bounds__(WAVE) void k_solve_totals(const double* __restrict__ totals, float* __restrict__ coef,
                                                       int* __restrict__ status)
{
    __shared__ double s_tot[NGRAM];
    const int frame = blockIdx.x, t = threadIdx.x;
    if (t < NGRAM) s_tot[t] = totals[(long long)frame * NGRAM + t];
    wave_lds_fence();
    lu_solve_wave(s_tot, t, frame, coef, status);
}

// ---- band mode with the totals resident in device memory (wm_band_*_dev): the small glue between the sweeps and the
// collectives the caller runs on the same stream
// the raw totals a stats / detect sweep's fold tail left (RawSums) as n consecutive doubles per frame
__global__ void k_band_pick(const RawSums* __restrict__ raw, int n, double* __restrict__ out)
{
    const int f = blockIdx.x, t = threadIdx.x;
    if (t < n) out[(long long)f * n + t] = raw[f].v[t];
}
// the strength from the parts every band contributed ({max|e|, sum} pairs, gathered: parts[part][frame][2]); parts are folded
// in index order, so every rank computes the same bits.  a = sF / (float)(||u|| / sqrt(N))   (Watermark.cpp:170)
__global__ void k_band_scalars(const double* __restrict__ parts, int nparts, int frames, int mask, float sF, double sqrt_n,
                               const int* __restrict__ status, EmbedScalars* __restrict__ scal, float* __restrict__ a_dev)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    double mx = 0.0, ss = 0.0;
    for (int p = 0; p < nparts; ++p) {
        const double* q = parts + ((long long)p * frames + f) * 2;
        mx = fmax(mx, q[0]);
        ss += q[1];
    }
    EmbedScalars sc;
    sc.maxe = mask == 0 ? (float)mx : 1.0f;
    const double nrm = mask == 0 ? sqrt(ss) / (double)sc.maxe : sqrt(ss);
    sc.a = sF / (float)(nrm / sqrt_n);
    scal[f] = sc;
    // unsolvable: the reference leaves the strength unset (Watermark.cpp:164-165); a NaN says so on the device
    if (a_dev) a_dev[f] = (status && status[f] != 0) ? __int_as_float(0x7fc00000) : sc.a;
}
// corr = (float)dot / (float)(||e_w|| ||e_u||)   (Watermark.cpp:230) from all-reduced sums [frames][3]; 0.0f when unsolvable (:246-247)
__global__ void k_band_corr(const double* __restrict__ sums, int frames, const int* __restrict__ status, float* __restrict__ corr)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    const double* q = sums + (long long)f * 3;
    corr[f] = status[f] != 0 ? 0.0f : (float)q[0] / (float)(sqrt(q[2]) * sqrt(q[1]));
}

// launchers
void launch_band_pick(hipStream_t s, int frames, const RawSums* raw, int n, double* out)
{
    hipLaunchKernelGGL(k_band_pick, dim3(frames), dim3(4), 0, s, raw, n, out);
}
void launch_band_scalars(hipStream_t s, int frames, const double* parts, int nparts, int mask, float sF, double sqrt_n, const int* status,
                         EmbedScalars* scal, float* a_dev)
{
    hipLaunchKernelGGL(k_band_scalars, dim3((frames + 63) / 64), dim3(64), 0, s, parts, nparts, frames, mask, sF, sqrt_n, status, scal, a_dev);
}
void launch_band_corr(hipStream_t s, int frames, const double* sums, const int* status, float* corr)
{
    hipLaunchKernelGGL(k_band_corr, dim3((frames + 63) / 64), dim3(64), 0, s, sums, frames, status, corr);
}
void launch_solve_totals(hipStream_t s, int frames, const double* totals, float* coef, int* status)
{
    hipLaunchKernelGGL(k_solve_totals, dim3(frames), dim3(WAVE), 0, s, totals, coef, status);
}

int handover_seam_blocks(const LaunchGeom& lg)
{
    const int n = ((lg.nstrips - 1) * lg.nsegs + WPB - 1) / WPB;
    return n > 0 ? n : 1;  // one strip: a block without seams still folds the wave records
}

void launch_gram_ho(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& y, const HandOver& ho, double* pborder,
                    unsigned* ticket, float* coef, int* status, double* gram_tot)
{
    // the geometry of the embed's aligned launch (every strip on the aligned path: launch_embed's hand-over condition)
    const SweepPart pv = sweep_part(lg, frames, true, 2);
    const int nsb = handover_seam_blocks(lg);
    const SolveTail tail{ticket, lg.nbb + nsb, lg.nbb, coef, status, gram_tot};
    const dim3 grid((unsigned)((lg.nbb + nsb) * frames), 1, 1);
    WM_KLAUNCH(k_gram_ho, grid, dim3(BLOCK), 0, s, (const float*)y.p, y.pitch, y.fstride, pv.g, lg.nbb, nsb, ho, pborder, tail);
}

void launch_gram(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, double* pmain, double* pborder,
                 unsigned* ticket, float* coef, int* status, double* gram_tot)
{
    // the border blocks ride in the first launch of the sweep (the aligned-path one when it exists)
    const int al = align_mode(lg, x.aligned != 0);
    const SweepPart pv = sweep_part(lg, frames, true, al);
    const SweepPart pg = sweep_part(lg, frames, false, al);
    const int nbb_v = pv.run ? lg.nbb : 0;
    const int nbb_g = pv.run ? 0 : lg.nbb;
    // every block of both launches takes a ticket of its frame; the last one folds the partials and solves
    const SolveTail tail{ticket, lg.nblk + lg.nbb, lg.nbb, coef, status, gram_tot};
    if (pv.run) {
        Geom g = pv.g;
        g.frame_fastest = 0;  // the Gram sweep reads no W: keep a frame's tiles together (halo rows stay in L2)
        const dim3 grid(pv.grid.x + nbb_v * frames, 1, 1);
        WM_DISPATCH_T(x.dtype, WM_KLAUNCH((k_gram<T, true>), grid, dim3(BLOCK), 0, s, (const T*)x.p, x.pitch, x.fstride, g, nbb_v,
                                                   pmain, pborder, tail));
    }
    if (pg.run) {
        Geom g = pg.g;
        g.frame_fastest = 0;
        const dim3 grid(pg.grid.x + nbb_g * frames, 1, 1);
        WM_DISPATCH_T(x.dtype, WM_KLAUNCH((k_gram<T, false>), grid, dim3(BLOCK), 0, s, (const T*)x.p, x.pitch, x.fstride, g, nbb_g,
                                                   pmain, pborder, tail));
    }
}

}  // namespace wmk
