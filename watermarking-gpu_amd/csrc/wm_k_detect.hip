// wm_k_detect.hip -- detector kernel k_detect with its fold tail corr_finalize_frame (see wm_k_gram.hip header)
#include "wm_march.hpp"
#include <cstdlib>

#ifndef WM_DET_RING
#define WM_DET_RING 6   // x rows of k_detect's aligned 3x3 path: ring length (rows in flight = ring - 3).  Measured: 9 and 12 (6 and
                        // 9 rows in flight) gain nothing and cost a wave per SIMD
#endif
#ifndef WM_DET_EXP
#define WM_DET_EXP 0   // timing experiments (wrong results): 1 no prediction chains, 3 no e_u chain
#endif
#ifndef WM_DET_WAVES
#define WM_DET_WAVES 4
#endif
#ifndef WM_PFW_DET
#define WM_PFW_DET 3   // W rows are L2 hits (the frames of a block share them): 3 in flight suffice and leave k_detect at 4 waves per SIMD
#endif


namespace wmk {

constexpr int PFWD = WM_PFW_DET;  // rows of W (and of W's halo column) prefetched per wave in k_detect

// =================================================================================================
// k_detect: one fused sweep over the test image and W:
//   e_w = x - c.nbrs(x);  u = m W  (ME: m ~ |e_w|, the max|e_w| normalisation cancels in the
//   correlation; NVF: m = nvf(x));  e_u = u - c.nbrs(u)  with u replicate-padded;
//   per block: <e_u,e_w>, ||e_u||^2, ||e_w||^2          (Watermark.cpp:221-250)
// =================================================================================================
template <typename T, int MASK, int PAD, int HC, bool VEC, bool EDGE>
__device__ __forceinline__ void detect_march(const T* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                             const Geom& g, const WaveJob& j, float* lds_x, float* lds_u,
                                             const float (&c)[8], float& dot, float& nu, float& nw)
{
    constexpr int HRX = MASK == 0 ? 1 : PAD;  // x rows needed above/below a u row
    constexpr int NR = 2 * HRX + 1;
    constexpr int O = 4 * HC;                 // own chunk offset in window rows
    constexpr int MID = HRX;                  // window row of the u row being produced
    const int R = g.rows, C = g.cols;
    float nc[8];  // the negated coefficients of residual4 (wave-uniform: SGPRs)
#pragma unroll
    for (int k = 0; k < 8; ++k) nc[k] = -c[k];
    // u rows t0..t1 are computed; x rows t0-HRX .. t1+HRX are streamed (clamped at load)
    const int t0 = j.rs > 0 ? j.rs - 1 : 0;
    const int t1 = j.re < R ? j.re : R - 1;
    const int nu_rows = t1 - t0 + 1;
    const int n = nu_rows + 2 * HRX;
    // (aligned path, 3x3 masks) OVERLAPPED STRIPS (Geom::sstride / lead, WaveJob::lo / hi): the wave loads 256 consecutive
    // columns of which lanes lo .. hi own theirs; lane 0 and lane 63 (when they are not owners: everywhere but at the image's
    // left border) evaluate e_w and u like every lane and exist to hand them to lanes 1 and 62 by DPP.  Nothing at a strip's
    // halo column is loaded or computed separately: the per-row halo loads (2 of 4 loads), the halo prediction every lane
    // evaluated for the two that used it and the selects that routed its inputs (~25 of ~120 vector instructions per row) are
    // gone, for 2 of 64 lanes that own nothing (4K: 16 strips of 248 columns instead of 15 of 256).  NVF with p = 5, 7 runs the same
    // way since round 4 (the mask's 2 or 3 halo columns lie inside the provider lane's 4 pixels; until then every lane evaluated the
    // mask at a halo column for the two lanes that used it: a third of the row's arithmetic); p = 9 takes the generic path.
    constexpr bool HALO1 = VEC && HC == 1 && (MASK == 0 || PAD <= 3);  // (p = 5, 7 as well: the mask's halo columns lie inside the provider lane)
    constexpr int DR = HALO1 && NR == 3 ? WM_DET_RING : UNROLL;
    XMarch<T, HC, HALO1 ? HRX : HRX + 1, NR, VEC, PFX, EDGE, HALO1, DR> xm;
    PMarch<float, VEC, PFWD> wm_;
    const int c0 = j.c0s + 4 * j.lane;
    const bool left_edge = EDGE && j.c0s == 0;
    const bool has_right = !EDGE || j.c0s + STRIP <= C - 1;  // column c0s+STRIP exists in the image
    xm.start(xf, pitch, g, j, lds_x, t0 - HRX, n);
    wm_.start(W, C, C, j, t0, nu_rows);
    // W at the strip's halo columns c0s-1 (lanes != 63) and c0s+STRIP (lane 63): loaded by every lane, no branch (variants
    // without the gather)
    const int wh_col = j.lane == WAVE - 1 ? (j.c0s + STRIP < C ? j.c0s + STRIP : C - 1) : (j.c0s > 0 ? j.c0s - 1 : 0);
    const unsigned wh_off = (unsigned)wh_col * 4u;
    const float* whp = W + wh_col;
    auto load_wh = [&](int r) -> float {
        if constexpr (HALO1) return 0.0f;
        else if constexpr (VEC) return buf_load<float>(wm_.ps.rs, wh_off, (unsigned)r * wm_.ps.pitch_b);  // (a row of W: scalar offset)
        else return whp[(long long)r * C];
    };
    float whpre[PFWD];
#pragma unroll
    for (int s = 0; s < PFWD; ++s) whpre[s] = load_wh(min(t0 + s, t1));
    // rolling window of u rows (left neighbour, 4 own, right neighbour) in rotating slots, e_w of two rows
    float uw[3][6];
    float eww[3][4];  // (three slots: the ring length DR may be odd)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) uw[a][b] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) eww[a][b] = 0.f;
    const int last_col_local = C - 1 - j.c0s;  // strip-local index of the image's last column
    // lanes that own their 4 columns: not the duplicate lanes of a shifted last strip (their sums belong to the previous strip),
    // and with overlapped strips (HALO1) only lanes lo .. hi -- those sum everything and are masked once, at the end
    const bool own = HALO1 || !EDGE || 4 * j.lane >= j.dup;
    march_n<2 * HRX, DR>(n, [&](int i, auto qc, auto emit) {
        constexpr int Q = decltype(qc)::value;
        xm.template step<Q>(i);
        if (decltype(emit)::value) {
            const int o = i - 2 * HRX;  // u row index t = t0 + o; its slots: uw[Q % 3], eww[Q % 2]
            const int t = t0 + o;
            constexpr int SLOT = (Q + 2 * DR - 2 * HRX) % PFWD;
            const float4 w = wm_.template take<SLOT>();
            const float wh = HALO1 ? 0.0f : pinned(whpre[SLOT]);
            const float* xup = xm.template row<Q>(MID - 1);
            const float* xmid = xm.template row<Q>(MID);
            const float* xdn = xm.template row<Q>(MID + 1);
            // ---- e_w and u of row t for the 4 own pixels
            float uu[4];
            float* ew = eww[Q % 3];
            float ewn[4];
#if WM_DET_EXP == 1
            ewn[0] = xup[O]; ewn[1] = xup[O + 1]; ewn[2] = xdn[O + 2]; ewn[3] = xdn[O + 3];  // (timing experiment: no prediction chains)
#else
            residual4<O>(xup, xmid, xdn, nc, ewn);  // e_w = x - c.nbrs(x), the subtraction folded into the chain (wm_device.hpp)
#endif
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ew[k] = ewn[k];
                const float m = MASK == 0 ? fabsf(ew[k]) : nvf_value<PAD, O, Q>(xm, k);
                uu[k] = m * f4get(w, k);
            }
            float* un = uw[Q % 3];
            if constexpr (HALO1) {
                // neighbours' u by DPP wave shifts.  Lane 0 / lane 63 keep their own border pixel (the "old" operand): that is
                // the replicate border u(-1) := u(0) where lane 0 owns the image's first column, and never used where they are
                // provider lanes; at the image's right border the lane that holds the last column takes u(C) := u(C-1) itself
                if constexpr (EDGE) {
                    un[0] = dpp_from_prev(uu[3], uu[0]);
                    const float nx = dpp_from_next(uu[0], uu[3]);
                    un[5] = xm.xs.rsel ? uu[3] : nx;
                } else {
                    un[0] = dpp_from_prev_any(uu[3]);
                    un[5] = dpp_from_next_any(uu[0]);
                }
            } else {
                // replicate border inside the own chunk: u(c) := u(C-1) for c >= C
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (c0 + k >= C) uu[k] = uu[k - 1];
                // ---- publish the u row through LDS: own chunk, strip halo columns, replicate border
                float* urow = lds_u + (Q & 1) * RowBuf<1>::N;
                reinterpret_cast<float4*>(urow)[1 + j.lane] = make_float4(uu[0], uu[1], uu[2], uu[3]);
                if (j.lane == 0) {
                    float uh;
                    if (left_edge) uh = uu[0];
                    else {
                        const float eh = residual1<O>(xup, xmid, xdn, -1, nc);
                        const float m = MASK == 0 ? fabsf(eh) : nvf_value<PAD, O, Q>(xm, -1);
                        uh = m * wh;
                    }
                    urow[3] = uh;
                }
                if (j.lane == WAVE - 1 && has_right) {
                    const float eh = residual1<O>(xup, xmid, xdn, 4, nc);
                    const float m = MASK == 0 ? fabsf(eh) : nvf_value<PAD, O, Q>(xm, 4);
                    urow[4 + STRIP] = m * wh;
                }
                if (!has_right) {
                    // image's last column lies in this strip: u(C) := u(C-1)
                    const int lk = last_col_local - 4 * j.lane;
                    if (lk >= 0 && lk < 4) urow[4 + last_col_local + 1] = uu[lk];
                }
                wave_lds_fence();
                un[0] = urow[3 + 4 * j.lane];
                un[5] = urow[8 + 4 * j.lane];
            }
            un[1] = uu[0]; un[2] = uu[1]; un[3] = uu[2]; un[4] = uu[3];
            if (o == 0 && j.rs == 0) {
                // u(-1) := u(0): the first computed row is image row 0; seed the slot the next step reads as "um"
#pragma unroll
                for (int b = 0; b < 6; ++b) uw[(Q + 2) % 3][b] = un[b];
            }
            // ---- emit e_u for row r = t-1: u rows r-1, r, r+1 are slots (Q+1)%3, (Q+2)%3, Q%3
            const int r = t - 1;
            if (r >= j.rs && r < j.re) {
                const float* um = uw[(Q + 1) % 3];
                const float* u0 = uw[(Q + 2) % 3];
                const float* ewp = eww[(Q + 2) % 3];
                float eun[4];
#if WM_DET_EXP == 1 || WM_DET_EXP == 3
                eun[0] = um[1]; eun[1] = um[2]; eun[2] = un[3]; eun[3] = un[4];  // (timing experiment)
#else
                residual4<1>(um, u0, un, nc, eun);  // e_u = u - c.nbrs(u)
#endif
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (VEC ? own : (c0 + k < C && c0 + k >= j.own_c0)) {
                        const float eu = eun[k];
                        dot = fmaf(eu, ewp[k], dot);
                        nu = fmaf(eu, eu, nu);
                        nw = fmaf(ewp[k], ewp[k], nw);
                    }
                }
            }
            if (j.re == R && t == R - 1) {
                // last image row: u(R) := u(R-1); window (u(R-2), u(R-1), u(R-1))
                const float* u0 = uw[(Q + 2) % 3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (VEC ? own : (c0 + k < C && c0 + k >= j.own_c0)) {
                        const float eu = residual1<1>(u0, un, un, k, nc);
                        dot = fmaf(eu, ew[k], dot);
                        nu = fmaf(eu, eu, nu);
                        nw = fmaf(ew[k], ew[k], nw);
                    }
                }
            }
            wm_.template refill<SLOT>(o);
            if constexpr (!HALO1) {
                __builtin_amdgcn_sched_barrier(0);
                whpre[SLOT] = load_wh(min(t + PFWD, t1));
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    });
    if constexpr (HALO1) {
        // the provider lanes (and the lanes beyond the image's last column) summed pixels other lanes own: drop their sums
        const bool mine = j.lane >= j.lo && j.lane <= j.hi;
        dot = mine ? dot : 0.0f; nu = mine ? nu : 0.0f; nw = mine ? nw : 0.0f;
    }
}

// corr_fold (tail of k_detect; take_ticket, wm_device.hpp): the last wave of a strip folds the strip's records, the last
// strip's wave folds the frame:
// corr = (float)dot / (float)(||e_w|| * ||e_u||)   (Watermark.cpp:230); unsolvable => 0.0f (:246-247)
__device__ __forceinline__ void corr_fold(int frame, const WaveJob& j, const double* pcorr, int nrec,
                                          const int* __restrict__ status, const CorrTail& tl)
{
    const int lane = j.lane;
    if (!take_ticket(tl.ticket_strip + (frame * tl.nstrips + j.strip) * TKS, (unsigned)tl.nsegs, lane)) return;
    // all loads of a batch are issued before the first is used (index clamped, surplus terms dropped): agent-scope loads
    // come from the memory side, a dependent chain of them costs a memory latency per term
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int s0 = lane; s0 < tl.nsegs; s0 += 2 * WAVE) {
        double v[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const double* p = pcorr + ((long long)frame * nrec + (long long)min(s0 + u * WAVE, tl.nsegs - 1) * tl.nstrips + j.strip) * 3;
            v[u][0] = ld_agent(p); v[u][1] = ld_agent(p + 1); v[u][2] = ld_agent(p + 2);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool in = s0 + u * WAVE < tl.nsegs;
            a0 += in ? v[u][0] : 0.0; a1 += in ? v[u][1] : 0.0; a2 += in ? v[u][2] : 0.0;
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) {
        double* q = tl.scorr + ((long long)frame * tl.nstrips + j.strip) * 3;
        st_agent(q, a0); st_agent(q + 1, a1); st_agent(q + 2, a2);
    }
    if (!take_ticket(tl.ticket + frame * TKS, (unsigned)tl.nstrips, lane)) return;
    a0 = 0.0; a1 = 0.0; a2 = 0.0;
    for (int s0 = lane; s0 < tl.nstrips; s0 += WAVE) {
        const double* q = tl.scorr + ((long long)frame * tl.nstrips + s0) * 3;
        const double v0 = ld_agent(q), v1 = ld_agent(q + 1), v2 = ld_agent(q + 2);
        a0 += v0; a1 += v1; a2 += v2;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) {
        const int st = status[frame];
        float corr = 0.0f;
        if (st == 0) corr = (float)a0 / (float)(sqrt(a2) * sqrt(a1));
        tl.res[frame].status = st;
        tl.res[frame].value = corr;
        RawSums rw;
        rw.v[0] = a0; rw.v[1] = a1; rw.v[2] = a2; rw.v[3] = 0.0;
        tl.raw[frame] = rw;
    }
}

// corr_finalize_frame (tail of k_detect, run by the frame's last block):
// corr = (float)dot / (float)(||e_w|| * ||e_u||)   (Watermark.cpp:230); unsolvable => 0.0f (:246-247)
__device__ __forceinline__ void corr_finalize_frame(int frame, const double* pcorr, int nblk, const int* __restrict__ status,
                                                    OpResult* __restrict__ res, RawSums* __restrict__ raw)
{
    __shared__ double s[3][BLOCK];
    const int t = threadIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    // 2 x 3 partials in flight per thread (index clamped, surplus terms dropped), see solve_frame
    for (int b0 = t; b0 < nblk; b0 += 2 * BLOCK) {
        double v[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const double* p = pcorr + ((long long)frame * nblk + min(b0 + u * BLOCK, nblk - 1)) * 3;
            v[u][0] = ld_agent(p); v[u][1] = ld_agent(p + 1); v[u][2] = ld_agent(p + 2);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool in = b0 + u * BLOCK < nblk;
            a0 += in ? v[u][0] : 0.0; a1 += in ? v[u][1] : 0.0; a2 += in ? v[u][2] : 0.0;
        }
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
        RawSums rw;
        rw.v[0] = s[0][0]; rw.v[1] = s[1][0]; rw.v[2] = s[2][0]; rw.v[3] = 0.0;
        raw[frame] = rw;
    }
}

template <typename T, int MASK, int PAD, int HC, bool VEC>
// occupancy floor: 4 waves per SIMD for the aligned 3x3 instances (105 / 106 / 91 VGPRs); the generic instances (LDS re-lay,
// halo predictions of their own) need ~150 registers -- bound to 4 they spilled 48-70 VGPRs to scratch, at 3 they do not
__global__ __launch_bounds__(BLOCK, (PAD == 1 && HC == 1 ? (VEC ? WM_DET_WAVES : 3) : 1)) void k_detect(const T* __restrict__ x, long long pitch, long long fstride,
                                                  const float* __restrict__ W, Geom g,
                                                  const float* __restrict__ coef, const int* __restrict__ status,
                                                  double* pcorr, CorrTail tail)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<HC>::N];
    __shared__ __attribute__((aligned(16))) float s_u[WPB][2 * RowBuf<1>::N];
    __shared__ double s_red[WPB][3];
    const WaveJob j = make_job(g);
    const int frame = j.frame;
    float dot = 0.0f, nu = 0.0f, nw = 0.0f;
    if (j.valid && status[frame] == 0) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = coef[frame * 8 + k];
        const T* xf = x + (long long)frame * fstride;
        constexpr bool DPP_OK = HC == 1;  // the halo of p = 9 (HC = 2) exceeds one neighbour chunk: LDS path only
        constexpr bool V = VEC && DPP_OK;
        if (MASK != 0 || strip_on_edge<V>(g, j)) detect_march<T, MASK, PAD, HC, V, true>(xf, pitch, W, g, j, s_row[j.wave], s_u[j.wave], c, dot, nu, nw);
        else detect_march<T, MASK, PAD, HC, V, (MASK != 0)>(xf, pitch, W, g, j, s_row[j.wave], s_u[j.wave], c, dot, nu, nw);
    }
    const double d0 = wave_sum((double)dot), d1 = wave_sum((double)nu), d2 = wave_sum((double)nw);
    if (g.quad) {
        // the waves of this block are 4 frames: one record per wave, folded per strip and then per frame (corr_fold)
        if (!j.valid) return;  // surplus wave of a short last quad (wave-uniform; no barrier below)
        if (j.lane == 0) {
            double* p = pcorr + ((long long)frame * g.nrec + j.rec) * 3;
            st_agent(p, d0); st_agent(p + 1, d1); st_agent(p + 2, d2);
        }
        corr_fold(frame, j, pcorr, g.nrec, status, tail);
        return;
    }
    // the waves of this block are 4 segments of one frame: one record per block, folded by the frame's last block
    if (j.lane == 0) { s_red[j.wave][0] = d0; s_red[j.wave][1] = d1; s_red[j.wave][2] = d2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        st_agent(pcorr + ((long long)frame * g.nblk_total + g.pb0 + j.tile) * 3 + k, ((s_red[0][k] + s_red[1][k]) + s_red[2][k]) + s_red[3][k]);
    }
    if (last_block_of_frame(tail.ticket + frame * TKS, (unsigned)tail.expected))
        corr_finalize_frame(frame, pcorr, g.nblk_total, status, tail.res, tail.raw);
}

// results of a mask-only op: status + coefficients
__global__ void k_mask_result(const int* __restrict__ status, const float* __restrict__ coef, OpResult* __restrict__ res,
                              float* __restrict__ coef_out)
{
    const int frame = blockIdx.x, t = threadIdx.x;
    if (t == 0) { res[frame].status = status ? status[frame] : 0; res[frame].value = 0.0f; }
    if (t < 8) coef_out[frame * 8 + t] = coef ? coef[frame * 8 + t] : 0.0f;
}

// launchers
template <typename T>
static void launch_detect_t(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                            const float* W, int aligned_w, const float* coef, const int* status, double* pcorr,
                            const CorrTail& tail, bool split)
{
#define DET(MASK, P, HC)                                                                                                      \
    WM_LAUNCH_SWEEP_Q(s, lg, frames, align_mode(lg, x.aligned && aligned_w && HC == 1), (k_detect<T, MASK, P, HC, true>), (k_detect<T, MASK, P, HC, false>), \
                    (const T*)x.p, x.pitch, x.fstride, W, g, coef, status, pcorr, tail)
    // 3x3 masks: the aligned instantiation works on overlapped strips (every strip, when all planes allow vector access and
    // the width is a multiple of 4); otherwise the whole image takes the generic instantiation
#define DET3P(MASK, P)                                                                                                          \
    do {                                                                                                                      \
        if (align_mode(lg, x.aligned && aligned_w) == 2) {                                                                    \
            const SweepPart pv_ = sweep_part_overlap(lg, frames, 1);                                                          \
            const Geom g = pv_.g;                                                                                             \
            WM_KLAUNCH((k_detect<T, MASK, P, 1, true>), pv_.grid, dim3(BLOCK), 0, s, (const T*)x.p, x.pitch, x.fstride, W, g, coef, status, pcorr, tail); \
        } else if (split) {                                                                                                   \
            /* a width that is not a multiple of 4: overlapped strips below column B + one generic strip (wm_march.hpp) */     \
            { const SweepPart pv_ = sweep_part_split_overlap(lg, frames, 1); const Geom g = pv_.g;                              \
              WM_KLAUNCH((k_detect<T, MASK, P, 1, true>), pv_.grid, dim3(BLOCK), 0, s, (const T*)x.p, x.pitch, x.fstride, W, g, coef, status, pcorr, tail); } \
            { const SweepPart pg_ = sweep_part_split_generic(lg, frames, 1); const Geom g = pg_.g;                              \
              WM_KLAUNCH((k_detect<T, MASK, P, 1, false>), pg_.grid, dim3(BLOCK), 0, s, (const T*)x.p, x.pitch, x.fstride, W, g, coef, status, pcorr, tail); } \
        } else {                                                                                                              \
            WM_LAUNCH_SWEEP_Q(s, lg, frames, 0, (k_detect<T, MASK, P, 1, true>), (k_detect<T, MASK, P, 1, false>),           \
                              (const T*)x.p, x.pitch, x.fstride, W, g, coef, status, pcorr, tail);                            \
        }                                                                                                                     \
    } while (0)
    if (mask == 0) { DET3P(0, 1); return; }
    switch (pad) {
        case 1: DET3P(1, 1); break;
        case 2: DET3P(1, 2); break;
        case 3: DET3P(1, 3); break;
        case 4: DET(1, 4, 2); break;
    }
#undef DET
#undef DET3P
}
void launch_detect(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                   int aligned_w, const float* coef, const int* status, double* pcorr, unsigned* ticket, unsigned* ticket_strip,
                   double* scorr, OpResult* res, RawSums* raw)
{
    // the aligned 3x3 path runs on overlapped strips: more, narrower strips than the other sweeps of the call (overlap_geom);
    // the records, the strip tickets and the fold follow that strip count
    const bool overlap = (mask == 0 || pad <= 3) && align_mode(lg, x.aligned && aligned_w) == 2;
    // ... and for widths that are not multiples of 4 on planes that allow vector access: overlapped strips + one generic strip
    const bool split = (mask == 0 || pad == 1) && !overlap && x.aligned && aligned_w && split_applies(lg.cols);
    const LaunchGeom ld = overlap ? overlap_geom(lg) : (split ? split_geom(lg) : lg);
    const CorrTail tail{ticket, ticket_strip, ld.nblk, ld.nsegs, ld.nstrips, scorr, res, raw};
    WM_DISPATCH_T(x.dtype, launch_detect_t<T>(s, ld, frames, mask, pad, x, W, aligned_w, coef, status, pcorr, tail, split));
}

// ---- W on the device: the counter-based N(0,1) generator of csrc/app/wm_genw.cpp (the replacement of the reference's
// CommonRandomMatrix tool, CommonRandomMatrix/main.cpp:34-51): element (r, c) is a pure function of (seed, r, c) -- a 32-bit
// hash, two uniforms, Box-Muller in f64 -- so every GPU of a node fills its own copy without a file, an upload or a broadcast
__device__ __forceinline__ uint32_t genw_mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t genw_hash(uint32_t seed, uint32_t stream, uint32_t r, uint32_t c)
{
    const uint32_t h = genw_mix(seed ^ genw_mix(stream * 0x9E3779B1u + r));
    return genw_mix(h ^ genw_mix(c + 0x85EBCA6Bu));
}
__global__ void k_gen_w(float* __restrict__ w, int rows, int cols, uint32_t seed)
{
    const long long n = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / cols), c = (uint32_t)(i - (long long)r * cols);
        const double u1 = ((double)genw_hash(seed, 0x5741u, r, c) + 1.0) / 4294967297.0;
        const double u2 = (double)genw_hash(seed, 0x5742u, r, c) / 4294967296.0;
        w[i] = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2));
    }
}
void launch_gen_w(hipStream_t s, float* w, int rows, int cols, uint32_t seed)
{
    hipLaunchKernelGGL(k_gen_w, dim3(2048), dim3(256), 0, s, w, rows, cols, seed);
}

// ---- exhaustive self-test of the NVF quotient (nvf_quot, wm_device.hpp): every f32 bit pattern in [lo, hi) as `var`,
// the sequence's var / (1 + var) against the compiler's IEEE division (hipcc divides f32 correctly rounded by default).
// out2[0] = values whose results differ in any bit (NaN results compare equal), out2[1] = the smallest such bit pattern
template <int VARIANT>
__global__ void k_selftest_quot(uint32_t lo, uint32_t hi, unsigned long long* out2)
{
    unsigned long long bad = 0, first = ~0ull;
    for (unsigned long long b = (unsigned long long)lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < hi;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float var = __uint_as_float((uint32_t)b);
        const float d = 1.0f + var;
        const float ref = var / d;
        const float got = nvf_quot_variant<VARIANT>(var, d);
        const bool same = __float_as_uint(ref) == __float_as_uint(got) || (ref != ref && got != got);
        if (!same) { ++bad; if (b < first) first = b; }
    }
    if (bad) { atomicAdd(out2, bad); atomicMin(out2 + 1, first); }
}
void launch_selftest_quot(hipStream_t s, int variant, uint32_t bits_lo, uint32_t bits_hi, unsigned long long* out2)
{
    if (variant == 0) hipLaunchKernelGGL(k_selftest_quot<0>, dim3(4096), dim3(256), 0, s, bits_lo, bits_hi, out2);
    else if (variant == 1) hipLaunchKernelGGL(k_selftest_quot<1>, dim3(4096), dim3(256), 0, s, bits_lo, bits_hi, out2);
    else if (variant == 2) hipLaunchKernelGGL(k_selftest_quot<2>, dim3(4096), dim3(256), 0, s, bits_lo, bits_hi, out2);
    else hipLaunchKernelGGL(k_selftest_quot<3>, dim3(4096), dim3(256), 0, s, bits_lo, bits_hi, out2);
}

// ---- memory-system yardstick (wm_membench, wm.h): grid-stride streams of 16-byte elements, 2048 blocks x 256 threads like
// the sweeps' grids; stores are non-temporal buffer stores (store4's form), loads plain 16-byte buffer-free global loads
template <int KIND, int UNR>
__global__ __launch_bounds__(256) void k_membench(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16, unsigned long long* sink)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    // UNR elements in flight per thread and trip
    for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (KIND == 0) v[u] = make_float4((float)i, 1.0f, 2.0f, (float)u);
            else v[u] = src[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (KIND == 2) acc += v[u].x + v[u].w;
            else {
                f4v w; w.x = v[u].x; w.y = v[u].y; w.z = v[u].z; w.w = v[u].w;
                __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(dst + i + u * stride));
            }
        }
    }
    for (; i < n16; i += stride) {
        const float4 v = KIND == 0 ? make_float4((float)i, 1.0f, 2.0f, 3.0f) : src[i];
        if (KIND == 2) acc += v.x + v.w;
        else { f4v w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(dst + i)); }
    }
    if (KIND == 2 && acc == 123456.789f) atomicAdd(sink, 1ull);  // (keeps the loads alive)
}
void launch_membench(hipStream_t s, int kind, const void* src, void* dst, size_t n16, unsigned long long* sink, hipEvent_t a, hipEvent_t b)
{
    // two grid shapes (kind / 3): 0 = 2048 blocks x 4 elements in flight per thread, a grid like the sweeps' own (a few waves per
    // SIMD that stay for the whole launch); 1 = 65536 blocks x 1 element per thread, the shape that streams fastest on MI355X
    // (tools/membench_sweep.py: store 6.6, copy 6.2, read 6.4 TB/s against 4.4-5.0 / 4.3-5.0 / 5.3 in shape 0).  The environment
    // overrides both for that sweep
    const int shape = kind / 3;
    kind %= 3;
    static const int env_blocks = getenv("WM_MEMBENCH_BLOCKS") ? atoi(getenv("WM_MEMBENCH_BLOCKS")) : 0;
    static const int env_unr = getenv("WM_MEMBENCH_UNROLL") ? atoi(getenv("WM_MEMBENCH_UNROLL")) : 0;
    const int blocks = env_blocks > 0 ? env_blocks : (shape ? 65536 : 2048);
    const int unr = env_unr > 0 ? env_unr : (shape ? 1 : 4);
    const dim3 grid(blocks), block(256);
#define MB_LAUNCH(K, U) hipExtLaunchKernelGGL((k_membench<K, U>), grid, block, 0, s, a, b, 0, (const float4*)src, (float4*)dst, n16, sink)
#define MB_KIND(U) do { if (kind == 0) MB_LAUNCH(0, U); else if (kind == 1) MB_LAUNCH(1, U); else MB_LAUNCH(2, U); } while (0)
    if (unr <= 1) MB_KIND(1); else if (unr == 2) MB_KIND(2); else if (unr <= 4) MB_KIND(4); else MB_KIND(8);
#undef MB_KIND
#undef MB_LAUNCH
}

void launch_mask_result(hipStream_t s, int frames, const int* status, const float* coef, OpResult* res, float* coef_out)
{
    hipLaunchKernelGGL(k_mask_result, dim3(frames), dim3(64), 0, s, status, coef, res, coef_out);
}

}  // namespace wmk
