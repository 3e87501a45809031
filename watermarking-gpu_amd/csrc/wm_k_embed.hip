// wm_k_embed.hip -- embed-side kernels: k_me_stats, k_nvf_stats (fold tail embed_scalars_frame), k_embed, k_mask (see wm_k_gram.hip header)
#include "wm_march.hpp"

#ifndef WM_RING3
#define WM_RING3 UNROLL   // ring length of the 3-row x windows of k_me_stats / k_embed (rows in flight per wave = ring - 3)
#endif

#ifndef WM_HO_PFW
#define WM_HO_PFW 3   // W / base rows in flight per wave in the hand-over instantiation of k_embed: 3 instead of 6 brings it from
                      // 174 to 162 VGPRs, i.e. three waves per SIMD instead of two (+0.8 % frames/s on the hand-over leg)
#endif

namespace wmk {

// =================================================================================================
// stats_fold (tail of k_me_stats / k_nvf_stats; take_ticket, wm_device.hpp): after a wave stored its record, the
// last wave of a strip folds the strip, the last strip's wave folds the frame:
//      a = sF / (float)(||u|| / sqrt(N))   (Watermark.cpp:170)
//   ME : ||u|| = sqrt(sum (|e| W)^2) / max|e|     NVF: ||u|| = sqrt(sum (m W)^2)
// Sums run in record order over the lanes, then through the fixed DPP tree: deterministic.
// =================================================================================================
__device__ __forceinline__ void stats_fold(int frame, const WaveJob& j, const float* pmax, const double* pss, int nrec,
                                           const int* __restrict__ status, const ScalarsTail& tl)
{
    const int lane = j.lane;
    if (!take_ticket(tl.ticket_strip + (frame * tl.nstrips + j.strip) * TKS, (unsigned)tl.nsegs, lane)) return;
    // ---- this strip's wave records: index seg * nstrips + strip.  All loads of a batch are issued before the first is
    // used (index clamped, surplus terms dropped): agent-scope loads come from the memory side, a dependent chain of
    // them costs a memory latency per term
    float mx = 0.0f;
    double ss = 0.0;
    for (int s0 = lane; s0 < tl.nsegs; s0 += 4 * WAVE) {
        float vm[4];
        double vs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long idx = (long long)frame * nrec + (long long)min(s0 + u * WAVE, tl.nsegs - 1) * tl.nstrips + j.strip;
            vm[u] = pmax ? ld_agent(pmax + idx) : 0.0f;
            vs[u] = ld_agent(pss + idx);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = s0 + u * WAVE < tl.nsegs;
            mx = fmaxf(mx, in ? vm[u] : 0.0f);
            ss += in ? vs[u] : 0.0;
        }
    }
    mx = wave_max(mx);
    ss = wave_sum(ss);
    if (lane == 0) {
        if (pmax) st_agent(tl.smax + frame * tl.nstrips + j.strip, mx);
        st_agent(tl.sss + frame * tl.nstrips + j.strip, ss);
    }
    if (!take_ticket(tl.ticket + frame * TKS, (unsigned)tl.nstrips, lane)) return;
    // ---- the frame's strip records
    mx = 0.0f; ss = 0.0;
    for (int s0 = lane; s0 < tl.nstrips; s0 += WAVE) {
        const float vm = pmax ? ld_agent(tl.smax + frame * tl.nstrips + s0) : 0.0f;
        const double vs = ld_agent(tl.sss + frame * tl.nstrips + s0);
        mx = fmaxf(mx, vm);
        ss += vs;
    }
    mx = wave_max(mx);
    ss = wave_sum(ss);
    if (lane == 0) {
        const int st = status ? status[frame] : 0;
        EmbedScalars s;
        s.maxe = pmax ? mx : 1.0f;
        const double nrm = pmax ? sqrt(ss) / (double)s.maxe : sqrt(ss);
        s.a = tl.sF / (float)(nrm / tl.sqrt_n);
        tl.scal[frame] = s;
        tl.res[frame].status = st;
        tl.res[frame].value = s.a;
        RawSums rw;
        rw.v[0] = (double)s.maxe; rw.v[1] = ss; rw.v[2] = 0.0; rw.v[3] = 0.0;
        tl.raw[frame] = rw;
    }
}

// =================================================================================================
// embed_scalars_frame (tail of k_me_stats / k_nvf_stats, run by the frame's last block): fold the stats partials
//   a = sF / (float)(||u|| / sqrt(N))   (Watermark.cpp:170)
//   ME : ||u|| = sqrt(sum (|e| W)^2) / max|e|     NVF: ||u|| = sqrt(sum (m W)^2)
// =================================================================================================
__device__ __forceinline__ void embed_scalars_frame(int frame, const float* pmax, const double* pss, int nblk,
                                                    const int* __restrict__ status, const ScalarsTail& tl)
{
    __shared__ float s_mx[BLOCK];
    __shared__ double s_ss[BLOCK];
    const int t = threadIdx.x;
    float mx = 0.0f;
    double ss = 0.0;
    // 4 partials in flight per thread (index clamped, surplus terms dropped), see solve_frame
    for (int b0 = t; b0 < nblk; b0 += 4 * BLOCK) {
        float vm[4];
        double vs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long idx = (long long)frame * nblk + min(b0 + u * BLOCK, nblk - 1);
            vm[u] = pmax ? ld_agent(pmax + idx) : 0.0f;
            vs[u] = ld_agent(pss + idx);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = b0 + u * BLOCK < nblk;
            mx = fmaxf(mx, in ? vm[u] : 0.0f);
            ss += in ? vs[u] : 0.0;
        }
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
        s.a = tl.sF / (float)(nrm / tl.sqrt_n);
        tl.scal[frame] = s;
        tl.res[frame].status = st;
        tl.res[frame].value = s.a;
        RawSums rw;
        rw.v[0] = (double)s.maxe; rw.v[1] = s_ss[0]; rw.v[2] = 0.0; rw.v[3] = 0.0;
        tl.raw[frame] = rw;
    }
}

// =================================================================================================
// k_me_stats: e = x - c.nbrs;  per block: max|e| and sum (|e| W)^2
// =================================================================================================
template <typename T, bool VEC, bool EDGE>
__device__ __forceinline__ void me_stats_march(const T* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                               const Geom& g, const WaveJob& j, float* lds, const float (&c)[8], float& mx,
                                               float& ss)
{
    constexpr int RG = VEC ? WM_RING3 : UNROLL;
    XMarch<T, 1, 1, 3, VEC, PFX, EDGE, false, RG> xm;
    PMarch<float, VEC, PFW> wm_;
    const int nout = j.re - j.rs, n = nout + 2;
    xm.start(xf, pitch, g, j, lds, j.rs - 1, n);
    wm_.start(W, g.cols, g.cols, j, j.rs, nout);
    const int c0 = j.c0s + 4 * j.lane;
    const bool own = !EDGE || 4 * j.lane >= j.dup;  // duplicate lanes of a shifted last strip do not count
    march_n<2, RG>(n, [&](int i, auto qc, auto emit) {
        constexpr int Q = decltype(qc)::value;
        xm.template step<Q>(i);
        if (decltype(emit)::value) {
            constexpr int SLOT = (Q + 4 * UNROLL - 2) % PFW;
            const float4 w = wm_.template take<SLOT>();
            const float* up = xm.template row<Q>(0);
            const float* mid = xm.template row<Q>(1);
            const float* dn = xm.template row<Q>(2);
            float pr[4];
            predict4<4>(up, mid, dn, c, pr);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (VEC ? own : c0 + k < g.cols) {
                    const float e = mid[4 + k] - pr[k];
                    const float ae = fabsf(e);
                    mx = fmaxf(mx, ae);
                    const float t = ae * f4get(w, k);
                    ss = fmaf(t, t, ss);
                }
            }
            wm_.template refill<SLOT>(i - 2);
        }
    });
}

template <typename T, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_me_stats(const T* __restrict__ x, long long pitch, long long fstride,
                                                    const float* __restrict__ W, Geom g,
                                                    const float* __restrict__ coef, const int* __restrict__ status,
                                                    float* pmax, double* pss, ScalarsTail tail)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    __shared__ float s_mx[WPB];
    __shared__ double s_ss[WPB];
    const WaveJob j = make_job(g);
    const int frame = j.frame;
    float mx = 0.0f, ss = 0.0f;
    if (j.valid && status[frame] == 0) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = coef[frame * 8 + k];
        const T* xf = x + (long long)frame * fstride;
        if (strip_on_edge<VEC>(g, j)) me_stats_march<T, VEC, true>(xf, pitch, W, g, j, s_row[j.wave], c, mx, ss);
        else me_stats_march<T, VEC, false>(xf, pitch, W, g, j, s_row[j.wave], c, mx, ss);
    }
    mx = wave_max(mx);
    const double ssd = wave_sum((double)ss);
    if (g.quad) {
        // the waves of this block are 4 frames: one record per wave, folded per strip and then per frame (stats_fold)
        if (!j.valid) return;  // surplus wave of a short last quad (wave-uniform; no barrier below)
        if (j.lane == 0) {
            const long long pb = (long long)frame * g.nrec + j.rec;
            st_agent(pmax + pb, mx);
            st_agent(pss + pb, ssd);
        }
        stats_fold(frame, j, pmax, pss, g.nrec, status, tail);
        return;
    }
    // the waves of this block are 4 segments of one frame: one record per block, folded by the frame's last block
    if (j.lane == 0) { s_mx[j.wave] = mx; s_ss[j.wave] = ssd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long pb = (long long)frame * g.nblk_total + g.pb0 + j.tile;
        st_agent(pmax + pb, fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3])));
        st_agent(pss + pb, ((s_ss[0] + s_ss[1]) + s_ss[2]) + s_ss[3]);
    }
    if (last_block_of_frame(tail.ticket + frame * TKS, (unsigned)tail.expected))
        embed_scalars_frame(frame, pmax, pss, g.nblk_total, status, tail);
}

// =================================================================================================
// k_nvf_stats: per block sum (m_nvf W)^2          (p = 2*PAD+1)
// =================================================================================================
template <typename T, int PAD, bool VEC>
__device__ __forceinline__ void nvf_stats_march(const T* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                                const Geom& g, const WaveJob& j, float* lds, float& ss)
{
    constexpr int NR = 2 * PAD + 1;
    XMarch<T, 1, PAD, NR, VEC, PFX> xm;
    PMarch<float, VEC, PFW> wm_;
    const int nout = j.re - j.rs, n = nout + 2 * PAD;
    xm.start(xf, pitch, g, j, lds, j.rs - PAD, n);
    wm_.start(W, g.cols, g.cols, j, j.rs, nout);
    const int c0 = j.c0s + 4 * j.lane;
    march<2 * PAD>(n, [&](int i, auto qc, auto emit) {
        constexpr int Q = decltype(qc)::value;
        xm.template step<Q>(i);
        if (decltype(emit)::value) {
            constexpr int SLOT = (Q + 2 * UNROLL - 2 * PAD) % PFW;
            const float4 w = wm_.template take<SLOT>();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (VEC ? 4 * j.lane >= j.dup : c0 + k < g.cols) {
                    const float t = nvf_value<PAD, 4, Q>(xm, k) * f4get(w, k);
                    ss = fmaf(t, t, ss);
                }
            }
            wm_.template refill<SLOT>(i - 2 * PAD);
        }
    });
}

template <typename T, int PAD, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_nvf_stats(const T* __restrict__ x, long long pitch, long long fstride,
                                                     const float* __restrict__ W, Geom g, double* pss, ScalarsTail tail)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    __shared__ double s_ss[WPB];
    const WaveJob j = make_job(g);
    const int frame = j.frame;
    float ss = 0.0f;
    if (j.valid) {
        const T* xf = x + (long long)frame * fstride;
        nvf_stats_march<T, PAD, VEC>(xf, pitch, W, g, j, s_row[j.wave], ss);
    }
    const double ssd = wave_sum((double)ss);
    if (g.quad) {
        if (!j.valid) return;
        if (j.lane == 0) st_agent(pss + (long long)frame * g.nrec + j.rec, ssd);
        stats_fold(frame, j, nullptr, pss, g.nrec, nullptr, tail);
        return;
    }
    if (j.lane == 0) s_ss[j.wave] = ssd;
    __syncthreads();
    if (threadIdx.x == 0) st_agent(pss + (long long)frame * g.nblk_total + g.pb0 + j.tile, ((s_ss[0] + s_ss[1]) + s_ss[2]) + s_ss[3]);
    if (last_block_of_frame(tail.ticket + frame * TKS, (unsigned)tail.expected))
        embed_scalars_frame(frame, nullptr, pss, g.nblk_total, nullptr, tail);
}

// =================================================================================================
// Gram hand-over (HandOver, wm_kernels.hpp): the lag products of y that stay inside this wave's tile, accumulated as
// k_gram's march accumulates them (f64 FMAs of exact products; window of rows q, q+1, q+2 x columns c0-2 .. c0+5 in rotating
// slots).  The partner rows behind the segment (q+1, q+2 of its last q rows) are computed here as well -- the march runs two
// rows further, without storing them -- so that no product is left open between vertically adjacent tiles; what a lane cannot
// see is y in other strips (lanes 0 / 63 get zeros for the neighbour they do not have): k_gram_ho's column seams (wm_k_gram.hip),
// for which the lanes at a strip's two ends also store their two outermost columns of every row to a compact array (read back
// from the plane, those 16 bytes per row and boundary would cost two 128-byte lines each).
// =================================================================================================
struct HoState {
    double w[3][8];
    double acc[13];
    bool cv[4];
};
// the products of q row `w0` with itself (dr = 0), with `w1` (dr = 1) and with `w2` (dr = 2)
__device__ __forceinline__ void ho_products(HoState& h, const double* w0, const double* w1, const double* w2)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double xq = h.cv[k] ? w0[2 + k] : 0.0;
        h.acc[0] = fma(xq, w0[2 + k], h.acc[0]);
        h.acc[1] = fma(xq, w0[3 + k], h.acc[1]);
        h.acc[2] = fma(xq, w0[4 + k], h.acc[2]);
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            h.acc[3 + b] = fma(xq, w1[k + b], h.acc[3 + b]);
            h.acc[8 + b] = fma(xq, w2[k + b], h.acc[8 + b]);
        }
    }
}
// row r of y enters slot S; q row r - 2 (slot S + 1) is complete when `qvalid` (a core row of this segment)
template <int S>
__device__ __forceinline__ void ho_row(HoState& h, const float4& y, bool qvalid)
{
    double* s2 = h.w[S];
    s2[0] = (double)dpp_from_prev(y.z, 0.0f); s2[1] = (double)dpp_from_prev(y.w, 0.0f);
    s2[2] = (double)y.x; s2[3] = (double)y.y; s2[4] = (double)y.z; s2[5] = (double)y.w;
    s2[6] = (double)dpp_from_next(y.x, 0.0f); s2[7] = (double)dpp_from_next(y.y, 0.0f);
    if (qvalid) ho_products(h, h.w[(S + 1) % 3], h.w[(S + 2) % 3], s2);
}

// =================================================================================================
// k_embed: y = clamp(base + a * m * W, 0, 255) with the mask recomputed on the fly
//   MASK 0 (ME): m = |e| / max|e|;  MASK 1 (NVF): m = nvf(x)
// =================================================================================================
template <typename TX, typename TB, int NCH, int MASK, int PAD, bool VEC, bool BX, bool EDGE, bool HO = false>
__device__ __forceinline__ void embed_march(const TX* __restrict__ xf, long long pitch, const float* __restrict__ W,
                                            const TB* __restrict__ bptr, TB* __restrict__ optr, const PlaneDesc& base,
                                            const PlaneDesc& out, const Geom& g, const WaveJob& j, float* lds, float* obuf,
                                            const float (&c)[8], float a, float maxe, bool pass = false, double* horec = nullptr,
                                            float* hoseam = nullptr)
{
    static_assert(!HO || (VEC && NCH == 1 && sizeof(TB) == 4), "hand-over: grey f32 planes on the aligned path");
    constexpr int NR = MASK == 0 ? 3 : 2 * PAD + 1;
    constexpr int HR = MASK == 0 ? 1 : PAD;  // halo rows above/below = halo columns left/right
    constexpr int RG = VEC && NR == 3 ? WM_RING3 : UNROLL;
    XMarch<TX, 1, HR, NR, VEC, PFX, EDGE, false, RG> xm;
    constexpr int PW = HO ? WM_HO_PFW : PFW;  // rows of W / base in flight per wave
    PMarch<float, VEC, PW> wm_;
    // m = |e| / max|e| (Watermark.cpp:213-214): one reciprocal per wave, then div_by() per pixel (same quotient)
    const float inv_maxe = 1.0f / maxe;
    // BX: the base IS the grey input plane (video frames, grey images): its pixels are already in the stencil window,
    // so the base stream -- a third of this kernel's loads -- is not issued at all
    PMarch<TB, VEC, PW> bm[BX ? 1 : NCH];
    // (hand-over: up to two rows of y behind the segment are computed, not stored -- the partner rows of its last q rows)
    const int nout = j.re - j.rs, nrow = nout + (HO ? min(2, g.rows - j.re) : 0), n = nrow + 2 * HR;
    const int c0 = j.c0s + 4 * j.lane;
    xm.start(xf, pitch, g, j, lds, j.rs - HR, n);
    wm_.start(W, g.cols, g.cols, j, j.rs, nrow);
    if (!BX) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) bm[ch].start(bptr + (long long)ch * base.cstride, base.pitch, g.cols, j, j.rs, nrow);
    }
    HoState ho;
    if constexpr (HO) {
        static_assert(!HO || (HR == 1 && RG % 3 == 0), "hand-over: 3x3 windows (one x row ahead of the output row)");
#pragma unroll
        for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 8; ++b_) ho.w[a_][b_] = 0.0;
#pragma unroll
        for (int l = 0; l < 13; ++l) ho.acc[l] = 0.0;
        // q pixels: the core columns 2 .. C-3 this lane owns (k_gram's column factor)
#pragma unroll
        for (int k = 0; k < 4; ++k) ho.cv[k] = !EDGE || (c0 + k >= 2 && c0 + k <= g.cols - 3 && 4 * j.lane >= j.dup);
    }
    // this lane's entry of the seam array, or null: lane 63 holds columns S-2, S-1 of the boundary behind its strip, the first
    // lane that owns pixels (lane 0, or dup / 4 in a shifted last strip) holds columns S, S+1 of the boundary in front of it
    float* seamp = nullptr;
    bool seam_right = false;  // this lane stores its LAST two columns (the boundary behind the strip), else its first two
    if constexpr (HO) {
        const long long per_frame = (long long)(g.nstrips_total - 1) * g.rows * 4;
        if (j.lane == WAVE - 1 && j.strip < g.nstrips_total - 1) { seamp = hoseam + (long long)j.frame * per_frame + (long long)j.strip * g.rows * 4; seam_right = true; }
        else if (4 * j.lane == j.dup && j.strip > 0) seamp = hoseam + (long long)j.frame * per_frame + (long long)(j.strip - 1) * g.rows * 4 + 2;
    }
    march_n<2 * HR, RG>(n, [&](int i, auto qc, auto emit) {
        constexpr int Q = decltype(qc)::value;
        xm.template step<Q>(i);
        if (decltype(emit)::value) {
            const int o = i - 2 * HR;
            constexpr int SLOT = (Q + 4 * UNROLL - 2 * HR) % PW;
            const float4 w = wm_.template take<SLOT>();
            float u[4];
            float pr[4] = {0.f, 0.f, 0.f, 0.f};
            if (MASK == 0) predict4<4>(xm.template row<Q>(0), xm.template row<Q>(1), xm.template row<Q>(2), c, pr);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float m;
                if (MASK == 0) {
                    const float* mid = xm.template row<Q>(1);
                    const float e = mid[4 + k] - pr[k];
                    m = div_by(fabsf(e), maxe, inv_maxe);
                } else {
                    m = nvf_value<PAD, 4, Q>(xm, k);
                }
                u[k] = m * f4get(w, k);  // Watermark.cpp:169
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                float4 b;
                if (BX) {
                    const float* ctr = xm.template row<Q>(HR);  // the output row itself
                    b = make_float4(ctr[4], ctr[5], ctr[6], ctr[7]);
                } else {
                    b = bm[ch].template take<SLOT>();
                }
                float4 y;
                y.x = fminf(fmaxf(fmaf(u[0], a, b.x), 0.0f), 255.0f);
                y.y = fminf(fmaxf(fmaf(u[1], a, b.y), 0.0f), 255.0f);
                y.z = fminf(fmaxf(fmaf(u[2], a, b.z), 0.0f), 255.0f);
                y.w = fminf(fmaxf(fmaf(u[3], a, b.w), 0.0f), 255.0f);
                if constexpr (HO) {
                    if (pass) y = b;  // unsolvable frame: out = base bit-exact (Watermark.cpp:164-165), and that is the plane the detector reads
                    const int rq = j.rs + o - 2;  // the q row that row o completes (core rows 1 .. R-3; rq < re by construction)
                    ho_row<Q % 3>(ho, y, o >= 2 && rq >= 1 && rq < g.rows - 2);
                    if (seamp && o < nout)
                        *reinterpret_cast<float2*>(seamp + (long long)(j.rs + o) * 4) = seam_right ? make_float2(y.z, y.w) : make_float2(y.x, y.y);
                }
                if constexpr (VEC) {
                    if ((!EDGE || 4 * j.lane >= j.dup) && (!HO || o < nout))  // duplicate lanes of a shifted last strip: the previous strip stores these pixels
                        store4<TB, true>(optr + (long long)ch * out.cstride, out.pitch, j.rs + o, c0, g.cols, y);
                } else {
                    store_row_generic<TB>(optr + (long long)ch * out.cstride, out.pitch, j.rs + o, j.c0s, j.lane, g.cols, y, obuf);
                }
                if (!BX) bm[ch].template refill<SLOT>(o);
            }
            wm_.template refill<SLOT>(o);
        }
    });
    if constexpr (HO) {
        int idx;
        const double t = wave_sum_multi<13>(ho.acc, j.lane, idx);
        if (idx < 13) horec[idx] = t;
    }
}

#ifndef WM_HO_BLOCKS
#define WM_HO_BLOCKS 1
#endif
template <typename TX, typename TB, int NCH, int MASK, int PAD, bool VEC, bool BX, bool HO = false>
__global__ __launch_bounds__(BLOCK, HO ? WM_HO_BLOCKS : 1) void k_embed(const TX* __restrict__ x, long long pitch, long long fstride,
                                                 const float* __restrict__ W, PlaneDesc base, PlaneDesc out, Geom g,
                                                 const float* __restrict__ coef, const int* __restrict__ status,
                                                 const EmbedScalars* __restrict__ scal, HandOver ho)
{
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    __shared__ __attribute__((aligned(16))) float s_out[VEC ? 1 : WPB][VEC ? 4 : STRIP];  // generic path: store re-layout rows
    const WaveJob j = make_job(g);
    const int frame = j.frame;
    if (!j.valid) return;
    const TB* bptr = static_cast<const TB*>(base.p) + (long long)frame * base.fstride;
    TB* optr = static_cast<TB*>(const_cast<void*>(out.p)) + (long long)frame * out.fstride;
    const int st = MASK == 0 ? status[frame] : 0;
    if (!HO && st != 0) {
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
    // hand-over: an unsolvable frame runs the march too (y = base, selected per row) -- its lag sums are the detector's
    const bool pass = HO && st != 0;
    double* horec = HO ? ho.rec + ((long long)frame * ho.stride + j.rec) * 13 : nullptr;
    float* hoseam = HO ? ho.seam : nullptr;
    // NVF windows (PAD > 1) keep the single instance: their halo fix-up is a small share of the step
    if (MASK != 0 || strip_on_edge<VEC>(g, j))
        embed_march<TX, TB, NCH, MASK, PAD, VEC, BX, true, HO>(xf, pitch, W, bptr, optr, base, out, g, j, s_row[j.wave], s_out[VEC ? 0 : j.wave], c, a, maxe, pass, horec, hoseam);
    else
        embed_march<TX, TB, NCH, MASK, PAD, VEC, BX, (MASK != 0), HO>(xf, pitch, W, bptr, optr, base, out, g, j, s_row[j.wave], s_out[VEC ? 0 : j.wave], c, a, maxe, pass, horec, hoseam);
}

// =================================================================================================
// k_mask: materialise the mask (and the error sequence) -- parity-test building block
// =================================================================================================
template <typename T, int MASK, int PAD, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_mask(const T* __restrict__ x, long long pitch, long long fstride, Geom g,
                                                const float* __restrict__ coef, const int* __restrict__ status,
                                                const EmbedScalars* __restrict__ scal, PlaneDesc mo, PlaneDesc eo)
{
    constexpr int NR = MASK == 0 ? 3 : 2 * PAD + 1;
    constexpr int HR = MASK == 0 ? 1 : PAD;
    __shared__ __attribute__((aligned(16))) float s_row[WPB][2 * RowBuf<1>::N];
    const WaveJob j = make_job(g);
    const int frame = j.frame;
    if (!j.valid) return;
    if (MASK == 0 && status[frame] != 0) return;
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float maxe = 1.0f;
    if (MASK == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = coef[frame * 8 + k];
        maxe = scal[frame].maxe;
    }
    const float inv_maxe = 1.0f / maxe;
    float* mptr = static_cast<float*>(const_cast<void*>(mo.p)) + (long long)frame * mo.fstride;
    float* eptr = eo.p ? static_cast<float*>(const_cast<void*>(eo.p)) + (long long)frame * eo.fstride : nullptr;
    const int nout = j.re - j.rs, n = nout + 2 * HR;
    const int c0 = j.c0s + 4 * j.lane;
    const T* xf = x + (long long)frame * fstride;
    {
        XMarch<T, 1, HR, NR, VEC, PFX> xm;
        xm.start(xf, pitch, g, j, s_row[j.wave], j.rs - HR, n);
        march<2 * HR>(n, [&](int i, auto qc, auto emit) {
            constexpr int Q = decltype(qc)::value;
            xm.template step<Q>(i);
            if (decltype(emit)::value) {
                float mv[4], ev[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (MASK == 0) {
                        const float* mid = xm.template row<Q>(1);
                        ev[k] = mid[4 + k] - predict<4>(xm.template row<Q>(0), mid, xm.template row<Q>(2), k, c);
                        mv[k] = div_by(fabsf(ev[k]), maxe, inv_maxe);  // as k_embed computes it
                    } else {
                        ev[k] = 0.0f;
                        mv[k] = nvf_value<PAD, 4, Q>(xm, k);
                    }
                }
                if (4 * j.lane < j.dup) return;  // duplicate lanes of a shifted last strip (j.dup = 0 otherwise)
                store4<float, false>(mptr, mo.pitch, j.rs + i - 2 * HR, c0, g.cols, make_float4(mv[0], mv[1], mv[2], mv[3]));
                if (MASK == 0 && eptr)
                    store4<float, false>(eptr, eo.pitch, j.rs + i - 2 * HR, c0, g.cols, make_float4(ev[0], ev[1], ev[2], ev[3]));
            }
        });
    }
}

// launchers
static ScalarsTail scalars_tail(const LaunchGeom& lg, unsigned* ticket, unsigned* ticket_strip, float* smax, double* sss, float sF,
                                double sqrt_n, EmbedScalars* scal, OpResult* res, RawSums* raw)
{
    // ticket: [frames] frame-level counters followed (at ticket_strip) by [frames][nstrips] strip-level counters
    return ScalarsTail{ticket, ticket_strip, lg.nblk, lg.nsegs, lg.nstrips, smax, sss, sF, sqrt_n, scal, res, raw};
}

void launch_me_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                     const float* coef, const int* status, float* pmax, double* pss, unsigned* ticket, unsigned* ticket_strip,
                     float* smax, double* sss, float sF, double sqrt_n, EmbedScalars* scal, OpResult* res, RawSums* raw)
{
    const int al = align_mode(lg, x.aligned && aligned_w);
    const ScalarsTail tail = scalars_tail(lg, ticket, ticket_strip, smax, sss, sF, sqrt_n, scal, res, raw);
    WM_DISPATCH_T(x.dtype, WM_LAUNCH_SWEEP_Q(s, lg, frames, al, (k_me_stats<T, true>), (k_me_stats<T, false>), (const T*)x.p, x.pitch,
                                           x.fstride, W, g, coef, status, pmax, pss, tail));
}

template <typename T>
static void launch_nvf_stats_t(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W,
                               int aligned_w, int pad, double* pss, const ScalarsTail& tail)
{
    const int al = align_mode(lg, x.aligned && aligned_w);
#define NVF_CASE(P)                                                                                                           \
    case P:                                                                                                                   \
        WM_LAUNCH_SWEEP_Q(s, lg, frames, al, (k_nvf_stats<T, P, true>), (k_nvf_stats<T, P, false>), (const T*)x.p, x.pitch, x.fstride, \
                        W, g, pss, tail);                                                                                     \
        break;
    switch (pad) { NVF_CASE(1) NVF_CASE(2) NVF_CASE(3) NVF_CASE(4) }
#undef NVF_CASE
}
void launch_nvf_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                      int pad, double* pss, unsigned* ticket, unsigned* ticket_strip, double* sss, float sF, double sqrt_n,
                      EmbedScalars* scal, OpResult* res, RawSums* raw)
{
    const ScalarsTail tail = scalars_tail(lg, ticket, ticket_strip, nullptr, sss, sF, sqrt_n, scal, res, raw);
    WM_DISPATCH_T(x.dtype, launch_nvf_stats_t<T>(s, lg, frames, x, W, aligned_w, pad, pss, tail));
}

template <typename TX, typename TB, int NCH>
static bool launch_embed_tt(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                            const float* W, int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef,
                            const int* status, const EmbedScalars* scal, const HandOver* ho)
{
    const int al = align_mode(lg, x.aligned && aligned_w && base.aligned && out.aligned);
    // the base is the grey input itself (same plane, same layout): k_embed then takes it from its stencil window
    const bool bx = NCH == 1 && std::is_same<TX, TB>::value && base.p == x.p && base.pitch == x.pitch && base.fstride == x.fstride;
    const HandOver none{nullptr, 0, nullptr};
    if constexpr (NCH == 1 && std::is_same<TX, float>::value && std::is_same<TB, float>::value) {
        // Gram hand-over: every strip on the aligned path (one launch), 3x3 windows, a core, segments of two rows or more; two
        // frames or more (measured at 4K: one frame -3 %, two +2 %, four +6..9 %, eight and more +9..11 % -- a one-frame launch of
        // k_gram_ho is as latency-bound as the k_gram it replaces)
        if (ho && ho->rec && frames >= 2 && al == 2 && lg.nfull > 0 && pad == 1 && lg.rps >= 2 && lg.rows >= 4 && lg.cols >= 5 && lg.row_lo == 0 && lg.row_hi == lg.rows) {
            const SweepPart pv_ = sweep_part(lg, frames, true, al, 1);
            const Geom g = pv_.g;
#define EMB_HO(MASK)                                                                                                            \
            do {                                                                                                                \
                if (bx) WM_KLAUNCH((k_embed<float, float, 1, MASK, 1, true, true, true>), pv_.grid, dim3(BLOCK), 0, s, (const float*)x.p, x.pitch, \
                                   x.fstride, W, base, out, g, coef, status, scal, *ho);                                         \
                else WM_KLAUNCH((k_embed<float, float, 1, MASK, 1, true, false, true>), pv_.grid, dim3(BLOCK), 0, s, (const float*)x.p, x.pitch,  \
                                x.fstride, W, base, out, g, coef, status, scal, *ho);                                            \
            } while (0)
            if (mask == 0) EMB_HO(0); else EMB_HO(1);
#undef EMB_HO
            return true;
        }
    }
#define EMB(MASK, P)                                                                                                            \
    do {                                                                                                                        \
        if (bx) WM_LAUNCH_SWEEP_Q(s, lg, frames, al, (k_embed<TX, TB, 1, MASK, P, true, true>), (k_embed<TX, TB, 1, MASK, P, false, true>),  \
                                (const TX*)x.p, x.pitch, x.fstride, W, base, out, g, coef, status, scal, none);                  \
        else WM_LAUNCH_SWEEP_Q(s, lg, frames, al, (k_embed<TX, TB, NCH, MASK, P, true, false>), (k_embed<TX, TB, NCH, MASK, P, false, false>), \
                             (const TX*)x.p, x.pitch, x.fstride, W, base, out, g, coef, status, scal, none);                     \
    } while (0)
    if (mask == 0) { EMB(0, 1); return false; }
    switch (pad) {
        case 1: EMB(1, 1); break;
        case 2: EMB(1, 2); break;
        case 3: EMB(1, 3); break;
        case 4: EMB(1, 4); break;
    }
#undef EMB
    return false;
}
template <typename TX, typename TB>
static bool launch_embed_t(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                           const float* W, int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef,
                           const int* status, const EmbedScalars* scal, const HandOver* ho)
{
    if (base.channels == 3) return launch_embed_tt<TX, TB, 3>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal, ho);
    return launch_embed_tt<TX, TB, 1>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal, ho);
}
bool launch_embed(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                  int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef, const int* status,
                  const EmbedScalars* scal, const HandOver* ho)
{
    if (x.dtype == 0 && base.dtype == 0) return launch_embed_t<float, float>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal, ho);
    if (x.dtype == 1 && base.dtype == 1) return launch_embed_t<uint8_t, uint8_t>(s, lg, frames, mask, pad, x, W, aligned_w, base, out, coef, status, scal, ho);
    // mixed f32/u8 planes are rejected by the API layer (the reference converts whole frames, main.cpp:355-357)
    return false;
}

template <typename T>
static void launch_mask_t(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x,
                          const float* coef, const int* status, const EmbedScalars* scal, const PlaneDesc& mo,
                          const PlaneDesc& eo)
{
    // exercises the same two input paths as the production kernels (DPP for aligned full strips, LDS otherwise)
    const int al = align_mode(lg, x.aligned != 0);
#define MSK(MASK, P)                                                                                                      \
    WM_LAUNCH_SWEEP(s, lg, frames, al, (k_mask<T, MASK, P, true>), (k_mask<T, MASK, P, false>), (const T*)x.p, x.pitch, x.fstride, g, \
                    coef, status, scal, mo, eo)
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

}  // namespace wmk
