// wm_march.hpp -- the strip-march skeleton shared by the kernel translation units
#pragma once
#include "wm_kernels.hpp"
#include "wm_device.hpp"
#include <type_traits>

namespace wmk {

// -------------------------------------------------------------------------------------------------
// The march.  A wave streams `n` rows; row i is consumed at step i.  Steps are executed as
//   prologue : NPRO steps that only fill the window (compile-time count),
//   steady   : groups of UNROLL steps, straight-line, every load unconditional (rows clamped),
//   epilogue : < UNROLL guarded steps.
// body(i, q_constant, emit_constant): q = step index mod UNROLL as a compile-time constant.
// -------------------------------------------------------------------------------------------------
template <int V>
using IC = std::integral_constant<int, V>;

// UNR = steps per straight-line group (the ring length of the row slots); the default is UNROLL.
template <int NPRO, int UNR, typename F, int... K>
__device__ __forceinline__ void march_prologue(int n, F& body, std::integer_sequence<int, K...>)
{
    ((K < n ? (void)body(K, IC<K % UNR>{}, std::false_type{}) : (void)0), ...);
}
template <int NPRO, int UNR, typename F, int... K>
__device__ __forceinline__ void march_group(int i, F& body, std::integer_sequence<int, K...>)
{
    (body(i + K, IC<(NPRO + K) % UNR>{}, std::true_type{}), ...);
}
template <int NPRO, int UNR, typename F, int... K>
__device__ __forceinline__ void march_epilogue(int i, int n, F& body, std::integer_sequence<int, K...>)
{
    ((i + K < n ? (void)body(i + K, IC<(NPRO + K) % UNR>{}, std::true_type{}) : (void)0), ...);
}
template <int NPRO, int UNR, typename F>
__device__ __forceinline__ void march_n(int n, F&& body)
{
    static_assert(NPRO <= 8, "prologue too long");
    // prologue (window fill; may be cut short by a tiny segment)
    march_prologue<NPRO, UNR>(n, body, std::make_integer_sequence<int, NPRO>{});
    int i = NPRO;
    for (; i + UNR <= n; i += UNR) march_group<NPRO, UNR>(i, body, std::make_integer_sequence<int, UNR>{});
    march_epilogue<NPRO, UNR>(i, n, body, std::make_integer_sequence<int, UNR - 1>{});
}
template <int NPRO, typename F>
__device__ __forceinline__ void march(int n, F&& body) { march_n<NPRO, UNROLL>(n, body); }

// Rolling window over the x row stream: NR rows of (4 + 8*HC) columns per lane, HN halo columns valid.
//
// Register lifetimes are arranged so that the 6-step group needs no register copies (a copy of a freshly
// loaded register at the loop back-edge costs an s_waitcnt vmcnt(0), i.e. a full memory latency per group):
//  * NR == 3: prefetch depth 3; loaded rows and window rows share one ring of 6 slots (row i lives in slot
//    i % 6): loaded at step i-3, window row during steps i..i+2, dead afterwards, reloaded at step i+3.
//  * NR == 1: the row is consumed (converted) at once, so the slot is simply reloaded for row i + PF.
//  * other NR (NVF p > 3): rows are kept in order and shifted.
// A slot is always CONSUMED BEFORE its new load is issued, so the loop-carried value and the new load can
// share registers.
//  * RING (NR == 3 only) = length of that ring = steps per group (march_n<.., RING>): RING - 3 rows are in flight ahead of
//    the consumer.  The default (UNROLL = 6) keeps 3; k_detect's aligned path runs a ring of 9 (6 rows in flight).  Odd
//    rings are for the aligned path only: the generic path alternates two LDS row buffers by step parity.
template <typename T, int HC, int HN, int NR, bool VEC, int PFREQ, bool EDGE = true, bool XH = false, int RING = UNROLL>
struct XMarch {
    static constexpr int WN = 4 + 8 * HC;
    static constexpr bool ROT = NR == 3;
    static constexpr int UNR = ROT ? RING : UNROLL;     // steps per group of the march this window lives in
    static constexpr int PF = ROT ? RING - 3 : PFREQ;   // rows in flight ahead of the consumer
    static constexpr int NSLOT = ROT ? RING : PF;       // load slots
    static constexpr int NWIN = ROT ? RING : NR;        // window row slots
    static_assert(UNR % NSLOT == 0, "slot ring must divide the group length");
    static_assert(RING == UNROLL || (ROT && VEC), "a ring of its own: 3-row windows on the aligned path");
    static_assert(VEC || UNR % 2 == 0, "the generic path alternates two LDS row buffers");
    XStream<T, HC, HN, VEC, EDGE, XH> xs;
    typename XStream<T, HC, HN, VEC, EDGE, XH>::Raw pre[NSLOT];
    float win[NWIN][WN];
    float* buf;  // this wave's LDS row buffers (generic path): 2 x RowBuf<HC>::N floats
    int s0, last;

    __device__ __forceinline__ void start(const T* base, long long pitch, const Geom& g, const WaveJob& j, float* lds,
                                          int first_row, int count)
    {
        xs.init(base, pitch, g.rows, g.cols, j);
        buf = lds; s0 = first_row; last = first_row + count - 1;
#pragma unroll
        for (int a = 0; a < NWIN; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) win[a][b] = 0.0f;
#pragma unroll
        for (int q = 0; q < PF; ++q) pre[q] = xs.issue(min(s0 + q, last));
    }
    // consume stream row i (Q = i % UNROLL), then prefetch row i + PF (clamped to the segment: the tail
    // re-reads its last row from L1 instead of branching around the load)
    template <int Q>
    __device__ __forceinline__ void step(int i)
    {
        if (!ROT && NR > 1) {
#pragma unroll
            for (int a = 0; a + 1 < NR; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) win[a][b] = win[a + 1][b];
        }
        xs.consume(pre[Q % NSLOT], buf + (Q & 1) * RowBuf<HC>::N, win[ROT ? Q : NR - 1]);
        // fence the issue on both sides: everything that still reads the slot's old registers stays above it
        // (so the new load can reuse them and the loop-carried value needs no copy), and the load itself stays here
        __builtin_amdgcn_sched_barrier(0);
        pre[(Q + PF) % NSLOT] = xs.issue(min(s0 + i + PF, last));
        // pin the prefetch here: left alone, the machine scheduler sinks the loads towards their use in the next
        // group (shorter live ranges) and the wave then waits a full memory latency per row
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // window row a (0 = oldest .. NR-1 = newest) after step<Q>
    template <int Q>
    __device__ __forceinline__ const float* row(int a) const { return win[ROT ? (Q + UNR - (NR - 1) + a) % UNR : a]; }
};

// PF-deep prefetch ring for a pointwise operand: take<SLOT>() reads the row, refill<SLOT>(o) -- called after
// the last use of the taken value -- reloads the slot with row o + PF (clamped)
template <typename T, bool VEC, int PF>
struct PMarch {
    PStream<T, VEC> ps;
    typename Elem<T>::vec4 pre[PF];
    int r0, last;
    __device__ __forceinline__ void start(const T* base, long long pitch, int cols, const WaveJob& j, int first_row, int count)
    {
        ps.init(base, pitch, cols, j);
        r0 = first_row; last = first_row + count - 1;
#pragma unroll
        for (int q = 0; q < PF; ++q) pre[q] = ps.issue(min(r0 + q, last));
    }
    template <int SLOT>
    __device__ __forceinline__ float4 take() const { return Elem<T>::cvt4_pinned(pre[SLOT]); }
    template <int SLOT>
    __device__ __forceinline__ void refill(int o)
    {
        __builtin_amdgcn_sched_barrier(0);
        pre[SLOT] = ps.issue(min(r0 + o + PF, last));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch where it is written (see XMarch::step)
    }
};

// wave-uniform: does this wave's strip touch the image's left or right border?  (the generic path always may)
template <bool VEC>
__device__ __forceinline__ bool strip_on_edge(const Geom& g, const WaveJob& j) { return !VEC || j.c0s == 0 || j.c0s + STRIP >= g.cols; }

__device__ __forceinline__ float f4get(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }

#ifndef WM_PFX
#define WM_PFX 6
#endif
#ifndef WM_PFW
#define WM_PFW 3   // W rows are L2 hits for all frames of a launch but the first (frame-fastest / frame-quad order): 3 in flight
                   // suffice, and 12 VGPRs less bring k_me_stats / k_nvf_stats (98 -> 86) and k_embed<NVF> (100 -> 88) to five
                   // waves per SIMD (must divide UNROLL)
#endif
static_assert(UNROLL % WM_PFW == 0, "the W prefetch ring must divide the march group");
constexpr int PFX = WM_PFX;  // rows of x prefetched per wave (kernels with a 3-row window use a fixed depth of 3)
constexpr int PFW = WM_PFW;  // rows of W / base prefetched per wave

// =================================================================================================
// NVF value of pixel k from a window of 2*PAD+1 rows (nvf.hpp:37-50): row-major taps,
// sum += v; sumSq = fma(v, v, sumSq); mean = sum / p^2; var = sumSq / p^2 - mean*mean; var / (1 + var)
// =================================================================================================
template <int PAD>
__device__ __forceinline__ float nvf_from_sums(float sum, float sumsq)
{
    // the three divisions of nvf.hpp:47-50 as correctly rounded quotients without the full division sequence:
    // a constant divisor (div_by with its reciprocal) and a divisor 1 + var >= ~1 (div_inrange)
    constexpr float psq = (float)((2 * PAD + 1) * (2 * PAD + 1));
    constexpr float rpsq = 1.0f / psq;
    const float mean = div_by(sum, psq, rpsq);
    const float var = div_by(sumsq, psq, rpsq) - (mean * mean);
    return nvf_quot(var, 1.0f + var);
}

template <int PAD, int O, int Q, typename XM>
__device__ __forceinline__ float nvf_value(const XM& xm, int k)
{
    // the first tap starts the two chains: 0 + v = v and fma(v, v, 0) = v * v exactly (pixels are never -0), so the chains are
    // the oracle's with one addition less per pixel
    float sum = 0.0f, sumsq = 0.0f;
#pragma unroll
    for (int a = 0; a < 2 * PAD + 1; ++a) {
        const float* rowp = xm.template row<Q>(a);
#pragma unroll
        for (int b = -PAD; b <= PAD; ++b) {
            const float v = rowp[O + k + b];
            if (a == 0 && b == -PAD) { sum = v; sumsq = v * v; }
            else { sum += v; sumsq = fmaf(v, v, sumsq); }
        }
    }
    return nvf_from_sums<PAD>(sum, sumsq);
}

// the same over an explicit 3x3 window (rows top to bottom, columns left to right)
__device__ __forceinline__ float nvf_3x3(const float* up, const float* mid, const float* dn)
{
    float sum = 0.0f, sumsq = 0.0f;
    const float* rows[3] = {up, mid, dn};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float v = rows[a][b];
            if (a == 0 && b == 0) { sum = v; sumsq = v * v; }  // (0 + v, fma(v, v, 0): see nvf_value)
            else { sum += v; sumsq = fmaf(v, v, sumsq); }
        }
    return nvf_from_sums<1>(sum, sumsq);
}


// A sweep is launched as up to two kernels: the aligned-path instantiation over the strips that lie fully inside the
// image (when every plane allows vector access), and the generic instantiation over the remaining strips.
// `aligned`: 0 = no plane access by vectors, 1 = vectors on the full strips, 2 = as 1 and the image's last, partial
// strip may be moved left to end at the last column (align_mode below): then every strip runs the aligned path and the
// generic launch disappears (1920 columns = 7 full strips + 1 shifted strip whose first 128 columns are duplicates).
struct SweepPart { bool run; Geom g; dim3 grid; };
static inline SweepPart sweep_part(const LaunchGeom& lg, int frames, bool vec_part, int aligned, int quad = 0)
{
    int nvec = aligned ? lg.nfull : 0;
    // the aligned path loads up to 4 halo columns right of its strip with one vector load: when fewer than 4 (but
    // more than 0) columns remain right of the last full strip, that strip goes to the generic path instead
    const int rem = lg.cols - lg.nfull * STRIP;
    const bool shift = aligned == 2 && rem > 0 && lg.nfull > 0;
    if (shift) nvec = lg.nstrips;
    else if (nvec > 0 && rem > 0 && rem < 4) nvec -= 1;
    const int seggroups = (lg.nsegs + WPB - 1) / WPB;
    SweepPart sp;
    Geom& g = sp.g;
    g.rows = lg.rows; g.cols = lg.cols; g.row_lo = lg.row_lo; g.row_hi = lg.row_hi;
    g.nsegs = lg.nsegs; g.rps = lg.rps; g.nblk_total = lg.nblk;
    if (vec_part) { g.strip0 = 0; g.nstrips = nvec; g.pb0 = 0; }
    else { g.strip0 = nvec; g.nstrips = lg.nstrips - nvec; g.pb0 = nvec * seggroups; }
    g.shift_last = shift && vec_part ? 1 : 0;
    g.frames = frames; g.frame_fastest = 1;
    g.nstrips_total = lg.nstrips; g.nrec = lg.nstrips * lg.nsegs;
    g.sstride = 0; g.lead = 0; g.own_cols = 0; g.c0s_fixed = -1; g.own_c0 = 0;
    g.quad = quad && frames >= 4 ? 1 : 0;
    // quad: one (strip, segment) per block, its 4 waves are 4 consecutive frames; else 4 segments of one frame per block
    g.ntiles = g.quad ? g.nstrips * lg.nsegs : g.nstrips * seggroups;
    sp.run = g.nstrips > 0;
    sp.grid = dim3((unsigned)(g.quad ? g.ntiles * ((frames + 3) / 4) : g.ntiles * frames), 1, 1);
    return sp;
}
// Overlapped strips (Geom::sstride / lead; k_detect's aligned 3x3 path): strips 248 columns apart, each loading 256 from 4
// columns before its own (the image's first strip from column 0).  All strips run the aligned instantiation: lanes beyond the
// image's last column re-read its last pixels and own nothing, so there is no shifted strip and no generic remainder.
constexpr int OV_STRIDE = STRIP - 8, OV_LEAD = 4;
static inline int overlap_strips(int cols) { return (cols + OV_STRIDE - 1) / OV_STRIDE; }
static inline LaunchGeom overlap_geom(const LaunchGeom& lg)
{
    LaunchGeom l2 = lg;
    l2.nstrips = overlap_strips(lg.cols);
    l2.nfull = l2.nstrips;
    l2.nblk = l2.nstrips * ((lg.nsegs + WPB - 1) / WPB);
    return l2;
}
static inline SweepPart sweep_part_overlap(const LaunchGeom& l2, int frames, int quad)
{
    SweepPart sp = sweep_part(l2, frames, true, 1, quad);
    sp.g.sstride = OV_STRIDE; sp.g.lead = OV_LEAD; sp.g.shift_last = 0;
    return sp;
}
// Widths that are not multiples of 4 (k_detect's 3x3 path on planes that allow vector access): the overlapped strips own the
// columns below B = cols - cols % 4 - 4 -- every 4-pixel group they load, the provider lane's [B, B + 4) included, lies inside
// the row -- and ONE generic strip of 256 columns ending at the last column owns the columns >= B (it brings the replicate border
// at the image's right edge with it).  Records: the overlapped strips 0 .. n - 1, the generic strip n.
static inline int split_own_cols(int cols) { return cols - cols % 4 - 4; }
static inline bool split_applies(int cols) { return cols % 4 != 0 && cols >= STRIP + 8; }
static inline LaunchGeom split_geom(const LaunchGeom& lg)
{
    LaunchGeom l2 = lg;
    l2.nstrips = overlap_strips(split_own_cols(lg.cols)) + 1;
    l2.nfull = l2.nstrips;
    l2.nblk = l2.nstrips * ((lg.nsegs + WPB - 1) / WPB);
    return l2;
}
static inline SweepPart sweep_part_split_overlap(const LaunchGeom& l2, int frames, int quad)
{
    LaunchGeom lv = l2;
    lv.nstrips = l2.nstrips - 1; lv.nfull = lv.nstrips;
    SweepPart sp = sweep_part(lv, frames, true, 1, quad);
    sp.g.sstride = OV_STRIDE; sp.g.lead = OV_LEAD; sp.g.shift_last = 0;
    sp.g.own_cols = split_own_cols(l2.cols);
    sp.g.nstrips_total = l2.nstrips; sp.g.nrec = l2.nstrips * l2.nsegs; sp.g.nblk_total = l2.nblk;
    return sp;
}
static inline SweepPart sweep_part_split_generic(const LaunchGeom& l2, int frames, int quad)
{
    LaunchGeom lv = l2;
    lv.nstrips = 1; lv.nfull = 0;
    SweepPart sp = sweep_part(lv, frames, false, 0, quad);
    Geom& g = sp.g;
    g.strip0 = l2.nstrips - 1;   // (its record / ticket index; the column comes from c0s_fixed)
    g.c0s_fixed = l2.cols - STRIP; g.own_c0 = split_own_cols(l2.cols);
    g.pb0 = (l2.nstrips - 1) * ((l2.nsegs + WPB - 1) / WPB);
    g.nstrips_total = l2.nstrips; g.nrec = l2.nstrips * l2.nsegs; g.nblk_total = l2.nblk;
    return sp;
}

// aligned: every plane of the sweep allows 4-pixel vector access at multiples of 4 columns (PlaneDesc::aligned);
// the shifted strip starts at column cols - 256, which must be a vector boundary of every plane as well:
// a multiple of 4 columns for f32 planes (16 B), of 16 columns when a u8 plane takes part (its vectors are 4 B, but the
// u8 Gram path and stores assume dword alignment of c0s only -- 4 columns -- so 4 suffices there too)
static inline int align_mode(const LaunchGeom& lg, bool aligned)
{
    if (!aligned) return 0;
    return lg.cols % 4 == 0 ? 2 : 1;
}
// launches KERNEL<..., true> and KERNEL<..., false> over their strips
#define WM_LAUNCH_SWEEP(stream, lg, frames, aligned, KVEC, KGEN, ...)                                   \
    do {                                                                                                \
        const SweepPart pv_ = sweep_part(lg, frames, true, aligned);                                    \
        if (pv_.run) { const Geom g = pv_.g; WM_KLAUNCH(KVEC, pv_.grid, dim3(BLOCK), 0, stream, __VA_ARGS__); } \
        const SweepPart pg_ = sweep_part(lg, frames, false, aligned);                                   \
        if (pg_.run) { const Geom g = pg_.g; WM_KLAUNCH(KGEN, pg_.grid, dim3(BLOCK), 0, stream, __VA_ARGS__); } \
    } while (0)

// the same with the 4-frames-per-block mapping where the batch allows it (sweeps that read W)
#define WM_LAUNCH_SWEEP_Q(stream, lg, frames, aligned, KVEC, KGEN, ...)                                 \
    do {                                                                                                \
        const SweepPart pv_ = sweep_part(lg, frames, true, aligned, 1);                                 \
        if (pv_.run) { const Geom g = pv_.g; WM_KLAUNCH(KVEC, pv_.grid, dim3(BLOCK), 0, stream, __VA_ARGS__); } \
        const SweepPart pg_ = sweep_part(lg, frames, false, aligned, 1);                                \
        if (pg_.run) { const Geom g = pg_.g; WM_KLAUNCH(KGEN, pg_.grid, dim3(BLOCK), 0, stream, __VA_ARGS__); } \
    } while (0)

#define WM_DISPATCH_T(dtype, ...)                   \
    do {                                            \
        if ((dtype) == 0) { using T = float; __VA_ARGS__; } \
        else { using T = uint8_t; __VA_ARGS__; }    \
    } while (0)

}  // namespace wmk
