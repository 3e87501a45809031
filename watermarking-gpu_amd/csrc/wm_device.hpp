// wm_device.hpp -- device-side building blocks shared by all hot-path kernels (gfx950 / CDNA4).
//
// Execution shape ("strip march"): the image is cut into column strips of 256 pixels
// (64 lanes x 4 consecutive pixels = one 1 KiB global_load_dwordx4 per row) and each strip into
// row segments.  ONE WAVEFRONT owns one (strip, segment) and marches down its rows:
//   HBM row (coalesced, prefetched PF rows ahead in registers)
//     -> neighbour columns: aligned strips exchange them between lanes with DPP wave shifts (no LDS);
//        ragged/unaligned strips re-lay the row through a per-wave LDS row buffer
//     -> a rolling register window of the last rows feeds the stencil.
// No workgroup barrier exists on the data path: the four waves of a block are independent and
// only meet once, at the end, to fold their partial sums.  LDS use per wave is a few KiB and
// independent of the image size.
//
// Code-generation rules this file follows (learned from the ISA, see DESIGN.md "kernel anatomy"):
//  * everything that is uniform per wave (segment bounds, row pointers, loop trip counts) is kept
//    in SGPRs: the wave index goes through readfirstlane, so loops are scalar branches and row
//    addresses are SGPR base + a per-lane 32-bit offset computed once;
//  * no global load sits under a divergent branch (halo loads are issued by every lane at a clamped
//    address), so the compiler can count vmcnt and leave the prefetched rows in flight;
//  * the march runs in straight-line groups of 6 rows with UNCONDITIONAL (row-clamped) prefetch: the
//    compiler can then count vmcnt exactly; one conditional load in the group makes it wait vmcnt(0);
//  * inside a group the prefetch slot (i mod PF), LDS buffer (i mod 2) and window row (i mod 3) are
//    compile-time constants, so the rolling window costs no register moves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace wmk {

constexpr int WAVE = 64;
constexpr int STRIP = 256;  // columns per strip
constexpr int WPB = 4;      // waves per block
constexpr int BLOCK = WAVE * WPB;
constexpr int UNROLL = 6;   // march steps per unrolled group: lcm(2 LDS buffers, 3 window rows); prefetch depth PF divides it

template <int HC>
struct RowBuf {
    static constexpr int N = STRIP + 8 * HC;  // floats; HC halo chunks (4 cols each) on both sides
};

// Orders LDS traffic between the lanes of ONE wave.  The DS unit executes a wave's instructions
// in issue order, so cross-lane visibility inside a wave needs no s_barrier; the fence only stops
// the compiler from moving LDS accesses across this point.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// A register copy the compiler cannot see through.  Prefetched rows are loop-carried values; consuming them
// through an opaque copy ends the loaded register's life at a point WE choose (the step that consumes the
// row, when the data is rows old).  Without it the window rows alias the loaded registers, the allocator
// cannot give the next load the same registers, and it resolves the loop-carried value with copies at the
// loop back-edge -- each guarded by an s_waitcnt that waits for the newest loads (a full memory latency).
__device__ __forceinline__ float opaque(float v)
{
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t opaque(uint32_t v)
{
    uint32_t r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float4 opaque(const float4& v) { return make_float4(opaque(v.x), opaque(v.y), opaque(v.z), opaque(v.w)); }
// The same fence without the copy, for a prefetched value that is used up inside the step that takes it (pointwise
// operands): the value stays in the registers the load wrote, the asm only pins the point of use.
__device__ __forceinline__ float pinned(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t pinned(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float4 pinned(const float4& v) { return make_float4(pinned(v.x), pinned(v.y), pinned(v.z), pinned(v.w)); }

// ---- "the last block finishes" --------------------------------------------------------------------
// A sweep's per-block partial sums are folded by the block of that frame that arrives last, inside the same kernel
// (no separate fold kernel, no launch gap).  Blocks of one frame run on different XCDs, whose L2s are not coherent
// with each other inside a kernel: partials are therefore written and read with agent-scope (sc1) accesses, which go
// to / come from the memory side whatever an XCD's L2 holds, and a per-frame ticket counter (agent-scope atomic) names
// the last block.  Summation order is fixed by the partial index, not by arrival order: results stay deterministic.
template <typename V>
__device__ __forceinline__ void st_agent(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename V>
__device__ __forceinline__ V ld_agent(const V* p) { return __hip_atomic_load(const_cast<V*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// NOTE on the memory model: the partial records are published with relaxed agent-scope (sc1, write-through) stores, an
// explicit s_waitcnt vmcnt(0) and a relaxed agent-scope ticket, and read back with sc1 loads -- the hand-off form the gfx950
// guide measures as valid (MI355X_MICROARCH.md, "Valid forms", row 1) and 2-3 us cheaper per hop than release / acquire
// fences (which write back / invalidate whole caches and would drop the W tiles the block order keeps in L2).  It relies on
// gfx950's sc1 semantics, hence the guard below; tests/test_gpu_soak.py stresses it under even and uneven load.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the in-kernel hand-offs of this engine are written for gfx950 (sc1 write-through stores / L1-bypassing loads)"
#endif

// Called by every thread of every block of the frame after its partial stores.  True in all threads of the last block.
__device__ __forceinline__ bool last_block_of_frame(unsigned* ticket, unsigned expected)
{
    __shared__ unsigned s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial stores are acknowledged by the memory side
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = prev + 1u == expected;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next op on this slot
        s_last = last ? 1u : 0u;
    }
    __syncthreads();
    return s_last != 0u;
}

// Sweeps whose partial records are per WAVE (k_*_stats, k_detect; the waves of a block may belong to different frames,
// Geom::quad) fold in two levels without a block barrier: the waves of a (frame, strip) take tickets of that strip, the
// one that draws the last folds the strip's records into a strip record and takes a ticket of the frame; the wave that
// draws the frame's last ticket folds the strip records.  Two levels because atomics on ONE address serialise (a 4K
// frame has 720 wave records: 16 frames x 720 tickets on 16 addresses cost ~60 us per launch) and because it spreads
// the fold over as many waves as there are strips.
// take_ticket: called by all lanes of a wave after lane 0's stores; true in all lanes if this wave drew the last ticket.
__device__ __forceinline__ bool take_ticket(unsigned* ticket, unsigned expected, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores are acknowledged by the memory side
    int last = 0;
    if (lane == 0) {
        const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = prev + 1u == expected ? 1 : 0;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next op on this slot
    }
    return __builtin_amdgcn_readfirstlane(last) != 0;
}

// a / b given rb = 1.0f / b (correctly rounded): product plus one residual correction (Markstein).  With an exact
// residual (fmaf) and a correctly rounded reciprocal the result is the correctly rounded quotient, i.e. the value the
// oracle's IEEE division gives, for 3 VALU operations instead of the 11 of the full division sequence.
// (0 / 0 stays NaN: rb = inf, 0 * inf.)
__device__ __forceinline__ float div_by(float a, float b, float rb)
{
    const float q0 = a * rb;
    const float r = fmaf(-q0, b, a);
    return fmaf(r, rb, q0);
}

// n / d for a finite |d| well inside the f32 range: the arithmetic of the IEEE division sequence the compiler emits
// (reciprocal, one refinement, quotient, two residual corrections) without its range scaling and special-case
// fix-up -- 8 operations instead of 11, the same correctly rounded quotient
__device__ __forceinline__ float div_inrange(float n, float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, r, 1.0f);
    r = fmaf(e, r, r);
    float q = n * r;
    const float e2 = fmaf(-d, q, n);
    q = fmaf(e2, r, q);
    const float e3 = fmaf(-d, q, n);
    return fmaf(e3, r, q);
}

// var / (1 + var) of the NVF mask (nvf.hpp:50) with d = 1 + var already formed: hardware reciprocal, product, ONE residual
// correction -- 4 operations instead of the 11 of the IEEE division sequence (8 in div_inrange).  No error bound of v_rcp_f32
// proves that in general; it does not have to: the divisor is a function of the dividend, so the inputs are a ONE-parameter
// family, and the mask can only produce var = sumSq/p^2 - mean^2 of pixels in [0, 255], i.e. values in [-0.5, 2^17) with
// room to spare -- 2.2e9 floats.  Every one of them is checked on the device against the compiler's correctly rounded
// division (wm_selftest_nvf_quotient, wm.h; tests/test_gpu_nvf_quotient.py, every round, on the hardware the kernels run on):
// 0 differing results for this sequence on gfx950.  VARIANT selects what the self-test compares: 0 = div_inrange (8
// operations), 1 = with a refined reciprocal (6), 2 = this one (4), 3 = n * rcp(d) alone (2; NOT exact -- kept so that the
// self-test is seen to fail when it should).
template <int VARIANT>
__device__ __forceinline__ float nvf_quot_variant(float n, float d)
{
    if constexpr (VARIANT == 0) return div_inrange(n, d);
    float r = __builtin_amdgcn_rcpf(d);
    if constexpr (VARIANT == 1) {
        const float e = fmaf(-d, r, 1.0f);
        r = fmaf(e, r, r);
    }
    const float q = n * r;
    if constexpr (VARIANT == 3) return q;
    const float e2 = fmaf(-d, q, n);
    return fmaf(e2, r, q);
}
#ifndef WM_NVF_QUOT
#define WM_NVF_QUOT 2
#endif
__device__ __forceinline__ float nvf_quot(float n, float d) { return nvf_quot_variant<WM_NVF_QUOT>(n, d); }

// ---- element type adapters -------------------------------------------------------------------
template <typename T>
struct Elem;
template <>
struct Elem<float> {
    using vec4 = float4;  // 4 consecutive pixels as loaded
    using one = float;
    static __device__ __forceinline__ float4 cvt4(const vec4& v) { return opaque(v); }
    static __device__ __forceinline__ float4 cvt4_pinned(const vec4& v) { return pinned(v); }
    static __device__ __forceinline__ float cvt1(one v) { return opaque(v); }
    static __device__ __forceinline__ vec4 pack(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
};
template <>
struct Elem<uint8_t> {
    using vec4 = uint32_t;
    using one = uint8_t;
    static __device__ __forceinline__ float4 cvt4(const vec4& v0)
    {
        const uint32_t v = opaque(v0);
        return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
    }
    static __device__ __forceinline__ float4 cvt4_pinned(const vec4& v0)
    {
        const uint32_t v = pinned(v0);
        return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
    }
    static __device__ __forceinline__ float cvt1(one v) { return (float)opaque((uint32_t)v); }
    static __device__ __forceinline__ vec4 pack(uint8_t a, uint8_t b, uint8_t c, uint8_t d)
    {
        return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
    }
};

// f32 -> output element.  u8 follows ArrayFire's .as(u8): truncation of an in-range value
// (the value is already clamped to [0,255], main.cpp:356,380).
template <typename T>
__device__ __forceinline__ T out_cvt(float v);
template <>
__device__ __forceinline__ float out_cvt<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint8_t out_cvt<uint8_t>(float v) { return (uint8_t)v; }

// ---- geometry of one wave's job --------------------------------------------------------------
struct Geom {
    int rows, cols;
    int row_lo, row_hi;   // rows owned by this launch (LaunchGeom): segments tile [row_lo, row_hi), loads may reach outside
    int strip0, nstrips;  // this launch covers strips [strip0, strip0 + nstrips): aligned full strips and ragged /
                          // unaligned strips are launched as separate kernels (one code path and one register budget each)
    int nsegs, rps;       // rps = rows per segment
    int nblk_total;       // blocks per frame over all launches of a sweep (stride of the per-block partial arrays)
    int pb0;              // index of this launch's block 0 in those arrays
    int frames;           // frames in this launch
    int ntiles;           // march blocks per frame in this launch (grid = ntiles * frames [+ extra leading blocks])
    int shift_last;       // aligned path: the image's last strip is not full, so it is moved left to end at the last column
                          // (c0s = cols - 256); its leading columns duplicate the previous strip's and are masked out
    int quad;             // 1: a block is ONE (strip, segment) of 4 consecutive frames (wave w = frame 4q + w) instead of 4
                          //    vertically adjacent segments of one frame: the 4 waves read the same W rows at the same time, so a
                          //    W row is fetched once per CU.  Used by the sweeps that read W when a launch has 4 frames or more
    int nstrips_total;    // strips of the whole image (all launches of the sweep)
    int nrec;             // per-wave partial records per frame = nstrips_total * nsegs (k_*_stats, k_detect)
    int frame_fastest;    // block order: 1 = same tile of consecutive frames back to back (kernels that read W),
                          //              0 = all tiles of a frame, then the next frame (k_gram: nothing is shared between frames)
    // Overlapped strips (k_detect's aligned 3x3 path): strips are `sstride` columns apart (0 = STRIP) and every strip but the
    // first starts `lead` columns early, so that a wave's 64 lanes load 256 consecutive columns of which the first and the
    // last lane only PROVIDE neighbours (sstride = 248, lead = 4: lanes 1..62 own the strip's columns).  Everything a pixel
    // needs from beyond its strip then arrives by DPP from those two lanes: no halo loads, no halo arithmetic of its own
    int sstride, lead;
    // (overlapped strips) columns this launch OWNS: [0, own_cols), 0 = the whole width.  Widths that are not multiples of 4 run
    // the overlapped instance over the columns below B = cols - cols % 4 - 4 (every 4-pixel group it touches lies inside a row)
    // and ONE generic strip over the rest:
    int own_cols;
    // (generic path) c0s_fixed >= 0: the launch is a single strip that starts at this column (any alignment) and owns the
    // columns >= own_c0 only -- the pixels in front of them are the overlapped launch's
    int c0s_fixed, own_c0;
};

struct WaveJob {
    bool valid;  // wave-uniform (SGPR)
    int c0s;     // first column of the strip (SGPR)
    int rs, re;  // row segment [rs, re) (SGPR)
    int lane;    // VGPR
    int wave;    // SGPR
    int frame;   // frame of the batch this block works on (SGPR)
    int tile;    // block index inside the frame, 0 .. ntiles-1 (SGPR)
    bool full;   // strip lies fully inside the image (c0s + STRIP <= cols)
    int dup;     // leading columns of this strip that belong to the previous strip (shifted last strip), else 0 (SGPR)
    int rec;     // this wave's partial record: segment * nstrips_total + strip (SGPR)
    int strip;   // strip index in the whole image (SGPR)
    int lo, hi;  // first / last lane that OWNS its 4 columns (0 / 63 unless the strips overlap: Geom::sstride) (SGPR)
    int own_c0;  // (generic path) first column whose pixels this wave owns (Geom::own_c0; 0 otherwise) (SGPR)
};

// Block order.  Hardware deals consecutive block ids round-robin over the 8 XCDs (placement is a speed matter
// only).  xcd_remap makes the ids one XCD receives a contiguous range of a logical index; the logical index runs
// FRAME-FASTEST over (tile, frame) pairs.  Two effects: (1) an XCD works on a contiguous band of the image, so halo
// rows shared by vertically adjacent segments stay in its L2; (2) the blocks that run back to back on an XCD are the
// SAME tile of consecutive frames of the batch, so the tile of W -- identical for every frame -- is fetched once and
// then served from that XCD's L2 for the rest of the batch instead of crossing the fabric once per frame.
__device__ __forceinline__ int xcd_remap(int b, int nblk)
{
    const int per = nblk >> 3, rem = nblk & 7;
    const int x = b & 7, i = b >> 3;
    return x * per + (x < rem ? x : rem) + i;
}

// block_id: 0 .. ntiles*frames-1 (callers subtract any leading extra blocks first)
__device__ __forceinline__ WaveJob make_job(const Geom& g, int block_id)
{
    WaveJob j;
    j.lane = threadIdx.x & (WAVE - 1);
    j.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int strip, seg;
    if (g.quad) {
        const int nq = (g.frames + 3) >> 2;       // the last quad may be short: its surplus waves are idle
        const int pidx = xcd_remap(block_id, g.ntiles * nq);
        j.tile = pidx / nq;                       // consecutive blocks: the frame quads of one (strip, segment)
        j.frame = 4 * (pidx - j.tile * nq) + j.wave;
        strip = g.strip0 + j.tile % g.nstrips;
        seg = j.tile / g.nstrips;
    } else {
        const int pidx = xcd_remap(block_id, g.ntiles * g.frames);
        if (g.frame_fastest) {
            j.tile = pidx / g.frames;
            j.frame = pidx - j.tile * g.frames;
        } else {
            j.frame = pidx / g.ntiles;
            j.tile = pidx - j.frame * g.ntiles;
        }
        // a block = 4 vertically adjacent segments of one strip; consecutive tiles = adjacent strips
        strip = g.strip0 + j.tile % g.nstrips;
        seg = (j.tile / g.nstrips) * WPB + j.wave;
    }
    j.rec = seg * g.nstrips_total + strip;
    j.strip = strip;
    j.valid = seg < g.nsegs && j.frame < g.frames;
    j.c0s = strip * STRIP;
    j.dup = 0;
    j.lo = 0; j.hi = WAVE - 1;
    if (g.sstride) {
        // overlapped strips: strip s owns columns [s * sstride, (s + 1) * sstride) and loads from `lead` columns before them
        // (strip 0 from column 0: the image's left border is the replicate case of lane 0)
        j.lo = strip > 0 ? g.lead / 4 : 0;
        j.c0s = strip * g.sstride - 4 * j.lo;
        const int last = ((g.own_cols ? g.own_cols : g.cols) - j.c0s) / 4 - 1;  // lane that holds the last owned column (a multiple of 4 columns)
        j.hi = j.lo + g.sstride / 4 - 1 < last ? j.lo + g.sstride / 4 - 1 : last;
    }
    j.own_c0 = 0;
    if (g.c0s_fixed >= 0) { j.c0s = g.c0s_fixed; j.own_c0 = g.own_c0; }
    if (g.shift_last && j.c0s + STRIP > g.cols) {
        j.dup = j.c0s - (g.cols - STRIP);
        j.c0s = g.cols - STRIP;
    }
    j.rs = g.row_lo + seg * g.rps;
    j.re = j.rs + g.rps < g.row_hi ? j.rs + g.rps : g.row_hi;
    j.full = j.c0s + STRIP <= g.cols;
    return j;
}
__device__ __forceinline__ WaveJob make_job(const Geom& g) { return make_job(g, (int)blockIdx.x); }

// ---- cross-lane neighbour exchange without LDS (aligned path) ----------------------------------
// DPP wave shifts: lane i receives lane i-1's (resp. i+1's) value; the lane with no source keeps `edge`
// (the DPP "old" operand), which is where the strip's halo column enters.
__device__ __forceinline__ float dpp_from_prev(float own, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(own), 0x138 /*wave_shr:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_next(float own, float edge)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(own), 0x130 /*wave_shl:1*/, 0xF, 0xF, false));
}
// the same shifts where the lane WITHOUT a source does not care what it gets (provider lanes of overlapped strips away from the
// image borders): bound_ctrl zero-fills it, so no "old" value has to be copied into the destination first -- one v_mov less
// per exchange, on the unit that bounds the u8 sweeps (4 per row in k_detect)
__device__ __forceinline__ float dpp_from_prev_any(float own)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(own), 0x138 /*wave_shr:1*/, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_from_next_any(float own)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(own), 0x130 /*wave_shl:1*/, 0xF, 0xF, true));
}
// whole-wave rotations by one lane (every lane has a source: no "old" operand): rol1: lane i <- lane i+1 (lane 63 <- lane 0),
// ror1: lane i <- lane i-1 (lane 0 <- lane 63)
__device__ __forceinline__ float wave_rol1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134 /*wave_rol:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_ror1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C /*wave_ror:1*/, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---- one plane as a row stream ---------------------------------------------------------------
// Delivers, per row, this lane's window: columns c0 - HN .. c0 + 3 + HN (c0 = c0s + 4*lane) at
// win[O - HN .. O + 3 + HN], O = 4*HC.
//   VEC  : one 16 B (f32) / 4 B (u8) load per lane + one halo element per lane (lanes 0..8*HC-1 matter);
//          neighbours' columns arrive by DPP wave shifts, the strip's halo columns by readlane -- no LDS.
//          Needs aligned planes, a strip fully inside the image, HN <= 4 and HC == 1.
//   !VEC : four coalesced element loads per lane (columns c0s+l+64k), column index clamped to the image
//          (this IS the replicate border), re-laid out through the wave's LDS row buffer.
// All per-lane offsets are row-invariant and computed once; a row costs one SGPR row base.
// halo vector of the aligned path: HV = 1, 2 or 4 elements per lane
template <typename T, int HV> struct HaloVec;
template <> struct HaloVec<float, 1> { using type = float; };
template <> struct HaloVec<float, 2> { using type = float2; };
template <> struct HaloVec<float, 4> { using type = float4; };
template <> struct HaloVec<uint8_t, 1> { using type = uint8_t; };
template <> struct HaloVec<uint8_t, 2> { using type = uint16_t; };
template <> struct HaloVec<uint8_t, 4> { using type = uint32_t; };
__device__ __forceinline__ void halo_unpack(float v, float (&o)[4]) { o[0] = opaque(v); }
__device__ __forceinline__ void halo_unpack(float2 v, float (&o)[4]) { o[0] = opaque(v.x); o[1] = opaque(v.y); }
__device__ __forceinline__ void halo_unpack(float4 v, float (&o)[4]) { o[0] = opaque(v.x); o[1] = opaque(v.y); o[2] = opaque(v.z); o[3] = opaque(v.w); }
__device__ __forceinline__ void halo_unpack(uint8_t v, float (&o)[4]) { o[0] = (float)opaque((uint32_t)v); }
__device__ __forceinline__ void halo_unpack(uint16_t v0, float (&o)[4]) { const uint32_t v = opaque((uint32_t)v0); o[0] = (float)(v & 0xffu); o[1] = (float)(v >> 8); }
__device__ __forceinline__ void halo_unpack(uint32_t v0, float (&o)[4])
{
    const uint32_t v = opaque(v0);
    o[0] = (float)(v & 0xffu); o[1] = (float)((v >> 8) & 0xffu); o[2] = (float)((v >> 16) & 0xffu); o[3] = (float)(v >> 24);
}

// ---- row loads of the aligned path: buffer loads --------------------------------------------------------------------
// A row load's address is {plane base} + {row offset, wave-uniform} + {column offset, per lane and row-invariant}.  As a
// flat global load the compiler keeps the column offset zero-extended in a VGPR pair and forms the address with a 64-bit
// VECTOR add per load (v_lshl_add_u64: four per row in k_detect, on the unit that bounds the sweeps).  A buffer load takes
// exactly these three parts -- descriptor (SGPRs, built once per wave), soffset (an SGPR: the row) and voffset (32-bit VGPR) --
// so a row costs scalar instructions only.  Offsets are 32-bit: callers take this path only for planes below 4 GiB
// (PlaneDesc::aligned, wm_api.hip vec_ok).  The descriptor is raw (stride 0) with the range check opened to the offset range.
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc make_rsrc(const void* base)
{
    // (the pointer is wave-uniform by construction: kernel argument + a frame offset derived through readfirstlane)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFF, 0x00020000);
}
// AUX: the instruction's cache-policy bits (0: default; 17 = sc0 | sc1: past the caches that are not coherent across the chip)
template <typename V, int AUX = 0>
__device__ __forceinline__ V buf_load(BufRsrc rs, unsigned voff, unsigned soff)
{
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    typedef int v3i_t __attribute__((ext_vector_type(3)));
    static_assert(sizeof(V) == 16 || sizeof(V) == 12 || sizeof(V) == 8 || sizeof(V) == 4 || sizeof(V) == 2 || sizeof(V) == 1, "buffer load width");
    V out;
    if constexpr (sizeof(V) == 12) { const v3i_t v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, AUX); __builtin_memcpy(&out, &v, 12); }
    else if constexpr (sizeof(V) == 16) { const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX); __builtin_memcpy(&out, &v, 16); }
    else if constexpr (sizeof(V) == 8) { const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX); __builtin_memcpy(&out, &v, 8); }
    else if constexpr (sizeof(V) == 4) { const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX); __builtin_memcpy(&out, &v, 4); }
    else if constexpr (sizeof(V) == 2) { const unsigned short v = __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, AUX); __builtin_memcpy(&out, &v, 2); }
    else { const unsigned char v = __builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, AUX); __builtin_memcpy(&out, &v, 1); }
    return out;
}

// EDGE = false: the strip touches neither image border (callers check), so the halo needs no replicate fix-up
// XH = true (aligned path, HN == 1): overlapped strips (Geom::sstride): no halo is loaded -- lanes 0 and 63 are neighbour
// providers, every lane's halo column comes by DPP; the lane offsets are clamped into the row (lanes beyond the image's last
// column re-read its last 4 pixels; they own nothing)
template <typename T, int HC, int HN, bool VEC, bool EDGE = true, bool XH = false>
struct XStream {
    using E = Elem<T>;
    static constexpr int WN = 4 + 8 * HC;
    static constexpr int O = 4 * HC;
    static constexpr int HV = HN <= 1 ? 1 : (HN == 2 ? 2 : 4);  // halo elements each lane loads on the aligned path
    static_assert(!VEC || (HC == 1 && HN <= 4), "DPP path covers one neighbour chunk per side");
    static_assert(!XH || (VEC && HN >= 1 && HN <= 3), "overlapped strips: aligned path, the halo columns lie in the neighbouring lane's 4 pixels");
    using HaloT = typename std::conditional<VEC, typename HaloVec<T, HV>::type, typename E::one>::type;
    const T* base;
    long long pitch;
    BufRsrc rs;         // (aligned path) the frame plane as a buffer
    unsigned pitch_b;   // (aligned path) bytes between rows
    int rows;
    int lane;
    unsigned off[VEC ? 1 : 4];  // per-lane BYTE offsets inside a row
    unsigned off_h;
    bool edge_l, edge_r;  // (aligned path) the strip touches the image's left / right border: halo = replicate
    bool rsel;            // (overlapped strips) this lane's right neighbour is the replicate border

    struct Raw {
        typename E::vec4 v;
        HaloT h;
    };

    __device__ __forceinline__ void init(const T* b, long long p, int r, int cols, const WaveJob& j)
    {
        base = b; pitch = p; rows = r; lane = j.lane;
        if constexpr (VEC) { rs = make_rsrc(b); pitch_b = (unsigned)p * (unsigned)sizeof(T); }
        edge_l = EDGE && j.c0s == 0;
        edge_r = EDGE && j.c0s + STRIP >= cols;
        if constexpr (VEC) {
            // (lanes beyond the image re-read its last whole 4-pixel group: (cols - 4) & ~3, a vector boundary for u8 planes too
            // when the width is not a multiple of 4)
            off[0] = (unsigned)(XH ? min(j.c0s + 4 * j.lane, (cols - 4) & ~3) : j.c0s + 4 * j.lane) * (unsigned)sizeof(T);
            rsel = XH && EDGE && j.lane == j.hi && j.c0s + 4 * (j.hi + 1) >= cols;  // this lane holds the image's last column
            // lane 63 loads the HV columns right of the strip, every other lane the HV columns left of it (only lane 0
            // and lane 63 use them, as the DPP "edge" operands); at the image border the address is pulled inside
            // and the value replaced by the replicated border pixel in consume()
            const int hc = j.lane == WAVE - 1 ? (edge_r ? cols - HV : j.c0s + STRIP) : (edge_l ? 0 : j.c0s - HV);
            off_h = (unsigned)hc * (unsigned)sizeof(T);
        } else {
#pragma unroll
            for (int k = 0; k < (VEC ? 1 : 4); ++k) off[k] = (unsigned)min(j.c0s + j.lane + 64 * k, cols - 1) * (unsigned)sizeof(T);
            // every lane loads a halo element (lanes >= 8*HC repeat the last one): no divergent load.
            // lanes 0..4HC-1: columns c0s-4HC .. c0s-1; lanes 4HC..8HC-1: columns c0s+STRIP .. ; clamped = replicate
            const int hl = min(j.lane, 8 * HC - 1);
            off_h = (unsigned)(hl < 4 * HC ? max(j.c0s - 4 * HC + hl, 0) : min(j.c0s + STRIP + hl - 4 * HC, cols - 1)) * (unsigned)sizeof(T);
        }
    }

    __device__ __forceinline__ Raw issue(int r) const
    {
        Raw raw;
        if constexpr (VEC) {
            const unsigned soff = (unsigned)clampi(r, 0, rows - 1) * pitch_b;  // scalar
            raw.v = buf_load<typename E::vec4>(rs, off[0], soff);
            if constexpr (!XH) raw.h = buf_load<HaloT>(rs, off_h, soff);
            return raw;
        }
        const char* rowp = reinterpret_cast<const char*>(base + (long long)clampi(r, 0, rows - 1) * pitch);  // scalar
        if (VEC) raw.v = *reinterpret_cast<const typename E::vec4*>(rowp + off[0]);
        else raw.v = E::pack(*reinterpret_cast<const T*>(rowp + off[0]), *reinterpret_cast<const T*>(rowp + off[VEC ? 0 : 1]),
                             *reinterpret_cast<const T*>(rowp + off[VEC ? 0 : 2]), *reinterpret_cast<const T*>(rowp + off[VEC ? 0 : 3]));
        raw.h = *reinterpret_cast<const HaloT*>(rowp + off_h);
        return raw;
    }

    __device__ __forceinline__ void consume(const Raw& raw, float* __restrict__ buf, float* __restrict__ win) const
    {
        const float4 f = E::cvt4(raw.v);
        if constexpr (XH) {
            // lane 0 keeps its own first pixel (the replicate border of strip 0; a provider lane elsewhere: never used), lane
            // 63 its own last one; at the image's right border the lane that holds the last column takes its own pixel
            win[O + 0] = f.x; win[O + 1] = f.y; win[O + 2] = f.z; win[O + 3] = f.w;
            const float comp[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int d = 1; d <= HN; ++d) {   // column c0 - d: lane - 1's pixel 4 - d; column c0 + 3 + d: lane + 1's pixel d - 1
                if constexpr (EDGE) {
                    win[O - d] = dpp_from_prev(comp[4 - d], f.x);
                    const float nx = dpp_from_next(comp[d - 1], f.w);
                    win[O + 3 + d] = rsel ? f.w : nx;
                } else {
                    // (no image border in this strip: lanes 0 and 63 only provide, what they receive is never used)
                    win[O - d] = dpp_from_prev_any(comp[4 - d]);
                    win[O + 3 + d] = dpp_from_next_any(comp[d - 1]);
                }
            }
        } else if constexpr (VEC) {
            win[O + 0] = f.x; win[O + 1] = f.y; win[O + 2] = f.z; win[O + 3] = f.w;
            const float comp[4] = {f.x, f.y, f.z, f.w};
            float h[4];
            halo_unpack(raw.h, h);  // h[0..HV-1]: columns (c0s-HV .. c0s-1) in lanes != 63, (c0s+STRIP .. ) in lane 63
#pragma unroll
            for (int d = 1; d <= HN; ++d) {
                // left neighbour column c0-d: lane-1's component 4-d; lane 0 keeps the strip halo column c0s-d = h[HV-d]
                const float el = (EDGE && edge_l) ? f.x : h[HV - d];
                win[O - d] = dpp_from_prev(comp[4 - d], el);
                // right neighbour column c0+3+d: lane+1's component d-1; lane 63 keeps column c0s+STRIP+d-1 = h[d-1]
                const float er = (EDGE && edge_r) ? f.w : h[d - 1];
                win[O + 3 + d] = dpp_from_next(comp[d - 1], er);
            }
        } else {
            const float hv = E::cvt1(raw.h);
            float4* b4 = reinterpret_cast<float4*>(buf);
            buf[4 * HC + lane] = f.x;
            buf[4 * HC + lane + 64] = f.y;
            buf[4 * HC + lane + 128] = f.z;
            buf[4 * HC + lane + 192] = f.w;
            if (lane < 8 * HC) buf[lane < 4 * HC ? lane : STRIP + lane] = hv;
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 1 + 2 * HC; ++k) {
                const float4 c = b4[lane + k];
                win[4 * k + 0] = c.x; win[4 * k + 1] = c.y; win[4 * k + 2] = c.z; win[4 * k + 3] = c.w;
            }
        }
    }
};

// ---- pointwise operand (W, base): this lane's 4 consecutive pixels of a row -----------------
template <typename T, bool VEC>
struct PStream {
    using E = Elem<T>;
    const T* base;
    long long pitch;
    BufRsrc rs;         // (aligned path) the plane as a buffer, see XStream
    unsigned pitch_b;
    unsigned off[VEC ? 1 : 4];  // byte offsets (see XStream)

    __device__ __forceinline__ void init(const T* b, long long p, int cols, const WaveJob& j)
    {
        base = b; pitch = p;
        if constexpr (VEC) { rs = make_rsrc(b); pitch_b = (unsigned)p * (unsigned)sizeof(T); }
        const int c0 = j.c0s + 4 * j.lane;
        if (VEC) off[0] = (unsigned)min(c0, cols - 4) * (unsigned)sizeof(T);  // (clamped: lanes beyond the image in overlapped strips)
        else {
            // clamped: out-of-image lanes read a valid address and are masked later
#pragma unroll
            for (int k = 0; k < (VEC ? 1 : 4); ++k) off[k] = (unsigned)min(c0 + k, cols - 1) * (unsigned)sizeof(T);
        }
    }
    __device__ __forceinline__ typename E::vec4 issue(int r) const
    {
        if constexpr (VEC) return buf_load<typename E::vec4>(rs, off[0], (unsigned)r * pitch_b);
        const char* rowp = reinterpret_cast<const char*>(base + (long long)r * pitch);  // scalar
        if (VEC) return *reinterpret_cast<const typename E::vec4*>(rowp + off[0]);
        return E::pack(*reinterpret_cast<const T*>(rowp + off[0]), *reinterpret_cast<const T*>(rowp + off[VEC ? 0 : 1]),
                       *reinterpret_cast<const T*>(rowp + off[VEC ? 0 : 2]), *reinterpret_cast<const T*>(rowp + off[VEC ? 0 : 3]));
    }
};

template <typename T, bool VEC>
__device__ __forceinline__ void store4(T* base, long long pitch, int r, int c0, int cols, float4 y)
{
    T* rowp = base + (long long)r * pitch;
    if constexpr (VEC) {
        // non-temporal (aux 2 = nt): the output plane is written once and next read by another sweep long after it has left
        // L2; marking it first-to-evict leaves the cache to the W tiles and halo rows (+1 % at 4K with 3 slots).  A buffer
        // store like the row loads (make_rsrc): descriptor + scalar row offset + per-lane column offset, no vector address add
        const BufRsrc rs = make_rsrc(base);
        const unsigned voff = (unsigned)c0 * (unsigned)sizeof(T), soff = (unsigned)r * (unsigned)pitch * (unsigned)sizeof(T);
        if constexpr (sizeof(T) == 4) {
            typedef int v4i_t __attribute__((ext_vector_type(4)));
            v4i_t v; v.x = __float_as_int(y.x); v.y = __float_as_int(y.y); v.z = __float_as_int(y.z); v.w = __float_as_int(y.w);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 2);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(Elem<T>::pack(out_cvt<T>(y.x), out_cvt<T>(y.y), out_cvt<T>(y.z), out_cvt<T>(y.w)), rs, voff, soff, 2);
        }
    } else {
        if (c0 + 0 < cols) rowp[c0 + 0] = out_cvt<T>(y.x);
        if (c0 + 1 < cols) rowp[c0 + 1] = out_cvt<T>(y.y);
        if (c0 + 2 < cols) rowp[c0 + 2] = out_cvt<T>(y.z);
        if (c0 + 3 < cols) rowp[c0 + 3] = out_cvt<T>(y.w);
    }
}

// Generic-path store of a whole strip row: the lanes hold 4 consecutive pixels each, which as four element stores per
// lane would write 4-byte pieces 16 bytes apart.  Re-laid through a 256-float LDS row (this wave's own) every store
// instruction writes 64 consecutive elements instead.  `obuf`: STRIP floats, 16-byte aligned, private to the wave.
template <typename T>
__device__ __forceinline__ void store_row_generic(T* base, long long pitch, int r, int c0s, int lane, int cols, float4 y,
                                                  float* obuf)
{
    T* rowp = base + (long long)r * pitch;
    reinterpret_cast<float4*>(obuf)[lane] = y;
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = c0s + lane + 64 * k;
        const float v = obuf[lane + 64 * k];
        if (col < cols) rowp[col] = out_cvt<T>(v);
    }
    wave_lds_fence();  // the next row's write must not overtake these reads
}

// ---- wave reductions (64 lanes) in DPP: no LDS, fixed order => deterministic ---------------------
// row_shr:1,2,4,8 build each 16-lane row's total in its lane 15, row_bcast15 / row_bcast31 chain the
// rows; lane 63 ends with the wave total, which is then broadcast with readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov0(float v)  // lanes without a source (or in masked rows) read 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov0(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov0<0x111, 0xF>(v);  // row_shr:1
    v += dpp_mov0<0x112, 0xF>(v);  // row_shr:2
    v += dpp_mov0<0x114, 0xF>(v);  // row_shr:4
    v += dpp_mov0<0x118, 0xF>(v);  // row_shr:8
    v += dpp_mov0<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_mov0<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float wave_max(float v)  // for values >= 0 (0 is the identity here)
{
    v = fmaxf(v, dpp_mov0<0x111, 0xF>(v));
    v = fmaxf(v, dpp_mov0<0x112, 0xF>(v));
    v = fmaxf(v, dpp_mov0<0x114, 0xF>(v));
    v = fmaxf(v, dpp_mov0<0x118, 0xF>(v));
    v = fmaxf(v, dpp_mov0<0x142, 0xA>(v));
    v = fmaxf(v, dpp_mov0<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- N sums over the wave at once ------------------------------------------------------------------------------------
// wave_sum() costs ~20 vector instructions per value (6 dependent DPP steps); a wave that has N values to reduce (the 13
// lag sums of the Gram sweep) pays N times that.  Recursive halving instead: at every level two registers are merged into
// one -- lanes whose level bit is 0 keep summing the first register's values, lanes whose bit is 1 the second's -- so the
// register count halves while the distance halves: 32 and 16 with gfx950's v_permlane32_swap / v_permlane16_swap (one swap
// per 32-bit half and ONE add for two values), 8 ... 1 with a select and a DPP shuffle.  13 values: 62 instructions
// instead of 260.  The total of value i ends in lane bitreverse6(i); every lane returns the total of value `idx` =
// bitreverse6(lane), which exists iff idx < N.  The order of the additions is fixed: deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_shuffle_d(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int D>
__device__ __forceinline__ double xor_shuffle_d(double v)  // lane i <- lane i ^ D, D in {8, 4, 2, 1}
{
    if constexpr (D == 8) return dpp_shuffle_d<0x128>(v);                         // row_ror:8
    else if constexpr (D == 4) return dpp_shuffle_d<0x1B>(dpp_shuffle_d<0x141>(v));  // row_half_mirror, then quad_perm [3,2,1,0]
    else if constexpr (D == 2) return dpp_shuffle_d<0x4E>(v);                     // quad_perm [2,3,0,1]
    else return dpp_shuffle_d<0xB1>(v);                                           // quad_perm [1,0,3,2]
}
// a + (a's other half), b + (b's other half): lanes 0..31 get a's sums, lanes 32..63 b's
__device__ __forceinline__ double merge_swap32(double a, double b)
{
    const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)(ba & 0xffffffffLL), (unsigned)(bb & 0xffffffffLL), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    const double x = __longlong_as_double(((long long)hi[0] << 32) | (unsigned)lo[0]);
    const double y = __longlong_as_double(((long long)hi[1] << 32) | (unsigned)lo[1]);
    return x + y;
}
// the same between the 16-lane rows of each half: even rows get a's sums, odd rows b's
__device__ __forceinline__ double merge_swap16(double a, double b)
{
    const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)(ba & 0xffffffffLL), (unsigned)(bb & 0xffffffffLL), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    const double x = __longlong_as_double(((long long)hi[0] << 32) | (unsigned)lo[0]);
    const double y = __longlong_as_double(((long long)hi[1] << 32) | (unsigned)lo[1]);
    return x + y;
}
template <int D>
__device__ __forceinline__ double merge_xor(double a, double b, int lane)
{
    const bool up = (lane & D) != 0;
    const double keep = up ? b : a;
    const double send = up ? a : b;
    return keep + xor_shuffle_d<D>(send);
}
template <int N, int LEVEL>
__device__ __forceinline__ double wave_sum_multi_level(const double (&v)[N], int lane)
{
    constexpr int M = (N + 1) / 2;
    double r[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const double a = v[2 * i];
        const double b = 2 * i + 1 < N ? v[2 * i + 1] : v[2 * i];  // a leftover pairs with itself: both sides then hold its sum
        if constexpr (LEVEL == 1) r[i] = merge_swap32(a, b);
        else if constexpr (LEVEL == 2) r[i] = merge_swap16(a, b);
        else if constexpr (LEVEL == 3) r[i] = merge_xor<8>(a, b, lane);
        else if constexpr (LEVEL == 4) r[i] = merge_xor<4>(a, b, lane);
        else if constexpr (LEVEL == 5) r[i] = merge_xor<2>(a, b, lane);
        else r[i] = merge_xor<1>(a, b, lane);
    }
    if constexpr (LEVEL == 6) return r[0];
    else return wave_sum_multi_level<M, LEVEL + 1>(r, lane);
}
template <int N>
__device__ __forceinline__ double wave_sum_multi(const double (&v)[N], int lane, int& idx)
{
    static_assert(N >= 1 && N <= 64, "one total per lane at most");
    idx = (int)(__builtin_bitreverse32((unsigned)lane) >> 26);
    return wave_sum_multi_level<N, 1>(v, lane);
}

// the 8 neighbour taps of pixel k (k = 0..3) in the reference's order (me_p3.hpp:46-54,
// scaled_neighbors_p3.hpp:35-42).  Rows are window arrays whose element [O + k] is the pixel's own
// column (O = columns of left halo carried in the array).
template <int O>
__device__ __forceinline__ float predict(const float* __restrict__ up, const float* __restrict__ mid,
                                         const float* __restrict__ dn, int k, const float (&c)[8])
{
    float dot = 0.0f;
    dot = fmaf(c[0], up[O + k - 1], dot);
    dot = fmaf(c[1], up[O + k], dot);
    dot = fmaf(c[2], up[O + k + 1], dot);
    dot = fmaf(c[3], mid[O + k - 1], dot);
    dot = fmaf(c[4], mid[O + k + 1], dot);
    dot = fmaf(c[5], dn[O + k - 1], dot);
    dot = fmaf(c[6], dn[O + k], dot);
    dot = fmaf(c[7], dn[O + k + 1], dot);
    return dot;
}

// the predictions of a lane's 4 pixels, tap-major: the four fmaf chains advance together, so consecutive instructions are
// independent (pixel-major order leaves every instruction waiting for its predecessor; the machine scheduler does not
// interleave the chains by itself).  Each chain is the reference's order of taps: results are bit-identical to predict().
template <int O>
__device__ __forceinline__ void predict4(const float* __restrict__ up, const float* __restrict__ mid,
                                         const float* __restrict__ dn, const float (&c)[8], float (&d)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[0], up[O + k - 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[1], up[O + k], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[2], up[O + k + 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[3], mid[O + k - 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[4], mid[O + k + 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[5], dn[O + k - 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[6], dn[O + k], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(c[7], dn[O + k + 1], d[k]);
}

// centre - (the 8-tap prediction) with the subtraction folded into the chain: the chain starts at the centre pixel and runs over
// the NEGATED coefficients, 8 operations per pixel instead of 9.  NOT the oracle's rounding sequence (that forms the prediction
// from 0 and subtracts once: predict4 above, which every kernel whose e / mask / y is compared element by element keeps); the
// two differ by a few ulp of the pixel value.  For the detector only, whose one output is a correlation with a stated
// tolerance of 1e-5: the sums over 8 M pixels move by ~1e-9 relative (tests/test_gpu_parity.py holds the score against the
// oracle's at every shape).  nc[k] = -c[k].
// (one pixel; the same rounding sequence as residual4, so that a pixel evaluated by two waves -- a strip's halo column -- gets
// the same value from both)
template <int O>
__device__ __forceinline__ float residual1(const float* __restrict__ up, const float* __restrict__ mid,
                                           const float* __restrict__ dn, int k, const float (&nc)[8])
{
    float d = fmaf(nc[0], up[O + k - 1], mid[O + k]);
    d = fmaf(nc[1], up[O + k], d);
    d = fmaf(nc[2], up[O + k + 1], d);
    d = fmaf(nc[3], mid[O + k - 1], d);
    d = fmaf(nc[4], mid[O + k + 1], d);
    d = fmaf(nc[5], dn[O + k - 1], d);
    d = fmaf(nc[6], dn[O + k], d);
    d = fmaf(nc[7], dn[O + k + 1], d);
    return d;
}
template <int O>
__device__ __forceinline__ void residual4(const float* __restrict__ up, const float* __restrict__ mid,
                                          const float* __restrict__ dn, const float (&nc)[8], float (&d)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[0], up[O + k - 1], mid[O + k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[1], up[O + k], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[2], up[O + k + 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[3], mid[O + k - 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[4], mid[O + k + 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[5], dn[O + k - 1], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[6], dn[O + k], d[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = fmaf(nc[7], dn[O + k + 1], d[k]);
}

}  // namespace wmk
