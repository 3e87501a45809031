// wm_k_fused.hip -- ONE launch per operation for one image per call (the reference's own call pattern:
// Watermark::makeWatermark / detectWatermark on a single image, Watermark.cpp:156-172,234-250; main.cpp:165-220).
//
// The streaming kernels (wm_k_gram / wm_k_embed / wm_k_detect) are built for batches: with one frame per launch each of
// the 3 (embed) / 2 (detect) dependent sweeps re-reads the frame from memory and pays its own ramp and fold tail
// (17-24 us per sweep at 3840x2160 whatever its size).  Here the whole chip works on ONE frame inside ONE launch:
//
//   * the frame is cut into tiles of 256 columns x up to 128 rows, one tile per workgroup of 16 wavefronts, one
//     workgroup per CU; a tile plus 2 halo rows / columns is loaded from HBM ONCE and stays in the CU's LDS (137 KB of
//     160) for every later phase; W is read once into registers (each thread owns RPW rows x 4 pixels);
//   * the global reductions between the phases (Gram sums -> coefficients, {max|e|, sum (|e| W)^2} -> strength) are folds
//     inside the launch: every workgroup stores its partial record write-through (sc1), takes a ticket (agent-scope
//     atomic add); the workgroup that arrives last folds the records in index order (deterministic), solves / finalises
//     and publishes the result behind a flag; the others poll the flag with one lane (MI355X_MICROARCH.md "Valid
//     forms", row 1: sc1 payload, drained, one atomic per workgroup, sc1 poll, workgroup barrier, sc1 loads);
//   * operands of the NEXT phase (W, the base planes) are requested before the wait, so their HBM latency hides
//     behind the fold.
//
// HBM traffic per call: embed {x, W -> y} = 12 N bytes, detect {y, W} = 8 N bytes (f32) -- SURVEY.md 8d's compulsory floor --
// against 24 N / 12 N for the sweep chain.  Arithmetic, operation order and the exact f64 Gram are those of the streaming
// kernels (same helpers), so results agree with them to the reduction-order rounding of the f64 / f32 partial sums.
//
// Every workgroup must be resident for the in-launch hand-offs to complete: the grid never exceeds the CU count, the
// host serialises fused launches per device, and every spin is bounded (s_memrealtime); a timed-out launch leaves the
// result record untouched and the host re-runs the call on the streaming kernels.
#include "wm_march.hpp"
#include "wm_gram_common.hpp"
#include <atomic>
#include <cstdlib>

namespace wmk {

// Shapes of a fused workgroup (RPW = rows per wavefront), chosen by the tile height (fused_geometry):
//   RPW = 4 or 8: 16 wavefronts (1024 threads, 128 VGPRs each), tiles of up to 64 / 128 rows;
//   RPW = 16    : 8 wavefronts (512 threads, 256 VGPRs each), every wavefront with all its 18 rows in flight at once.
//                 Measured at 3840x2160: 1.7 us SLOWER per Gram phase than RPW = 8 (the first row arrives later behind
//                 the larger burst and 2 wavefronts per SIMD hide the f64 chain worse); not instantiated.
constexpr int fw_of(int rpw) { return rpw >= 16 ? 8 : 16; }  // wavefronts per fused workgroup
constexpr int FW_MAX = 16;
constexpr int FNT = 57;             // partial-record terms of the Gram phase: 13 lag sums + 44 border terms
// Bounds of the spins, in ticks of the 100 MHz s_memrealtime clock.  A hand-off normally completes in a few microseconds; it
// cannot complete when a workgroup is not resident (another process's kernels on the device, a CU mask).  Before the first
// store of the output a time-out is harmless -- the launch ends without a result and the host takes the batched sweeps -- so
// those limits are short (a missing workgroup costs milliseconds, not the 50 ms it did): the ONE workgroup that folds the
// statistics gives up first, the workgroups that wait for a published value later (a value published late would otherwise
// reach only some of them).  After the output stores only the completion flags are left to wait for: that limit stays long,
// and its time-out is reported as FUSED_INCOMPLETE (the output plane may be partly written: the host must not re-run an
// in-place call).
constexpr unsigned long long SPIN_FOLD_TICKS = 100000ull;    // 1 ms: the folding workgroup's polls of the statistics records
constexpr unsigned long long SPIN_WAIT_TICKS = 400000ull;    // 4 ms: waits for published granules; the detector's final fold
constexpr unsigned long long SPIN_DONE_TICKS = 5000000ull;   // 50 ms: the end-of-embed flags (after the output stores)

template <int RPW>
struct FTile {
    static constexpr int FW = fw_of(RPW);
    static constexpr int TH = FW * RPW;         // tile rows at most
    static constexpr int NROW = TH + 4;         // LDS rows: tile-local rows -2 .. TH+1
    static constexpr int TILE_F = NROW * STRIP; // floats
    static constexpr int HALO_F = NROW * 4;     // per LDS row: columns c0s-2, c0s-1, c0s+256, c0s+257
    // f64 scratch (reductions, fold, solve), in doubles, after the tile and halo floats
    static constexpr int RED_D = FW_MAX * 13;   // per-wave lag sums
    static constexpr int BOR_D = FW_MAX * NGRAM;  // border terms of the workgroup's chunks (2 for all but tiny images)
    static constexpr int FOLD_D = 969;          // fold scratch; before the hand-off: 40 doubles per wave for the border chunks' terms
    static constexpr int MISC_D = 13 + NGRAM + 8 * 9 + 4 * FW_MAX + 48;  // the last 48 doubles: small unsigned words (flags, granule values)
    static constexpr size_t BYTES = (size_t)(TILE_F + HALO_F) * 4 + (size_t)(RED_D + BOR_D + FOLD_D + MISC_D) * 8;
};

struct LdsView {
    float* tile;    // [NROW][256]
    float* halo;    // [NROW][4]
    double* red;    // [FW][13]
    double* bor;    // [FW][44]
    double* fold;   // [FGROUPS][57]
    double* s_m;    // [13]
    double* s_tot;  // [44]
    double* A;      // [8][9]
    double* wred;   // [4][FW_MAX] per-wave scalars of the later phases
    unsigned* flags;  // [..] small words: last / ok
};

template <int RPW>
__device__ __forceinline__ LdsView carve(char* smem)
{
    using FT = FTile<RPW>;
    LdsView v;
    v.tile = reinterpret_cast<float*>(smem);
    v.halo = v.tile + FT::TILE_F;
    double* d = reinterpret_cast<double*>(v.halo + FT::HALO_F);
    v.red = d; d += FT::RED_D;
    v.bor = d; d += FT::BOR_D;
    v.fold = d; d += FT::FOLD_D;
    v.s_m = d; d += 13;
    v.s_tot = d; d += NGRAM;
    v.A = d; d += 72;
    v.wred = d; d += 4 * FW_MAX;
    v.flags = reinterpret_cast<unsigned*>(d);
    return v;
}

// ---- per-lane row loader of the aligned layout: 4 own pixels per row.  The two columns left and right of the strip (used
// by lanes 0 / 63 only) come from ONE gather per wavefront: lane 2k + side requests the pair of the wave's k-th row.  (A
// pair per row and lane, as the streaming kernels load it, doubles the wave's memory instructions; what a CU keeps in
// flight is bounded in requests, and the first rows of a workgroup's younger waves left ~1.2 us later for it.)
template <typename T>
struct FLoad {
    using E = Elem<T>;
    using H2 = typename HaloVec<T, 2>::type;
    const T* base;
    long long pitch;
    int rows, cols, c0s;
    int off;
    bool edge_l, edge_r;
    struct Raw { typename E::vec4 v; };
    __device__ __forceinline__ void init(const T* b, long long p, int r, int cl, int c0s_, int lane)
    {
        base = b; pitch = p; rows = r; cols = cl; c0s = c0s_;
        edge_l = c0s == 0;
        // (widths that are not multiples of 4 can leave ONE column right of a full strip: its halo pair is that column and its
        // replicate, like the pair of a strip that ends at the image's last column)
        edge_r = c0s + STRIP >= cols - 1;
        off = c0s + 4 * lane;
    }
    __device__ __forceinline__ Raw issue(int r) const
    {
        Raw raw;
        const T* rowp = base + (long long)clampi(r, 0, rows - 1) * pitch;
        raw.v = *reinterpret_cast<const typename E::vec4*>(rowp + off);
        return raw;
    }
    // The same loads past the caches that are not coherent across the chip (sc0 sc1, as the hand-off records are read): for
    // lines this launch may hold in their state of BEFORE a workgroup elsewhere rewrote them -- k_fused_pair's detector half
    // reads rows of y at addresses where, in an in-place call, its embed half read x
    // (one buffer load per vector, cache policy sc0 sc1 = aux bits 0 and 4 on gfx940+; the fused path takes planes below 4 GiB)
    template <typename V>
    __device__ __forceinline__ V load_coherent(long long elem_off) const
    {
        return buf_load<V, 17>(make_rsrc(base), (unsigned)(elem_off * (long long)sizeof(T)), 0u);
    }
    __device__ __forceinline__ Raw issue_coherent(int r) const
    {
        Raw raw;
        raw.v = load_coherent<typename E::vec4>((long long)clampi(r, 0, rows - 1) * pitch + off);
        return raw;
    }
    // halo pairs of rows r_first .. r_first + n - 1 (n <= 32): lane 2k: columns c0s-2, c0s-1 of row k; lane 2k+1: c0s+256, c0s+257
    __device__ __forceinline__ H2 issue_halos(int r_first, int n, int lane) const
    {
        const int k = min(lane >> 1, n - 1);
        // (inside the image; replicate below.  c0s == 1 -- a 257-column image's shifted strip, whose lane 0 owns nothing -- stays inside too)
        const int col = (lane & 1) ? (edge_r ? cols - 2 : c0s + STRIP) : (edge_l ? 0 : max(c0s - 2, 0));
        return *reinterpret_cast<const H2*>(base + (long long)clampi(r_first + k, 0, rows - 1) * pitch + col);
    }
    __device__ __forceinline__ H2 issue_halos_coherent(int r_first, int n, int lane) const
    {
        const int k = min(lane >> 1, n - 1);
        const int col = (lane & 1) ? (edge_r ? cols - 2 : c0s + STRIP) : (edge_l ? 0 : max(c0s - 2, 0));
        return load_coherent<H2>((long long)clampi(r_first + k, 0, rows - 1) * pitch + col);
    }
};
__device__ __forceinline__ float4 fcvt4(const float4& v) { return v; }
__device__ __forceinline__ float4 fcvt4(uint32_t v)
{
    return make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
}
__device__ __forceinline__ float2 fcvt2(const float2& v) { return v; }
__device__ __forceinline__ float2 fcvt2(uint16_t v) { return make_float2((float)(v & 0xffu), (float)(v >> 8)); }

// the 8-wide window of a row: columns c0-2 .. c0+5; the strip's halo pairs of the row come from the LDS side array
// (replicate already applied there)
__device__ __forceinline__ void row8(const LdsView& L, int tl, int lane, const float4& f, float (&v)[8])
{
    const float4 hh = *reinterpret_cast<const float4*>(L.halo + tl * 4);  // (one address for the wave: a broadcast read)
    const float hx = lane == WAVE - 1 ? hh.z : hh.x, hy = lane == WAVE - 1 ? hh.w : hh.y;
    v[2] = f.x; v[3] = f.y; v[4] = f.z; v[5] = f.w;
    v[0] = dpp_from_prev(f.z, hx);
    v[1] = dpp_from_prev(f.w, hy);
    v[6] = dpp_from_next(f.x, hx);
    v[7] = dpp_from_next(f.y, hy);
}
// the gathered halo pairs (FLoad::issue_halos) into the side array: pair k belongs to LDS row tl_first + k
template <typename LD>
__device__ __forceinline__ void lds_put_halos(const LdsView& L, const LD& ld, int tl_first, int n, int lane, float2 h)
{
    const bool right = lane & 1;
    if (right ? ld.edge_r : ld.edge_l) { if (right) h.x = h.y; else h.y = h.x; }  // replicate: both columns := the border pixel
    if (lane < 2 * n) *reinterpret_cast<float2*>(L.halo + (tl_first + (lane >> 1)) * 4 + (right ? 2 : 0)) = h;
}

// store the own 4 pixels of one row into the LDS tile
__device__ __forceinline__ void lds_put_row(const LdsView& L, int tl, int lane, const float (&v)[8])
{
    reinterpret_cast<float4*>(L.tile + tl * STRIP)[lane] = make_float4(v[2], v[3], v[4], v[5]);
}

// a row of the LDS tile as a 6-wide window: columns c0-1 .. c0+4
__device__ __forceinline__ void lds_row6(const LdsView& L, int tl, int lane, float (&o)[6])
{
    const float4 f = reinterpret_cast<const float4*>(L.tile + tl * STRIP)[lane];
    const float hv = L.halo[tl * 4 + (lane == WAVE - 1 ? 2 : 1)];
    o[1] = f.x; o[2] = f.y; o[3] = f.z; o[4] = f.w;
    o[0] = dpp_from_prev(f.w, hv);
    o[5] = dpp_from_next(f.x, hv);
}
// the 3 columns around this lane's halo column (lane 63: c0s+255 .. c0s+257, others: c0s-2 .. c0s) of an LDS row
__device__ __forceinline__ void lds_halo3(const LdsView& L, int tl, int lane, float (&o)[3])
{
    const bool last = lane == WAVE - 1;
    const float2 p = *reinterpret_cast<const float2*>(L.halo + tl * 4 + (last ? 2 : 0));
    const float s = L.tile[tl * STRIP + (last ? STRIP - 1 : 0)];
    o[0] = last ? s : p.x;
    o[1] = last ? p.x : p.y;
    o[2] = last ? p.y : s;
}

struct FusedArgs {
    int rows, cols;
    int nstrips, nbands, th;  // grid = nstrips * nbands workgroups; workgroup b: strip b % nstrips, row band b / nstrips
    int G;
    int nfull_rows, cpr, rpc, nchunks;  // border frame of the Gram matrix in 64-element chunks (gram_border_block's layout)
    int nbw, nbc_base, nbc_rem;         // border workgroups (border ranks 0 .. nbw-1), nchunks / nbw, nchunks % nbw
    int bx0, bx1, bn0;                  // border rank of a workgroup: on XCD bx0 (index & 7) index >> 3, on XCD bx1 bn0 + (index >> 3);
                                        // bx0 < 0: the workgroup index itself
    unsigned inv_cpr, inv_rpc;          // reciprocals for chunk_pos (wm_gram_common.hpp)
    unsigned epoch;           // value of this call's flags (never 0)
    float sF;
    double sqrt_n;
    // scratch of the slot (device memory), one buffer per phase: no address is read twice with different contents inside a launch
    double* pmain;    // [13][G] lag sums of every workgroup, then [44][nbw] border terms of the border workgroups (term-major)
    unsigned long long* gstat;  // [G][4] per workgroup, as {epoch, 32 bits} granules: max|e| (f32, or 0), the halves of sum (m W)^2 (f64)
    int folder;                 // workgroup that folds them
    unsigned long long* gdone;  // [G] flags: the workgroup's output stores are at the memory side
    unsigned long long* gcorr;  // [G][8] per workgroup, as granule pairs: <e_u,e_w>, |e_u|^2, |e_w|^2 (6 granules used)
    unsigned long long* gran;  // published values as {epoch, value} granules: [0..8] coefficients + status, [16..17] a, max|e|
    unsigned* cnt;    // arrival counters, one per 128-byte line: 3 hand-offs x (NSH shard counters + 1 top counter); zero between calls
    OpResult* res;    // result record (device-mapped pinned host memory)
    unsigned long long* stamps;  // development aid: [G][16] s_memrealtime stamps of the phase boundaries, or null
    int dbg;                     // development aid: bit 0 skip the lag products, bit 1 skip the border chunks (timing only, wrong
                                 // results; honoured only together with WM_FUSED_STAMPS); bit 2: workgroup 0 never arrives at a
                                 // hand-off (exercises the time-out and the fallback); bit 3: workgroup 0 never raises its
                                 // end-of-embed flag (the folder reports FUSED_INCOMPLETE)
};

struct FJob {
    int lane, wave;
    int c0s, dup, c0;
    int r0, rend;     // rows of this workgroup's tile
    int rs, nv;       // this wave's first row and number of valid rows (0: idle wave)
    int tl0;          // LDS row of rs
    bool last_active; // this wave holds the tile's last row
    bool own;         // lane owns at least one of its 4 columns (false in the duplicate lanes of a shifted last strip): it stores
    bool ok[4];       // pixel k of this lane is owned (a width that is not a multiple of 4 splits one lane of the shifted strip)
};

// grid = (strips, row bands): the workgroup's tile follows from blockIdx without a division (an integer division is ~25
// vector instructions per wave in front of the first row request); WG_ID is the linear index records and tickets use
#define WG_ID ((int)(blockIdx.y * gridDim.x + blockIdx.x))
template <int RPW>
__device__ __forceinline__ FJob make_fjob(const FusedArgs& a)
{
    FJob j;
    j.lane = threadIdx.x & (WAVE - 1);
    j.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int band = blockIdx.y, strip = blockIdx.x;
    j.c0s = strip * STRIP; j.dup = 0;
    // last strip moved left to end at the last column.  (f32 planes of any width: 16-byte accesses at 4-byte aligned addresses are
    // correct on gfx950; u8 planes only with cols % 4 == 0 -- a lane's 4 pixels are one dword -- the host checks)
    if (j.c0s + STRIP > a.cols) { j.dup = j.c0s - (a.cols - STRIP); j.c0s = a.cols - STRIP; }
    j.c0 = j.c0s + 4 * j.lane;
    j.r0 = band * a.th;
    j.rend = j.r0 + a.th < a.rows ? j.r0 + a.th : a.rows;
    j.rs = j.r0 + j.wave * RPW;
    const int left = j.rend - j.rs;
    j.nv = left <= 0 ? 0 : (left < RPW ? left : RPW);
    j.tl0 = j.rs - j.r0 + 2;
    j.last_active = j.nv > 0 && j.rs + RPW >= j.rend;
    j.own = 4 * j.lane + 3 >= j.dup;  // (duplicate pixels of a split lane are stored twice, with identical values)
#pragma unroll
    for (int k = 0; k < 4; ++k) j.ok[k] = 4 * j.lane + k >= j.dup;
    return j;
}

// phase-boundary time stamp of this workgroup (100 MHz clock), only when the host asked for them
#define FSTAMP8(a, k) do { if ((a).stamps && threadIdx.x == 512) (a).stamps[WG_ID * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FSTAMP(a, k) do { if ((a).stamps && threadIdx.x == 0) (a).stamps[WG_ID * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// ---- hand-offs inside the launch ------------------------------------------------------------------------------------
// Fan-in: 255 returning atomics on ONE address serialise (~13 ns each, 3.3 us for the last), so arrivals are counted in
// NSH shards (workgroups with equal blockIdx & 7) and the shards' last arrivers in a top counter; the fold follows the same
// two levels (a shard's last workgroup folds the shard's records in index order, the top's last workgroup folds the shard
// records), which also keeps every fold to ONE round of loads.  Every counter sits on a 128-byte line of its own.
// Fan-out: the published values are 8-byte {epoch, value} granules (the data is the flag): one wave per workgroup polls
// them, one round trip instead of flag-then-data.
static_assert(FUSED_MAX_WG == 4 * WAVE, "the folding wavefronts take workgroups l, l + 64, l + 128, l + 192: four per lane");
constexpr int NSH = 8;
constexpr int CNT_STRIDE = 32;  // unsigned words per counter line
__device__ __forceinline__ unsigned* cnt_shard(const FusedArgs& a, int handoff, int sh) { return a.cnt + (handoff * (NSH + 1) + sh) * CNT_STRIDE; }
__device__ __forceinline__ unsigned* cnt_top(const FusedArgs& a, int handoff) { return a.cnt + (handoff * (NSH + 1) + NSH) * CNT_STRIDE; }

// one arrival on `cnt` for the whole workgroup (after its record stores); true in all threads if it was the last of `expected`
__device__ __forceinline__ bool arrive(unsigned* cnt, unsigned expected, unsigned* s_word, bool stored = true)
{
    // every wave that stored a part of the record drains its stores (`stored` is wave-uniform); the others must not wait
    // here: their outstanding requests are the next phase's operands
    if (stored) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = prev + 1u == expected;
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
        *s_word = last ? 1u : 0u;
    }
    __syncthreads();
    return *s_word != 0u;
}
// Two-level convergence: the workgroup's arrival in its shard (the workgroups with the same index & 7), the shard's last
// arriver in the top counter.  Returns true in the one workgroup that arrived last overall (every record is visible to it).
__device__ __forceinline__ bool converge(const FusedArgs& a, int handoff, unsigned* s_word, bool stored = true)
{
    const int sh = WG_ID & (NSH - 1);
    const int n = (a.G - sh + NSH - 1) / NSH;
    if (!arrive(cnt_shard(a, handoff, sh), (unsigned)n, s_word, stored)) return false;
    return arrive(cnt_top(a, handoff), (unsigned)(a.G < NSH ? a.G : NSH), s_word, stored);
}

__device__ __forceinline__ void put_granule(unsigned long long* g, unsigned epoch, unsigned value)
{
    __hip_atomic_store(g, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a 64-bit value as two granules (fan-in: the reader needs no ticket, no drained stores and no second round trip -- every
// half says by itself whether it belongs to this call)
__device__ __forceinline__ void put_pair(unsigned long long* g, unsigned epoch, double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    put_granule(g, epoch, (unsigned)b);
    put_granule(g + 1, epoch, (unsigned)(b >> 32));
}
// wave 0 of every workgroup polls the n (<= 64) granules until all carry this call's epoch, then hands the values to the
// workgroup through LDS.  False in all threads on time-out.
__device__ __forceinline__ bool fetch_granules(const unsigned long long* g, int n, unsigned epoch, unsigned* s_vals, unsigned* s_ok)
{
    if (threadIdx.x < WAVE) {
        const int lane = threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long x = 0;
        bool ok = false;
        for (;;) {
            if (lane < n) x = __hip_atomic_load(const_cast<unsigned long long*>(g) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = lane >= n || (unsigned)(x >> 32) == epoch;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_WAIT_TICKS) break;
        }
        if (lane < n) s_vals[lane] = (unsigned)x;
        if (lane == 0) *s_ok = __all(ok) ? 1u : 0u;
    }
    __syncthreads();
    return *s_ok != 0u;
}

// =================================================================================================
// Phase A: this wave's rows from HBM into the LDS tile; with GRAM the 13 exact lag sums of its core pixels on the way
// (gram_march_impl's arithmetic: f64 products of the f32 / u8 pixels, wm_k_gram.hip)
// =================================================================================================
struct NoOp { __device__ __forceinline__ void operator()() const {} };
// RELOAD (the detector half of k_fused_pair): the tile's own rows are in LDS already -- the plane the embed half has just
// written -- so only the halo comes from memory: the two rows above (wave 0), the two rows below the tile (every wave requests
// them, two L2 hits; the waves whose window reaches them use them) and the side pairs of every row.
template <typename T, int RPW, bool GRAM, bool RELOAD = false, typename AFTER = NoOp>
__device__ __forceinline__ void phase_load(const T* xf, long long pitch, const FusedArgs& a, const FJob& j, const LdsView& L,
                                           double (&acc)[13], AFTER&& after_issue = NoOp())
{
    constexpr int NS = RPW + 2;  // rows streamed: rs .. rs + RPW + 1
    // rows in flight per wavefront: with the Gram sums and RPW = 8, what 128 VGPRs leave beside the f64 window
    // (u8 frames take the integer march below: its window is 15 dwords, every row can be in flight)
    constexpr bool IGRAM = GRAM && sizeof(T) == 1;
    constexpr int PF = RELOAD ? 2 : ((GRAM && !IGRAM && RPW == 8) ? 4 : NS);  // (RELOAD: the two rows below the tile)
    // The first row requests leave as early as the wave can form them: the four wavefronts of a SIMD issue oldest first, so
    // whatever a wave executes before its requests also delays the requests of the younger waves behind it.
    FLoad<T> ld;
    ld.init(xf, pitch, a.rows, a.cols, j.c0s, j.lane);
    const int R = a.rows, C = a.cols;
    typename FLoad<T>::Raw pre[PF], t0, t1;
    // wave 0 also takes the tile's two halo rows above (row index clamped at the image's top): LDS rows 0 and 1
    const int up = j.wave == 0 ? 2 : 0;
    const typename FLoad<T>::H2 hraw = RELOAD ? ld.issue_halos_coherent(j.rs - up, NS + up, j.lane) : ld.issue_halos(j.rs - up, NS + up, j.lane);
    if (j.wave == 0) {
        if constexpr (RELOAD) { t0 = ld.issue_coherent(j.r0 - 2); t1 = ld.issue_coherent(j.r0 - 1); }
        else { t0 = ld.issue(j.r0 - 2); t1 = ld.issue(j.r0 - 1); }
    }
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        if constexpr (RELOAD) pre[q] = ld.issue_coherent(j.rend + q);
        else pre[q] = ld.issue(j.rs + (q < NS ? q : NS - 1));
    }
    FSTAMP8(a, 11);
    after_issue();  // requests that should queue behind the first image rows (the next phase's operands)
    FSTAMP(a, 8);
    if (j.nv == 0) return;
    lds_put_halos(L, ld, j.tl0 - up, NS + up, j.lane, fcvt2(hraw));
    if (j.wave == 0) {
        float v[8];
        row8(L, 0, j.lane, fcvt4(t0.v), v);
        lds_put_row(L, 0, j.lane, v);
        row8(L, 1, j.lane, fcvt4(t1.v), v);
        lds_put_row(L, 1, j.lane, v);
    }
    double w[3][8];
    bool cv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cv[k] = j.c0 + k >= 2 && j.c0 + k <= C - 3 && j.ok[k];
    // u8 frames: the 13 lag sums in exact integer arithmetic, as k_gram's aligned path does (gram_march_u8, wm_k_gram.hip): a
    // lane's 4 pixels are one packed dword, the partner pixels byte-shifted dwords (v_alignbyte of the neighbours' dwords by DPP),
    // one v_dot4_u32_u8 per lag and row instead of 4 f64 FMAs + conversions; u32 sums of a wave's <= 8 rows cannot overflow.
    uint32_t ish[3][5], iacc[13], cmask = 0;
    if constexpr (IGRAM) {
#pragma unroll
        for (int k = 0; k < 4; ++k) cmask |= cv[k] ? 0xffu << (8 * k) : 0u;
#pragma unroll
        for (int l = 0; l < 13; ++l) iacc[l] = 0u;
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
            for (int b = 0; b < 5; ++b) ish[a2][b] = 0u;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        typename FLoad<T>::Raw rawr;
        float4 f;
        bool put;
        if constexpr (RELOAD) {
            // rows of the tile: LDS (own pixels as floats); row rend / rend + 1: the two requested rows (beyond: irrelevant)
            const int row = j.rs + s;
            const bool inside = row < j.rend;
            const float4 fl = reinterpret_cast<const float4*>(L.tile + (j.tl0 + s) * STRIP)[j.lane];
            const float4 fg = fcvt4(row == j.rend ? pre[0].v : pre[1].v);
            f = inside ? fl : fg;
            rawr = pre[0];
            if constexpr (IGRAM) rawr.v = (typename Elem<T>::vec4)Elem<T>::pack(out_cvt<T>(f.x), out_cvt<T>(f.y), out_cvt<T>(f.z), out_cvt<T>(f.w));
            put = !inside && j.last_active;
        } else {
            rawr = pre[s % PF];
            f = fcvt4(rawr.v);
            if (s + PF < NS) pre[s % PF] = ld.issue(j.rs + s + PF);
            put = s < RPW || j.last_active;
        }
        const auto rawv = rawr.v;
        float v[8];
        row8(L, j.tl0 + s, j.lane, f, v);
        if (put) lds_put_row(L, j.tl0 + s, j.lane, v);
        if constexpr (IGRAM) {
            const uint32_t own = rawv;
            // the strip's halo pairs of this row as the bytes a neighbour's dword would hold: left pair = bytes 2, 3 of "lane -1",
            // right pair = bytes 0, 1 of "lane 64" (the side array holds them as floats, replicate already applied)
            const float4 hh = *reinterpret_cast<const float4*>(L.halo + (j.tl0 + s) * 4);
            const uint32_t hl = ((uint32_t)hh.x << 16) | ((uint32_t)hh.y << 24), hr = (uint32_t)hh.z | ((uint32_t)hh.w << 8);
            const uint32_t Lw = (uint32_t)__builtin_amdgcn_update_dpp((int)hl, (int)own, 0x138, 0xF, 0xF, false);
            const uint32_t Rw = (uint32_t)__builtin_amdgcn_update_dpp((int)hr, (int)own, 0x130, 0xF, 0xF, false);
            uint32_t* s2 = ish[s % 3];
            s2[0] = __builtin_amdgcn_alignbyte(own, Lw, 2);
            s2[1] = __builtin_amdgcn_alignbyte(own, Lw, 3);
            s2[2] = own;
            s2[3] = __builtin_amdgcn_alignbyte(Rw, own, 1);
            s2[4] = __builtin_amdgcn_alignbyte(Rw, own, 2);
            if (s == 0) { FSTAMP(a, 9); FSTAMP8(a, 12); }
            if (s >= 2 && !(a.dbg & 1)) {
                const int q = j.rs + s - 2;
                const bool vq = q >= 1 && q <= R - 3 && s - 2 < j.nv;
                const uint32_t* w0 = ish[(s - 2) % 3];
                const uint32_t* w1 = ish[(s - 1) % 3];
                const uint32_t A = vq ? (w0[2] & cmask) : 0u;
                iacc[0] = __builtin_amdgcn_udot4(A, w0[2], iacc[0], false);
                iacc[1] = __builtin_amdgcn_udot4(A, w0[3], iacc[1], false);
                iacc[2] = __builtin_amdgcn_udot4(A, w0[4], iacc[2], false);
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    iacc[3 + b] = __builtin_amdgcn_udot4(A, w1[b], iacc[3 + b], false);
                    iacc[8 + b] = __builtin_amdgcn_udot4(A, s2[b], iacc[8 + b], false);
                }
            }
        } else if constexpr (GRAM) {
#pragma unroll
            for (int b = 0; b < 8; ++b) w[s % 3][b] = (double)v[b];
            if (s == 0) { FSTAMP(a, 9); FSTAMP8(a, 12); }
            if (s >= 2 && !(a.dbg & 1)) {
                // q row = rs + s - 2 (window rows s-2, s-1, s); in the core 1 <= q <= R-3 and one of this wave's valid rows
                const int q = j.rs + s - 2;
                const bool vq = q >= 1 && q <= R - 3 && s - 2 < j.nv;
                const double* w0 = w[(s - 2) % 3];
                const double* w1 = w[(s - 1) % 3];
                const double* w2 = w[s % 3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double xq = (vq && cv[k]) ? w0[2 + k] : 0.0;
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
        }
    }
    if constexpr (IGRAM) {
#pragma unroll
        for (int l = 0; l < 13; ++l) acc[l] = (double)iacc[l];
    }
}

// The Gram matrix's border frame in 64-element chunks (wm_gram_common.hpp: border_chunk_issue / border_chunk_terms).
// nbw workgroups take them (border rank r < nbw), chunk ch = r + nbw * ci for ci < nbc <= 16, one per wavefront (the oldest
// ones, which finish their march ~5 us before the workgroup's barrier), after the wave's march; only these workgroups leave
// border records, so the exposed fold reads 13 G + 44 nbw doubles (4K: 49 KB through one CU, at the ~66 GB/s a CU reads
// other XCDs' fresh lines with; 116 KB when every workgroup carries border terms).  8 chunks cost a border workgroup 0.9 us:
// the border workgroups are those of the two XCDs whose workgroups start first (a launch reaches the XCDs one after the
// other, 0.2 us apart, in the order 1 2 3 4 7 0 5 6 on every box seen; tools/fused_skew.py) -- they have that time, the
// workgroups of the last XCD, which everybody waits for at the hand-off, do not.  Nothing but that microsecond depends on
// the order; small grids take the first nbw workgroups.
constexpr int CHUNKS_PER_BORDER_WG = 8;
__device__ __forceinline__ int border_rank(const FusedArgs& a)
{
    if (a.bx0 < 0) return WG_ID;
    const int xk = WG_ID & 7;
    return xk == a.bx0 ? (WG_ID >> 3) : (xk == a.bx1 ? a.bn0 + (WG_ID >> 3) : a.nbw);
}

// Gram phase of a workgroup up to the coefficients: load + lag sums + border chunks, workgroup record, two-level
// convergence with the folds, solve by the last workgroup, granules.  On return (true) c[] / st hold the frame's
// coefficients / status in every thread.  `prefetch` runs between the ticket and the wait: it issues the next phase's
// global loads, whose latency then hides behind the fold and the solve.
template <typename T, int RPW, bool RELOAD = false, typename PF>
__device__ __forceinline__ bool gram_phase(const T* xf, long long pitch, const FusedArgs& a, const FJob& j, const LdsView& L,
                                           float (&c)[8], int& st, PF&& prefetch)
{
    constexpr int FW = fw_of(RPW);
    double acc[13];
#pragma unroll
    for (int l = 0; l < 13; ++l) acc[l] = 0.0;
    const int brank = border_rank(a);
    const int nbc = (a.dbg & 2) || brank >= a.nbw ? 0 : a.nbc_base + (brank < a.nbc_rem ? 1 : 0);  // border chunks of this workgroup (<= FW)
    double* sc = L.fold + j.wave * 40;  // 39 doubles of scratch per wave (the fold scratch is free until the hand-off)
    const bool loader = j.wave < nbc;
    const BorderGeom bg{a.rows, a.cols, a.nfull_rows, a.cpr, a.rpc, 0, a.rows, false, a.inv_cpr, a.inv_rpc, a.cols % 4 == 0};  // (side columns by row loads: widths that are multiples of 4)
    phase_load<T, RPW, true, RELOAD>(xf, pitch, a, j, L, acc);
    // the next phase's operands stream in behind the image rows, under the reductions.  Waves 0 and 1 store the workgroup's
    // record and must see those stores acknowledged before the ticket (one in-order counter covers loads and stores): they
    // are the oldest waves of their SIMDs, finish the march ~5 us before the workgroup's barrier, and their operands are in
    // by then.  (Requested after the ticket, as they were, these loads were in flight during the fold: one round trip of
    // the folding workgroup, the longer the more the memory system carries.)
    prefetch();
    FSTAMP(a, 10);
    FSTAMP8(a, 13);
    {
        int idx;
        const double s = wave_sum_multi<13>(acc, j.lane, idx);
        if (idx < 13) L.red[j.wave * 13 + idx] = s;
    }
    if (loader) {
        const int ch = brank + a.nbw * j.wave;
        const BorderVals<T> b2 = border_chunk_issue<T, RELOAD>(xf, pitch, bg, ch, j.lane);  // (RELOAD: coherent loads, see FLoad::load_coherent)
        const double t2 = border_chunk_terms<T>(b2, bg, ch, j.lane, sc);
        if (j.lane < NGRAM) L.bor[j.wave * NGRAM + j.lane] = t2;
    }
    FSTAMP8(a, 14);
    FSTAMP(a, 15);
    __syncthreads();
    FSTAMP(a, 1);
    // the workgroup's record, stored TERM-major ([13][G], then [44][nbw]) so that the fold reads whole cache lines: 13 lag
    // sums (waves in index order) + 44 border terms (chunks in index order) in the border workgroups
    const int t = threadIdx.x;
    if (t < 13) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < FW; ++w) s += L.red[w * 13 + t];
        st_agent(a.pmain + (long long)t * a.G + WG_ID, s);
    } else if (t >= WAVE && t < WAVE + NGRAM && brank < a.nbw) {
        const int k = t - WAVE;
        double s = 0.0;
        for (int ci = 0; ci < nbc; ++ci) s += L.bor[ci * NGRAM + k];
        st_agent(a.pmain + 13LL * a.G + (long long)k * a.nbw + brank, s);
    }
    if ((a.dbg & 4) && WG_ID == 0) return false;  // test hook: a workgroup that never arrives (the others time out)
    // the 13 G + 44 nbw doubles are read in ONE round by the last workgroup; the shards only spread the tickets
    const bool is_last = converge(a, 0, L.flags + 0, j.wave < 2);
    FSTAMP(a, 2);
    if (is_last) {
        // term k is folded by the 16 lanes of one DPP row: lane q sums records q, q + 16, ... (index order, all loads in
        // flight at once; a row reads 128 contiguous bytes per step), then the row is summed in lane order
        for (int k = t >> 4; k < FNT; k += FW * WAVE / 16) {  // (uniform trip count per 16-lane row)
            const int q = t & 15;
            double s = 0.0;
            const int n = k < 13 ? a.G : a.nbw;  // records of term k
            const double* p = k < 13 ? a.pmain + (long long)k * a.G : a.pmain + 13LL * a.G + (long long)(k - 13) * a.nbw;
            for (int b0 = q; b0 < n; b0 += 16 * 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {  // (no request beyond the term's records: these loads bypass L1, a clamped duplicate costs a full one)
                    v[u] = 0.0;
                    if (b0 + 16 * u < n) v[u] = ld_agent(p + b0 + 16 * u);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) s += v[u];
            }
            s += dpp_mov0<0x111, 0xF>(s);  // row_shr:1
            s += dpp_mov0<0x112, 0xF>(s);  // row_shr:2
            s += dpp_mov0<0x114, 0xF>(s);  // row_shr:4
            s += dpp_mov0<0x118, 0xF>(s);  // row_shr:8  -> lane 15 of the row holds the term's total
            if (q == 15) L.fold[k] = s;
        }
        __syncthreads();
        if (a.stamps && t == 0) a.stamps[16 * a.G + 0] = __builtin_amdgcn_s_memrealtime();
        if (t < WAVE) {  // one wave: the 44 totals, then the solve
            if (t < NGRAM) {
                constexpr GramTab tab = make_gram_tab();
                L.s_tot[t] = L.fold[13 + t] + L.fold[tab.lag[t]];
                if (a.stamps) reinterpret_cast<double*>(a.stamps + 16 * (a.G + 1))[t] = L.s_tot[t];  // (tests compare the folded sums)
            }
            wave_lds_fence();
            if (a.stamps && t == 0) a.stamps[16 * a.G + 1] = __builtin_amdgcn_s_memrealtime();
            float cc[8];
            const int stt = spd_solve_lanes(L.s_tot, t, cc);
            float v = cc[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) v = t == k ? cc[k] : v;
            if (t < 8) put_granule(a.gran + t, a.epoch, __float_as_uint(v));
            if (t == 8) put_granule(a.gran + 8, a.epoch, (unsigned)stt);
            if (a.stamps && t == 0) a.stamps[16 * a.G + 2] = __builtin_amdgcn_s_memrealtime();
        }
    }
    unsigned* vals = L.flags + 8;
    if (!fetch_granules(a.gran, 9, a.epoch, vals, L.flags + 1)) return false;
    FSTAMP(a, 3);
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = __uint_as_float(vals[k]);
    st = (int)vals[8];
    return true;
}

// W (or a base plane) of this thread's RPW rows x 4 pixels
template <int RPW>
__device__ __forceinline__ void load_rows4(const float* P, long long pitch, const FJob& j, int rows, float4 (&w)[RPW])
{
#pragma unroll
    for (int i = 0; i < RPW; ++i) w[i] = *reinterpret_cast<const float4*>(P + (long long)min(j.rs + i, rows - 1) * pitch + j.c0);
}

template <typename TB>
__device__ __forceinline__ float4 ld_base4(const TB* p)
{
    if constexpr (sizeof(TB) == 4) return *reinterpret_cast<const float4*>(p);
    else return fcvt4(*reinterpret_cast<const uint32_t*>(p));
}

// ---- the output plane, written THROUGH to memory (sc0 sc1): when the last workgroup reports the frame done, every byte
// of y is at the memory side, visible to the host, the copy engines and any later kernel -- the host may then return from
// the call on the record's status word (a poll of device-mapped pinned memory) instead of waiting for the stream, which
// costs 5.5 us more per call (tools/ubench/launch_sync.hip).  hipcc does not count asm stores: the callers drain with
// s_waitcnt vmcnt(0) (arrive()) before they signal.
template <typename T>
__device__ __forceinline__ void store4_through(T* p, float4 y)
{
    if constexpr (sizeof(T) == 4) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v v; v.x = y.x; v.y = y.y; v.z = y.z; v.w = y.w;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
        const uint32_t v = Elem<T>::pack(out_cvt<T>(y.x), out_cvt<T>(y.y), out_cvt<T>(y.z), out_cvt<T>(y.w));
        asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
    }
}
template <typename V>
__device__ __forceinline__ void copy_through(V* dst, const V* src)
{
    if constexpr (sizeof(V) == 16) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = *reinterpret_cast<const f4v*>(src);
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    } else {
        static_assert(sizeof(V) == 4, "4 pixels of an f32 or u8 plane");
        const uint32_t v = *reinterpret_cast<const uint32_t*>(src);
        asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(dst), "v"(v) : "memory");
    }
}
// every workgroup after its last store: the one that arrives last reports the frame.
// The result record lives in device-mapped pinned host memory and the host polls its status word.  Status and value leave
// as ONE 8-byte store (one write across the host link; two stores with a release between them cost a second crossing and
// the wait for the first).  Nothing has to be ordered in front of it: every workgroup's stores were acknowledged before it
// took its ticket, and the reporter has seen all tickets.
__device__ __forceinline__ void report(OpResult* res, int status, float value)
{
    static_assert(sizeof(OpResult) == 8 && alignof(OpResult) == 8, "the result record is one naturally aligned 8-byte store");
    const unsigned long long rec = (unsigned long long)(unsigned)status | ((unsigned long long)__float_as_uint(value) << 32);
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(res), "v"(rec) : "memory");
}
// End of an embed: every workgroup drains its y stores and then raises its flag (one granule); the folding workgroup polls
// the flags and reports (one flag store + a poll instead of a ticket and, for a shard's last workgroup, a second one).
__device__ __forceinline__ void finish_frame(const FusedArgs& a, const LdsView& L, int status, float value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores are acknowledged by the memory side
    __syncthreads();
    // (test hook, dbg bit 3: workgroup 0's completion is never seen although its output IS written -- only its flag is withheld;
    // were it the folder itself, it still polls and reports)
    if (threadIdx.x == 0 && !((a.dbg & 8) && WG_ID == 0)) put_granule(a.gdone + WG_ID, a.epoch, 1u);
    if (WG_ID != a.folder || threadIdx.x >= WAVE) return;
    const int l = threadIdx.x;
    unsigned pend = 0u;
#pragma unroll
    for (int u = 0; u < FUSED_MAX_WG / WAVE; ++u)
        if (l + u * WAVE < a.G) pend |= 1u << u;  // (fused_geometry: G <= FUSED_MAX_WG)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool timed_out = false;
    while (__any(pend != 0u)) {
        unsigned long long g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pend >> u & 1u) g[u] = ld_agent(a.gdone + l + u * WAVE);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((pend >> u & 1u) && (unsigned)(g[u] >> 32) == a.epoch) pend &= ~(1u << u);
        if (!__any(pend != 0u)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_DONE_TICKS) { timed_out = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    // Output stores have been issued by now (this workgroup's at least), so a time-out here must NOT look like "nothing
    // happened": the host would re-run the call on the sweeps, and for an in-place frame (the video contract, main.cpp:356,380)
    // that would watermark an already watermarked frame.  Say what is known: the end of the frame was not observed.
    if (l == 0) report(a.res, timed_out ? FUSED_INCOMPLETE : status, value);
}

// k_fused_pair, between its halves: like finish_frame, but EVERY workgroup waits until all have raised their flags (the detector
// half reads its neighbours' rows of y, the border workgroups any part of the frame's border); the folding workgroup reports
// the embed's record.  False (in all threads) on a time-out: the workgroup ends, the host sees an incomplete pair.
__device__ __forceinline__ bool pair_barrier(const FusedArgs& a, const LdsView& L, int status, float value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && !((a.dbg & 8) && WG_ID == 0)) put_granule(a.gdone + WG_ID, a.epoch, 1u);
    if (threadIdx.x < WAVE) {
        const int l = threadIdx.x;
        // Whose flags: the folding workgroup (it reports the embed) and the border workgroups (their chunks lie anywhere along
        // the frame's border) wait for everybody; the others for the (up to) 8 workgroups around their tile, whose rows and
        // columns of y are their halo -- they may start the detector half while distant tiles are still being written
        const bool all = WG_ID == a.folder || border_rank(a) < a.nbw;
        unsigned pend = 0u;
        int nb = -1;
        if (all) {
#pragma unroll
            for (int u = 0; u < FUSED_MAX_WG / WAVE; ++u)
                if (l + u * WAVE < a.G) pend |= 1u << u;
        } else if (l < 9 && l != 4) {
            const int bx = (int)blockIdx.x + l % 3 - 1, by = (int)blockIdx.y + l / 3 - 1;
            if (bx >= 0 && bx < (int)gridDim.x && by >= 0 && by < (int)gridDim.y) { nb = by * (int)gridDim.x + bx; pend = 1u; }
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool timed_out = false;
        while (__any(pend != 0u)) {
            unsigned long long g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pend >> u & 1u) g[u] = ld_agent(a.gdone + (all ? l + u * WAVE : nb));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((pend >> u & 1u) && (unsigned)(g[u] >> 32) == a.epoch) pend &= ~(1u << u);
            if (!__any(pend != 0u)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_DONE_TICKS) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (WG_ID == a.folder && l == 0) report(a.res, timed_out ? FUSED_INCOMPLETE : status, value);
        if (l == 0) L.flags[1] = timed_out ? 0u : 1u;
    }
    __syncthreads();
    return L.flags[1] != 0u;
}

// =================================================================================================
// k_fused_embed: makeWatermark of ONE frame in one launch (Watermark.cpp:156-172)
//   MASK 0 (ME): Gram -> c -> e, max|e|, sum (|e| W)^2 -> a -> y;  MASK 1 (NVF, p = 3): m, sum (m W)^2 -> a -> y
//   BX: the base is the grey input plane itself (taken from the LDS tile)
// =================================================================================================
// PAIR (k_fused_pair): y (as the detector will read it: after the output conversion) also replaces x in the LDS tile, and the
// body ends in pair_barrier instead of finish_frame; true = go on with the detector half
template <typename T, typename TB, int NCH, int MASK, int RPW, bool BX, bool PAIR>
__device__ __forceinline__ bool fused_embed_body(const T* __restrict__ x, long long pitch, const float* __restrict__ W,
                                                 const PlaneDesc& base, const PlaneDesc& out, const FusedArgs& a, const LdsView& L, const FJob& j)
{
    constexpr int FW = fw_of(RPW);
    FSTAMP(a, 0);
    float4 w[RPW];
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int st = 0;
    if (MASK == 0) {
        if (!gram_phase<T, RPW>(x, pitch, a, j, L, c, st, [&]() { load_rows4<RPW>(W, a.cols, j, a.rows, w); })) return false;
    } else {
        double unused[13];
        phase_load<T, RPW, false>(x, pitch, a, j, L, unused, [&]() { load_rows4<RPW>(W, a.cols, j, a.rows, w); });
        __syncthreads();
    }
    const TB* bptr = static_cast<const TB*>(base.p);
    TB* optr = static_cast<TB*>(const_cast<void*>(out.p));
    if (st != 0) {
        // unsolvable: out = base bit-exact, strength untouched (Watermark.cpp:164-165)
        if (bptr != optr && j.own) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                for (int i = 0; i < j.nv; ++i) {
                    const long long ro = (long long)(j.rs + i);
                    copy_through(reinterpret_cast<typename Elem<TB>::vec4*>(optr + (long long)ch * out.cstride + ro * out.pitch + j.c0),
                                 reinterpret_cast<const typename Elem<TB>::vec4*>(bptr + (long long)ch * base.cstride + ro * base.pitch + j.c0));
                }
        }
        if constexpr (PAIR) {
            // the detector half scores the plane as it now is: the base (in the tile already when the base is the input)
            if (!BX) {
                for (int i = 0; i < j.nv; ++i)
                    reinterpret_cast<float4*>(L.tile + (j.tl0 + i) * STRIP)[j.lane] = ld_base4<TB>(bptr + (long long)(j.rs + i) * base.pitch + j.c0);
            }
            return pair_barrier(a, L, st, 0.0f);
        }
        finish_frame(a, L, st, 0.0f);
        return false;
    }
    // ---- mask values of the own pixels from the LDS tile: m[i][k] = |e| (ME, before the 1/max|e|) or nvf (NVF)
    float m[RPW][4];
    float mx = 0.0f, ss = 0.0f;
    {
        float r6[3][6];
        lds_row6(L, j.tl0 - 1, j.lane, r6[0]);
        lds_row6(L, j.tl0, j.lane, r6[1]);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            lds_row6(L, j.tl0 + i + 1, j.lane, r6[(i + 2) % 3]);
            const float* up = r6[i % 3];
            const float* mid = r6[(i + 1) % 3];
            const float* dn = r6[(i + 2) % 3];
            const bool use = i < j.nv;
            float pr[4] = {0.f, 0.f, 0.f, 0.f};
            if (MASK == 0) predict4<1>(up, mid, dn, c, pr);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float mv;
                if (MASK == 0) mv = fabsf(mid[1 + k] - pr[k]);
                else mv = nvf_3x3(up + k, mid + k, dn + k);
                m[i][k] = mv;
                if (use && j.ok[k]) {
                    mx = fmaxf(mx, mv);
                    const float tt = mv * f4get(w[i], k);
                    ss = fmaf(tt, tt, ss);
                }
            }
        }
    }
    mx = wave_max(mx);
    const double ssd = wave_sum((double)ss);
    if (j.lane == 0) { L.wred[j.wave] = (double)mx; L.wred[FW + j.wave] = ssd; }
    __syncthreads();
    FSTAMP(a, 4);
    if ((a.dbg & 4) && WG_ID == 0) return false;  // test hook, see gram_phase
    // The workgroup's two statistics travel as {epoch, 32 bits} granules (max|e| is an f32, the sum two halves of an f64): ONE
    // workgroup polls them, so this hand-off needs no drained stores, no tickets and no separate round of fold loads behind
    // the last arrival -- a record is complete when its three granules carry this call's epoch.  (With 3 granules per
    // workgroup a poll round is one short round trip; for the 57-term Gram records it is not, see gram_phase.)
    if (threadIdx.x == 0) {
        double bm = 0.0, bs = 0.0;
#pragma unroll
        for (int q = 0; q < FW; ++q) { bm = fmax(bm, L.wred[q]); bs += L.wred[FW + q]; }
        const unsigned long long sb = (unsigned long long)__double_as_longlong(bs);
        unsigned long long* g = a.gstat + 4LL * WG_ID;
        put_granule(g, a.epoch, __float_as_uint((float)bm));
        put_granule(g + 1, a.epoch, (unsigned)sb);
        put_granule(g + 2, a.epoch, (unsigned)(sb >> 32));
    }
    FSTAMP(a, 5);
    if (WG_ID == a.folder && threadIdx.x < WAVE) {
        // the frame's strength (embed_scalars_frame, wm_k_embed.hip): lane l takes workgroups l, l + 64, l + 128, l + 192 (the
        // grid never exceeds 256), polls until each has delivered, folds them in index order, then the fixed wave trees
        const int l = threadIdx.x;
        float vm[4];
        double vs[4];
        unsigned pend = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            vm[u] = 0.0f; vs[u] = 0.0;
            if (l + u * WAVE < a.G) pend |= 1u << u;
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool timed_out = false;
        while (__any(pend != 0u)) {
            unsigned long long g0[4], g1[4], g2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pend >> u & 1u) {
                    const unsigned long long* g = a.gstat + 4LL * (l + u * WAVE);
                    g0[u] = ld_agent(g); g1[u] = ld_agent(g + 1); g2[u] = ld_agent(g + 2);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((pend >> u & 1u) && (unsigned)(g0[u] >> 32) == a.epoch && (unsigned)(g1[u] >> 32) == a.epoch && (unsigned)(g2[u] >> 32) == a.epoch) {
                    vm[u] = __uint_as_float((unsigned)g0[u]);
                    vs[u] = __longlong_as_double((long long)((g2[u] << 32) | (g1[u] & 0xffffffffull)));
                    pend &= ~(1u << u);
                }
            if (!__any(pend != 0u)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_FOLD_TICKS) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        double fm = 0.0, fs = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { fm = fmax(fm, (double)vm[u]); fs += vs[u]; }
        fm = wave_max_d(fm);
        fs = wave_sum(fs);
        const float maxe_f = MASK == 0 ? (float)fm : 1.0f;
        const double nrm = MASK == 0 ? sqrt(fs) / (double)maxe_f : sqrt(fs);
        const float a_f = a.sF / (float)(nrm / a.sqrt_n);
        if (!timed_out) {  // (nothing is published after a time-out: every workgroup then times out, the host takes the sweeps)
            if (l == 0) put_granule(a.gran + 16, a.epoch, __float_as_uint(a_f));
            if (l == 1) put_granule(a.gran + 17, a.epoch, __float_as_uint(maxe_f));
        }
    }
    // operands of the last phase, requested before the wait (behind the folding wave's polls: its registers are taken until then): the first base plane (unless it is the LDS tile)
    float4 b0[RPW];
    if (!BX) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) b0[i] = ld_base4<TB>(bptr + (long long)min(j.rs + i, a.rows - 1) * base.pitch + j.c0);
    }
    unsigned* vals = L.flags + 8;
    if (!fetch_granules(a.gran + 16, 2, a.epoch, vals, L.flags + 1)) return false;
    FSTAMP(a, 6);
    const float sa = __uint_as_float(vals[0]);
    const float maxe = __uint_as_float(vals[1]);
    const float inv_maxe = 1.0f / maxe;
    // ---- y = clamp(base + a * m * W, 0, 255)   (Watermark.cpp:169-171)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            float4 b;
            if (BX) b = reinterpret_cast<const float4*>(L.tile + (j.tl0 + i) * STRIP)[j.lane];
            else if (ch == 0) b = b0[i];
            else b = ld_base4<TB>(bptr + (long long)ch * base.cstride + (long long)min(j.rs + i, a.rows - 1) * base.pitch + j.c0);
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float mk = MASK == 0 ? div_by(m[i][k], maxe, inv_maxe) : m[i][k];
                u[k] = mk * f4get(w[i], k);
            }
            float4 y;
            y.x = fminf(fmaxf(fmaf(u[0], sa, b.x), 0.0f), 255.0f);
            y.y = fminf(fmaxf(fmaf(u[1], sa, b.y), 0.0f), 255.0f);
            y.z = fminf(fmaxf(fmaf(u[2], sa, b.z), 0.0f), 255.0f);
            y.w = fminf(fmaxf(fmaf(u[3], sa, b.w), 0.0f), 255.0f);
            if (j.own && i < j.nv) store4_through<TB>(optr + (long long)ch * out.cstride + (long long)(j.rs + i) * out.pitch + j.c0, y);
            if constexpr (PAIR) {
                // (the row's base was read above; every lane -- duplicate lanes hold the same pixels -- keeps its y in the tile)
                if (i < j.nv)
                    reinterpret_cast<float4*>(L.tile + (j.tl0 + i) * STRIP)[j.lane] =
                        make_float4((float)out_cvt<TB>(y.x), (float)out_cvt<TB>(y.y), (float)out_cvt<TB>(y.z), (float)out_cvt<TB>(y.w));
            }
        }
    }
    if constexpr (PAIR) {
        const bool go = pair_barrier(a, L, 0, sa);
        FSTAMP(a, 7);
        return go;
    }
    finish_frame(a, L, 0, sa);  // (every workgroup knows the strength; the one that arrives last reports it)
    FSTAMP(a, 7);
    return false;
}

template <typename T, typename TB, int NCH, int MASK, int RPW, bool BX>
__global__ __launch_bounds__(fw_of(RPW) * WAVE) void k_fused_embed(const T* __restrict__ x, long long pitch, const float* __restrict__ W,
                                                        PlaneDesc base, PlaneDesc out, FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsView L = carve<RPW>(smem);
    const FJob j = make_fjob<RPW>(a);
    (void)fused_embed_body<T, TB, NCH, MASK, RPW, BX, false>(x, pitch, W, base, out, a, L, j);
}

// =================================================================================================
// k_fused_detect: detectWatermark of ONE frame in one launch (Watermark.cpp:234-250)
//   Gram -> c;  e_w and u = m W of the own pixels (registers);  u replaces x in the LDS tile (replicate-padded like the
//   reference's u image);  e_u = u - c.nbrs(u);  <e_u,e_w>, |e_u|^2, |e_w|^2 -> corr by the last workgroup
// =================================================================================================
// RELOAD: the detector half of k_fused_pair (the tile's own rows are in LDS, see phase_load)
template <typename T, int MASK, int RPW, bool RELOAD>
__device__ __forceinline__ void fused_detect_body(const T* __restrict__ x, long long pitch, const float* __restrict__ W, const FusedArgs& a,
                                                  const LdsView& L, const FJob& j)
{
    constexpr int FW = fw_of(RPW);
    if (!RELOAD) FSTAMP(a, 0);
    const int R = a.rows, C = a.cols;
    // W of the u rows this wave may produce: i = -1 .. RPW (row -1 / row nv only matter at the tile's top / bottom), and
    // W at this lane's halo column (lane 63: c0s+256, others: c0s-1; clamped, the replicate cases never use it)
    float4 w[RPW + 2];
    float wh[RPW + 2];
    const int wh_col = j.lane == WAVE - 1 ? min(j.c0s + STRIP, C - 1) : max(j.c0s - 1, 0);
    float c[8];
    int st = 0;
    if (!gram_phase<T, RPW, RELOAD>(x, pitch, a, j, L, c, st, [&]() {
#pragma unroll
            for (int i = 0; i < RPW + 2; ++i) {
                const long long ro = (long long)clampi(j.rs + i - 1, 0, R - 1) * C;
                w[i] = *reinterpret_cast<const float4*>(W + ro + j.c0);
                wh[i] = W[ro + wh_col];
            }
        })) return;
    if (st != 0) {
        if (WG_ID == 0 && threadIdx.x == 0) report(a.res, st, 0.0f);  // Watermark.cpp:246-247
        return;
    }
    float nc[8];  // negated coefficients of residual4 / residual1
#pragma unroll
    for (int k = 0; k < 8; ++k) nc[k] = -c[k];
    // ---- one pass over the wave's rows, k_detect's rolling scheme with the x rows coming from the LDS tile: step ii
    // produces e_w and u of row i = ii - 1 (rows -1 and RPW are the neighbours' -- recomputed here, 2 of RPW + 2, so that
    // u never has to be exchanged), then emits e_u of row ii - 2 from the three newest u rows
    float dot = 0.0f, nu = 0.0f, nw = 0.0f;
    if (j.nv > 0) {
        const bool edge_l = j.c0s == 0, edge_r = j.c0s + STRIP >= C;
        const bool top_rep = j.wave == 0 && j.r0 == 0;      // u(-1) := u(0)   (the reference pads u, not x, for e_u)
        const bool bot_rep = j.last_active && j.rend == R;  // u(R) := u(R-1)
        float r6[3][6], h3[3][3];
        float uw[3][6], eww[2][4];
        lds_row6(L, j.tl0 - 2, j.lane, r6[0]); lds_halo3(L, j.tl0 - 2, j.lane, h3[0]);
        lds_row6(L, j.tl0 - 1, j.lane, r6[1]); lds_halo3(L, j.tl0 - 1, j.lane, h3[1]);
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) uw[a2][b2] = 0.0f;
#pragma unroll
        for (int ii = 0; ii < RPW + 2; ++ii) {
            lds_row6(L, j.tl0 + ii, j.lane, r6[(ii + 2) % 3]);
            lds_halo3(L, j.tl0 + ii, j.lane, h3[(ii + 2) % 3]);
            const float* up = r6[ii % 3];
            const float* mid = r6[(ii + 1) % 3];
            const float* dn = r6[(ii + 2) % 3];
            float* ew = eww[ii % 2];
            float uu[4];
            float ewn[4];
            residual4<1>(up, mid, dn, nc, ewn);  // (the detector's form of e = x - c.nbrs: wm_device.hpp residual4, as k_detect)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ew[k] = ewn[k];
                const float mv = MASK == 0 ? fabsf(ew[k]) : nvf_3x3(up + k, mid + k, dn + k);
                uu[k] = mv * f4get(w[ii], k);
            }
            float mh;
            if (MASK == 0) mh = fabsf(residual1<1>(h3[ii % 3], h3[(ii + 1) % 3], h3[(ii + 2) % 3], 0, nc));
            else mh = nvf_3x3(h3[ii % 3], h3[(ii + 1) % 3], h3[(ii + 2) % 3]);
            const float uh = mh * wh[ii];
            float* un = uw[ii % 3];
            un[0] = dpp_from_prev(uu[3], edge_l ? uu[0] : uh);  // replicate border of u: u(., -1) := u(., 0)
            un[5] = dpp_from_next(uu[0], edge_r ? uu[3] : uh);  //                        u(., C) := u(., C-1)
            un[1] = uu[0]; un[2] = uu[1]; un[3] = uu[2]; un[4] = uu[3];
            if (ii == 1 && top_rep) {
#pragma unroll
                for (int b2 = 0; b2 < 6; ++b2) uw[0][b2] = un[b2];
            }
            if (ii >= 2) {
                // e_u of own row i = ii - 2: u rows i-1, i, i+1 are slots (ii+1)%3, (ii+2)%3, ii%3; its e_w is eww[(ii+1)%2]
                const int i = ii - 2;
                if (i < j.nv && !(bot_rep && i == j.nv - 1)) {
                    const float* um = uw[(ii + 1) % 3];
                    const float* u0 = uw[(ii + 2) % 3];
                    const float* ewp = eww[(ii + 1) % 2];
                    float eun[4];
                    residual4<1>(um, u0, un, nc, eun);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float eu = eun[k];
                        if (j.ok[k]) {
                            dot = fmaf(eu, ewp[k], dot);
                            nu = fmaf(eu, eu, nu);
                            nw = fmaf(ewp[k], ewp[k], nw);
                        }
                    }
                }
            }
            if (ii >= 1 && ii <= RPW) {
                // the image's last row (own row i = ii - 1 = nv - 1): window (u(R-2), u(R-1), u(R-1))
                if (bot_rep && ii == j.nv) {
                    const float* u0 = uw[(ii + 2) % 3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float eu = residual1<1>(u0, un, un, k, nc);
                        if (j.ok[k]) {
                            dot = fmaf(eu, ew[k], dot);
                            nu = fmaf(eu, eu, nu);
                            nw = fmaf(ew[k], ew[k], nw);
                        }
                    }
                }
            }
        }
    }
    const double d0 = wave_sum((double)dot), d1 = wave_sum((double)nu), d2 = wave_sum((double)nw);
    if (j.lane == 0) { L.wred[j.wave] = d0; L.wred[FW + j.wave] = d1; L.wred[2 * FW + j.wave] = d2; }
    __syncthreads();
    FSTAMP(a, 4);
    // the workgroup's three sums as {epoch, half} granule pairs; ONE workgroup polls them, folds and reports to the host (no
    // drained stores, no tickets, no separate round of fold loads: see the statistics hand-off of k_fused_embed)
    if (threadIdx.x < 3) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < FW; ++q) s += L.wred[threadIdx.x * FW + q];
        put_pair(a.gcorr + 8LL * WG_ID + 2 * threadIdx.x, a.epoch, s);  // [G][8] granules, 6 used
    }
    FSTAMP(a, 5);
    if (WG_ID != a.folder) return;
    // corr = (float)dot / (float)(||e_w|| ||e_u||)   (Watermark.cpp:230)
    if (threadIdx.x < WAVE) {
        const int l = threadIdx.x;
        double v[4][3];
        unsigned pend = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u][0] = v[u][1] = v[u][2] = 0.0;
            if (l + u * WAVE < a.G) pend |= 1u << u;  // (the grid never exceeds 256 workgroups)
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool timed_out = false;
        while (__any(pend != 0u)) {
            unsigned long long g[4][6];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pend >> u & 1u) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) g[u][k] = ld_agent(a.gcorr + 8LL * (l + u * WAVE) + k);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pend >> u & 1u) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 6; ++k) ok = ok && (unsigned)(g[u][k] >> 32) == a.epoch;
                    if (ok) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) v[u][k] = __longlong_as_double((long long)((g[u][2 * k + 1] << 32) | (g[u][2 * k] & 0xffffffffull)));
                        pend &= ~(1u << u);
                    }
                }
            if (!__any(pend != 0u)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_WAIT_TICKS) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0 += v[u][0]; a1 += v[u][1]; a2 += v[u][2]; }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (l == 0 && !timed_out) report(a.res, 0, (float)a0 / (float)(sqrt(a2) * sqrt(a1)));  // (a time-out leaves the record alone: the host takes the sweeps)
    }
    if (!RELOAD) FSTAMP(a, 7);
}

template <typename T, int MASK, int RPW>
__global__ __launch_bounds__(fw_of(RPW) * WAVE) void k_fused_detect(const T* __restrict__ x, long long pitch, const float* __restrict__ W,
                                                         FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsView L = carve<RPW>(smem);
    const FJob j = make_fjob<RPW>(a);
    fused_detect_body<T, MASK, RPW, false>(x, pitch, W, a, L, j);
}

// =================================================================================================
// k_fused_pair: makeWatermark and detectWatermark of its result (wm_embed_detect on one image) in ONE launch.  The embed half is
// k_fused_embed's; y stays in the LDS tile; when every workgroup's y stores are at the memory side (pair_barrier) the detector
// half takes its halo rows and columns from memory, its own rows from LDS, and runs on a second set of records, counters and
// granules (a2).  Grey output; results are those of the two launches bit for bit (the same arithmetic on the same values).
// =================================================================================================
// The detector half's records, arrival counters, coefficient granules and result record are the second halves / the next
// element of the embed half's (FusedScratch; the host queues the two result records next to each other), its epoch the next one:
// derived here rather than passed, the kernel's scalar state is tight as it is
template <typename T, int MASK, int RPW, bool BX>
__global__ __launch_bounds__(fw_of(RPW) * WAVE) void k_fused_pair(const T* __restrict__ x, long long pitch, const float* __restrict__ W,
                                                       PlaneDesc base, PlaneDesc out, FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsView L = carve<RPW>(smem);
    const FJob j = make_fjob<RPW>(a);
    if (!fused_embed_body<T, T, 1, MASK, RPW, BX, true>(x, pitch, W, base, out, a, L, j)) return;
    FusedArgs a2 = a;
    a2.pmain = a.pmain + (long long)a.G * FNT;
    a2.gran = a.gran + 32;
    a2.cnt = a.cnt + (NSH + 1) * CNT_STRIDE;
    a2.res = a.res + 1;
    a2.epoch = a.epoch + 1u == 0u ? 1u : a.epoch + 1u;
    fused_detect_body<T, MASK, RPW, true>(static_cast<const T*>(out.p), out.pitch, W, a2, L, j);
}

// ---- launchers -------------------------------------------------------------------------------------------------------
template <typename K>
static hipError_t fused_attr(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static FusedArgs fused_args(const FusedGeom& fg, const FusedScratch& sc, unsigned epoch, float sF, double sqrt_n, OpResult* res)
{
    FusedArgs a;
    a.rows = fg.rows; a.cols = fg.cols; a.nstrips = fg.nstrips; a.nbands = fg.nbands; a.th = fg.th; a.G = fg.G;
    a.nfull_rows = 5;
    a.cpr = (fg.cols + 2 + WAVE - 1) / WAVE;
    a.rpc = (fg.rows - 3 + WAVE - 1) / WAVE;
    a.nchunks = a.nfull_rows * a.cpr + 6 * a.rpc;
    a.nbw = fg.nbw; a.nbc_base = a.nchunks / a.nbw; a.nbc_rem = a.nchunks % a.nbw;
    a.bx0 = fg.bx0; a.bx1 = fg.bx1; a.bn0 = fg.bn0;
    a.inv_cpr = div_magic(a.cpr); a.inv_rpc = div_magic(a.rpc);
    a.epoch = epoch; a.sF = sF; a.sqrt_n = sqrt_n;
    a.pmain = sc.pmain; a.gstat = sc.gstat; a.gcorr = sc.gcorr; a.gdone = sc.gdone; a.folder = fg.folder;
    a.gran = sc.gran; a.cnt = sc.cnt;
    a.res = res; a.stamps = sc.stamps; a.dbg = sc.dbg;
    return a;
}

// geometry of the fused path, or fusable = 0: one workgroup per CU at most, tiles of 256 columns x th <= 16 * RPW rows
FusedGeom fused_geometry(int rows, int cols, int ncu)
{
    FusedGeom fg{};
    fg.rows = rows; fg.cols = cols;
    if (cols < STRIP || rows < 4 || ncu < 1) return fg;  // (cols % 4 != 0: f32 planes only, the callers check)
    fg.nstrips = (cols + STRIP - 1) / STRIP;
    const int bands_max = (ncu < FUSED_MAX_WG ? ncu : FUSED_MAX_WG) / fg.nstrips;
    if (bands_max < 1) return fg;
    fg.th = (rows + bands_max - 1) / bands_max;
    fg.rpw = fg.th <= 64 ? 4 : 8;
    if (fg.th > 128) return fg;
    fg.nbands = (rows + fg.th - 1) / fg.th;
    fg.G = fg.nstrips * fg.nbands;
    if (fg.G > FUSED_MAX_WG) return fg;  // (a device with more CUs than the folding wavefronts cover: cap the row bands instead)
    const int nchunks = 5 * ((cols + 2 + WAVE - 1) / WAVE) + 6 * ((rows - 3 + WAVE - 1) / WAVE);
    // the workgroup that folds the statistics of an embed: on the XCD that starts first, in the last row band (the shortest
    // tiles); any workgroup would do
    fg.folder = 0;
    for (int id = fg.G - 1; id >= 0; --id)
        if ((id & 7) == 1) { fg.folder = id; break; }
    fg.nbw = (nchunks + CHUNKS_PER_BORDER_WG - 1) / CHUNKS_PER_BORDER_WG;  // border workgroups
    if (fg.nbw > fg.G) fg.nbw = fg.G;
    // ... on the two XCDs that start first, when they hold that many workgroups (XCD k holds (G - k + 7) / 8)
    fg.bx0 = -1; fg.bx1 = -1; fg.bn0 = 0;
    {
        const int n1 = (fg.G - 1 + 7) / 8, n2 = (fg.G - 2 + 7) / 8;
        if (fg.G >= 16 && fg.nbw <= n1 + n2) { fg.bx0 = 1; fg.bx1 = 2; fg.bn0 = n1; }
    }
    if ((nchunks + fg.nbw - 1) / fg.nbw > fw_of(fg.rpw)) return fg;  // border chunks per workgroup: one per wavefront at most
    fg.fusable = 1;
    return fg;
}

// (the dynamic-LDS limit is a per-device attribute of the function: set once per device and kernel instance)
#define FUSED_LAUNCH(KERNEL, RPWV, ...)                                                                                       \
    do {                                                                                                                      \
        static std::atomic<unsigned long long> attr_done{0};                                                                  \
        int dev_ = 0;                                                                                                         \
        if (hipGetDevice(&dev_) != hipSuccess) return -1;                                                                     \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                                  \
        if (!(attr_done.load(std::memory_order_acquire) & bit_)) {                                                            \
            if (fused_attr(KERNEL, FTile<RPWV>::BYTES) != hipSuccess) return -1;                                              \
            attr_done.fetch_or(bit_, std::memory_order_release);                                                              \
        }                                                                                                                     \
        WM_KLAUNCH(KERNEL, dim3(fg.nstrips, fg.nbands), dim3(fw_of(RPWV) * WAVE), FTile<RPWV>::BYTES, s, __VA_ARGS__);                 \
    } while (0)

template <typename T, typename TB, int NCH, bool BX>
static int launch_fused_embed_t(hipStream_t s, const FusedGeom& fg, int mask, const PlaneDesc& x, const float* W,
                                const PlaneDesc& base, const PlaneDesc& out, const FusedArgs& a)
{
    if (mask == 0) {
        if (fg.rpw == 4) FUSED_LAUNCH((k_fused_embed<T, TB, NCH, 0, 4, BX>), 4, (const T*)x.p, x.pitch, W, base, out, a);
        else FUSED_LAUNCH((k_fused_embed<T, TB, NCH, 0, 8, BX>), 8, (const T*)x.p, x.pitch, W, base, out, a);
    } else {
        if (fg.rpw == 4) FUSED_LAUNCH((k_fused_embed<T, TB, NCH, 1, 4, BX>), 4, (const T*)x.p, x.pitch, W, base, out, a);
        else FUSED_LAUNCH((k_fused_embed<T, TB, NCH, 1, 8, BX>), 8, (const T*)x.p, x.pitch, W, base, out, a);
    }
    return 0;
}

int launch_fused_embed(hipStream_t s, const FusedGeom& fg, const FusedScratch& sc, unsigned epoch, int mask, const PlaneDesc& x,
                       const float* W, const PlaneDesc& base, const PlaneDesc& out, float sF, double sqrt_n, OpResult* res)
{
    const FusedArgs a = fused_args(fg, sc, epoch, sF, sqrt_n, res);
    const bool bx = base.channels == 1 && base.dtype == x.dtype && base.p == x.p && base.pitch == x.pitch;
    if (x.dtype == 0 && base.dtype == 0) {
        if (bx) return launch_fused_embed_t<float, float, 1, true>(s, fg, mask, x, W, base, out, a);
        if (base.channels == 3) return launch_fused_embed_t<float, float, 3, false>(s, fg, mask, x, W, base, out, a);
        return launch_fused_embed_t<float, float, 1, false>(s, fg, mask, x, W, base, out, a);
    }
    if (x.dtype == 1 && base.dtype == 1) {
        if (bx) return launch_fused_embed_t<uint8_t, uint8_t, 1, true>(s, fg, mask, x, W, base, out, a);
        if (base.channels == 3) return launch_fused_embed_t<uint8_t, uint8_t, 3, false>(s, fg, mask, x, W, base, out, a);
        return launch_fused_embed_t<uint8_t, uint8_t, 1, false>(s, fg, mask, x, W, base, out, a);
    }
    return -1;
}

template <typename T>
static int launch_fused_detect_t(hipStream_t s, const FusedGeom& fg, int mask, const PlaneDesc& x, const float* W, const FusedArgs& a)
{
    if (mask == 0) {
        if (fg.rpw == 4) FUSED_LAUNCH((k_fused_detect<T, 0, 4>), 4, (const T*)x.p, x.pitch, W, a);
        else FUSED_LAUNCH((k_fused_detect<T, 0, 8>), 8, (const T*)x.p, x.pitch, W, a);
    } else {
        if (fg.rpw == 4) FUSED_LAUNCH((k_fused_detect<T, 1, 4>), 4, (const T*)x.p, x.pitch, W, a);
        else FUSED_LAUNCH((k_fused_detect<T, 1, 8>), 8, (const T*)x.p, x.pitch, W, a);
    }
    return 0;
}

int launch_fused_detect(hipStream_t s, const FusedGeom& fg, const FusedScratch& sc, unsigned epoch, int mask, const PlaneDesc& x,
                        const float* W, OpResult* res)
{
    const FusedArgs a = fused_args(fg, sc, epoch, 0.0f, 0.0, res);
    if (x.dtype == 0) return launch_fused_detect_t<float>(s, fg, mask, x, W, a);
    return launch_fused_detect_t<uint8_t>(s, fg, mask, x, W, a);
}

template <typename T, bool BX>
static int launch_fused_pair_t(hipStream_t s, const FusedGeom& fg, int mask, const PlaneDesc& x, const float* W, const PlaneDesc& base,
                               const PlaneDesc& out, const FusedArgs& a)
{
    if (mask == 0) {
        if (fg.rpw == 4) FUSED_LAUNCH((k_fused_pair<T, 0, 4, BX>), 4, (const T*)x.p, x.pitch, W, base, out, a);
        else FUSED_LAUNCH((k_fused_pair<T, 0, 8, BX>), 8, (const T*)x.p, x.pitch, W, base, out, a);
    } else {
        if (fg.rpw == 4) FUSED_LAUNCH((k_fused_pair<T, 1, 4, BX>), 4, (const T*)x.p, x.pitch, W, base, out, a);
        else FUSED_LAUNCH((k_fused_pair<T, 1, 8, BX>), 8, (const T*)x.p, x.pitch, W, base, out, a);
    }
    return 0;
}

// -2: the combination of planes is not one the pair kernel takes (the caller launches the two kernels instead)
int launch_fused_pair(hipStream_t s, const FusedGeom& fg, const FusedScratch& sc, unsigned epoch_embed, unsigned epoch_detect, int mask,
                      const PlaneDesc& x, const float* W, const PlaneDesc& base, const PlaneDesc& out, float sF, double sqrt_n,
                      OpResult* res_embed, OpResult* res_detect)
{
    if (base.channels != 1 || out.channels != 1 || x.dtype != base.dtype || x.dtype != out.dtype) return -2;
    const FusedArgs a = fused_args(fg, sc, epoch_embed, sF, sqrt_n, res_embed);
    // the detector half: records, arrival counters and coefficient granules of its own (FusedScratch: second halves)
    // (the kernel derives the detector half's scratch and record from the embed half's: see k_fused_pair)
    if (res_detect != res_embed + 1 || epoch_detect != (epoch_embed + 1u == 0u ? 1u : epoch_embed + 1u)) return -2;
    const bool bx = base.p == x.p && base.pitch == x.pitch;
    if (x.dtype == 0) return bx ? launch_fused_pair_t<float, true>(s, fg, mask, x, W, base, out, a) : launch_fused_pair_t<float, false>(s, fg, mask, x, W, base, out, a);
    return bx ? launch_fused_pair_t<uint8_t, true>(s, fg, mask, x, W, base, out, a) : launch_fused_pair_t<uint8_t, false>(s, fg, mask, x, W, base, out, a);
}

}  // namespace wmk
