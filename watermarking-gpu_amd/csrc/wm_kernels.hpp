// wm_kernels.hpp -- host-visible launch interface of the HIP kernels (internal to libwm_hip.so)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace wmk {

// Per-kernel timing (wm_prof_*): the events of a profiled launch are attached to the dispatch itself (hipExtLaunchKernelGGL's
// start / stop events = the begin and end time stamps of that dispatch, what rocprofv3's kernel trace reports).  Events
// recorded around a launch (hipEventRecord before / after) also time the marker packets and the dispatch gap between two
// dependent kernels: 5-25 us on top of a ~110 us kernel, different for every kernel.  The launcher functions do not know
// about profiling: wm_api.hip's ProfScope parks itself in a thread-local slot, every launch made through WM_KLAUNCH while it
// is set draws a pair of events from it.
struct LaunchProf {
    static constexpr int MAX = 4;          // launches of one sweep at most (aligned part + generic remainder, march + border ...)
    hipEvent_t a[MAX], b[MAX];
    int n = 0;
    hipEvent_t (*get)(void*) = nullptr;    // hands out an event (the context's pool)
    void* owner = nullptr;
};
LaunchProf*& launch_prof_slot();  // (thread-local; defined in wm_api.hip)
// every launch made while a scope is set gets its own start / stop pair (a sweep may be two launches: the aligned strips and
// the generic remainder; their durations are summed into one call of that kernel)
#define WM_KLAUNCH(KERNEL, GRID, BLOCKDIM, SHMEM, STREAM, ...)                                                         \
    do {                                                                                                               \
        ::wmk::LaunchProf* lp_ = ::wmk::launch_prof_slot();                                                            \
        if (lp_ && lp_->n < ::wmk::LaunchProf::MAX) {                                                                  \
            const int i_ = lp_->n++;                                                                                   \
            lp_->a[i_] = lp_->get(lp_->owner); lp_->b[i_] = lp_->get(lp_->owner);                                      \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, SHMEM, STREAM, lp_->a[i_], lp_->b[i_], 0, __VA_ARGS__);      \
        } else {                                                                                                       \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCKDIM, SHMEM, STREAM, __VA_ARGS__);                                    \
        }                                                                                                              \
    } while (0)

constexpr int NGRAM = 44;  // 36 unique Rx entries + 8 rx entries (me_p3.hpp:8-21)
// Ticket counters of the fold tails ("the last block / wave finishes") sit one per 128-byte line: TKS words apart.  Packed,
// the 16 frame counters of a launch shared ONE line and the 256 strip counters eight: every arrival of a launch (2 880 blocks
// in k_gram, 11 520 waves in k_detect, each a returning atomic the wave waits for before it may leave) queued on the same
// few lines, and how long that queue was depended on where the allocation happened to land.
constexpr int TKS = 32;

struct PlaneDesc {
    const void* p;
    long long pitch;    // elements
    long long fstride;  // elements between frames
    long long cstride;  // elements between channels
    int dtype;          // 0 f32, 1 u8
    int channels;
    int aligned;        // base/pitch/strides allow 4-pixel vector access
};

struct LaunchGeom {
    int rows, cols;
    int nstrips, nfull, nsegs, rps;  // nfull = strips lying fully inside the image (cols / 256)
    int nblk;  // strip-march blocks per frame (all strips)
    int nbb;   // blocks per frame of k_gram_border
    // rows whose pixels this launch owns: sums and stores cover [row_lo, row_hi) only.  The whole image unless the
    // context works on a row band of a larger image (intra-frame sharding, wm_band_*): then the plane is the band plus
    // its halo rows, row_lo == 0 / row_hi == rows mark the sides that are true image borders
    int row_lo, row_hi;
};

struct EmbedScalars {
    float a;     // watermark strength (Watermark.cpp:170)
    float maxe;  // max|e| (ME) or 1
};

// raw totals of the stats and detect sweeps, written by the fold tails beside the finished scalars: what a row band of a
// sharded image contributes to the all-reduce (wm_band_*)
struct RawSums {
    double v[4];  // stats: {max|e| (or 1), sum (m W)^2, -, -}; detect: {<e_u,e_w>, ||e_u||^2, ||e_w||^2, -}
};

struct alignas(8) OpResult {  // (8 bytes, naturally aligned: the fused kernels deliver it with one store)
    int status;   // 0 OK, 1 unsolvable
    float value;  // a (embed) or correlation (detect)
};

// Fold tails ("the last block finishes", wm_device.hpp): what the block that arrives last does with the partials.
struct SolveTail {       // k_gram: fold Gram partials, solve the 8x8 system (Watermark.cpp:203)
    unsigned* ticket;    // [frames] zero between ops
    int expected;        // blocks per frame over all launches of the sweep (march blocks + border blocks)
    int nbb_total;       // border partial records per frame
    float* coef;         // [frames][8]
    int* status;         // [frames]
    double* gram_tot;    // [frames][44]
};
struct ScalarsTail {     // k_me_stats / k_nvf_stats: a = sF / (float)(||u|| / sqrt(N))   (Watermark.cpp:170)
    unsigned* ticket;        // [frames] frame-level tickets: blocks of the frame, or (Geom::quad) its strips
    unsigned* ticket_strip;  // quad: [frames][nstrips] strip-level tickets (count the waves = segments of a strip)
    int expected;            // blocks per frame over all launches of the sweep (block-level fold)
    int nsegs, nstrips;
    float* smax;             // [frames][nstrips] strip records: max|e| ...
    double* sss;             // ... and sum
    float sF;
    double sqrt_n;
    EmbedScalars* scal;
    OpResult* res;
    RawSums* raw;        // [frames]
};
struct CorrTail {        // k_detect: corr = (float)dot / (float)(||e_w|| ||e_u||)   (Watermark.cpp:230)
    unsigned* ticket;
    unsigned* ticket_strip;
    int expected;
    int nsegs, nstrips;
    double* scorr;           // [frames][nstrips][3] strip records
    OpResult* res;
    RawSums* raw;        // [frames]
};

// ---- fused single-frame path (wm_k_fused.hip): one launch per operation, the frame's tiles stay in LDS ----------------
struct FusedGeom {
    int rows, cols;
    int nstrips, nbands, th;  // tiles of 256 columns x th rows; grid = nstrips * nbands workgroups (<= CU count)
    int rpw;                  // rows per wavefront: 4 (th <= 64) or 8 (th <= 128) with 16 wavefronts; 16 with 8 (experiments)
    int G;
    int nbw;                  // workgroups that take the Gram matrix's border chunks
    int folder;               // workgroup that folds the statistics of an embed
    int bx0, bx1, bn0;        // ... those of XCDs bx0, bx1 (bn0 = workgroups on bx0), or bx0 < 0: the first nbw
    int fusable;              // 0: shape not supported by the fused kernels (the caller takes the streaming kernels)
};
struct FusedScratch {        // per slot, device memory (one allocation; layout in wm_api.hip)
    double* pmain;            // 2 x ([13][G] + [44][nbw])  workgroup records of the Gram phase (term-major); second half: k_fused_pair's detector
    unsigned long long* gstat;  // [G][4]  statistics of an embed as {epoch, 32 bits} granules
    unsigned long long* gcorr;  // [G][8]  the detector's sums as {epoch, half} granule pairs
    unsigned long long* gdone;  // [G]     end-of-embed flags
    unsigned long long* gran; // 2 x [32] published {epoch, value} granules (second half: k_fused_pair's detector)
    unsigned* cnt;            // [27][32] arrival counters, one per 128-byte line (zero between calls)
    unsigned long long* stamps;  // [G][16] phase time stamps (development aid) or null
    int dbg;                     // development switches of the fused kernels (timing experiments; 0 in production)
};
constexpr size_t FUSED_CNT_BYTES = 27 * 128;
constexpr int FUSED_MAX_WG = 256;        // workgroups of a fused grid at most: what the folding wavefronts cover (4 per lane)
constexpr int FUSED_INCOMPLETE = -98;    // result-record status: output stores were issued but the end of the frame was not observed
FusedGeom fused_geometry(int rows, int cols, int ncu);
// return 0 when the launch was issued (errors of the launch itself surface through hipGetLastError)
// embed + detect of its output in one launch (grey planes of one type): -2 = planes the pair kernel does not take
int launch_fused_pair(hipStream_t s, const FusedGeom& fg, const FusedScratch& sc, unsigned epoch_embed, unsigned epoch_detect, int mask,
                      const PlaneDesc& x, const float* W, const PlaneDesc& base, const PlaneDesc& out, float sF, double sqrt_n,
                      OpResult* res_embed, OpResult* res_detect);
int launch_fused_embed(hipStream_t s, const FusedGeom& fg, const FusedScratch& sc, unsigned epoch, int mask, const PlaneDesc& x,
                       const float* W, const PlaneDesc& base, const PlaneDesc& out, float sF, double sqrt_n, OpResult* res);
int launch_fused_detect(hipStream_t s, const FusedGeom& fg, const FusedScratch& sc, unsigned epoch, int mask, const PlaneDesc& x,
                        const float* W, OpResult* res);

void launch_gram(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, double* pmain, double* pborder,
                 unsigned* ticket, float* coef, int* status, double* gram_tot);
void launch_me_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                     const float* coef, const int* status, float* pmax, double* pss, unsigned* ticket, unsigned* ticket_strip,
                     float* smax, double* sss, float sF, double sqrt_n, EmbedScalars* scal, OpResult* res, RawSums* raw);
void launch_nvf_stats(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& x, const float* W, int aligned_w,
                      int pad, double* pss, unsigned* ticket, unsigned* ticket_strip, double* sss, float sF, double sqrt_n,
                      EmbedScalars* scal, OpResult* res, RawSums* raw);
// Gram hand-over from an embed to the detector that reads its output (wm_set_handover, WM_MEM_SLOT_OUT): k_embed holds every
// row of y in registers, so it also accumulates y's 13 lag sums over the products that stay inside a wave's tile (its strip's
// columns x its segment's rows and the two rows behind them) and leaves one record per wave, plus a compact copy of the two
// columns on either side of every strip boundary; k_gram_ho adds the products across strip boundaries from that copy, the
// border frame and the solve -- the detector's Gram sweep over y is not run.
//   rec:  [frames][stride][13]            wave records at [0, nstrips * nsegs), k_gram_ho's seam-block records behind them
//   seam: [frames][nstrips - 1][rows][4]  y at columns S-2, S-1, S, S+1 of the boundary S in front of strip k (k = 1 ..)
struct HandOver {
    double* rec;
    int stride;
    float* seam;
};
int handover_seam_blocks(const LaunchGeom& lg);   // seam blocks per frame of k_gram_ho for this geometry
// returns true when the hand-over instantiation ran (f32 grey planes on the aligned path, p = 3, segments of >= 2 rows)
bool launch_embed(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                  int aligned_w, const PlaneDesc& base, const PlaneDesc& out, const float* coef, const int* status,
                  const EmbedScalars* scal, const HandOver* ho = nullptr);
// lg: the geometry of the embed that left the wave records
void launch_gram_ho(hipStream_t s, const LaunchGeom& lg, int frames, const PlaneDesc& y, const HandOver& ho, double* pborder,
                    unsigned* ticket, float* coef, int* status, double* gram_tot);
void launch_mask(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* coef,
                 const int* status, const EmbedScalars* scal, const PlaneDesc& mo, const PlaneDesc& eo);
void launch_detect(hipStream_t s, const LaunchGeom& lg, int frames, int mask, int pad, const PlaneDesc& x, const float* W,
                   int aligned_w, const float* coef, const int* status, double* pcorr, unsigned* ticket, unsigned* ticket_strip,
                   double* scorr, OpResult* res, RawSums* raw);
// band mode: solve the 8x8 system from all-reduced Gram totals [frames][44]; writes coef / status like k_gram's tail
void launch_solve_totals(hipStream_t s, int frames, const double* totals, float* coef, int* status);
// band mode, device-resident exchange (wm_band_*_dev): glue kernels between the sweeps and the caller's collectives
void launch_band_pick(hipStream_t s, int frames, const RawSums* raw, int n, double* out);
void launch_band_scalars(hipStream_t s, int frames, const double* parts, int nparts, int mask, float sF, double sqrt_n, const int* status,
                         EmbedScalars* scal, float* a_dev);
void launch_band_corr(hipStream_t s, int frames, const double* sums, const int* status, float* corr);
// W generated on the device (wm_create_generated): same values as csrc/app/wm_genw.cpp writes
void launch_gen_w(hipStream_t s, float* w, int rows, int cols, uint32_t seed);
// exhaustive check of the NVF quotient sequence against the IEEE division (wm_selftest_nvf_quotient)
void launch_selftest_quot(hipStream_t s, int variant, uint32_t bits_lo, uint32_t bits_hi, unsigned long long* out2);
// memory-system yardstick (wm_membench): kind 0 store, 1 copy, 2 read; n16 = 16-byte elements
void launch_membench(hipStream_t s, int kind, const void* src, void* dst, size_t n16, unsigned long long* sink, hipEvent_t a, hipEvent_t b);
void launch_mask_result(hipStream_t s, int frames, const int* status, const float* coef, OpResult* res, float* coef_out);

}  // namespace wmk
