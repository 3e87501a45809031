// wm_api.hip -- the C ABI of include/wm.h: context, slots, staging, launch sequencing.
//
// No CPU fallback exists here by design: without a HIP device wm_create() fails with
// WM_ERR_NO_DEVICE.  (The CPU oracle lives in oracle/ and is test infrastructure only.)
#include "../../include/wm.h"
#include "wm_kernels.hpp"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace wmk;

namespace {

constexpr int RES_CAP = 4096;  // result records a slot can hold between two wm_sync calls
// wavefronts a launch should have at least.  Measured at 4K with one frame per launch: the ME sweeps are fastest with ~1 wave
// per SIMD (segments of 24-32 rows), the NVF sweeps (three times the arithmetic per pixel) with ~2 (16 rows)
constexpr int TARGET_WAVES_ME = 1280;
constexpr int TARGET_WAVES_NVF = 2048;

// the fold steps (solve, embed scalars, correlation) are tails of k_gram / k_*_stats / k_detect: no kernels of their own
enum KernelId { K_GRAM = 0, K_ME_STATS, K_NVF_STATS, K_EMBED, K_DETECT, K_MASK, K_FUSED_EMBED, K_FUSED_DETECT, K_GRAM_HO, K_FUSED_PAIR, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"k_gram", "k_me_stats", "k_nvf_stats", "k_embed", "k_detect", "k_mask", "k_fused_embed", "k_fused_detect", "k_gram_ho", "k_fused_pair"};

// fused single-frame launches use every CU and wait for each other inside the launch: two of them in flight on one device
// could each hold a part of the CUs and starve the other (their spins are bounded, so that would be a slow fallback, not a
// hang).  Synchronous calls hold this lock from launch to completion: a mutex between the threads of this process and an
// advisory file lock (flock on a per-device file named after the device's PCI address, so the same GPU has the same name
// under any HIP_VISIBLE_DEVICES) between processes that share the device.
constexpr int MAX_DEVICES = 64;
std::mutex g_fused_mu[MAX_DEVICES];
constexpr int FUSED_PENDING = -99;  // result record status while a fused launch has not delivered
constexpr int PAIR_RETRY = 100;     // internal: the fused pair of wm_embed_detect did not complete, take the sweeps

struct FusedGuard {
    std::lock_guard<std::mutex> lk;
    int fd;
    bool ok = true;  // false: another process kept the device's lock beyond the deadline -- the caller takes the sweeps
    FusedGuard(int device, int lock_fd) : lk(g_fused_mu[device % MAX_DEVICES]), fd(lock_fd)
    {
        if (fd < 0) return;
        // a holder is normally gone within one call (~30 us); one that is stopped (a debugger, SIGSTOP) must not hang every other
        // process's synchronous calls: non-blocking attempts up to a deadline, then the call proceeds on the sweeps
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            if (flock(fd, LOCK_EX | LOCK_NB) == 0) return;
            if (errno != EWOULDBLOCK && errno != EINTR) break;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        }
        ok = false; fd = -1;
    }
    ~FusedGuard() { if (fd >= 0) (void)flock(fd, LOCK_UN); }
};

struct WShared {
    float* d_w = nullptr;
    size_t n = 0;
    ~WShared() { if (d_w) (void)hipFree(d_w); }
};

struct Pending {
    bool keep_value_when_unsolvable = false;  // embed: `a` stays untouched (Watermark.cpp:164-165)
    int frames;
    int res_off;
    float* value_out;
    int* status_out;
    float* coef_out;  // host destination for 8*frames coefficients (mask-only ops)
    int coef_off;
};

struct Slot {
    hipStream_t own = nullptr, stream = nullptr;
    void* arena = nullptr;  // the one device allocation behind the scratch arrays below (alloc_slots)
    // scratch (sized for max_frames and the worst-case block count)
    double* d_gram = nullptr;    // k_gram main partials [frames][nblk][13]
    double* d_gramb = nullptr;   // k_gram border partials [frames][nbb][44]
    double* d_gramtot = nullptr; // folded sums [frames][44]
    float* d_coef = nullptr;
    int* d_status = nullptr;
    float* d_pmax = nullptr;
    double* d_pss = nullptr;
    double* d_pcorr = nullptr;
    EmbedScalars* d_scal = nullptr;
    float* d_smax = nullptr;       // [max_frames][nstrips] strip records of the stats sweep
    double* d_sss = nullptr;
    double* d_scorr = nullptr;     // [max_frames][nstrips][3] strip records of the detect sweep
    RawSums* d_raw = nullptr;      // [2][max_frames] raw totals of the stats / detect sweeps (band mode reads them)
    double* d_totals = nullptr;    // [max_frames][44] all-reduced Gram totals handed back by wm_band_solve
    unsigned* d_ticket = nullptr;  // [3][max_frames] last-block tickets of the Gram, stats and detect sweeps (zero between ops)
    // result records live in pinned, device-mapped host memory: the finalising kernels store them straight over
    // PCIe, so a call needs no D2H copy node and wm_sync only waits for the stream
    OpResult* h_res = nullptr;   // host view
    OpResult* d_res = nullptr;   // device view of the same memory
    float* h_coefres = nullptr;
    float* d_coefres = nullptr;
    int res_used = 0;
    std::deque<Pending> pending;
    // the device copy of the last embed's output on this slot (WM_MEM_SLOT_OUT planes name it)
    PlaneDesc last_out{};
    int last_out_frames = 0, last_out_dtype = 0;
    // fused single-frame path (wm_k_fused.hip)
    FusedScratch fz{};
    void* fz_block = nullptr;  // one allocation behind fz
    unsigned fz_epoch = 0;
    // Gram hand-over (wm_set_handover): wave + seam records [max_frames][ho_stride_max][13]; ho.valid: the last embed on this
    // slot left the tile-internal lag sums of its output (= last_out) there, for the geometry ho.lg
    double* d_ho = nullptr;
    float* d_hoseam = nullptr;   // [max_frames][strips - 1][rows][4]: the columns at the strip boundaries (HandOver::seam)
    struct HoInfo { bool valid = false; LaunchGeom lg{}; int frames = 0; int stride = 0; } ho;
    // wm_embed_detect: a fused embed whose wait was deferred to the detector's record (the two launches go out back to back)
    struct PairEmbed {
        bool armed = false; int res_index = 0; bool host_out = false; bool out_overlaps_inputs = false;
        // one launch for both halves (k_fused_pair): the embed was NOT launched -- the detector's call launches both
        bool deferred = false; int mask = 0; unsigned epoch = 0; PlaneDesc xd{}, bd{}, od{};
    } pair;
    // staging for WM_MEM_HOST planes
    void* st_in = nullptr; size_t st_in_bytes = 0;
    void* st_base = nullptr; size_t st_base_bytes = 0;
    void* st_out = nullptr; size_t st_out_bytes = 0;
};

struct ProfRec { int kid; hipEvent_t a, b; bool first; };  // first: the sweep's first launch (counts the call)

}  // namespace

struct wm_ctx {
    int device = 0;
    int rows = 0, cols = 0, p = 3;
    float psnr = 0.f, sF = 0.f;
    std::shared_ptr<WShared> w;
    int nslots = 0, max_frames = 1;
    int rps_override = 0;
    int ncu = 0;
    int fused_mode = 1;  // 1: synchronous one-frame calls take the fused kernels when the shape allows (wm_set_fused)
    // wm_embed_detect on one image: 1 = ONE launch for both halves (k_fused_pair).  Off unless WM_FUSED_PAIR=1: measured 1.5-2 us
    // SLOWER than the two launches back to back at 4K, equal at 1080p
    int fused_pair = (getenv("WM_FUSED_PAIR") && getenv("WM_FUSED_PAIR")[0] == '1') ? 1 : 0;
    FusedGeom fg{};
    unsigned long long fused_fallbacks = 0;  // fused launches that timed out and were re-run on the sweeps
    unsigned long long fused_lock_skips = 0; // synchronous calls that took the sweeps because another process held the device's lock
    // after a fallback the fused path is skipped for `fused_backoff` calls (8, doubling up to 4096 while the re-probes keep
    // failing; a probe that succeeds clears it): a device on which the workgroups cannot all be resident -- another
    // process's kernels, a CU mask -- costs one time-out per window, not one per call
    int fused_backoff = 0, fused_skip = 0;
    int handover = 0;        // wm_set_handover
    int handover_verify = (getenv("WM_HANDOVER_VERIFY") && getenv("WM_HANDOVER_VERIFY")[0] == '1') ? 1 : 0;  // debug: re-check every hand-over
    int pair_handover = 0;   // 1 inside wm_embed_detect: its detector reads the embed's output by construction
    int pair_mode = 0;       // 1 inside wm_embed_detect: the fused embed does not wait (and the caller holds the FusedGuard)
    int fused_lock_fd = -1;  // per-device lock file shared with other processes (FusedGuard), -1: none
    int max_nblk = 0, max_nrec = 0;  // per-frame capacity of the slots' partial-record arrays (alloc_slots)
    // row band of a larger image (wm_band_configure): planes are the band plus halo rows, sums and stores cover the owned rows
    int band_lo = 0, band_hi = 0;       // owned rows in plane coordinates; band_hi == 0: no band (the whole plane is owned)
    long long band_rows_global = 0;     // rows of the whole image (the strength needs sqrt(N) of the whole image)
    std::vector<Slot> slots;
    std::string last_error;
    bool prof = false;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_free;
    uint64_t prof_n[K_COUNT] = {0};
    double prof_ms[K_COUNT] = {0};
    ~wm_ctx();  // releases streams, scratch and events (also on the error paths of wm_create / wm_clone)
};

namespace {

// Wait for a fused launch by polling the result record its folding workgroup writes to device-mapped pinned memory (ONE 8-byte
// store: status and value; embed: after every byte of the output has been written through to memory).  The record is read
// with one 64-bit acquire load, so status and value belong together and nothing that follows is read ahead of it.  Returns
// true when the record arrived (copy in *got).  A launch whose hand-off timed out ends WITHOUT writing the record: once
// the call is older than a normal one (150 us) the stream is queried every ~20 us, and a finished stream with the record
// still pending returns false at once (it used to spin for 200 ms).  hipStreamSynchronize costs 5.5 us more per call than
// this poll (tools/ubench/launch_sync.hip), so the normal path never touches the stream.
bool poll_record(const OpResult* rec, int pending, hipStream_t stream, OpResult* got)
{
    static_assert(sizeof(OpResult) == 8 && sizeof(std::atomic<uint64_t>) == 8, "the record is one 8-byte word");
    const std::atomic<uint64_t>* word = reinterpret_cast<const std::atomic<uint64_t>*>(rec);
    auto look = [&]() {
        const uint64_t v = word->load(std::memory_order_acquire);
        std::memcpy(got, &v, 8);
        return got->status != pending;
    };
    const auto t0 = std::chrono::steady_clock::now();
    auto next_query = std::chrono::microseconds(150);
    for (unsigned spins = 0;; ++spins) {
        if (look()) return true;
        if ((spins & 0xff) != 0xff) continue;
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt < next_query) continue;
        next_query = std::chrono::duration_cast<std::chrono::microseconds>(dt) + std::chrono::microseconds(20);
        if (hipStreamQuery(stream) != hipErrorNotReady) return look();  // the launch has ended (or failed): the record is final
        if (dt > std::chrono::milliseconds(500)) return false;             // backstop; the caller synchronises the stream
    }
}

}  // namespace
namespace wmk {
LaunchProf*& launch_prof_slot()
{
    static thread_local LaunchProf* slot = nullptr;
    return slot;
}
}  // namespace wmk
namespace {

int fail(wm_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->last_error = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                           \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, WM_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

int ceil_div(int a, int b) { return (a + b - 1) / b; }
// strips the per-strip record and ticket arrays are sized for: k_detect's overlapped strips are 248 columns apart
// (wm_march.hpp OV_STRIDE), every other sweep's 256
int strips_alloc(int cols) { return ceil_div(cols, 248) + 1; }  // (+1: the generic strip of a width that is not a multiple of 4)
int border_blocks(int rows, int cols, int frames = 1);

// sqrt(N) of Watermark.cpp:170: N counts the pixels of the whole image (a row band knows the image's row count)
double sqrt_n(const wm_ctx* ctx)
{
    const double rows = ctx->band_hi > 0 ? (double)ctx->band_rows_global : (double)ctx->rows;
    return sqrt(rows * (double)ctx->cols);
}

// geometry of one launch: strips of 256 columns, segments of rps rows, 4 segments per block
LaunchGeom make_geom(const wm_ctx* ctx, int frames, int mask = WM_MASK_ME)
{
    const int TARGET_WAVES = mask == WM_MASK_NVF ? TARGET_WAVES_NVF : TARGET_WAVES_ME;
    LaunchGeom lg;
    lg.rows = ctx->rows; lg.cols = ctx->cols;
    lg.nstrips = ceil_div(ctx->cols, 256);
    lg.nfull = ctx->cols / 256;
    lg.row_lo = ctx->band_hi > 0 ? ctx->band_lo : 0;
    lg.row_hi = ctx->band_hi > 0 ? ctx->band_hi : ctx->rows;
    const int owned = lg.row_hi - lg.row_lo;
    int rps = ctx->rps_override;
    if (rps <= 0) {
        // enough wavefronts to fill 256 CUs several times over, but segments long enough to amortise their halo rows
        // (2 of rps for the 3x3 sweeps, 4 of rps for k_detect): 8 .. 48 rows, measured flat from 40 to 64 at 4K
        const long long want = (long long)owned * lg.nstrips * frames;
        rps = (int)((want + TARGET_WAVES - 1) / TARGET_WAVES);
        if (rps < 8) rps = 8;
        if (rps > 48) rps = 48;
        // balance: blocks own 4 segments, so make the segments equal parts of a whole number of blocks
        const int groups = ceil_div(owned, 4 * rps);
        rps = ceil_div(owned, 4 * groups);
        if (rps < 1) rps = 1;
    }
    if (rps > owned) rps = owned;
    lg.rps = rps;
    lg.nsegs = ceil_div(owned, rps);
    lg.nblk = lg.nstrips * ceil_div(lg.nsegs, 4);
    lg.nbb = border_blocks(ctx->rows, ctx->cols, frames);
    return lg;
}

// blocks of 256 threads for the border frame of k_gram: 5 full rows + 6 side columns (or everything for tiny images)
int border_blocks(int rows, int cols, int frames)
{
    const bool core_empty = rows < 4 || cols < 5;
    const long long nfull = core_empty ? rows + 2 : 5;
    const long long cpr = (cols + 2 + 63) / 64;
    const long long rpc = core_empty ? 0 : (rows - 3 + 63) / 64;
    long long nb = (nfull * cpr + 6 * rpc + 7) / 8;  // 2 chunks per wave: the pass is latency-bound, so short waves, many of them
    // ... but every block leaves a 44-sum record that the frame's last block folds before it can solve, and that fold
    // is the exposed tail of the launch: at most 32 blocks per frame (4 chunks per wave at 4K), 16 in batched launches,
    // where the other frames' blocks hide the longer border waves (4K: k_gram 7.6 -> 7.3 us per frame; one frame per
    // launch loses 7 % with 16)
    const long long cap = frames >= 8 ? 16 : 32;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// make_geom + the guarantee the kernels index by: a launch's per-block / per-wave record counts fit the slot's arrays
int geom_checked(wm_ctx* ctx, int frames, int mask, LaunchGeom* lg)
{
    *lg = make_geom(ctx, frames, mask);
    if (lg->nblk > ctx->max_nblk || lg->nstrips * lg->nsegs > ctx->max_nrec || lg->nbb > border_blocks(ctx->rows, ctx->cols))
        return fail(ctx, WM_ERR_RUNTIME, "launch geometry exceeds the slot's partial-record arrays (nblk " + std::to_string(lg->nblk) + "/" +
                                             std::to_string(ctx->max_nblk) + ", nrec " + std::to_string(lg->nstrips * lg->nsegs) + "/" +
                                             std::to_string(ctx->max_nrec) + ")");
    return WM_OK;
}

int worst_nsegs(int rows, int rps_override);
int worst_nblk(int rows, int cols, int rps_override)
{
    return strips_alloc(cols) * ceil_div(worst_nsegs(rows, rps_override), 4);
}

void free_slot(Slot& s)
{
    if (s.own) (void)hipStreamDestroy(s.own);
    (void)hipFree(s.arena);  // (d_gram ... d_ticket point into it)
    if (s.h_res) (void)hipHostFree(s.h_res);
    if (s.h_coefres) (void)hipHostFree(s.h_coefres);
    (void)hipFree(s.st_in); (void)hipFree(s.st_base); (void)hipFree(s.st_out); (void)hipFree(s.fz_block); (void)hipFree(s.d_ho); (void)hipFree(s.d_hoseam);
    s = Slot();
}

// last-block / last-wave tickets of a slot: [3][max_frames] frame-level (Gram, stats, detect) + [2][max_frames][nstrips]
// strip-level (stats, detect)
size_t ticket_words(const wm_ctx* ctx)
{
    return ((size_t)3 * ctx->max_frames + (size_t)2 * ctx->max_frames * strips_alloc(ctx->cols)) * TKS;  // one counter per 128-byte line
}
unsigned* strip_tickets(const wm_ctx* ctx, const Slot& s, int which)  // which: 0 stats, 1 detect
{
    return s.d_ticket + ((size_t)3 * ctx->max_frames + (size_t)which * ctx->max_frames * strips_alloc(ctx->cols)) * TKS;
}

// per-wave partial records of the stats / detect sweeps: strips x segments, for the LARGEST segment count make_geom can
// produce.  make_geom starts from rps0 >= 8 rows and then balances: groups = ceil(owned / (4 rps0)), rps = ceil(owned /
// (4 groups)), which may end below 8, so nsegs = ceil(owned / rps) <= 4 groups <= 4 ceil(rows / 32) (not ceil(rows / 8)).
int worst_nsegs(int rows, int rps_override)
{
    if (rps_override > 0) return ceil_div(rows, rps_override > rows ? rows : rps_override);
    return 4 * ceil_div(rows, 32);
}
int worst_nrec(int rows, int cols, int rps_override) { return strips_alloc(cols) * worst_nsegs(rows, rps_override); }

// Gram hand-over: records per frame of a slot's hand-over array = the wave records + the seam-block records of the largest geometry
int ho_stride_max(const wm_ctx* ctx) { return ctx->max_nrec + 2 * ((ctx->max_nrec + 3) / 4) + 2; }
// both arrays of a slot or neither: a slot with records but no seam array would send k_embed<HO>'s edge lanes to address 0
int ho_alloc(wm_ctx* ctx)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (auto& s : ctx->slots) {
        s.ho.valid = false;
        if (s.d_ho && s.d_hoseam) continue;
        (void)hipFree(s.d_ho); (void)hipFree(s.d_hoseam);
        s.d_ho = nullptr; s.d_hoseam = nullptr;
        void* rec = nullptr; void* seam = nullptr;
        hipError_t e = hipMalloc(&rec, (size_t)ctx->max_frames * ho_stride_max(ctx) * 13 * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&seam, (size_t)ctx->max_frames * ceil_div(ctx->cols, 256) * ctx->rows * 4 * sizeof(float));
        if (e != hipSuccess) {
            (void)hipFree(rec); (void)hipFree(seam);
            (void)hipGetLastError();
            return fail(ctx, WM_ERR_ALLOC, std::string("hand-over arrays: ") + hipGetErrorString(e));
        }
        s.d_ho = (double*)rec; s.d_hoseam = (float*)seam;
    }
    return WM_OK;
}

int alloc_slots(wm_ctx* ctx, int nslots, int max_frames)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (auto& s : ctx->slots) free_slot(s);
    ctx->slots.clear();
    ctx->nslots = nslots; ctx->max_frames = max_frames;
    ctx->max_nblk = worst_nblk(ctx->rows, ctx->cols, ctx->rps_override);
    ctx->max_nrec = worst_nrec(ctx->rows, ctx->cols, ctx->rps_override);
    ctx->slots.resize(nslots);
    const size_t nb = (size_t)ctx->max_nblk * max_frames;
    if (ctx->ncu == 0 && hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) ctx->ncu = 0;
    ctx->fg = fused_geometry(ctx->rows, ctx->cols, ctx->ncu);
    if (ctx->fg.fusable && ctx->fused_lock_fd < 0 && !(getenv("WM_FUSED_XPROC_LOCK") && getenv("WM_FUSED_XPROC_LOCK")[0] == '0')) {
        char bus[64] = "dev";
        if (hipDeviceGetPCIBusId(bus, sizeof bus, ctx->device) != hipSuccess) snprintf(bus, sizeof bus, "ordinal%d", ctx->device);
        for (char* q = bus; *q; ++q) if (*q == ':' || *q == '/' || *q == '.') *q = '_';
        // a lock directory that every user of the machine shares: /run/lock where it exists, else TMPDIR / /tmp.  The file is
        // opened read-only (flock works on read-only descriptors, so a file another user created with any umask can still be
        // locked), never through a symbolic link, and made 0666 by whoever creates it
        const char* dir = getenv("WM_FUSED_LOCK_DIR") ? getenv("WM_FUSED_LOCK_DIR")
                          : access("/run/lock", W_OK | X_OK) == 0 ? "/run/lock" : (getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp");
        const std::string path = std::string(dir) + "/wm_fused_" + bus + ".lock";
        ctx->fused_lock_fd = open(path.c_str(), O_CREAT | O_RDONLY | O_NOFOLLOW | O_CLOEXEC, 0666);  // (-1: no cross-process serialisation, the bounded spins still hold)
        if (ctx->fused_lock_fd >= 0) (void)fchmod(ctx->fused_lock_fd, 0666);  // (fails quietly when the file is another user's: it is 0666 already)
    }
    for (auto& s : ctx->slots) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&s.own, hipStreamNonBlocking));
        s.stream = s.own;
        // The slot's device scratch is ONE allocation (a multiple of 2 MiB, the arrays on 4 KiB boundaries inside it), not a
        // dozen small ones: small hipMallocs are sub-allocated wherever the runtime's pools have room, and where the partial
        // records and ticket counters of the fold tails landed decided 10-20 % of k_gram / k_detect -- the first context a
        // process created ran them in 120 / 130 us, the second and third in 105 / 110 us, same code, same frames
        // (tools/data_probe2.py).  An arena of its own gets its own large, aligned mapping, the same for every context.
        const size_t nr = (size_t)ctx->max_nrec * max_frames;
        const size_t nsr = (size_t)max_frames * strips_alloc(ctx->cols);
        struct Part { void** p; size_t bytes; };
        const Part parts[] = {
            {(void**)&s.d_gram, nb * 13 * sizeof(double)},
            {(void**)&s.d_gramb, (size_t)border_blocks(ctx->rows, ctx->cols) * max_frames * NGRAM * sizeof(double)},
            {(void**)&s.d_gramtot, (size_t)max_frames * NGRAM * sizeof(double)},
            {(void**)&s.d_coef, (size_t)max_frames * 8 * sizeof(float)},
            {(void**)&s.d_status, (size_t)max_frames * sizeof(int)},
            {(void**)&s.d_pmax, nr * sizeof(float)},
            {(void**)&s.d_pss, nr * sizeof(double)},
            {(void**)&s.d_pcorr, nr * 3 * sizeof(double)},
            {(void**)&s.d_scal, (size_t)max_frames * sizeof(EmbedScalars)},
            {(void**)&s.d_smax, nsr * sizeof(float)},
            {(void**)&s.d_sss, nsr * sizeof(double)},
            {(void**)&s.d_scorr, nsr * 3 * sizeof(double)},
            {(void**)&s.d_raw, (size_t)2 * max_frames * sizeof(RawSums)},
            {(void**)&s.d_totals, (size_t)max_frames * NGRAM * sizeof(double)},
            {(void**)&s.d_ticket, ticket_words(ctx) * sizeof(unsigned)},
        };
        auto up = [](size_t v, size_t a) { return (v + a - 1) / a * a; };
        size_t total = 0;
        for (const Part& pt : parts) total += up(pt.bytes, 4096);
        total = up(total, (size_t)2 << 20);
        HIPCHK(ctx, hipMalloc(&s.arena, total));
        HIPCHK(ctx, hipMemsetAsync(s.arena, 0, total, s.stream));  // (tickets and status words start at zero)
        {
            char* q = (char*)s.arena;
            for (const Part& pt : parts) { *pt.p = q; q += up(pt.bytes, 4096); }
        }
        HIPCHK(ctx, hipHostMalloc((void**)&s.h_res, (size_t)RES_CAP * sizeof(OpResult), hipHostMallocMapped));
        HIPCHK(ctx, hipHostMalloc((void**)&s.h_coefres, (size_t)RES_CAP * 8 * sizeof(float), hipHostMallocMapped));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&s.d_res, s.h_res, 0));
        HIPCHK(ctx, hipHostGetDevicePointer((void**)&s.d_coefres, s.h_coefres, 0));
        if (ctx->fg.fusable) {
            // [27 counter lines | 32 granules | workgroup records | stamps]
            const size_t G = (size_t)ctx->fg.G;
            const bool want_stamps = getenv("WM_FUSED_STAMPS") != nullptr;
            const size_t ndbl = G * (2 * (13 + NGRAM) + 4 + 8 + 1) + (want_stamps ? G * 16 + 16 + NGRAM : 0);
            const size_t bytes = up(FUSED_CNT_BYTES + 64 * 8 + ndbl * sizeof(double), (size_t)2 << 20);  // (an arena of its own, like the sweeps' scratch)
            HIPCHK(ctx, hipMalloc(&s.fz_block, bytes));
            HIPCHK(ctx, hipMemsetAsync(s.fz_block, 0, bytes, s.stream));
            char* b = (char*)s.fz_block;
            s.fz.cnt = (unsigned*)b;
            s.fz.gran = (unsigned long long*)(b + FUSED_CNT_BYTES);
            double* d = (double*)(b + FUSED_CNT_BYTES + 64 * 8);
            s.fz.pmain = d; d += 2 * G * (13 + NGRAM);
            s.fz.gstat = (unsigned long long*)d; d += G * 4;
            s.fz.gcorr = (unsigned long long*)d; d += G * 8;
            s.fz.gdone = (unsigned long long*)d; d += G;
            s.fz.stamps = want_stamps ? (unsigned long long*)d : nullptr;
            s.fz.dbg = getenv("WM_FUSED_DBG") ? atoi(getenv("WM_FUSED_DBG")) : 0;  // development / test switches of the fused kernels
            if (!want_stamps) s.fz.dbg &= ~3;  // bits 0, 1 give wrong results (timing experiments): only with the stamps switched on
        }
    }
    HIPCHK(ctx, hipDeviceSynchronize());
    if (ctx->handover) {
        const int rc = ho_alloc(ctx);
        if (rc != WM_OK) { ctx->handover = 0; return rc; }
    }
    return WM_OK;
}

int upload_w(wm_ctx* ctx, const float* w)
{
    auto ws = std::make_shared<WShared>();
    ws->n = (size_t)ctx->rows * ctx->cols;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc((void**)&ws->d_w, ws->n * sizeof(float)));
    HIPCHK(ctx, hipMemcpy(ws->d_w, w, ws->n * sizeof(float), hipMemcpyHostToDevice));
    ctx->w = ws;
    return WM_OK;
}

// loadRandomMatrix (Watermark.cpp:62-75): raw f32 file, size must be rows*cols*4
int read_w_file(wm_ctx* ctx, const char* path, int rows, int cols, std::vector<float>& out)
{
    if (!path) return fail(ctx, WM_ERR_BAD_ARG, "null W path");
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return fail(ctx, WM_ERR_W_OPEN, std::string("Error opening '") + path + "' file for Random noise W array");
    f.seekg(0, std::ios::end);
    const long long total = (long long)f.tellg();
    f.seekg(0, std::ios::beg);
    if ((long long)rows * cols * (long long)sizeof(float) != total)
        return fail(ctx, WM_ERR_W_SIZE,
                    "Error: W file total elements != image dimensions! W file total elements: " +
                        std::to_string(total / (long long)sizeof(float)) + ", Image width: " + std::to_string(cols) +
                        ", Image height: " + std::to_string(rows));
    out.resize((size_t)rows * cols);
    f.read(reinterpret_cast<char*>(out.data()), total);
    if (!f) return fail(ctx, WM_ERR_W_OPEN, std::string("short read on '") + path + "'");
    return WM_OK;
}

int check_params(int rows, int cols, int p, float psnr)
{
    if (p != 3 && p != 5 && p != 7 && p != 9) return WM_ERR_BAD_P;
    if (!(psnr > 0.0f)) return WM_ERR_PSNR;
    if (rows < 1 || cols < 1 || rows > 32768 || cols > 32768) return WM_ERR_BAD_ARG;
    return WM_OK;
}

// the aligned path addresses a plane as a buffer with 32-bit byte offsets (wm_device.hpp make_rsrc): one plane must stay
// below 4 GiB (a 32768 x 32768 f32 plane is not: it takes the generic path)
bool fits_32bit(int rows, long long pitch, int dtype)
{
    return (long long)rows * pitch * (dtype == WM_F32 ? 4 : 1) < (1LL << 32) - 4096;
}

bool vec_ok(const void* p, int rows, long long pitch, long long fstride, long long cstride, int dtype, int frames, int channels)
{
    // f32 planes: the 16-byte row loads and stores of the aligned path only need 4-byte alignment (measured on gfx950: values
    // correct, 2-5 % off the rate of naturally aligned accesses, tools/ubench/unaligned.hip), so a dense plane whose width is not
    // a multiple of 4 -- every other row 8 bytes off a 16-byte boundary -- still takes the aligned path on its full strips
    // instead of the LDS re-lay.  u8 planes keep their rule: a lane's 4 pixels are one dword
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    if (a % 4) return false;
    if (!fits_32bit(rows, pitch, dtype)) return false;
    if (dtype == WM_F32) return true;
    if (pitch % 4) return false;
    if (frames > 1 && fstride % 4) return false;
    if (channels > 1 && cstride % 4) return false;
    return true;
}

int check_plane(wm_ctx* ctx, const wm_plane* pl, int frames_expected, bool allow_rgb, const char* what, bool allow_slot_out = false)
{
    if (!pl || (!pl->data && pl->mem != WM_MEM_SLOT_OUT)) return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": null plane");
    // WM_MEM_SLOT_OUT names the slot's last grey output: an INPUT plane (wm.h).  Anywhere else it has no address behind it
    if (pl->mem == WM_MEM_SLOT_OUT && !allow_slot_out)
        return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": WM_MEM_SLOT_OUT is valid for the grey input plane only");
    if (pl->rows != ctx->rows || pl->cols != ctx->cols)
        return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": plane is " + std::to_string(pl->rows) + "x" + std::to_string(pl->cols) +
                                             ", engine was initialised for " + std::to_string(ctx->rows) + "x" + std::to_string(ctx->cols));
    if (pl->dtype != WM_F32 && pl->dtype != WM_U8) return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": bad dtype");
    if (pl->mem != WM_MEM_DEVICE && pl->mem != WM_MEM_HOST && pl->mem != WM_MEM_SLOT_OUT) return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": bad mem");
    if (pl->channels != 1 && !(allow_rgb && pl->channels == 3)) return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": channels must be 1" + (allow_rgb ? " or 3" : ""));
    if (pl->pitch < pl->cols) return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": pitch < cols");
    if (pl->frames < 1 || pl->frames > ctx->max_frames)
        return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": frames=" + std::to_string(pl->frames) + " exceeds wm_configure max_frames=" + std::to_string(ctx->max_frames));
    if (frames_expected > 0 && pl->frames != frames_expected) return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": frame count mismatch");
    if (pl->channels > 1 && pl->channel_stride < (int64_t)pl->rows * pl->pitch) return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": channel_stride too small");
    if (pl->frames > 1 && pl->frame_stride < (int64_t)(pl->channels - 1) * (pl->channels > 1 ? pl->channel_stride : 0) + (int64_t)pl->rows * pl->pitch)
        return fail(ctx, WM_ERR_BAD_ARG, std::string(what) + ": frame_stride too small (frames overlap)");
    return WM_OK;
}

size_t elem_size(int dtype) { return dtype == WM_F32 ? 4 : 1; }

PlaneDesc desc_device(const wm_plane* pl)
{
    PlaneDesc d;
    d.p = pl->data; d.pitch = pl->pitch; d.fstride = pl->frames > 1 ? pl->frame_stride : 0;
    d.cstride = pl->channels > 1 ? pl->channel_stride : 0;
    d.dtype = pl->dtype; d.channels = pl->channels;
    d.aligned = vec_ok(pl->data, pl->rows, d.pitch, d.fstride, d.cstride, pl->dtype, pl->frames, pl->channels) ? 1 : 0;
    return d;
}

// dense device staging layout for a host plane: pitch = cols rounded up to 4 elements
struct Staged { PlaneDesc d; size_t bytes; long long pitch; };
Staged staged_layout(const wm_plane* pl)
{
    Staged s;
    s.pitch = (pl->cols + 3) & ~3LL;
    const long long plane = s.pitch * pl->rows;
    s.d.pitch = s.pitch; s.d.cstride = plane; s.d.fstride = plane * pl->channels;
    s.d.dtype = pl->dtype; s.d.channels = pl->channels; s.d.aligned = fits_32bit(pl->rows, s.pitch, pl->dtype) ? 1 : 0; s.d.p = nullptr;
    s.bytes = (size_t)plane * pl->channels * pl->frames * elem_size(pl->dtype);
    return s;
}

int ensure(wm_ctx* ctx, void** buf, size_t* have, size_t need)
{
    if (*have >= need) return WM_OK;
    if (*buf) HIPCHK(ctx, hipFree(*buf));
    *buf = nullptr; *have = 0;
    HIPCHK(ctx, hipMalloc(buf, need));
    *have = need;
    return WM_OK;
}

// host plane -> staging (H2D), every (frame, channel) plane as one 2D copy (de-pitching like main.cpp:348-353)
int stage_in(wm_ctx* ctx, Slot& s, const wm_plane* pl, void* dst, const Staged& st)
{
    const size_t es = elem_size(pl->dtype);
    for (int f = 0; f < pl->frames; ++f)
        for (int ch = 0; ch < pl->channels; ++ch) {
            const char* src = (const char*)pl->data + ((size_t)f * (pl->frames > 1 ? pl->frame_stride : 0) + (size_t)ch * (pl->channels > 1 ? pl->channel_stride : 0)) * es;
            char* d = (char*)dst + ((size_t)f * st.d.fstride + (size_t)ch * st.d.cstride) * es;
            HIPCHK(ctx, hipMemcpy2DAsync(d, st.pitch * es, src, pl->pitch * es, pl->cols * es, pl->rows, hipMemcpyHostToDevice, s.stream));
        }
    return WM_OK;
}
int stage_out(wm_ctx* ctx, Slot& s, const wm_plane* pl, const void* src, const Staged& st)
{
    const size_t es = elem_size(pl->dtype);
    for (int f = 0; f < pl->frames; ++f)
        for (int ch = 0; ch < pl->channels; ++ch) {
            char* d = (char*)pl->data + ((size_t)f * (pl->frames > 1 ? pl->frame_stride : 0) + (size_t)ch * (pl->channels > 1 ? pl->channel_stride : 0)) * es;
            const char* sp = (const char*)src + ((size_t)f * st.d.fstride + (size_t)ch * st.d.cstride) * es;
            HIPCHK(ctx, hipMemcpy2DAsync(d, pl->pitch * es, sp, st.pitch * es, pl->cols * es, pl->rows, hipMemcpyDeviceToHost, s.stream));
        }
    return WM_OK;
}

// device -> device snapshot of a RESOLVED grey plane (a device plane, or the slot's last output behind WM_MEM_SLOT_OUT) into
// the dense staging layout `st` (staged_layout of the caller's plane: same rows, cols, frames, dtype)
int snapshot(wm_ctx* ctx, Slot& s, const PlaneDesc& src_d, const wm_plane* shape, void* dst, const Staged& st)
{
    const size_t es = elem_size(shape->dtype);
    for (int f = 0; f < shape->frames; ++f) {
        const char* src = (const char*)src_d.p + (size_t)f * src_d.fstride * es;
        char* d = (char*)dst + (size_t)f * st.d.fstride * es;
        HIPCHK(ctx, hipMemcpy2DAsync(d, st.pitch * es, src, src_d.pitch * es, shape->cols * es, shape->rows, hipMemcpyDeviceToDevice, s.stream));
    }
    return WM_OK;
}

bool planes_overlap(const wm_plane* a, const wm_plane* b)
{
    auto extent = [](const wm_plane* p) {
        const size_t es = elem_size(p->dtype);
        size_t n = (size_t)(p->rows - 1) * p->pitch + p->cols;
        if (p->channels > 1) n += (size_t)(p->channels - 1) * p->channel_stride;
        if (p->frames > 1) n += (size_t)(p->frames - 1) * p->frame_stride;
        return n * es;
    };
    const char* a0 = (const char*)a->data; const char* a1 = a0 + extent(a);
    const char* b0 = (const char*)b->data; const char* b1 = b0 + extent(b);
    return a0 < b1 && b0 < a1;
}

// the same question for RESOLVED planes (what the kernels will address: a WM_MEM_SLOT_OUT input is the slot's last output,
// host planes are their staging buffers)
bool descs_overlap(const PlaneDesc& a, const PlaneDesc& b, int rows, int cols, int frames)
{
    auto extent = [&](const PlaneDesc& d) {
        size_t n = (size_t)(rows - 1) * d.pitch + cols;
        if (d.channels > 1) n += (size_t)(d.channels - 1) * d.cstride;
        if (frames > 1) n += (size_t)(frames - 1) * d.fstride;
        return n * (d.dtype == WM_F32 ? 4 : 1);
    };
    const char* a0 = (const char*)a.p; const char* a1 = a0 + extent(a);
    const char* b0 = (const char*)b.p; const char* b1 = b0 + extent(b);
    return a0 < b1 && b0 < a1;
}

// A hand-over describes the plane a slot's last embed wrote (Slot::last_out).  Whatever the library itself writes over that
// plane afterwards -- an embed on ANOTHER slot into the same buffer, a band embed, wm_compute_mask's mask / error planes --
// ends the description; writes by the caller are the caller's contract (wm.h, WM_HANDOVER_VERIFY checks it)
void invalidate_handovers(wm_ctx* ctx, const PlaneDesc& written, int frames)
{
    if (!written.p) return;
    for (auto& t : ctx->slots) {
        if (!t.ho.valid || t.last_out_frames == 0) continue;
        auto extent = [&](const PlaneDesc& d, int f) {
            size_t n = (size_t)(ctx->rows - 1) * d.pitch + ctx->cols;
            if (d.channels > 1) n += (size_t)(d.channels - 1) * d.cstride;
            if (f > 1) n += (size_t)(f - 1) * d.fstride;
            return n * (d.dtype == WM_F32 ? 4 : 1);
        };
        const char* a0 = (const char*)t.last_out.p; const char* a1 = a0 + extent(t.last_out, t.last_out_frames);
        const char* b0 = (const char*)written.p; const char* b1 = b0 + extent(written, frames);
        if (a0 < b1 && b0 < a1) t.ho.valid = false;
    }
}

// a profiled sweep: while the scope is set, every kernel launched through WM_KLAUNCH draws its own start / stop event pair
// (wm_kernels.hpp: the events are attached to the dispatch, so they bracket the kernel and nothing else); a sweep that is two
// launches (aligned strips + generic remainder) counts as ONE call of its kernel with the two durations added
struct ProfScope {
    wm_ctx* ctx; int kid; hipStream_t st;
    LaunchProf lp;
    ProfScope(wm_ctx* c, int k, hipStream_t s) : ctx(c), kid(k), st(s)
    {
        if (!ctx->prof) return;
        lp.get = &ProfScope::get; lp.owner = ctx;
        launch_prof_slot() = &lp;
    }
    ~ProfScope()
    {
        if (!ctx->prof) return;
        launch_prof_slot() = nullptr;
        for (int i = 0; i < lp.n; ++i) ctx->prof_recs.push_back({kid, lp.a[i], lp.b[i], i == 0});
    }
    static hipEvent_t get(void* owner)
    {
        wm_ctx* c = static_cast<wm_ctx*>(owner);
        if (!c->prof_free.empty()) { hipEvent_t e = c->prof_free.back(); c->prof_free.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
};

int prof_collect(wm_ctx* ctx)
{
    if (ctx->prof_recs.empty()) return WM_OK;
    HIPCHK(ctx, hipDeviceSynchronize());
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { ctx->prof_n[r.kid] += r.first ? 1 : 0; ctx->prof_ms[r.kid] += ms; }
        ctx->prof_free.push_back(r.a); ctx->prof_free.push_back(r.b);
    }
    ctx->prof_recs.clear();
    return WM_OK;
}

int get_slot(wm_ctx* ctx, int slot, Slot** out, bool* sync_after)
{
    *sync_after = slot == WM_SLOT_SYNC;
    const int idx = slot == WM_SLOT_SYNC ? 0 : slot;
    if (idx < 0 || idx >= ctx->nslots) return fail(ctx, WM_ERR_BAD_ARG, "bad slot " + std::to_string(slot));
    *out = &ctx->slots[idx];
    return WM_OK;
}

// after the launches of one op: a failed launch may leave the sweep's last-block tickets half counted, so clear them
// (stream-ordered) before the slot is used again
int launch_check(wm_ctx* ctx, Slot& s)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipMemsetAsync(s.d_ticket, 0, ticket_words(ctx) * sizeof(unsigned), s.stream);
        return fail(ctx, WM_ERR_RUNTIME, std::string("kernel launch: ") + hipGetErrorString(e));
    }
    return WM_OK;
}

// remember where to deliver the result records of one op (the kernels write them to mapped host memory)
int push_pending(wm_ctx* ctx, Slot& s, int frames, float* value_out, int* status_out, float* coef_out)
{
    Pending pd;
    pd.frames = frames; pd.res_off = s.res_used; pd.value_out = value_out; pd.status_out = status_out;
    pd.coef_out = coef_out; pd.coef_off = s.res_used * 8;
    s.res_used += frames;
    s.pending.push_back(pd);
    return WM_OK;
}

// hand the result records of everything queued on the slot to the callers' pointers (the stream has completed)
int deliver(Slot& s)
{
    int rc = WM_OK;
    for (auto& pd : s.pending) {
        for (int f = 0; f < pd.frames; ++f) {
            const OpResult& r = s.h_res[pd.res_off + f];
            if (pd.status_out) pd.status_out[f] = r.status;
            // unsolvable embed leaves `a` untouched (Watermark.cpp:164-165); detect delivers 0.0f (:246-247)
            if (pd.value_out && !(r.status != 0 && pd.keep_value_when_unsolvable)) pd.value_out[f] = r.value;
            if (r.status != 0) rc = WM_UNSOLVABLE;
        }
        if (pd.coef_out) std::memcpy(pd.coef_out, s.h_coefres + (size_t)pd.res_off * 8, (size_t)pd.frames * 8 * sizeof(float));
    }
    s.pending.clear();
    s.res_used = 0;
    return rc;
}

int do_sync(wm_ctx* ctx, Slot& s)
{
    HIPCHK(ctx, hipStreamSynchronize(s.stream));
    return deliver(s);
}

}  // namespace

wm_ctx::~wm_ctx()
{
    for (auto& s : slots) free_slot(s);
    for (auto& r : prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : prof_free) (void)hipEventDestroy(e);
    if (fused_lock_fd >= 0) (void)close(fused_lock_fd);
}

extern "C" {

int wm_create(wm_ctx** out, int device, int rows, int cols, int p, float psnr, const float* w_rowmajor)
{
    if (!out) return WM_ERR_BAD_ARG;
    *out = nullptr;
    int rc = check_params(rows, cols, p, psnr);
    if (rc != WM_OK) return rc;
    if (!w_rowmajor) return WM_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WM_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) device = 0;  // main.cpp:73-78: invalid device -> default 0
    std::unique_ptr<wm_ctx> ctx(new wm_ctx);
    ctx->device = device; ctx->rows = rows; ctx->cols = cols; ctx->p = p; ctx->psnr = psnr;
    ctx->sF = 255.0f / sqrtf(powf(10.0f, psnr / 10.0f));  // Watermark.cpp:22
    if (const char* e = getenv("WM_FUSED")) ctx->fused_mode = e[0] == '0' ? 0 : 1;
    if (hipSetDevice(device) != hipSuccess) return WM_ERR_NO_DEVICE;
    rc = upload_w(ctx.get(), w_rowmajor);
    if (rc != WM_OK) return rc;
    rc = alloc_slots(ctx.get(), 2, 1);
    if (rc != WM_OK) return rc;
    *out = ctx.release();
    return WM_OK;
}

int wm_create_generated(wm_ctx** out, int device, int rows, int cols, int p, float psnr, uint32_t seed)
{
    if (!out) return WM_ERR_BAD_ARG;
    *out = nullptr;
    int rc = check_params(rows, cols, p, psnr);
    if (rc != WM_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WM_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) device = 0;
    std::unique_ptr<wm_ctx> ctx(new wm_ctx);
    ctx->device = device; ctx->rows = rows; ctx->cols = cols; ctx->p = p; ctx->psnr = psnr;
    ctx->sF = 255.0f / sqrtf(powf(10.0f, psnr / 10.0f));  // Watermark.cpp:22
    if (const char* e = getenv("WM_FUSED")) ctx->fused_mode = e[0] == '0' ? 0 : 1;
    if (hipSetDevice(device) != hipSuccess) return WM_ERR_NO_DEVICE;
    auto ws = std::make_shared<WShared>();
    ws->n = (size_t)rows * cols;
    if (hipMalloc((void**)&ws->d_w, ws->n * sizeof(float)) != hipSuccess) return WM_ERR_ALLOC;
    launch_gen_w(nullptr, ws->d_w, rows, cols, seed);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return WM_ERR_RUNTIME;
    ctx->w = ws;
    rc = alloc_slots(ctx.get(), 2, 1);
    if (rc != WM_OK) return rc;
    *out = ctx.release();
    return WM_OK;
}

int wm_create_from_file(wm_ctx** out, int device, int rows, int cols, int p, float psnr, const char* w_path)
{
    if (!out) return WM_ERR_BAD_ARG;
    *out = nullptr;
    int rc = check_params(rows, cols, p, psnr);
    if (rc != WM_OK) return rc;
    std::vector<float> w;
    rc = read_w_file(nullptr, w_path, rows, cols, w);
    if (rc != WM_OK) return rc;
    return wm_create(out, device, rows, cols, p, psnr, w.data());
}

int wm_clone(const wm_ctx* src, wm_ctx** out)
{
    if (!src || !out) return WM_ERR_BAD_ARG;
    *out = nullptr;
    std::unique_ptr<wm_ctx> ctx(new wm_ctx);
    ctx->device = src->device; ctx->rows = src->rows; ctx->cols = src->cols; ctx->p = src->p; ctx->psnr = src->psnr;
    ctx->sF = src->sF; ctx->w = src->w; ctx->rps_override = src->rps_override; ctx->fused_mode = src->fused_mode;
    ctx->band_lo = src->band_lo; ctx->band_hi = src->band_hi; ctx->band_rows_global = src->band_rows_global;
    int rc = alloc_slots(ctx.get(), src->nslots, src->max_frames);
    if (rc != WM_OK) return rc;
    *out = ctx.release();
    return WM_OK;
}

int wm_reinit(wm_ctx* ctx, int rows, int cols, const float* w_rowmajor)
{
    if (!ctx || !w_rowmajor) return WM_ERR_BAD_ARG;
    int rc = check_params(rows, cols, ctx->p, ctx->psnr);
    if (rc != WM_OK) return fail(ctx, rc, "bad dimensions");
    for (auto& s : ctx->slots)
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    // commit the new shape only once the new W is on the device; a band configuration belongs to the old shape
    const int old_rows = ctx->rows, old_cols = ctx->cols;
    ctx->rows = rows; ctx->cols = cols;
    rc = upload_w(ctx, w_rowmajor);
    if (rc != WM_OK) { ctx->rows = old_rows; ctx->cols = old_cols; return rc; }
    ctx->band_lo = ctx->band_hi = 0; ctx->band_rows_global = 0;
    return alloc_slots(ctx, ctx->nslots, ctx->max_frames);
}

int wm_reinit_from_file(wm_ctx* ctx, int rows, int cols, const char* w_path)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    std::vector<float> w;
    int rc = read_w_file(ctx, w_path, rows, cols, w);
    if (rc != WM_OK) return rc;
    return wm_reinit(ctx, rows, cols, w.data());
}

void wm_destroy(wm_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

int wm_configure(wm_ctx* ctx, int nslots, int max_frames)
{
    if (!ctx || nslots < 1 || nslots > 64 || max_frames < 1 || max_frames > RES_CAP) return fail(ctx, WM_ERR_BAD_ARG, "wm_configure: bad arguments");
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    return alloc_slots(ctx, nslots, max_frames);
}

int wm_set_fused(wm_ctx* ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 1) return fail(ctx, WM_ERR_BAD_ARG, "wm_set_fused: mode must be 0 or 1");
    ctx->fused_mode = mode;
    return WM_OK;
}

int wm_set_handover(wm_ctx* ctx, int on)
{
    if (!ctx || on < 0 || on > 1) return fail(ctx, WM_ERR_BAD_ARG, "wm_set_handover: 0 or 1");
    for (auto& s : ctx->slots) s.ho.valid = false;
    if (on) {
        // switched on only when every slot has both of its arrays
        const int rc = ho_alloc(ctx);
        if (rc != WM_OK) { ctx->handover = 0; return rc; }
    }
    ctx->handover = on;
    return WM_OK;
}

int wm_fused_info(const wm_ctx* ctx, int* workgroups, int* tile_rows, unsigned long long* fallbacks)
{
    if (!ctx) return 0;
    if (workgroups) *workgroups = ctx->fg.fusable ? ctx->fg.G : 0;
    if (tile_rows) *tile_rows = ctx->fg.fusable ? ctx->fg.th : 0;
    if (fallbacks) *fallbacks = ctx->fused_fallbacks;
    return ctx->fg.fusable && ctx->fused_mode != 0 && ctx->band_hi == 0 && ctx->p == 3 ? 1 : 0;
}

unsigned long long wm_fused_lock_skips(const wm_ctx* ctx) { return ctx ? ctx->fused_lock_skips : 0; }

int wm_fused_stamps(wm_ctx* ctx, unsigned long long* out, int cap)
{
    if (!ctx || !out || ctx->slots.empty() || !ctx->slots[0].fz.stamps) return 0;
    const int n = ctx->fg.G * 16 + 16 < cap ? ctx->fg.G * 16 + 16 : cap;  // [G][16] + 16 stamps of the workgroup that folded and solved
    // synchronous fused calls return on the result record, before the kernel has retired, and the stamps are plain stores
    // (visible at the end of the kernel), the slot streams are non-blocking: wait for the slot's stream, not for the null stream
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->slots[0].stream) != hipSuccess) return 0;
    if (hipMemcpy(out, ctx->slots[0].fz.stamps, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

int wm_fused_gram(wm_ctx* ctx, double* out44)
{
    if (!ctx || !out44 || ctx->slots.empty() || !ctx->slots[0].fz.stamps) return 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->slots[0].stream) != hipSuccess) return 0;
    if (hipMemcpy(out44, ctx->slots[0].fz.stamps + 16 * (ctx->fg.G + 1), NGRAM * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return NGRAM;
}

int wm_set_rows_per_segment(wm_ctx* ctx, int rps)
{
    if (!ctx || rps < 0 || rps > 4096) return fail(ctx, WM_ERR_BAD_ARG, "wm_set_rows_per_segment: bad value");
    ctx->rps_override = rps;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    return alloc_slots(ctx, ctx->nslots, ctx->max_frames);
}

// does this call take the fused single-frame kernels?  Synchronous one-frame calls on whole images with the 3x3 window,
// when the shape fits the LDS tiling (fused_geometry) -- everything else takes the batched sweeps
static bool fused_call(wm_ctx* ctx, bool sync_after, int frames)
{
    if (!(sync_after && frames == 1 && ctx->fused_mode != 0 && ctx->fg.fusable && ctx->band_hi == 0 && ctx->p == 3)) return false;
    if (ctx->fused_skip > 0) { --ctx->fused_skip; return false; }  // inside a back-off window after a fallback
    return true;
}

// a fused launch ended without delivering: count it, clear the arrival counters (stream-ordered, behind the launch) and keep
// the next calls off the fused path for a window that doubles while the re-probes keep failing
static int fused_failed(wm_ctx* ctx, Slot& s)
{
    ctx->fused_fallbacks++;
    ctx->fused_backoff = ctx->fused_backoff == 0 ? 8 : (ctx->fused_backoff >= 2048 ? 4096 : 2 * ctx->fused_backoff);
    ctx->fused_skip = ctx->fused_backoff;
    HIPCHK(ctx, hipMemsetAsync(s.fz.cnt, 0, FUSED_CNT_BYTES, s.stream));
    return WM_OK;
}

// launch -> record: wait for a fused launch's result record (poll; the stream only when the output is staged to the host
// behind the kernel, or when the poll gave up).  *got holds the record as read by ONE acquire load.
static int fused_wait(wm_ctx* ctx, Slot& s, const OpResult* hres, bool need_stream, OpResult* got)
{
    got->status = FUSED_PENDING; got->value = 0.0f;
    const bool arrived = poll_record(hres, FUSED_PENDING, s.stream, got);
    if (need_stream || !arrived) {
        HIPCHK(ctx, hipStreamSynchronize(s.stream));
        const uint64_t v = reinterpret_cast<const std::atomic<uint64_t>*>(hres)->load(std::memory_order_acquire);
        std::memcpy(got, &v, 8);
    }
    return WM_OK;
}

// shared front half of embed / detect / mask: stage the grey input if needed and describe it
static int prep_input(wm_ctx* ctx, Slot& s, const wm_plane* in, PlaneDesc* xd)
{
    if (in->mem == WM_MEM_SLOT_OUT) {
        // the device copy of what the last embed on this slot wrote (grey): a streamed frame is detected without
        // crossing the host link again
        if (s.last_out_frames == 0 || s.last_out.channels != 1 || in->frames != s.last_out_frames || in->dtype != s.last_out_dtype)
            return fail(ctx, WM_ERR_BAD_ARG, "WM_MEM_SLOT_OUT: no matching grey embed output on this slot (frames / dtype must equal the last wm_embed's)");
        *xd = s.last_out;
        return WM_OK;
    }
    if (in->mem == WM_MEM_HOST) {
        Staged st = staged_layout(in);
        int rc = ensure(ctx, &s.st_in, &s.st_in_bytes, st.bytes);
        if (rc != WM_OK) return rc;
        rc = stage_in(ctx, s, in, s.st_in, st);
        if (rc != WM_OK) return rc;
        *xd = st.d; xd->p = s.st_in;
    } else {
        *xd = desc_device(in);
    }
    return WM_OK;
}

int wm_embed(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out, float* a_out,
             int* status_out, int slot)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    if (mask != WM_MASK_ME && mask != WM_MASK_NVF) return fail(ctx, WM_ERR_BAD_ARG, "bad mask type");
    if (mask == WM_MASK_ME && ctx->p != 3) return fail(ctx, WM_ERR_BAD_P, "ME mask needs p == 3 (main.cpp:89)");
    Slot* sp; bool sync_after;
    int rc = get_slot(ctx, slot, &sp, &sync_after);
    if (rc != WM_OK) return rc;
    Slot& s = *sp;
    if ((rc = check_plane(ctx, in_gray, 0, false, "in_gray", true)) != WM_OK) return rc;
    const int frames = in_gray->frames;
    if ((rc = check_plane(ctx, base, frames, true, "base")) != WM_OK) return rc;
    if ((rc = check_plane(ctx, out, frames, true, "out")) != WM_OK) return rc;
    if (out->channels != base->channels || out->dtype != base->dtype) return fail(ctx, WM_ERR_BAD_ARG, "out must have the shape and dtype of base");
    if (in_gray->dtype != base->dtype) return fail(ctx, WM_ERR_BAD_ARG, "in_gray and base must have the same dtype (the reference converts whole frames, main.cpp:355-357)");
    if (s.res_used + frames > RES_CAP) return fail(ctx, WM_ERR_BUSY, "too many un-synced results on this slot");
    HIPCHK(ctx, hipSetDevice(ctx->device));

    PlaneDesc xd, bd, od;
    if ((rc = prep_input(ctx, s, in_gray, &xd)) != WM_OK) return rc;
    Staged st_out_l;
    const bool base_is_in = base->data == in_gray->data && base->mem == in_gray->mem && base->mem != WM_MEM_SLOT_OUT && base->channels == 1 &&
                            base->dtype == in_gray->dtype && base->pitch == in_gray->pitch;
    if (base->mem == WM_MEM_HOST) {
        if (base_is_in) { bd = xd; }
        else {
            Staged st = staged_layout(base);
            if ((rc = ensure(ctx, &s.st_base, &s.st_base_bytes, st.bytes)) != WM_OK) return rc;
            if ((rc = stage_in(ctx, s, base, s.st_base, st)) != WM_OK) return rc;
            bd = st.d; bd.p = s.st_base;
        }
    } else bd = desc_device(base);
    if (out->mem == WM_MEM_HOST) {
        st_out_l = staged_layout(out);
        if ((rc = ensure(ctx, &s.st_out, &s.st_out_bytes, st_out_l.bytes)) != WM_OK) return rc;
        od = st_out_l.d; od.p = s.st_out;
    } else od = desc_device(out);
    // in place = the output overlaps the plane the stencil reads, judged on the RESOLVED addresses (a WM_MEM_SLOT_OUT input is
    // the slot's last output buffer, which the caller may well pass as `out` again)
    const bool inplace = descs_overlap(xd, od, ctx->rows, ctx->cols, frames);
    s.ho.valid = false;  // (whatever this call writes replaces the plane a hand-over described)
    invalidate_handovers(ctx, od, frames);  // (... and that of any other slot whose last output this call overwrites)

    // one image per synchronous call (the reference's call pattern): ONE launch with the frame's tiles resident in LDS
    // (wm_k_fused.hip).  Its y stores come after two chip-wide hand-offs behind every read of x, so an in-place call
    // needs no snapshot of the input.
    std::optional<FusedGuard> guard;
    // (a width that is not a multiple of 4: f32 planes only -- their 16-byte accesses need 4-byte alignment, a u8 lane's dword does not exist there)
    const bool width_ok = ctx->cols % 4 == 0 || (xd.dtype == WM_F32 && bd.dtype == WM_F32 && od.dtype == WM_F32);
    bool take_fused = width_ok && fused_call(ctx, sync_after, frames) && xd.aligned && bd.aligned && od.aligned;
    if (take_fused && !ctx->pair_mode) {  // (wm_embed_detect holds the lock over both launches)
        guard.emplace(ctx->device, ctx->fused_lock_fd);
        if (!guard->ok) { guard.reset(); ctx->fused_lock_skips++; take_fused = false; }
    }
    if (take_fused) {
        OpResult* hres = s.h_res + s.res_used;
        hres->status = FUSED_PENDING;
        if (++s.fz_epoch == 0) s.fz_epoch = 1;
        int lrc;
        // wm_embed_detect on device planes of one type: ONE launch for the pair (k_fused_pair) -- nothing is launched here, the
        // detector's call that follows at once launches both halves (WM_FUSED_PAIR=0: the two kernels back to back, as before)
        s.pair.deferred = ctx->pair_mode && ctx->fused_pair && out->mem != WM_MEM_HOST && od.channels == 1 && bd.channels == 1 &&
                          xd.dtype == bd.dtype && xd.dtype == od.dtype;
        if (s.pair.deferred) { s.pair.mask = mask; s.pair.epoch = s.fz_epoch; s.pair.xd = xd; s.pair.bd = bd; s.pair.od = od; lrc = 0; }
        else { ProfScope ps(ctx, K_FUSED_EMBED, s.stream); lrc = launch_fused_embed(s.stream, ctx->fg, s.fz, s.fz_epoch, mask, xd, ctx->w->d_w, bd, od, ctx->sF, sqrt_n(ctx), s.d_res + s.res_used); }
        if (lrc == 0) {
            if ((rc = launch_check(ctx, s)) != WM_OK) return rc;
            if (out->mem == WM_MEM_HOST && (rc = stage_out(ctx, s, out, s.st_out, st_out_l)) != WM_OK) return rc;
            if (ctx->pair_mode) {
                // the detector's launch follows at once on the same stream; its record completes both (wm_detect's fused branch)
                s.pair.armed = true; s.pair.res_index = s.res_used; s.pair.host_out = out->mem == WM_MEM_HOST;
                s.pair.out_overlaps_inputs = inplace || descs_overlap(bd, od, ctx->rows, ctx->cols, frames);
                s.last_out = od; s.last_out_frames = frames; s.last_out_dtype = out->dtype;
                if ((rc = push_pending(ctx, s, frames, a_out, status_out, nullptr)) != WM_OK) return rc;
                s.pending.back().keep_value_when_unsolvable = true;
                return WM_OK;
            }
            // device output: the kernel writes y through to memory and reports last, so the record is the completion
            // signal; host output: the staging copy behind the kernel has to finish as well
            OpResult got;
            if ((rc = fused_wait(ctx, s, hres, out->mem == WM_MEM_HOST, &got)) != WM_OK) return rc;
            if (got.status != FUSED_PENDING && got.status != FUSED_INCOMPLETE) {
                ctx->fused_backoff = 0;
                s.last_out = od; s.last_out_frames = frames; s.last_out_dtype = out->dtype;
                if ((rc = push_pending(ctx, s, frames, a_out, status_out, nullptr)) != WM_OK) return rc;
                s.pending.back().keep_value_when_unsolvable = true;
                return deliver(s);  // the record has arrived
            }
            // PENDING: a hand-off timed out before any output store (the workgroups were not all resident): nothing was
            // written, the sweeps take the call.  INCOMPLETE: output stores were issued but the end of the frame was not
            // observed -- the output plane may be partly written.  If it overlaps the input or the base, they are no longer
            // the caller's frame: re-running would watermark a watermarked frame, so the call fails instead.
            if ((rc = fused_failed(ctx, s)) != WM_OK) return rc;
            if (got.status == FUSED_INCOMPLETE) {
                HIPCHK(ctx, hipStreamSynchronize(s.stream));
                if (inplace || descs_overlap(bd, od, ctx->rows, ctx->cols, frames))
                    return fail(ctx, WM_ERR_RUNTIME, "fused embed: the completion of the output stores was not observed and the output overlaps the input "
                                                     "or the base (in-place call): the frame may be partly watermarked and cannot be re-run");
            }
        }
    }
    guard.reset();  // (the sweeps below do not need the device to themselves)
    if (inplace) {
        // in-place embed (the video path hands the same frame as input, base and output, main.cpp:356,380):
        // the stencil must keep reading the ORIGINAL pixels while rows of `out` are being written, so the mask
        // source is snapshotted into the slot's staging buffer first (one extra device copy of the grey plane)
        Staged st = staged_layout(in_gray);
        if ((rc = ensure(ctx, &s.st_in, &s.st_in_bytes, st.bytes)) != WM_OK) return rc;
        if ((rc = snapshot(ctx, s, xd, in_gray, s.st_in, st)) != WM_OK) return rc;
        // the base IS the input plane (video frames: input, base and output are one plane): read it from the snapshot too --
        // k_embed then takes the base from its stencil window (no base stream) and nothing reads the plane being overwritten
        const bool base_is_input = bd.p == xd.p && bd.pitch == xd.pitch && bd.fstride == xd.fstride && bd.dtype == xd.dtype && bd.channels == 1;
        xd = st.d; xd.p = s.st_in;
        if (base_is_input) bd = xd;
    }

    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, mask, &lg)) != WM_OK) return rc;
    const float* W = ctx->w->d_w;
    const int aligned_w = fits_32bit(ctx->rows, ctx->cols, WM_F32) ? 1 : 0  /* (W is a dense f32 plane: 4-byte aligned rows suffice, vec_ok) */;
    const int pad = ctx->p / 2;
    OpResult* res = s.d_res + s.res_used;
    // Gram hand-over (wm_set_handover): k_embed also leaves the tile-internal lag sums of y for a detector that reads this
    // output as WM_MEM_SLOT_OUT (grey f32 planes on the aligned path; launch_embed says whether it applied)
    // (its tiles reach two rows behind their segment: not when the output overwrites the base those rows are read from)
    if (ctx->pair_handover && !(s.d_ho && s.d_hoseam) && ho_alloc(ctx) != WM_OK) { (void)hipGetLastError(); ctx->last_error.clear(); }  // (no memory: no hand-over)
    const HandOver ho{s.d_ho, lg.nstrips * lg.nsegs + handover_seam_blocks(lg), s.d_hoseam};
    const HandOver* hop = (ctx->handover || ctx->pair_handover) && s.d_ho && s.d_hoseam && out->channels == 1 && ho.stride <= ho_stride_max(ctx) &&
                                  !descs_overlap(bd, od, ctx->rows, ctx->cols, frames) ? &ho : nullptr;
    bool handed = false;
    if (mask == WM_MASK_ME) {
        { ProfScope ps(ctx, K_GRAM, s.stream); launch_gram(s.stream, lg, frames, xd, s.d_gram, s.d_gramb, s.d_ticket, s.d_coef, s.d_status, s.d_gramtot); }
        { ProfScope ps(ctx, K_ME_STATS, s.stream); launch_me_stats(s.stream, lg, frames, xd, W, aligned_w, s.d_coef, s.d_status, s.d_pmax, s.d_pss, s.d_ticket + ctx->max_frames * TKS, strip_tickets(ctx, s, 0), s.d_smax, s.d_sss, ctx->sF, sqrt_n(ctx), s.d_scal, res, s.d_raw); }
        { ProfScope ps(ctx, K_EMBED, s.stream); handed = launch_embed(s.stream, lg, frames, 0, 1, xd, W, aligned_w, bd, od, s.d_coef, s.d_status, s.d_scal, hop); }
    } else {
        { ProfScope ps(ctx, K_NVF_STATS, s.stream); launch_nvf_stats(s.stream, lg, frames, xd, W, aligned_w, pad, s.d_pss, s.d_ticket + ctx->max_frames * TKS, strip_tickets(ctx, s, 0), s.d_sss, ctx->sF, sqrt_n(ctx), s.d_scal, res, s.d_raw); }
        { ProfScope ps(ctx, K_EMBED, s.stream); handed = launch_embed(s.stream, lg, frames, 1, pad, xd, W, aligned_w, bd, od, nullptr, nullptr, s.d_scal, hop); }
    }
    if (handed) { s.ho.valid = true; s.ho.lg = lg; s.ho.frames = frames; s.ho.stride = ho.stride; }
    if ((rc = launch_check(ctx, s)) != WM_OK) return rc;
    if (out->mem == WM_MEM_HOST && (rc = stage_out(ctx, s, out, s.st_out, st_out_l)) != WM_OK) return rc;
    s.last_out = od; s.last_out_frames = frames; s.last_out_dtype = out->dtype;
    if ((rc = push_pending(ctx, s, frames, a_out, status_out, nullptr)) != WM_OK) return rc;
    s.pending.back().keep_value_when_unsolvable = true;
    return sync_after && !ctx->pair_mode ? do_sync(ctx, s) : WM_OK;  // (wm_embed_detect: the detector's wait covers the embed)
}

// the Gram sweep of a detector-side call: k_gram over the plane -- or, when the plane is the slot's last embed output and that
// embed left its tile-internal lag sums (wm_set_handover), only the seams, the border frame and the solve (k_gram_ho)
static int gram_sweep(wm_ctx* ctx, Slot& s, const LaunchGeom& lg, int frames, const PlaneDesc& xd, const wm_plane* img)
{
    if (img->mem == WM_MEM_SLOT_OUT && s.ho.valid && s.ho.frames == frames && s.d_ho && s.d_hoseam && ctx->band_hi == 0) {
        {
            ProfScope ps(ctx, K_GRAM_HO, s.stream);
            // (the border blocks are this short launch's longest: as many of them as the record array holds, not the batched sweep's 16)
            LaunchGeom l2 = s.ho.lg;
            l2.nbb = border_blocks(ctx->rows, ctx->cols);
            launch_gram_ho(s.stream, l2, frames, xd, HandOver{s.d_ho, s.ho.stride, s.d_hoseam}, s.d_gramb, s.d_ticket, s.d_coef, s.d_status, s.d_gramtot);
        }
        if (!ctx->handover_verify) return WM_OK;
        // WM_HANDOVER_VERIFY=1 (a debug mode, it synchronises): the hand-over rests on the caller's promise that the plane
        // behind WM_MEM_SLOT_OUT is still what the embed wrote.  Run the ordinary Gram sweep over the plane as it is NOW and
        // hold the 44 totals against the handed-over ones: the same exact products, so they agree to the f64 summation order
        // (<= 1e-14 relative, tests/test_gpu_handover.py) unless a single pixel changed
        std::vector<double> t_ho((size_t)frames * NGRAM), t_now((size_t)frames * NGRAM);
        HIPCHK(ctx, hipMemcpyAsync(t_ho.data(), s.d_gramtot, t_ho.size() * sizeof(double), hipMemcpyDeviceToHost, s.stream));
        launch_gram(s.stream, lg, frames, xd, s.d_gram, s.d_gramb, s.d_ticket, s.d_coef, s.d_status, s.d_gramtot);
        HIPCHK(ctx, hipMemcpyAsync(t_now.data(), s.d_gramtot, t_now.size() * sizeof(double), hipMemcpyDeviceToHost, s.stream));
        HIPCHK(ctx, hipStreamSynchronize(s.stream));
        for (int f = 0; f < frames; ++f)
            for (int t = 0; t < NGRAM; ++t) {
                const double a = t_ho[(size_t)f * NGRAM + t], b = t_now[(size_t)f * NGRAM + t];
                const double scale = std::fmax(1.0, std::fmax(std::fabs(a), std::fabs(b)));
                if (!(std::fabs(a - b) <= 1e-12 * scale)) {
                    s.ho.valid = false;
                    char msg[320];
                    snprintf(msg, sizeof msg, "WM_HANDOVER_VERIFY: the plane behind WM_MEM_SLOT_OUT is not the plane the slot's last wm_embed wrote "
                             "(frame %d, Gram term %d: handed over %.17g, the plane now gives %.17g): it was modified after the embed", f, t, a, b);
                    return fail(ctx, WM_ERR_RUNTIME, msg);
                }
            }
        return WM_OK;
    }
    ProfScope ps(ctx, K_GRAM, s.stream);
    launch_gram(s.stream, lg, frames, xd, s.d_gram, s.d_gramb, s.d_ticket, s.d_coef, s.d_status, s.d_gramtot);
    return WM_OK;
}

int wm_detect(wm_ctx* ctx, int mask, const wm_plane* img, float* corr_out, int* status_out, int slot)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    if (mask != WM_MASK_ME && mask != WM_MASK_NVF) return fail(ctx, WM_ERR_BAD_ARG, "bad mask type");
    if (ctx->p != 3 && mask == WM_MASK_ME) return fail(ctx, WM_ERR_BAD_P, "ME mask needs p == 3 (main.cpp:89)");
    Slot* sp; bool sync_after;
    int rc = get_slot(ctx, slot, &sp, &sync_after);
    if (rc != WM_OK) return rc;
    Slot& s = *sp;
    if ((rc = check_plane(ctx, img, 0, false, "image", true)) != WM_OK) return rc;
    const int frames = img->frames;
    if (s.res_used + frames > RES_CAP) return fail(ctx, WM_ERR_BUSY, "too many un-synced results on this slot");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlaneDesc xd;
    if ((rc = prep_input(ctx, s, img, &xd)) != WM_OK) return rc;
    const bool paired = s.pair.armed;  // a fused embed of wm_embed_detect is in flight in front of this call
    std::optional<FusedGuard> guard;
    bool take_fused = paired || ((ctx->cols % 4 == 0 || xd.dtype == WM_F32) && fused_call(ctx, sync_after, frames) && xd.aligned);
    if (take_fused && !ctx->pair_mode) {
        guard.emplace(ctx->device, ctx->fused_lock_fd);
        if (!guard->ok) { guard.reset(); ctx->fused_lock_skips++; take_fused = false; }
    }
    if (take_fused) {
        // one image per synchronous call: one launch, the frame's tiles resident in LDS (wm_k_fused.hip)
        OpResult* hres = s.h_res + s.res_used;
        hres->status = FUSED_PENDING;
        if (++s.fz_epoch == 0) s.fz_epoch = 1;
        int lrc;
        if (paired && s.pair.deferred) {
            ProfScope ps(ctx, K_FUSED_PAIR, s.stream);
            lrc = launch_fused_pair(s.stream, ctx->fg, s.fz, s.pair.epoch, s.fz_epoch, s.pair.mask, s.pair.xd, ctx->w->d_w, s.pair.bd, s.pair.od, ctx->sF,
                                    sqrt_n(ctx), s.d_res + s.pair.res_index, s.d_res + s.res_used);
        } else { ProfScope ps(ctx, K_FUSED_DETECT, s.stream); lrc = launch_fused_detect(s.stream, ctx->fg, s.fz, s.fz_epoch, mask, xd, ctx->w->d_w, s.d_res + s.res_used); }
        s.pair.deferred = false;
        OpResult got; got.status = FUSED_PENDING; got.value = 0.f;
        if (lrc == 0) {
            if ((rc = launch_check(ctx, s)) != WM_OK) { s.pair.armed = false; return rc; }
            if ((rc = fused_wait(ctx, s, hres, paired && s.pair.host_out, &got)) != WM_OK) { s.pair.armed = false; return rc; }
        }
        if (paired) {
            // the stream is in order: with the detector's record in (or the stream drained), the embed's record is final
            s.pair.armed = false;
            if (got.status == FUSED_PENDING) HIPCHK(ctx, hipStreamSynchronize(s.stream));
            OpResult ge;
            const uint64_t v = reinterpret_cast<const std::atomic<uint64_t>*>(s.h_res + s.pair.res_index)->load(std::memory_order_acquire);
            std::memcpy(&ge, &v, 8);
            if (ge.status == FUSED_PENDING || ge.status == FUSED_INCOMPLETE) {
                // the embed did not complete: forget its queued result, back off, and let wm_embed_detect redo both on the
                // sweeps -- unless output stores went out over the call's own input (wm_embed's rule).  (A completed embed
                // with a detector that timed out is NOT re-run: the detector alone takes the sweeps below.)
                s.pending.pop_back();
                s.res_used = s.pair.res_index;
                s.last_out_frames = 0;
                if ((rc = fused_failed(ctx, s)) != WM_OK) return rc;
                if (ge.status == FUSED_INCOMPLETE && s.pair.out_overlaps_inputs)
                    return fail(ctx, WM_ERR_RUNTIME, "fused embed: the completion of the output stores was not observed and the output overlaps the input "
                                                     "or the base (in-place call): the frame may be partly watermarked and cannot be re-run");
                return PAIR_RETRY;
            }
        }
        if (lrc == 0) {
            if (got.status != FUSED_PENDING) {
                ctx->fused_backoff = 0;
                if ((rc = push_pending(ctx, s, frames, corr_out, status_out, nullptr)) != WM_OK) return rc;
                return deliver(s);  // the record has arrived
            }
            if ((rc = fused_failed(ctx, s)) != WM_OK) return rc;  // (a detector writes nothing: the sweeps can always take the call)
        }
    }
    guard.reset();
    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, mask, &lg)) != WM_OK) return rc;
    const float* W = ctx->w->d_w;
    const int aligned_w = fits_32bit(ctx->rows, ctx->cols, WM_F32) ? 1 : 0  /* (W is a dense f32 plane: 4-byte aligned rows suffice, vec_ok) */;
    OpResult* res = s.d_res + s.res_used;
    if ((rc = gram_sweep(ctx, s, lg, frames, xd, img)) != WM_OK) return rc;
    { ProfScope ps(ctx, K_DETECT, s.stream); launch_detect(s.stream, lg, frames, mask, ctx->p / 2, xd, W, aligned_w, s.d_coef, s.d_status, s.d_pcorr, s.d_ticket + 2 * ctx->max_frames * TKS, strip_tickets(ctx, s, 1), s.d_scorr, res, s.d_raw + ctx->max_frames); }
    if ((rc = launch_check(ctx, s)) != WM_OK) return rc;
    if ((rc = push_pending(ctx, s, frames, corr_out, status_out, nullptr)) != WM_OK) return rc;
    return sync_after ? do_sync(ctx, s) : WM_OK;
}

int wm_embed_detect(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out, float* a_out,
                    float* corr_out, int* status_out, int slot)
{
    if (!ctx || !out) return WM_ERR_BAD_ARG;
    if (out->channels != 1) return fail(ctx, WM_ERR_BAD_ARG, "wm_embed_detect: grey output only (the detector reads the plane the embed wrote)");
    Slot* sp; bool sync_after;
    int rc = get_slot(ctx, slot, &sp, &sync_after);
    if (rc != WM_OK) return rc;
    Slot& s = *sp;
    // the detector's input: the device copy of what the embed writes (WM_MEM_SLOT_OUT) -- so on the sweeps the embed can hand
    // its output's lag sums over whether or not wm_set_handover is on (the detector's Gram sweep is not run)
    wm_plane slot_plane = *out;
    slot_plane.data = nullptr; slot_plane.mem = WM_MEM_SLOT_OUT;
    struct PairHo { wm_ctx* c; explicit PairHo(wm_ctx* c_) : c(c_) { c->pair_handover = 1; } ~PairHo() { c->pair_handover = 0; } } pair_ho(ctx);
    if (!sync_after) {
        // a slot in flight: the two operations queue behind each other, wm_sync delivers both
        if ((rc = wm_embed(ctx, mask, in_gray, base, out, a_out, status_out, slot)) < 0) return rc;
        return wm_detect(ctx, mask, &slot_plane, corr_out, nullptr, slot);
    }
    bool deferred;
    {
        // synchronous: when the fused kernels take the call, both launches go out back to back and the host waits ONCE, for
        // the detector's record (an in-order stream: that record completes the embed's too) -- one launch-to-poll round trip
        // and one host re-arm less than the two calls; the watermarked plane is read back from the caches
        FusedGuard guard(ctx->device, ctx->fused_lock_fd);
        // (the device's lock not obtained within the deadline: this pair runs on the sweeps)
        const int saved_mode = ctx->fused_mode;
        if (!guard.ok) { ctx->fused_lock_skips++; ctx->fused_mode = 0; }
        ctx->pair_mode = 1;
        s.pair.armed = false;
        rc = wm_embed(ctx, mask, in_gray, base, out, a_out, status_out, WM_SLOT_SYNC);
        deferred = rc == WM_OK && s.pair.armed;
        if (deferred) rc = wm_detect(ctx, mask, &slot_plane, corr_out, nullptr, WM_SLOT_SYNC);
        if (s.pair.deferred) {
            // the detector's call failed before it launched the pair: the embed it was to launch never ran -- forget its queued result
            s.pair.deferred = false;
            if (!s.pending.empty()) s.pending.pop_back();
            s.res_used = s.pair.res_index;
            s.last_out_frames = 0;
        }
        ctx->pair_mode = 0;
        ctx->fused_mode = saved_mode;
        s.pair.armed = false;
    }
    if (deferred && rc != PAIR_RETRY) return rc;
    if (deferred) {
        // the fused embed did not complete (a time-out; the context now backs off): the embed again, on the sweeps
        ctx->pair_mode = 1;
        rc = wm_embed(ctx, mask, in_gray, base, out, a_out, status_out, WM_SLOT_SYNC);
        ctx->pair_mode = 0;
    }
    // the embed went to the sweeps (queued, not waited for): the detector follows on the same stream and its wait delivers both
    if (rc < 0) return rc;
    const int rd = wm_detect(ctx, mask, &slot_plane, corr_out, nullptr, WM_SLOT_SYNC);
    if (rd < 0 && !s.pending.empty()) (void)do_sync(ctx, s);  // (never leave the embed's result queued behind a failed call)
    return rd;
}

int wm_compute_mask(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* mask_out, const wm_plane* e_out,
                    float* coef_out, int* status_out, int slot)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    if (mask != WM_MASK_ME && mask != WM_MASK_NVF) return fail(ctx, WM_ERR_BAD_ARG, "bad mask type");
    if (mask == WM_MASK_ME && ctx->p != 3) return fail(ctx, WM_ERR_BAD_P, "ME mask needs p == 3 (main.cpp:89)");
    Slot* sp; bool sync_after;
    int rc = get_slot(ctx, slot, &sp, &sync_after);
    if (rc != WM_OK) return rc;
    Slot& s = *sp;
    if ((rc = check_plane(ctx, in_gray, 0, false, "in_gray", true)) != WM_OK) return rc;
    const int frames = in_gray->frames;
    if ((rc = check_plane(ctx, mask_out, frames, false, "mask_out")) != WM_OK) return rc;
    if (mask_out->dtype != WM_F32 || mask_out->mem != WM_MEM_DEVICE) return fail(ctx, WM_ERR_BAD_ARG, "mask_out must be a device f32 plane");
    if (e_out) {
        if ((rc = check_plane(ctx, e_out, frames, false, "e_out")) != WM_OK) return rc;
        if (e_out->dtype != WM_F32 || e_out->mem != WM_MEM_DEVICE) return fail(ctx, WM_ERR_BAD_ARG, "e_out must be a device f32 plane");
    }
    if (s.res_used + frames > RES_CAP) return fail(ctx, WM_ERR_BUSY, "too many un-synced results on this slot");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlaneDesc xd;
    if ((rc = prep_input(ctx, s, in_gray, &xd)) != WM_OK) return rc;
    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, mask, &lg)) != WM_OK) return rc;
    const float* W = ctx->w->d_w;
    const int aligned_w = fits_32bit(ctx->rows, ctx->cols, WM_F32) ? 1 : 0  /* (W is a dense f32 plane: 4-byte aligned rows suffice, vec_ok) */;
    PlaneDesc mo = desc_device(mask_out), eo;
    if (e_out) eo = desc_device(e_out); else { eo = mo; eo.p = nullptr; }
    invalidate_handovers(ctx, mo, frames);
    invalidate_handovers(ctx, eo, frames);
    OpResult* res = s.d_res + s.res_used;
    float* coefres = s.d_coefres + (size_t)s.res_used * 8;
    if (mask == WM_MASK_ME) {
        { ProfScope ps(ctx, K_GRAM, s.stream); launch_gram(s.stream, lg, frames, xd, s.d_gram, s.d_gramb, s.d_ticket, s.d_coef, s.d_status, s.d_gramtot); }
        { ProfScope ps(ctx, K_ME_STATS, s.stream); launch_me_stats(s.stream, lg, frames, xd, W, aligned_w, s.d_coef, s.d_status, s.d_pmax, s.d_pss, s.d_ticket + ctx->max_frames * TKS, strip_tickets(ctx, s, 0), s.d_smax, s.d_sss, ctx->sF, sqrt_n(ctx), s.d_scal, res, s.d_raw); }
        { ProfScope ps(ctx, K_MASK, s.stream); launch_mask(s.stream, lg, frames, 0, 1, xd, s.d_coef, s.d_status, s.d_scal, mo, eo); }
        launch_mask_result(s.stream, frames, s.d_status, s.d_coef, res, coefres);
    } else {
        { ProfScope ps(ctx, K_MASK, s.stream); launch_mask(s.stream, lg, frames, 1, ctx->p / 2, xd, nullptr, nullptr, nullptr, mo, eo); }
        launch_mask_result(s.stream, frames, nullptr, nullptr, res, coefres);
    }
    if ((rc = launch_check(ctx, s)) != WM_OK) return rc;
    if ((rc = push_pending(ctx, s, frames, nullptr, status_out, coef_out)) != WM_OK) return rc;
    return sync_after ? do_sync(ctx, s) : WM_OK;
}

int wm_gram(wm_ctx* ctx, const wm_plane* img, double* gram_out, int slot)
{
    if (!ctx || !gram_out) return WM_ERR_BAD_ARG;
    Slot* sp; bool sync_after;
    int rc = get_slot(ctx, slot, &sp, &sync_after);
    if (rc != WM_OK) return rc;
    Slot& s = *sp;
    if ((rc = check_plane(ctx, img, 0, false, "image", true)) != WM_OK) return rc;
    const int frames = img->frames;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlaneDesc xd;
    if ((rc = prep_input(ctx, s, img, &xd)) != WM_OK) return rc;
    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, WM_MASK_ME, &lg)) != WM_OK) return rc;
    if ((rc = gram_sweep(ctx, s, lg, frames, xd, img)) != WM_OK) return rc;
    if ((rc = launch_check(ctx, s)) != WM_OK) return rc;
    HIPCHK(ctx, hipStreamSynchronize(s.stream));
    HIPCHK(ctx, hipMemcpy(gram_out, s.d_gramtot, (size_t)frames * NGRAM * sizeof(double), hipMemcpyDeviceToHost));
    return WM_OK;
}

// ---- intra-frame sharding: the context works on a row band of a larger image (wm.h) ----------------------------
int wm_band_configure(wm_ctx* ctx, int own_lo, int own_hi, long long rows_global)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    for (auto& t : ctx->slots) t.ho.valid = false;  // (a hand-over holds whole-image lag sums: not what a band's Gram sweep adds up)
    if (own_hi == 0) { ctx->band_lo = ctx->band_hi = 0; ctx->band_rows_global = 0; return WM_OK; }
    if (own_lo < 0 || own_hi > ctx->rows || own_lo >= own_hi) return fail(ctx, WM_ERR_BAD_ARG, "wm_band_configure: bad row range");
    // a side that is not an image border needs p/2 + 1 halo rows of image data: k_detect scores e_u = u - c.nbrs(u), i.e. it
    // reads the mask one row away from the pixel, and the mask reads x another p/2 rows away (2 rows for p = 3)
    const int need = ctx->p / 2 + 1 > 2 ? ctx->p / 2 + 1 : 2;
    if ((own_lo > 0 && own_lo < need) || (own_hi < ctx->rows && ctx->rows - own_hi < need))
        return fail(ctx, WM_ERR_BAD_ARG, "wm_band_configure: an interior side needs " + std::to_string(need) + " halo rows (p/2 + 1)");
    if (rows_global < own_hi - own_lo) return fail(ctx, WM_ERR_BAD_ARG, "wm_band_configure: rows_global smaller than the band");
    if (ctx->rows < 4 || ctx->cols < 5) return fail(ctx, WM_ERR_BAD_ARG, "wm_band_configure: band too small");
    ctx->band_lo = own_lo; ctx->band_hi = own_hi; ctx->band_rows_global = rows_global;
    return WM_OK;
}

int wm_band_solve(wm_ctx* ctx, const double* totals, int frames, int* status_out, int slot)
{
    if (!ctx || !totals || frames < 1 || frames > ctx->max_frames) return fail(ctx, WM_ERR_BAD_ARG, "wm_band_solve: bad arguments");
    Slot* sp; bool sync_after;
    int rc = get_slot(ctx, slot, &sp, &sync_after);
    if (rc != WM_OK) return rc;
    Slot& s = *sp;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(s.d_totals, totals, (size_t)frames * NGRAM * sizeof(double), hipMemcpyHostToDevice, s.stream));
    launch_solve_totals(s.stream, frames, s.d_totals, s.d_coef, s.d_status);
    if ((rc = launch_check(ctx, s)) != WM_OK) return rc;
    std::vector<int> st((size_t)frames);
    HIPCHK(ctx, hipMemcpyAsync(st.data(), s.d_status, (size_t)frames * sizeof(int), hipMemcpyDeviceToHost, s.stream));
    HIPCHK(ctx, hipStreamSynchronize(s.stream));
    rc = WM_OK;
    for (int f = 0; f < frames; ++f) {
        if (status_out) status_out[f] = st[f];
        if (st[f] != 0) rc = WM_UNSOLVABLE;
    }
    return rc;
}

// ---- the launches of the band phases, shared by the host-exchange calls (totals copied to the caller's host arrays) and the
// device-resident ones (wm_band_*_dev: totals handed over in device memory, nothing synchronises)
static int band_check_mask(wm_ctx* ctx, int mask)
{
    if (mask != WM_MASK_ME && mask != WM_MASK_NVF) return fail(ctx, WM_ERR_BAD_ARG, "bad mask type");
    if (mask == WM_MASK_ME && ctx->p != 3) return fail(ctx, WM_ERR_BAD_P, "ME mask needs p == 3 (main.cpp:89)");
    return WM_OK;
}
static int aligned_w_of(const wm_ctx* ctx) { return fits_32bit(ctx->rows, ctx->cols, WM_F32) ? 1 : 0  /* (W is a dense f32 plane: 4-byte aligned rows suffice, vec_ok) */; }

// stats sweep of the owned rows; the fold tail leaves {max|e| (or 1), sum} per frame in s.d_raw[0 .. frames)
static int band_stats_launch(wm_ctx* ctx, Slot& s, int mask, const wm_plane* in_gray, int* frames_out)
{
    int rc;
    if ((rc = band_check_mask(ctx, mask)) != WM_OK) return rc;
    if ((rc = check_plane(ctx, in_gray, 0, false, "inputImage", true)) != WM_OK) return rc;
    const int frames = in_gray->frames;
    if (s.res_used + frames > RES_CAP) return fail(ctx, WM_ERR_BUSY, "too many un-synced results on this slot");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlaneDesc xd;
    if ((rc = prep_input(ctx, s, in_gray, &xd)) != WM_OK) return rc;
    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, mask, &lg)) != WM_OK) return rc;
    const float* W = ctx->w->d_w;
    OpResult* res = s.d_res + s.res_used;  // written by the tail, not delivered (no pending record)
    if (mask == WM_MASK_ME)
        launch_me_stats(s.stream, lg, frames, xd, W, aligned_w_of(ctx), s.d_coef, s.d_status, s.d_pmax, s.d_pss, s.d_ticket + ctx->max_frames * TKS, strip_tickets(ctx, s, 0), s.d_smax, s.d_sss, ctx->sF, sqrt_n(ctx), s.d_scal, res, s.d_raw);
    else
        launch_nvf_stats(s.stream, lg, frames, xd, W, aligned_w_of(ctx), ctx->p / 2, s.d_pss, s.d_ticket + ctx->max_frames * TKS, strip_tickets(ctx, s, 0), s.d_sss, ctx->sF, sqrt_n(ctx), s.d_scal, res, s.d_raw);
    *frames_out = frames;
    return launch_check(ctx, s);
}

// embed sweep of the owned rows with the scalars in s.d_scal (device planes, out must not overlap the input)
static int band_embed_launch(wm_ctx* ctx, Slot& s, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out)
{
    int rc;
    const int frames = in_gray->frames;
    if ((rc = check_plane(ctx, base, frames, true, "outputImage")) != WM_OK) return rc;
    if ((rc = check_plane(ctx, out, frames, true, "out")) != WM_OK) return rc;
    if (out->channels != base->channels || out->dtype != base->dtype) return fail(ctx, WM_ERR_BAD_ARG, "out must match outputImage in channels and dtype");
    if (in_gray->mem != WM_MEM_DEVICE || base->mem != WM_MEM_DEVICE || out->mem != WM_MEM_DEVICE)
        return fail(ctx, WM_ERR_BAD_ARG, "wm_band_embed: device planes only");
    if (planes_overlap(in_gray, out)) return fail(ctx, WM_ERR_BAD_ARG, "wm_band_embed: out must not overlap the input (halo rows are shared)");
    const PlaneDesc xd = desc_device(in_gray), bd = desc_device(base), od = desc_device(out);
    invalidate_handovers(ctx, od, frames);
    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, mask, &lg)) != WM_OK) return rc;
    if (mask == WM_MASK_ME) launch_embed(s.stream, lg, frames, 0, 1, xd, ctx->w->d_w, aligned_w_of(ctx), bd, od, s.d_coef, s.d_status, s.d_scal);
    else launch_embed(s.stream, lg, frames, 1, ctx->p / 2, xd, ctx->w->d_w, aligned_w_of(ctx), bd, od, nullptr, nullptr, s.d_scal);
    return launch_check(ctx, s);
}

// detect sweep of the owned rows; the fold tail leaves {<e_u,e_w>, |e_u|^2, |e_w|^2} per frame in s.d_raw[max_frames ..)
static int band_detect_launch(wm_ctx* ctx, Slot& s, int mask, const wm_plane* img, int* frames_out)
{
    int rc;
    if ((rc = band_check_mask(ctx, mask)) != WM_OK) return rc;
    if ((rc = check_plane(ctx, img, 0, false, "image", true)) != WM_OK) return rc;
    const int frames = img->frames;
    if (s.res_used + frames > RES_CAP) return fail(ctx, WM_ERR_BUSY, "too many un-synced results on this slot");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlaneDesc xd;
    if ((rc = prep_input(ctx, s, img, &xd)) != WM_OK) return rc;
    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, mask, &lg)) != WM_OK) return rc;
    OpResult* res = s.d_res + s.res_used;
    launch_detect(s.stream, lg, frames, mask, ctx->p / 2, xd, ctx->w->d_w, aligned_w_of(ctx), s.d_coef, s.d_status, s.d_pcorr, s.d_ticket + 2 * ctx->max_frames * TKS, strip_tickets(ctx, s, 1), s.d_scorr, res, s.d_raw + ctx->max_frames);
    *frames_out = frames;
    return launch_check(ctx, s);
}

#define BAND_SLOT(ctx, slot)                                   \
    Slot* sp; bool sync_after;                                 \
    int rc = get_slot(ctx, slot, &sp, &sync_after);            \
    if (rc != WM_OK) return rc;                                \
    Slot& s = *sp; (void)sync_after

int wm_band_stats(wm_ctx* ctx, int mask, const wm_plane* in_gray, double* out, int slot)
{
    if (!ctx || !out) return WM_ERR_BAD_ARG;
    BAND_SLOT(ctx, slot);
    int frames = 0;
    if ((rc = band_stats_launch(ctx, s, mask, in_gray, &frames)) != WM_OK) return rc;
    std::vector<RawSums> raw((size_t)frames);
    HIPCHK(ctx, hipMemcpyAsync(raw.data(), s.d_raw, (size_t)frames * sizeof(RawSums), hipMemcpyDeviceToHost, s.stream));
    HIPCHK(ctx, hipStreamSynchronize(s.stream));
    for (int f = 0; f < frames; ++f) { out[2 * f] = raw[f].v[0]; out[2 * f + 1] = raw[f].v[1]; }
    return WM_OK;
}

int wm_band_embed(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out,
                  const double* max_sum, float* a_out, int slot)
{
    if (!ctx || !max_sum) return WM_ERR_BAD_ARG;
    BAND_SLOT(ctx, slot);
    if ((rc = band_check_mask(ctx, mask)) != WM_OK) return rc;
    if ((rc = check_plane(ctx, in_gray, 0, false, "inputImage")) != WM_OK) return rc;
    const int frames = in_gray->frames;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the strength from the all-reduced totals, as embed_scalars_frame computes it (Watermark.cpp:170)
    std::vector<EmbedScalars> sc((size_t)frames);
    for (int f = 0; f < frames; ++f) {
        const double mx = max_sum[2 * f], ss = max_sum[2 * f + 1];
        sc[f].maxe = mask == WM_MASK_ME ? (float)mx : 1.0f;
        const double nrm = mask == WM_MASK_ME ? sqrt(ss) / (double)sc[f].maxe : sqrt(ss);
        sc[f].a = ctx->sF / (float)(nrm / sqrt_n(ctx));
        if (a_out) a_out[f] = sc[f].a;
    }
    HIPCHK(ctx, hipMemcpyAsync(s.d_scal, sc.data(), (size_t)frames * sizeof(EmbedScalars), hipMemcpyHostToDevice, s.stream));
    rc = band_embed_launch(ctx, s, mask, in_gray, base, out);
    HIPCHK(ctx, hipStreamSynchronize(s.stream));  // sc must outlive the copy
    return rc;
}

int wm_band_detect_sums(wm_ctx* ctx, int mask, const wm_plane* img, double* out, int slot)
{
    if (!ctx || !out) return WM_ERR_BAD_ARG;
    BAND_SLOT(ctx, slot);
    int frames = 0;
    if ((rc = band_detect_launch(ctx, s, mask, img, &frames)) != WM_OK) return rc;
    std::vector<RawSums> raw((size_t)frames);
    HIPCHK(ctx, hipMemcpyAsync(raw.data(), s.d_raw + ctx->max_frames, (size_t)frames * sizeof(RawSums), hipMemcpyDeviceToHost, s.stream));
    HIPCHK(ctx, hipStreamSynchronize(s.stream));
    for (int f = 0; f < frames; ++f) { out[3 * f] = raw[f].v[0]; out[3 * f + 1] = raw[f].v[1]; out[3 * f + 2] = raw[f].v[2]; }
    return WM_OK;
}

// ---- the same phases with the exchange resident in device memory: every call only ENQUEUES on the slot's stream (give the
// slot the stream the caller's collectives are ordered on: wm_set_stream), every total is handed over in device memory the
// caller owns, so the collectives (RCCL all-reduce / all-gather on that stream) run between them without the host in the chain
int wm_band_gram_dev(wm_ctx* ctx, const wm_plane* img, double* totals_dev, int slot)
{
    if (!ctx || !totals_dev) return WM_ERR_BAD_ARG;
    BAND_SLOT(ctx, slot);
    if ((rc = check_plane(ctx, img, 0, false, "image", true)) != WM_OK) return rc;
    const int frames = img->frames;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    PlaneDesc xd;
    if ((rc = prep_input(ctx, s, img, &xd)) != WM_OK) return rc;
    LaunchGeom lg;
    if ((rc = geom_checked(ctx, frames, WM_MASK_ME, &lg)) != WM_OK) return rc;
    // (the sweep's solve tail also solves from the band's own partial totals into the slot: overwritten by wm_band_solve_dev)
    launch_gram(s.stream, lg, frames, xd, s.d_gram, s.d_gramb, s.d_ticket, s.d_coef, s.d_status, totals_dev);
    return launch_check(ctx, s);
}

int wm_band_solve_dev(wm_ctx* ctx, const double* totals_dev, int frames, int slot)
{
    if (!ctx || !totals_dev || frames < 1 || frames > ctx->max_frames) return fail(ctx, WM_ERR_BAD_ARG, "wm_band_solve_dev: bad arguments");
    BAND_SLOT(ctx, slot);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    launch_solve_totals(s.stream, frames, totals_dev, s.d_coef, s.d_status);
    return launch_check(ctx, s);
}

int wm_band_stats_dev(wm_ctx* ctx, int mask, const wm_plane* in_gray, double* max_sum_dev, int slot)
{
    if (!ctx || !max_sum_dev) return WM_ERR_BAD_ARG;
    BAND_SLOT(ctx, slot);
    int frames = 0;
    if ((rc = band_stats_launch(ctx, s, mask, in_gray, &frames)) != WM_OK) return rc;
    launch_band_pick(s.stream, frames, s.d_raw, 2, max_sum_dev);
    return launch_check(ctx, s);
}

int wm_band_embed_dev(wm_ctx* ctx, int mask, const wm_plane* in_gray, const wm_plane* base, const wm_plane* out,
                      const double* gathered_max_sum_dev, int nparts, float* a_dev, int slot)
{
    if (!ctx || !gathered_max_sum_dev || nparts < 1) return WM_ERR_BAD_ARG;
    BAND_SLOT(ctx, slot);
    if ((rc = band_check_mask(ctx, mask)) != WM_OK) return rc;
    if ((rc = check_plane(ctx, in_gray, 0, false, "inputImage")) != WM_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    launch_band_scalars(s.stream, in_gray->frames, gathered_max_sum_dev, nparts, mask, ctx->sF, sqrt_n(ctx), mask == WM_MASK_ME ? s.d_status : nullptr, s.d_scal, a_dev);
    return band_embed_launch(ctx, s, mask, in_gray, base, out);
}

int wm_band_detect_sums_dev(wm_ctx* ctx, int mask, const wm_plane* img, double* sums_dev, int slot)
{
    if (!ctx || !sums_dev) return WM_ERR_BAD_ARG;
    BAND_SLOT(ctx, slot);
    int frames = 0;
    if ((rc = band_detect_launch(ctx, s, mask, img, &frames)) != WM_OK) return rc;
    launch_band_pick(s.stream, frames, s.d_raw + ctx->max_frames, 3, sums_dev);
    return launch_check(ctx, s);
}

int wm_band_corr_dev(wm_ctx* ctx, const double* sums_dev, int frames, float* corr_dev, int slot)
{
    if (!ctx || !sums_dev || !corr_dev || frames < 1 || frames > ctx->max_frames) return fail(ctx, WM_ERR_BAD_ARG, "wm_band_corr_dev: bad arguments");
    BAND_SLOT(ctx, slot);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    launch_band_corr(s.stream, frames, sums_dev, s.d_status, corr_dev);
    return launch_check(ctx, s);
}

int wm_sync(wm_ctx* ctx, int slot)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    Slot* sp; bool unused_sync_flag;
    int rc = get_slot(ctx, slot, &sp, &unused_sync_flag);
    if (rc != WM_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return do_sync(ctx, *sp);
}

int wm_set_stream(wm_ctx* ctx, int slot, void* hip_stream)
{
    if (!ctx || slot < 0 || slot >= ctx->nslots) return fail(ctx, WM_ERR_BAD_ARG, "bad slot");
    Slot& s = ctx->slots[slot];
    if (!s.pending.empty()) { int rc = do_sync(ctx, s); if (rc < 0) return rc; }
    const hipStream_t next = hip_stream ? (hipStream_t)hip_stream : s.own;
    if (next != s.stream) {
        // the slot's scratch (coefficients, status words, raw totals, tickets) may still be in use by work enqueued on the old
        // stream that nobody waits for (the wm_band_*_dev phases only enqueue): the new stream starts behind it
        HIPCHK(ctx, hipSetDevice(ctx->device));
        hipEvent_t ev;
        HIPCHK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, s.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(next, ev, 0);
        (void)hipEventDestroy(ev);  // (released once the recorded work has completed)
        if (e != hipSuccess) return fail(ctx, WM_ERR_RUNTIME, std::string("wm_set_stream: ") + hipGetErrorString(e));
        s.stream = next;
    }
    return WM_OK;
}

void* wm_get_stream(wm_ctx* ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= ctx->nslots) return nullptr;
    return (void*)ctx->slots[slot].stream;
}

void* wm_dev_alloc(int device, size_t bytes)
{
    void* p = nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return p;
}
void wm_dev_free(void* p) { if (p) (void)hipFree(p); }
int wm_memcpy_h2d(void* dst, const void* src, size_t bytes)
{
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? WM_OK : WM_ERR_RUNTIME;
}
int wm_memcpy_d2h(void* dst, const void* src, size_t bytes)
{
    return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? WM_OK : WM_ERR_RUNTIME;
}
int wm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void* wm_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void wm_host_free(void* p) { if (p) (void)hipHostFree(p); }

int wm_selftest_nvf_quotient(int device, int variant, uint32_t bits_lo, uint32_t bits_hi, unsigned long long* mismatches, uint32_t* first_bad)
{
    if (variant < 0 || variant > 3 || bits_lo > bits_hi) return WM_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WM_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) device = 0;
    if (hipSetDevice(device) != hipSuccess) return WM_ERR_NO_DEVICE;
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, 16) != hipSuccess) return WM_ERR_ALLOC;
    const unsigned long long init[2] = {0ull, ~0ull};
    unsigned long long got[2] = {0ull, ~0ull};
    int rc = WM_OK;
    if (hipMemcpy(d, init, 16, hipMemcpyHostToDevice) != hipSuccess) rc = WM_ERR_RUNTIME;
    if (rc == WM_OK) {
        launch_selftest_quot(nullptr, variant, bits_lo, bits_hi, d);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemcpy(got, d, 16, hipMemcpyDeviceToHost) != hipSuccess) rc = WM_ERR_RUNTIME;
    }
    (void)hipFree(d);
    if (mismatches) *mismatches = got[0];
    if (first_bad) *first_bad = (uint32_t)got[1];
    return rc;
}

int wm_membench(int device, int kind, size_t bytes, double seconds, double* mean_us, int* launches)
{
    if (kind < 0 || kind > 5 || bytes < 4096 || !(seconds >= 0.0) || seconds > 30.0) return WM_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WM_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) device = 0;
    if (hipSetDevice(device) != hipSuccess) return WM_ERR_NO_DEVICE;
    const size_t n16 = bytes / 16;
    void *src = nullptr, *dst = nullptr;
    unsigned long long* sink = nullptr;
    hipStream_t st = nullptr;
    constexpr int NEV = 32;   // launches in flight between two waits
    hipEvent_t ea[NEV], eb[NEV];
    int nev = 0, rc = WM_OK;
    double total_ms = 0.0;
    int count = 0;
    do {
        if (kind % 3 != 0 && hipMalloc(&src, n16 * 16) != hipSuccess) { rc = WM_ERR_ALLOC; break; }
        if (kind % 3 != 2 && hipMalloc(&dst, n16 * 16) != hipSuccess) { rc = WM_ERR_ALLOC; break; }
        if (hipMalloc((void**)&sink, 8) != hipSuccess) { rc = WM_ERR_ALLOC; break; }
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = WM_ERR_RUNTIME; break; }
        if (src && hipMemsetAsync(src, 0x3c, n16 * 16, st) != hipSuccess) { rc = WM_ERR_RUNTIME; break; }
        if (hipMemsetAsync(sink, 0, 8, st) != hipSuccess) { rc = WM_ERR_RUNTIME; break; }
        for (; nev < NEV; ++nev)
            if (hipEventCreate(&ea[nev]) != hipSuccess || hipEventCreate(&eb[nev]) != hipSuccess) { rc = WM_ERR_RUNTIME; break; }
        if (rc != WM_OK) break;
        for (int w = 0; w < 3; ++w) launch_membench(st, kind, src, dst, n16, sink, ea[0], eb[0]);  // warm-up
        if (hipStreamSynchronize(st) != hipSuccess) { rc = WM_ERR_RUNTIME; break; }
        const auto t0 = std::chrono::steady_clock::now();
        do {
            for (int k = 0; k < NEV; ++k) launch_membench(st, kind, src, dst, n16, sink, ea[k], eb[k]);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = WM_ERR_RUNTIME; break; }
            for (int k = 0; k < NEV; ++k) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ea[k], eb[k]) == hipSuccess) { total_ms += ms; ++count; }
            }
        } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
    } while (false);
    for (int k = 0; k < nev; ++k) { (void)hipEventDestroy(ea[k]); (void)hipEventDestroy(eb[k]); }
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(sink);
    if (rc != WM_OK) { (void)hipGetLastError(); return rc; }
    if (mean_us) *mean_us = count ? 1e3 * total_ms / count : 0.0;
    if (launches) *launches = count;
    return WM_OK;
}

int wm_rows(const wm_ctx* ctx) { return ctx ? ctx->rows : 0; }
int wm_cols(const wm_ctx* ctx) { return ctx ? ctx->cols : 0; }
int wm_p(const wm_ctx* ctx) { return ctx ? ctx->p : 0; }
float wm_strength_factor(const wm_ctx* ctx) { return ctx ? ctx->sF : 0.0f; }
int wm_device(const wm_ctx* ctx) { return ctx ? ctx->device : -1; }
const float* wm_w_device(const wm_ctx* ctx) { return ctx && ctx->w ? ctx->w->d_w : nullptr; }

int wm_prof_enable(wm_ctx* ctx, int on)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    if (!on) { int rc = prof_collect(ctx); if (rc != WM_OK) return rc; }
    ctx->prof = on != 0;
    return WM_OK;
}
int wm_prof_reset(wm_ctx* ctx)
{
    if (!ctx) return WM_ERR_BAD_ARG;
    int rc = prof_collect(ctx);
    for (int k = 0; k < K_COUNT; ++k) { ctx->prof_n[k] = 0; ctx->prof_ms[k] = 0.0; }
    return rc;
}
int wm_prof_kernel_count(void) { return K_COUNT; }
const char* wm_prof_kernel_name(int k) { return k >= 0 && k < K_COUNT ? kKernelNames[k] : ""; }
int wm_prof_get(wm_ctx* ctx, int k, uint64_t* launches, double* total_ms)
{
    if (!ctx || k < 0 || k >= K_COUNT) return WM_ERR_BAD_ARG;
    int rc = prof_collect(ctx);
    if (launches) *launches = ctx->prof_n[k];
    if (total_ms) *total_ms = ctx->prof_ms[k];
    return rc;
}

const char* wm_strerror(int code)
{
    switch (code) {
        case WM_OK: return "ok";
        case WM_UNSOLVABLE: return "prediction system not solvable (image passed through / correlation 0)";
        case WM_ERR_BAD_P: return "Wrong p parameter";
        case WM_ERR_W_OPEN: return "Error opening file for Random noise W array";
        case WM_ERR_W_SIZE: return "Error: W file total elements != image dimensions!";
        case WM_ERR_RUNTIME: return "HIP runtime or kernel failure";
        case WM_ERR_BAD_ARG: return "bad argument";
        case WM_ERR_NO_DEVICE: return "no usable HIP device (the engine has no CPU fallback)";
        case WM_ERR_ALLOC: return "allocation failed";
        case WM_ERR_PSNR: return "PSNR must be a positive number";
        case WM_ERR_BUSY: return "too many un-synced operations on this slot";
        default: return "unknown error";
    }
}
const char* wm_last_error(const wm_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
const char* wm_version(void) { return "wm-hip 0.1 (gfx950)"; }

}  // extern "C"
