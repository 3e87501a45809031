// wm_stream: a video stream sharded frame-parallel over the GPUs of one node, host side in C++.
//
// The reference embeds and detects frame by frame on ONE device (main.cpp:319-340: testForVideo's loops over
// embedWatermarkFrame / detectFrameWatermark); frames are independent, so here frame i goes to device i mod G:
//   * one wm_ctx and one host thread per device (W is uploaded once per device), `slots` batches in flight per device;
//   * a batch = `batch` frames of that device's shard: staged in from pinned host memory, embedded, detected on the device
//     copy of the output (WM_MEM_SLOT_OUT), staged out -- one trip over the host link each way;
//   * per-frame detector scores are gathered with RCCL (ncclCommInitAll + one ncclAllGather of `batch` floats per round)
//     or, with --gather host (and always when a device is listed twice, which RCCL refuses), handed over in host memory.
//     Threading: after ncclCommInitAll every communicator is driven by exactly ONE thread (its device's worker), which is the
//     "different threads" form rccl.h describes -- ncclGroupStart/End is for one thread driving several devices
//     (rccl.h "Group semantics") and is not needed.  Every worker runs the same number of rounds, so the collectives pair up.
//     The wait for a gather is bounded and watches the shared error state: a worker that left early on an error (or a gather
//     older than --gather-timeout seconds) makes the others ncclCommAbort their communicator instead of waiting for ever.
//     REVIEWED, NOT RUN with more than one device: the build box has one GPU and RCCL refuses duplicate devices;
//   * the main thread is the in-order re-sequencer: frames and scores leave in stream order whatever device finished first.
// Input: synthetic u8 Y planes (a counter hash, the same for any device list) or a raw Y-plane file (--in, frames of
// rows*cols bytes).  Output: optional raw Y-plane file (--out) and a score per line (--scores), plus a summary line.
//
//   * host placement: before it creates its context or pins a byte, a device's worker thread pins itself to the CPUs of the
//     NUMA node its GPU hangs off (placement.hpp: hipDeviceGetPCIBusId -> sysfs numa_node / local_cpulist), so its pinned
//     staging buffers are first-touched next to the GPU; --pin 0 switches it off (default: on when more than one device);
//     the summary line names every device's NUMA node.  `wm_stream --placement-of <pci> [--sysfs <root>]` prints the plan
//     for one PCI address without touching a GPU (the CPU test of this code path).
//
//   * --ring N: the source is a ring of N distinct frames held in PINNED memory (what a decoder that writes into pinned buffers
//     looks like): frame i of the stream has the content of source frame i mod N, every device keeps the batches of its shard
//     of the ring pinned and hands them to wm_embed as they are -- no per-frame host copy, so the tool runs at the engine's
//     host-staged rate instead of one memcpy thread's (N is rounded down to a multiple of devices x batch).
//
//   wm_stream --devices 0,1,2,3 --rows 2160 --cols 3840 --frames 960 --batch 8 [--slots 3] [--mask ME|NVF] [--psnr 40]
//             [--gather rccl|host] [--interval 1] [--in y.raw] [--out y_marked.raw] [--scores scores.txt] [--pin 0|1] [--ring N]
#include "../../../include/wm.h"
#include "placement.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

struct Args {
    std::vector<int> devices{0};
    int rows = 1080, cols = 1920, frames = 240, batch = 8, slots = 3, interval = 1, mask = WM_MASK_ME;
    float psnr = 40.0f;
    std::string gather = "rccl", in, out, scores;
    double gather_timeout = 30.0;  // seconds a worker waits for one RCCL score gather before it aborts its communicator
    uint32_t seed = 28390211u;     // W seed (samples/make_w.bat)
};

// a finished batch on its way to the re-sequencer
struct Done {
    int worker, buf;                 // output buffer of that worker (returned to its pool after the frames left)
    std::vector<long long> frame;    // stream indices
    std::vector<float> a, corr;
    std::vector<int> marked;         // watermark_interval gating: frames that were not marked pass through
};

struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    std::map<long long, std::pair<Done*, int>> ready;  // frame index -> (batch, position in batch)
    std::vector<std::vector<int>> free_bufs;           // per worker
    std::string error;
    int ready_workers = 0;                              // workers whose set-up is done
    std::chrono::steady_clock::time_point stream_t0{};  // when the last of them was
};

// Synthetic Y planes: a handful of base frames (smooth pattern + texture noise, generated once) and, for frame f, base
// frame f % NB shifted circularly by f / NB columns and rows -- distinct frames at memcpy cost, the same for any device list.
struct Source {
    static constexpr int NB = 8;
    int rows, cols;
    std::vector<std::vector<uint8_t>> base;
    Source(int r, int c) : rows(r), cols(c), base(NB, std::vector<uint8_t>((size_t)r * c))
    {
        std::vector<float> sr(rows), cr(cols), s2(rows + 2 * cols);
        for (int i = 0; i < rows; ++i) sr[i] = std::sin(i * 0.0648f);
        for (int i = 0; i < cols; ++i) cr[i] = std::cos(i * 0.103f);
        for (int i = 0; i < rows + 2 * cols; ++i) s2[i] = std::sin(i * 0.01615f);
        for (int b = 0; b < NB; ++b)
            for (int r2 = 0; r2 < rows; ++r2)
                for (int c2 = 0; c2 < cols; ++c2) {
                    float v = 128.0f + (40.0f + 2.0f * b) * sr[r2] * cr[c2] + 36.0f * s2[r2 + 2 * c2] +
                              44.0f * (unit(hash32((uint32_t)b * 0x9E3779B9u + (uint32_t)(r2 * cols + c2))) - 0.5f);
                    base[b][(size_t)r2 * cols + c2] = (uint8_t)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));
                }
    }
    void get(uint8_t* dst, long long f) const
    {
        const std::vector<uint8_t>& src = base[f % NB];
        const int sh = (int)((f / NB) * 7 % cols), rsh = (int)((f / NB) * 3 % rows);
        for (int r2 = 0; r2 < rows; ++r2) {
            const uint8_t* row = src.data() + (size_t)((r2 + rsh) % rows) * cols;
            std::memcpy(dst + (size_t)r2 * cols, row + sh, (size_t)(cols - sh));
            std::memcpy(dst + (size_t)r2 * cols + (cols - sh), row, (size_t)sh);
        }
    }
};

#define CHK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail(std::string(#x) + ": " + hipGetErrorString(e_)); return; } } while (0)
#define CHK_NCCL(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fail(std::string(#x) + ": " + ncclGetErrorString(e_)); return; } } while (0)

int main(int argc, char** argv)
{
    Args A;
    int pin_opt = -1;  // -1: pin when more than one device
    int ring_opt = 0;  // > 0: a pinned ring of this many distinct source frames (per node)
    std::string placement_of, sysfs_root = "/sys";
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--devices") {
            A.devices.clear();
            size_t p = 0;
            while (p <= v.size()) { size_t q = v.find(',', p); if (q == std::string::npos) q = v.size(); A.devices.push_back(std::atoi(v.substr(p, q - p).c_str())); p = q + 1; }
        } else if (k == "--rows") A.rows = std::atoi(v.c_str());
        else if (k == "--cols") A.cols = std::atoi(v.c_str());
        else if (k == "--frames") A.frames = std::atoi(v.c_str());
        else if (k == "--batch") A.batch = std::atoi(v.c_str());
        else if (k == "--slots") A.slots = std::atoi(v.c_str());
        else if (k == "--interval") A.interval = std::atoi(v.c_str());
        else if (k == "--psnr") A.psnr = (float)std::atof(v.c_str());
        else if (k == "--mask") A.mask = v == "NVF" ? WM_MASK_NVF : WM_MASK_ME;
        else if (k == "--gather") A.gather = v;
        else if (k == "--gather-timeout") A.gather_timeout = std::atof(v.c_str());
        else if (k == "--seed") A.seed = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (k == "--in") A.in = v;
        else if (k == "--out") A.out = v;
        else if (k == "--scores") A.scores = v;
        else if (k == "--pin") pin_opt = std::atoi(v.c_str());
        else if (k == "--ring") ring_opt = std::atoi(v.c_str());
        else if (k == "--placement-of") placement_of = v;
        else if (k == "--sysfs") sysfs_root = v;
        else { std::fprintf(stderr, "wm_stream: unknown option %s\n", k.c_str()); return 2; }
    }
    if (!placement_of.empty()) {
        // no GPU is touched: the placement a worker would choose for the device at this PCI address
        const wmplace::Plan pl = wmplace::plan_for_pci(sysfs_root, placement_of, wmplace::allowed_cpus());
        std::printf("{\"pci\": \"%s\", \"valid\": %s, \"numa_node\": %d, \"cpus\": \"%s\"}\n", placement_of.c_str(), pl.valid ? "true" : "false",
                    pl.numa_node, wmplace::format_cpulist(pl.cpus).c_str());
        return 0;
    }
    const int G = (int)A.devices.size();
    if (G < 1 || A.batch < 1 || A.slots < 1 || A.frames < 1 || A.interval < 1) { std::fprintf(stderr, "wm_stream: bad arguments\n"); return 2; }
    const int R = A.rows, Cc = A.cols, B = A.batch;
    const size_t n = (size_t)R * Cc;
    bool dup = false;
    for (int i = 0; i < G; ++i) for (int j = i + 1; j < G; ++j) dup |= A.devices[i] == A.devices[j];
    const bool use_rccl = A.gather == "rccl" && !dup;
    if (A.gather == "rccl" && dup) std::fprintf(stderr, "wm_stream: a device is listed twice, RCCL refuses that: scores are gathered in host memory\n");

    // W: generated on every device from the seed (wm_create_generated: element (r,c) is a function of (seed, r, c) only, so all
    // devices hold the same matrix with no file, upload or broadcast)
    FILE* fin = A.in.empty() ? nullptr : std::fopen(A.in.c_str(), "rb");
    if (!A.in.empty() && !fin) { std::fprintf(stderr, "wm_stream: cannot open %s\n", A.in.c_str()); return 2; }
    std::mutex fin_mu;
    const Source source(R, Cc);

    // rounds: in every round each device takes one batch of its shard; the last round may be short / empty for some devices
    const long long per_round = (long long)G * B;
    const int rounds = (int)((A.frames + per_round - 1) / per_round);
    const int nbuf = A.slots + 2;  // output buffers per worker: `slots` in flight + those waiting in the re-sequencer
    // pinned ring: batches of the ring per device (0: frames are copied into the slot's input buffer every round)
    const int ring_batches = (ring_opt > 0 && fin == nullptr) ? (int)std::max<long long>(A.slots, ring_opt / per_round) : 0;

    Shared S;
    S.free_bufs.resize(G);
    std::vector<std::vector<uint8_t*>> out_tab(G);  // per worker: its pinned output buffers (written once, under S.mu)
    std::vector<wm_ctx*> ctxs(G, nullptr);
    std::vector<ncclComm_t> comms(G);
    if (use_rccl) {
        ncclResult_t e = ncclCommInitAll(comms.data(), G, A.devices.data());
        if (e != ncclSuccess) { std::fprintf(stderr, "wm_stream: ncclCommInitAll: %s\n", ncclGetErrorString(e)); return 1; }
    }
    std::vector<double> busy_s(G, 0.0);
    std::vector<int> numa_of(G, -1);    // NUMA node every worker pinned itself to (-1: not pinned)
    const bool pin = pin_opt < 0 ? G > 1 : pin_opt != 0;
    const std::vector<int> allowed = wmplace::allowed_cpus();  // (of the process, read before any worker narrows its own mask)
    std::vector<char> aborted(G, 0);  // communicators already destroyed by ncclCommAbort
    auto worker = [&](int g) {
        auto fail = [&](const std::string& m) { std::lock_guard<std::mutex> lk(S.mu); if (S.error.empty()) S.error = "device " + std::to_string(A.devices[g]) + ": " + m; S.cv.notify_all(); };
        if (pin) {
            // first of all: live next to the GPU (the context's mapped result records and the pinned ring below are first-touched
            // by this thread, and the threads the runtime starts for it inherit the mask)
            char bus[64] = "";
            if (hipDeviceGetPCIBusId(bus, sizeof bus, A.devices[g]) == hipSuccess) {
                const wmplace::Plan pl = wmplace::plan_for_pci(sysfs_root, bus, allowed);
                if (wmplace::apply_to_this_thread(pl)) numa_of[g] = pl.numa_node;
            }
        }
        wm_ctx* ctx = nullptr;
        int rc = wm_create_generated(&ctx, A.devices[g], R, Cc, 3, A.psnr, A.seed);
        if (rc != WM_OK) { fail(std::string("wm_create_generated: ") + wm_strerror(rc)); return; }
        if ((rc = wm_configure(ctx, A.slots, B)) != WM_OK) { fail(std::string("wm_configure: ") + wm_strerror(rc)); return; }
        CHK_HIP(hipSetDevice(A.devices[g]));
        std::vector<uint8_t*> hin(ring_batches > 0 ? ring_batches : A.slots), hout(nbuf);
        for (auto& p : hin) if (!(p = (uint8_t*)wm_host_alloc(n * B))) { fail("pinned allocation"); return; }
        for (auto& p : hout) if (!(p = (uint8_t*)wm_host_alloc(n * B))) { fail("pinned allocation"); return; }
        if (ring_batches > 0) {
            // this device's shard of the ring, filled once: ring batch k holds the frames of round k (k < ring_batches)
            for (int k = 0; k < ring_batches; ++k)
                for (int j = 0; j < B; ++j) source.get(hin[k] + (size_t)j * n, ((long long)k * B + j) * G + g);
        }
        { std::lock_guard<std::mutex> lk(S.mu); for (int b = 0; b < nbuf; ++b) S.free_bufs[g].push_back(b); out_tab[g] = hout; ctxs[g] = ctx; }
        hipStream_t gs = nullptr;
        float *d_send = nullptr, *d_recv = nullptr, *h_recv = nullptr;
        if (use_rccl) {
            CHK_HIP(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
            CHK_HIP(hipMalloc((void**)&d_send, B * sizeof(float)));
            CHK_HIP(hipMalloc((void**)&d_recv, (size_t)G * B * sizeof(float)));
            CHK_HIP(hipHostMalloc((void**)&h_recv, (size_t)G * B * sizeof(float), hipHostMallocDefault));
        }
        struct InFlight { Done* d = nullptr; std::vector<float> a, corr; std::vector<int> st; int count = 0; const uint8_t* in = nullptr; };
        std::vector<InFlight> fl(A.slots);
        auto retire = [&](int slot) {
            InFlight& f = fl[slot];
            if (!f.d) return true;
            if (f.count > 0 && wm_sync(ctx, slot) < 0) { fail(std::string("wm_sync: ") + wm_last_error(ctx)); return false; }
            for (int j = 0; j < (int)f.d->frame.size(); ++j) {
                f.d->a[j] = f.a[j]; f.d->corr[j] = f.corr[j];
                if (!f.d->marked[j]) {
                    // outside the watermark interval (main.cpp:346,395): the frame leaves as it came, without a score
                    std::memcpy(hout[f.d->buf] + (size_t)j * n, f.in + (size_t)j * n, n);
                    f.d->a[j] = 0.0f; f.d->corr[j] = 0.0f; f.corr[j] = 0.0f;
                }
            }
            if (use_rccl) {
                // the round's scores of every device through RCCL; what leaves here is this device's slice of the gathered vector
                std::vector<float> send(B, 0.0f);
                for (int j = 0; j < (int)f.d->frame.size(); ++j) send[j] = f.corr[j];
                if (hipMemcpyAsync(d_send, send.data(), B * sizeof(float), hipMemcpyHostToDevice, gs) != hipSuccess ||
                    ncclAllGather(d_send, d_recv, B, ncclFloat, comms[g], gs) != ncclSuccess ||
                    hipMemcpyAsync(h_recv, d_recv, (size_t)G * B * sizeof(float), hipMemcpyDeviceToHost, gs) != hipSuccess) { fail("RCCL score gather"); return false; }
                // bounded wait: the gather completes only if all G workers reach it.  One that has failed never will, so watch
                // the shared error state and a deadline, and abort the communicator rather than sit in the collective for ever
                const auto g0 = std::chrono::steady_clock::now();
                for (;;) {
                    const hipError_t q = hipStreamQuery(gs);
                    if (q == hipSuccess) break;
                    bool other_failed;
                    { std::lock_guard<std::mutex> lk(S.mu); other_failed = !S.error.empty(); }
                    ncclResult_t async = ncclSuccess;
                    ncclCommGetAsyncError(comms[g], &async);
                    const bool late = std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count() > A.gather_timeout;
                    if (q != hipErrorNotReady || other_failed || late || (async != ncclSuccess && async != ncclInProgress)) {
                        ncclCommAbort(comms[g]);
                        aborted[g] = 1;
                        fail(other_failed ? "RCCL score gather abandoned (another device failed)" : late ? "RCCL score gather timed out" : "RCCL score gather failed");
                        return false;
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
                for (int j = 0; j < (int)f.d->frame.size(); ++j) f.d->corr[j] = h_recv[(size_t)g * B + j];
            }
            {
                std::lock_guard<std::mutex> lk(S.mu);
                for (int j = 0; j < (int)f.d->frame.size(); ++j) S.ready[f.d->frame[j]] = {f.d, j};
                S.cv.notify_all();
            }
            f.d = nullptr;
            return true;
        };
        // set-up is done (context, W, pinned buffers, ring): wait for the other devices, so that the stream clock measures the
        // stream and not the slowest hipHostMalloc
        {
            std::unique_lock<std::mutex> lk(S.mu);
            if (++S.ready_workers == G) { S.stream_t0 = std::chrono::steady_clock::now(); S.cv.notify_all(); }
            S.cv.wait(lk, [&] { return S.ready_workers >= G || !S.error.empty(); });
            if (!S.error.empty()) return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int round = 0; round < rounds; ++round) {
            const int slot = round % A.slots;
            if (!retire(slot)) return;
            // frames of this batch: stream index = (round * B + j) * G + g
            Done* d = new Done;
            d->worker = g;
            for (int j = 0; j < B; ++j) {
                const long long fi = ((long long)round * B + j) * G + g;
                if (fi < A.frames) d->frame.push_back(fi);
            }
            const int cnt = (int)d->frame.size();
            d->a.assign(cnt, 0.0f); d->corr.assign(cnt, 0.0f); d->marked.assign(cnt, 1);
            {
                std::unique_lock<std::mutex> lk(S.mu);
                S.cv.wait(lk, [&] { return !S.free_bufs[g].empty() || !S.error.empty(); });
                if (!S.error.empty()) return;
                d->buf = S.free_bufs[g].back(); S.free_bufs[g].pop_back();
            }
            InFlight& f = fl[slot];
            f.d = d; f.count = cnt; f.a.assign(B, 0.0f); f.corr.assign(B, 0.0f); f.st.assign(B, 0);
            uint8_t* const inbuf = ring_batches > 0 ? hin[round % ring_batches] : hin[slot];  // (ring: read-only, filled once)
            f.in = inbuf;
            if (cnt > 0) {
                for (int j = 0; j < cnt; ++j) {
                    uint8_t* dst = inbuf + (size_t)j * n;
                    if (ring_batches == 0) {  // (ring batches were filled once, before the first round)
                        if (fin) {
                            std::lock_guard<std::mutex> lk(fin_mu);
                            if (std::fseek(fin, (long)(d->frame[j] * (long long)n), SEEK_SET) != 0 || std::fread(dst, 1, n, fin) != n) { fail("short read on the input file"); return; }
                        } else source.get(dst, d->frame[j]);
                    }
                    d->marked[j] = d->frame[j] % A.interval == 0 ? 1 : 0;  // main.cpp:346: framesCount % watermarkInterval
                }
                // a short last batch is padded with copies of its first frame: every frame of the stream then runs in a launch of
                // exactly B frames (the launch geometry, and with it the grouping of the partial sums, depends on the frame
                // count), so its result does not depend on how many devices share the stream
                if (ring_batches == 0) for (int j = cnt; j < B; ++j) std::memcpy(inbuf + (size_t)j * n, inbuf, n);
                wm_plane pin{inbuf, R, Cc, 1, WM_U8, WM_MEM_HOST, B, Cc, 0, (int64_t)n};
                wm_plane pout{hout[d->buf], R, Cc, 1, WM_U8, WM_MEM_HOST, B, Cc, 0, (int64_t)n};
                wm_plane pslot{nullptr, R, Cc, 1, WM_U8, WM_MEM_SLOT_OUT, B, Cc, 0, (int64_t)n};
                rc = wm_embed(ctx, A.mask, &pin, &pin, &pout, f.a.data(), f.st.data(), slot);
                if (rc == WM_OK) rc = wm_detect(ctx, A.mask, &pslot, f.corr.data(), nullptr, slot);
                if (rc < 0) { fail(std::string("enqueue: ") + wm_last_error(ctx)); return; }
            }
        }
        for (int s = 0; s < A.slots; ++s)
            if (!retire((rounds + s) % A.slots)) return;
        busy_s[g] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // the pinned buffers and the context outlive this thread: main emits from them and destroys the contexts after the join
    };

    const auto T0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) th.emplace_back(worker, g);

    // ---- the in-order re-sequencer ---------------------------------------------------------------------------------
    FILE* fout = A.out.empty() ? nullptr : std::fopen(A.out.c_str(), "wb");
    FILE* fsc = A.scores.empty() ? nullptr : std::fopen(A.scores.c_str(), "w");
    uint64_t checksum = 1469598103934665603ull;  // FNV-1a over the emitted frames, in stream order
    double sum_corr = 0.0;
    std::map<Done*, int> left;  // frames of a batch not emitted yet
    for (long long next = 0; next < A.frames; ++next) {
        Done* d; int j; const uint8_t* src;
        {
            std::unique_lock<std::mutex> lk(S.mu);
            S.cv.wait(lk, [&] { return S.ready.count(next) || !S.error.empty(); });
            if (!S.error.empty()) {
                // the workers notice the error themselves (their waits watch it); leave without running destructors under them
                std::fprintf(stderr, "wm_stream: %s\n", S.error.c_str());
                std::fflush(nullptr);
                for (auto& t : th) t.detach();
                std::_Exit(1);
            }
            d = S.ready[next].first; j = S.ready[next].second;
            S.ready.erase(next);
            src = out_tab[d->worker][d->buf] + (size_t)j * n;
            if (!left.count(d)) left[d] = (int)d->frame.size();
        }
        if (fout) std::fwrite(src, 1, n, fout);
        // (sampled, one byte per ~4 KiB: a 97-byte stride touched every other cache line of the frame -- 4 MB of host reads per 4K
        // frame in the ONE re-sequencer thread, which capped the tool at ~2 k frames/s)
        for (size_t i = 0; i < n; i += 4099) { checksum ^= src[i]; checksum *= 1099511628211ull; }
        if (fsc) std::fprintf(fsc, "%lld %.9g %.9g %d\n", next, (double)d->a[j], (double)d->corr[j], d->marked[j]);
        sum_corr += d->corr[j];
        if (--left[d] == 0) {
            std::lock_guard<std::mutex> lk(S.mu);
            S.free_bufs[d->worker].push_back(d->buf);
            left.erase(d);
            delete d;
            S.cv.notify_all();
        }
    }
    for (auto& t : th) t.join();
    const auto T1 = std::chrono::steady_clock::now();
    const double wall = std::chrono::duration<double>(T1 - T0).count();
    const double stream_s = std::chrono::duration<double>(T1 - S.stream_t0).count();  // first enqueue .. last frame emitted
    if (fout) std::fclose(fout);
    if (fsc) std::fclose(fsc);
    if (fin) std::fclose(fin);
    for (auto c : ctxs) wm_destroy(c);
    if (use_rccl) for (int g = 0; g < G; ++g) if (!aborted[g]) ncclCommDestroy(comms[g]);
    std::string devs;
    for (int g = 0; g < G; ++g) devs += (g ? "," : "") + std::to_string(A.devices[g]);
    std::string numas;
    for (int g = 0; g < G; ++g) numas += (g ? "," : "") + std::to_string(numa_of[g]);
    std::printf("{\"devices\": \"%s\", \"rows\": %d, \"cols\": %d, \"frames\": %d, \"batch\": %d, \"slots\": %d, \"mask\": \"%s\", \"gather\": \"%s\", "
                "\"frames_per_s\": %.1f, \"stream_s\": %.3f, \"wall_s\": %.3f, \"mean_corr\": %.7f, \"checksum\": \"%016llx\", \"pinned_to_numa_node\": \"%s\", \"ring_batches_per_device\": %d}\n",
                devs.c_str(), R, Cc, A.frames, B, A.slots, A.mask == WM_MASK_ME ? "ME" : "NVF", use_rccl ? "rccl" : "host", A.frames / stream_s, stream_s, wall,
                sum_corr / A.frames, (unsigned long long)checksum, numas.c_str(), ring_batches);
    return 0;
}

