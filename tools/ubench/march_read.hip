// dev micro-benchmark: what does the ACCESS SHAPE of the strip march cost against a plain stream?
// The read-only sweeps (k_gram, k_me_stats, k_detect) move 5.1-5.2 TB/s on a box whose pure-read kernel reaches 6.4-6.7 TB/s in
// its best shapes (tools/membench_sweep.py).  This program reads F planes of R x C f32 the way the sweeps do -- one wave per
// (strip of 256 columns, segment of rps rows), PF rows of 1 KiB in flight per wave, a trivial sum per row -- and varies what
// the waves of a block are and how blocks are ordered:
//   map 0: a block = 4 vertically adjacent SEGMENTS of one strip (the sweeps without frame quads), tiles strip-fastest
//   map 1: a block = 4 FRAMES of one (strip, segment) (frame quads: the sweeps that read W)
//   map 2: a block = 4 horizontally adjacent STRIPS of one (frame, segment): the block's waves read 4 KiB contiguous per row
//   map 3: as map 2 with 8 waves per block (8 KiB contiguous per row)
//   map 4: one wave per block, blocks in (frame, segment, strip) order, strip fastest
// and the block order: frame-fastest (the sweeps that read W) or tile-fastest, with or without the XCD remap.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/march_read.hip -o tools/ubench/march_read
//   tools/ubench/march_read [rows cols frames]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct P {
    int rows, cols, frames, rps, nsegs, nstrips, map, frame_fastest, remap, wpb, linear;
    long long pitch, fstride;
};

__device__ __forceinline__ int xcd_remap(int b, int nblk)
{
    const int per = nblk >> 3, rem = nblk & 7;
    const int x = b & 7, i = b >> 3;
    return x * per + (x < rem ? x : rem) + i;
}

template <int PF>
__global__ __launch_bounds__(512) void k_march(const float* __restrict__ src, float* __restrict__ out, P p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = gridDim.x;
    const int b = p.remap ? xcd_remap(blockIdx.x, nblk) : (int)blockIdx.x;
    int strip, seg, frame;
    if (p.map == 0) {
        const int seggroups = (p.nsegs + 3) / 4, ntiles = p.nstrips * seggroups;
        int tile, f;
        if (p.frame_fastest) { tile = b / p.frames; f = b - tile * p.frames; } else { f = b / ntiles; tile = b - f * ntiles; }
        strip = tile % p.nstrips; seg = (tile / p.nstrips) * 4 + wave; frame = f;
    } else if (p.map == 1) {
        const int nq = (p.frames + 3) / 4;
        const int tile = b / nq;
        frame = 4 * (b - tile * nq) + wave; strip = tile % p.nstrips; seg = tile / p.nstrips;
    } else if (p.map == 2 || p.map == 3) {
        const int sg = (p.nstrips + p.wpb - 1) / p.wpb, ntiles = sg * p.nsegs;
        int tile, f;
        if (p.frame_fastest) { tile = b / p.frames; f = b - tile * p.frames; } else { f = b / ntiles; tile = b - f * ntiles; }
        strip = (tile % sg) * p.wpb + wave; seg = tile / sg; frame = f;
    } else {
        const int ntiles = p.nstrips * p.nsegs;
        int tile, f;
        if (p.frame_fastest) { tile = b / p.frames; f = b - tile * p.frames; } else { f = b / ntiles; tile = b - f * ntiles; }
        strip = tile % p.nstrips; seg = tile / p.nstrips; frame = f;
    }
    if (strip >= p.nstrips || seg >= p.nsegs || frame >= p.frames) return;
    const int rs = seg * p.rps, re = min(rs + p.rps, p.rows), n = re - rs;
    const float4* base = reinterpret_cast<const float4*>(src + (long long)frame * p.fstride + (long long)rs * p.pitch + strip * 256) + lane;
    // linear: the wave's "rows" are consecutive 1 KiB chunks (its tile is one contiguous piece of memory): what a per-wave
    // sequential stream reaches with the same loop
    const long long rstep = p.linear ? 64 : p.pitch / 4;
    if (p.linear) base = reinterpret_cast<const float4*>(src + (long long)frame * p.fstride) + ((long long)(seg * p.nstrips + strip) * p.rps) * 64 + lane;
    float4 pre[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) pre[q] = base[(long long)min(q, n - 1) * rstep];
    float acc = 0.f;
    int i = 0;
    for (; i + PF <= n; i += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const float4 v = pre[q];
            acc += v.x + v.y + v.z + v.w;
            asm volatile("" : "+v"(acc));
            pre[q] = base[(long long)min(i + q + PF, n - 1) * rstep];
        }
    }
    for (int q = 0; i + q < n; ++q) { const float4 v = pre[q]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1234.5f) out[blockIdx.x] = acc;
}

// 512-column strips: a wave-row is 2 KiB (two 16-byte loads per lane, 1 KiB apart so that each instruction stays a 1 KiB burst)
template <int PF>
__global__ __launch_bounds__(256) void k_march_wide(const float* __restrict__ src, float* __restrict__ out, P p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = p.remap ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int ns2 = (p.nstrips + 1) / 2;
    const int seggroups = (p.nsegs + 3) / 4, ntiles = ns2 * seggroups;
    const int f = b / ntiles, tile = b - f * ntiles;
    const int strip2 = tile % ns2, seg = (tile / ns2) * 4 + wave;
    if (seg >= p.nsegs || f >= p.frames) return;
    const int rs = seg * p.rps, re = min(rs + p.rps, p.rows), n = re - rs;
    const int c1 = min(strip2 * 512 + 256, p.cols - 256);
    const float4* b0 = reinterpret_cast<const float4*>(src + (long long)f * p.fstride + (long long)rs * p.pitch + strip2 * 512) + lane;
    const float4* b1 = reinterpret_cast<const float4*>(src + (long long)f * p.fstride + (long long)rs * p.pitch + c1) + lane;
    const long long rstep = p.pitch / 4;
    float4 pa[PF], pb[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) { pa[q] = b0[(long long)min(q, n - 1) * rstep]; pb[q] = b1[(long long)min(q, n - 1) * rstep]; }
    float acc = 0.f;
    int i = 0;
    for (; i + PF <= n; i += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const float4 v = pa[q], w = pb[q];
            acc += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w;
            asm volatile("" : "+v"(acc));
            pa[q] = b0[(long long)min(i + q + PF, n - 1) * rstep];
            pb[q] = b1[(long long)min(i + q + PF, n - 1) * rstep];
        }
    }
    if (acc == 1234.5f) out[blockIdx.x] = acc;
}
// closer to a real sweep (k_me_stats): NS strips of 256 columns per wave (their row loads issued back to back), per row and
// strip one 1 KiB load of the frame, one 1 KiB load of a plane shared by all frames (W: L2 hits for all frames but the first),
// one halo dword per lane, and VALU fused multiply-adds per lane-row; 3 rows in flight
template <int NS, int VALU>
__global__ __launch_bounds__(256) void k_sweep(const float* __restrict__ src, const float* __restrict__ wpl, float* __restrict__ out, P p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int nsg = (p.nstrips + NS - 1) / NS;
    // frame quads: a block = one (strip group, segment) of 4 consecutive frames
    const int nq = (p.frames + 3) / 4;
    const int tile = b / nq;
    const int frame = 4 * (b - tile * nq) + wave, sg = tile % nsg, seg = tile / nsg;
    if (seg >= p.nsegs || frame >= p.frames) return;
    const int rs = seg * p.rps, re = min(rs + p.rps, p.rows), n = re - rs;
    const float4* xb[NS]; const float4* wb[NS]; const float* hb[NS];
#pragma unroll
    for (int h = 0; h < NS; ++h) {
        const int c0s = min((sg * NS + h) * 256, p.cols - 256);
        xb[h] = reinterpret_cast<const float4*>(src + (long long)frame * p.fstride + (long long)rs * p.pitch + c0s) + lane;
        wb[h] = reinterpret_cast<const float4*>(wpl + (long long)rs * p.pitch + c0s) + lane;
        hb[h] = src + (long long)frame * p.fstride + (long long)rs * p.pitch + (lane == 63 ? min(c0s + 256, p.cols - 1) : max(c0s - 1, 0));
    }
    const long long rstep = p.pitch / 4;
    constexpr int PF = 3;
    float4 px[NS][PF], pw[NS][PF]; float ph[NS][PF];
#pragma unroll
    for (int q = 0; q < PF; ++q)
#pragma unroll
        for (int h = 0; h < NS; ++h) { px[h][q] = xb[h][(long long)min(q, n - 1) * rstep]; pw[h][q] = wb[h][(long long)min(q, n - 1) * rstep]; ph[h][q] = hb[h][(long long)min(q, n - 1) * p.pitch]; }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + PF <= n; i += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
#pragma unroll
            for (int h = 0; h < NS; ++h) {
                const float4 v = px[h][q], w = pw[h][q]; const float hh = ph[h][q];
                float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < VALU / 4; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fmaf(t[k], (k & 1) ? w.x : w.y, acc[k] + hh);
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < NS; ++h) {
                px[h][q] = xb[h][(long long)min(i + q + PF, n - 1) * rstep];
                pw[h][q] = wb[h][(long long)min(i + q + PF, n - 1) * rstep];
                ph[h][q] = hb[h][(long long)min(i + q + PF, n - 1) * p.pitch];
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[blockIdx.x] = acc[0];
}
// the same with NF FRAMES per wave instead of strips: the wave marches the same (strip, segment) of NF consecutive frames and
// loads each W row ONCE for all of them (W is one plane shared by the batch); WLOAD = 0: no W stream at all (the bound)
template <int NF, int VALU, int WLOAD>
__global__ __launch_bounds__(256) void k_sweep_frames(const float* __restrict__ src, const float* __restrict__ wpl, float* __restrict__ out, P p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int per_block = 4 * NF;                      // frames of a block: wave w takes frames w*NF .. w*NF+NF-1 of the group
    const int ng = (p.frames + per_block - 1) / per_block;
    const int tile = b / ng;
    const int f0 = per_block * (b - tile * ng) + wave * NF, strip = tile % p.nstrips, seg = tile / p.nstrips;
    if (seg >= p.nsegs || f0 >= p.frames) return;
    const int rs = seg * p.rps, re = min(rs + p.rps, p.rows), n = re - rs;
    const int c0s = strip * 256;
    const float4* xb[NF]; const float* hb[NF];
#pragma unroll
    for (int h = 0; h < NF; ++h) {
        const int f = min(f0 + h, p.frames - 1);
        xb[h] = reinterpret_cast<const float4*>(src + (long long)f * p.fstride + (long long)rs * p.pitch + c0s) + lane;
        hb[h] = src + (long long)f * p.fstride + (long long)rs * p.pitch + (lane == 63 ? min(c0s + 256, p.cols - 1) : max(c0s - 1, 0));
    }
    const float4* wb = reinterpret_cast<const float4*>(wpl + (long long)rs * p.pitch + c0s) + lane;
    const long long rstep = p.pitch / 4;
    constexpr int PF = 3;
    float4 px[NF][PF], pw[PF]; float ph[NF][PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        pw[q] = WLOAD ? wb[(long long)min(q, n - 1) * rstep] : make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
        for (int h = 0; h < NF; ++h) { px[h][q] = xb[h][(long long)min(q, n - 1) * rstep]; ph[h][q] = hb[h][(long long)min(q, n - 1) * p.pitch]; }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + PF <= n; i += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const float4 w = pw[q];
#pragma unroll
            for (int h = 0; h < NF; ++h) {
                const float4 v = px[h][q]; const float hh = ph[h][q];
                float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < VALU / 4; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fmaf(t[k], (k & 1) ? w.x : w.y, acc[k] + hh);
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WLOAD) pw[q] = wb[(long long)min(i + q + PF, n - 1) * rstep];
#pragma unroll
            for (int h = 0; h < NF; ++h) {
                px[h][q] = xb[h][(long long)min(i + q + PF, n - 1) * rstep];
                ph[h][q] = hb[h][(long long)min(i + q + PF, n - 1) * p.pitch];
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[blockIdx.x] = acc[0];
}
// frame quads with the W row SHARED THROUGH LDS: the block's 4 waves are 4 frames of one (strip, segment); wave (row mod 4)
// loads the W row for all four (one global request per row and block instead of four L1 hits), stores it to a ring of LDS rows,
// a block barrier per row hands it over.  What it costs to couple the four waves, against the W requests it saves.
template <int VALU>
__global__ __launch_bounds__(256) void k_sweep_ldsw(const float* __restrict__ src, const float* __restrict__ wpl, float* __restrict__ out, P p)
{
    __shared__ float4 s_w[4][64];   // ring of 4 W rows
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    const int nq = (p.frames + 3) / 4;
    const int tile = b / nq;
    const int frame = min(4 * (b - tile * nq) + wave, p.frames - 1), strip = tile % p.nstrips, seg = tile / p.nstrips;
    const int rs = seg * p.rps, re = min(rs + p.rps, p.rows), n = re - rs;   // (block-uniform: no wave leaves before the barriers)
    const int c0s = strip * 256;
    const float4* xb = reinterpret_cast<const float4*>(src + (long long)frame * p.fstride + (long long)rs * p.pitch + c0s) + lane;
    const float* hb = src + (long long)frame * p.fstride + (long long)rs * p.pitch + (lane == 63 ? min(c0s + 256, p.cols - 1) : max(c0s - 1, 0));
    const float4* wb = reinterpret_cast<const float4*>(wpl + (long long)rs * p.pitch + c0s) + lane;
    const long long rstep = p.pitch / 4;
    constexpr int PF = 3;
    float4 px[PF]; float ph[PF];
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);   // the W row this wave is responsible for next (row i with i % 4 == wave), in flight
#pragma unroll
    for (int q = 0; q < PF; ++q) { px[q] = xb[(long long)min(q, n - 1) * rstep]; ph[q] = hb[(long long)min(q, n - 1) * p.pitch]; }
    int mine = wave;                               // next row index this wave loads W for
    if (mine < n) wq = wb[(long long)mine * rstep];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) {
        // hand-over of W row i: its loader stores it, everybody reads it after the barrier (ring slot i % 4; slot reuse is 4 rows
        // later, behind 3 more barriers)
        if ((i & 3) == wave) {
            s_w[i & 3][lane] = wq;
            mine = i + 4;
            if (mine < n) wq = wb[(long long)mine * rstep];
        }
        __syncthreads();
        const float4 w = s_w[i & 3][lane];
        const int q = i % PF;
        float4 v; float hh;
        // (runtime ring index: keep it simple, three-way select)
        v = q == 0 ? px[0] : (q == 1 ? px[1] : px[2]);
        hh = q == 0 ? ph[0] : (q == 1 ? ph[1] : ph[2]);
        float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < VALU / 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(t[k], (k & 1) ? w.x : w.y, acc[k] + hh);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        const float4 nx = xb[(long long)min(i + PF, n - 1) * rstep];
        const float nh = hb[(long long)min(i + PF, n - 1) * p.pitch];
        if (q == 0) { px[0] = nx; ph[0] = nh; } else if (q == 1) { px[1] = nx; ph[1] = nh; } else { px[2] = nx; ph[2] = nh; }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[blockIdx.x] = acc[0];
}
template <int VALU>
static double run_sweep_ldsw(const float* src, const float* wpl, float* out, P p, int reps)
{
    const int grid = p.nstrips * p.nsegs * ((p.frames + 3) / 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double total = 0;
    for (int r = 0; r < reps + 2; ++r) {
        hipExtLaunchKernelGGL((k_sweep_ldsw<VALU>), dim3(grid), dim3(256), 0, 0, a, b, 0, src, wpl, out, p);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 2) total += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 1e3 * total / reps;
}

template <int NF, int VALU, int WLOAD>
static double run_sweep_frames(const float* src, const float* wpl, float* out, P p, int reps)
{
    const int grid = p.nstrips * p.nsegs * ((p.frames + 4 * NF - 1) / (4 * NF));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double total = 0;
    for (int r = 0; r < reps + 2; ++r) {
        hipExtLaunchKernelGGL((k_sweep_frames<NF, VALU, WLOAD>), dim3(grid), dim3(256), 0, 0, a, b, 0, src, wpl, out, p);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 2) total += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 1e3 * total / reps;
}

template <int NS, int VALU>
static double run_sweep(const float* src, const float* wpl, float* out, P p, int reps)
{
    const int grid = ((p.nstrips + NS - 1) / NS) * p.nsegs * ((p.frames + 3) / 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double total = 0;
    for (int r = 0; r < reps + 2; ++r) {
        hipExtLaunchKernelGGL((k_sweep<NS, VALU>), dim3(grid), dim3(256), 0, 0, a, b, 0, src, wpl, out, p);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 2) total += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 1e3 * total / reps;
}

template <int PF>
static double run_wide(const float* src, float* out, P p, int reps)
{
    const int grid = ((p.nstrips + 1) / 2) * ((p.nsegs + 3) / 4) * p.frames;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double total = 0;
    for (int r = 0; r < reps + 2; ++r) {
        hipExtLaunchKernelGGL((k_march_wide<PF>), dim3(grid), dim3(256), 0, 0, a, b, 0, src, out, p);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (r >= 2) total += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 1e3 * total / reps;
}

template <int PF>
static double run(const float* src, float* out, P p, int reps)
{
    int grid;
    if (p.map == 0) grid = p.nstrips * ((p.nsegs + 3) / 4) * p.frames;
    else if (p.map == 1) grid = p.nstrips * p.nsegs * ((p.frames + 3) / 4);
    else if (p.map == 2 || p.map == 3) grid = ((p.nstrips + p.wpb - 1) / p.wpb) * p.nsegs * p.frames;
    else grid = p.nstrips * p.nsegs * p.frames;
    const int threads = 64 * p.wpb;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    double total = 0;
    for (int r = 0; r < reps + 2; ++r) {
        hipExtLaunchKernelGGL((k_march<PF>), dim3(grid), dim3(threads), 0, 0, a, b, 0, src, out, p);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r >= 2) total += ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return 1e3 * total / reps;  // us
}

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 2160, cols = argc > 2 ? atoi(argv[2]) : 3840, frames = argc > 3 ? atoi(argv[3]) : 16;
    const size_t n = (size_t)rows * (cols + (argc > 4 ? atoi(argv[4]) : 0)) * frames;
    float *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 1 << 22);
    hipMemset(src, 0, n * 4);
    const double mb = n * 4 / 1e6;
    std::printf("%d x %d f32 x %d frames = %.0f MB per launch\n", rows, cols, frames, mb);
    const int pad = argc > 4 ? atoi(argv[4]) : 0;  // extra floats per row (pitch = cols + pad)
    auto mk = [&](int rps, int map, int linear) {
        P p;
        p.rows = rows; p.cols = cols; p.frames = frames; p.rps = rps; p.nsegs = (rows + rps - 1) / rps; p.nstrips = cols / 256;
        p.map = map; p.frame_fastest = 0; p.remap = 1; p.wpb = 4; p.linear = linear;
        p.pitch = cols + pad; p.fstride = (long long)rows * (cols + pad);
        return p;
    };
    float* wpl;
    (void)hipMalloc(&wpl, (size_t)rows * (cols + pad) * 4);
    (void)hipMemset(wpl, 0, (size_t)rows * (cols + pad) * 4);
    for (int rps : {45, 24}) {
        P p = mk(rps, 0, 0);
        std::printf("rps %d sweep-like, valu64: 1 frame/wave %5.2f | 2 frames/wave sharing W %5.2f | 4 frames/wave %5.2f | no W stream at all: 1 frame %5.2f, 2 frames %5.2f TB/s\n", rps,
                    mb / run_sweep_frames<1, 64, 1>(src, wpl, out, p, 10), mb / run_sweep_frames<2, 64, 1>(src, wpl, out, p, 10), mb / run_sweep_frames<4, 64, 1>(src, wpl, out, p, 10),
                    mb / run_sweep_frames<1, 64, 0>(src, wpl, out, p, 10), mb / run_sweep_frames<2, 64, 0>(src, wpl, out, p, 10));
        std::printf("rps %d sweep-like, W shared through LDS with a block barrier per row: valu8 %5.2f valu64 %5.2f TB/s\n", rps, mb / run_sweep_ldsw<8>(src, wpl, out, p, 10),
                    mb / run_sweep_ldsw<64>(src, wpl, out, p, 10));
        std::printf("rps %d sweep-like (x + W + halo, frame quads), TB/s of the frame planes: 1 strip/wave: valu8 %5.2f valu64 %5.2f valu128 %5.2f | 2 strips/wave: valu8 %5.2f valu64 %5.2f valu128 %5.2f\n", rps,
                    mb / run_sweep<1, 8>(src, wpl, out, p, 10), mb / run_sweep<1, 64>(src, wpl, out, p, 10), mb / run_sweep<1, 128>(src, wpl, out, p, 10),
                    mb / run_sweep<2, 8>(src, wpl, out, p, 10), mb / run_sweep<2, 64>(src, wpl, out, p, 10), mb / run_sweep<2, 128>(src, wpl, out, p, 10));
        std::printf("rps %d march (pitch %lld B)      : PF1 %5.2f  PF2 %5.2f  PF3 %5.2f  PF4 %5.2f TB/s\n", rps, p.pitch * 4, mb / run<1>(src, out, p, 10), mb / run<2>(src, out, p, 10),
                    mb / run<3>(src, out, p, 10), mb / run<4>(src, out, p, 10));
        P q = mk(rps, 0, 1);
        std::printf("rps %d per-wave contiguous chunks : PF1 %5.2f  PF2 %5.2f  PF3 %5.2f  PF4 %5.2f TB/s\n", rps, mb / run<1>(src, out, q, 10), mb / run<2>(src, out, q, 10),
                    mb / run<3>(src, out, q, 10), mb / run<4>(src, out, q, 10));
        std::printf("rps %d 512-column strips          : PF1 %5.2f  PF2 %5.2f  PF3 %5.2f TB/s\n", rps, mb / run_wide<1>(src, out, p, 10), mb / run_wide<2>(src, out, p, 10),
                    mb / run_wide<3>(src, out, p, 10));
        std::fflush(stdout);
    }
    return 0;
}
