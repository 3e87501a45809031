// dev micro-benchmark: how many 1-KiB row requests does a CU keep in flight?  One 1024-thread workgroup per CU; every wave
// requests K rows (global_load_dwordx4, 1 KiB per wave-instruction) of a private region back to back, then waits.  Per wave:
// when its last request had been issued, when its data had arrived; chip-wide: bytes / time of the slowest wave.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/inflight.hip -o tools/ubench/inflight && tools/ubench/inflight
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int K>
__global__ void k_rows(const float4* __restrict__ src, float* out, unsigned long long* st, long long row_f4, int waves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wid = (long long)blockIdx.x * (blockDim.x / 64) + wave;
    // the fused kernels' tiling: workgroup b = (strip b % 15, band b / 15) of a 3840-wide plane, wave w: rows band*16K + wK ...
    const int strip = blockIdx.x % 15, band = blockIdx.x / 15;
    const float4* p = src + ((long long)band * (blockDim.x / 64) * K + (long long)wave * K) * row_f4 + strip * 64 + lane;
    float4 v[K];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = p[k * row_f4];
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
    asm volatile("" ::"v"(s));
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { st[wid * 3] = t0; st[wid * 3 + 1] = t1; st[wid * 3 + 2] = t2; }
    out[wid * 64 + lane] = s;
}

template <int K>
static void run(int threads, const float4* src, long long row_f4)
{
    const int blocks = 255, wpb = threads / 64, waves = blocks * wpb;
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * waves * 64); hipMalloc(&st, 8 * 3 * waves);
    static char* junk = nullptr; if (!junk) hipMalloc(&junk, 512u << 20);
    std::vector<unsigned long long> h(3 * waves);
    double best = 1e9, issue_last = 0, first_data = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(junk, rep, 512u << 20);  // the rows come from HBM
        hipDeviceSynchronize();
        hipLaunchKernelGGL((k_rows<K>), dim3(blocks), dim3(threads), 0, 0, src, out, st, row_f4, waves);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0, imax = 0, dmin = ~0ull;
        for (int w = 0; w < waves; ++w) { tmin = std::min(tmin, h[3 * w]); tmax = std::max(tmax, h[3 * w + 2]); imax = std::max(imax, h[3 * w + 1]); dmin = std::min(dmin, h[3 * w + 2]); }
        const double us = (tmax - tmin) / 100.0;
        if (us < best) { best = us; issue_last = (imax - tmin) / 100.0; first_data = (dmin - tmin) / 100.0; }
    }
    const double mb = (double)waves * K * 1024 / 1e6;
    std::printf("%2d waves/CU x %2d rows: %6.1f MB in %5.2f us = %4.2f TB/s   (last request issued at %5.2f us, first wave done at %5.2f us)\n",
                wpb, K, mb, best, mb / best, issue_last, first_data);
    hipFree(out); hipFree(st);
}

// a wave streams NR rows with at most PF in flight (consume one, request the next): the fused kernels' load phase without its
// arithmetic
template <int PF, int NR>
__global__ void k_stream(const float4* __restrict__ src, float* out, unsigned long long* st, long long row_f4, int waves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wid = (long long)blockIdx.x * (blockDim.x / 64) + wave;
    const int strip = blockIdx.x % 15, band = blockIdx.x / 15;
    const float4* p = src + ((long long)band * (blockDim.x / 64) * 8 + (long long)wave * 8) * row_f4 + strip * 64 + lane;
    float4 v[PF];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < PF; ++k) v[k] = p[k * row_f4];
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const float4 f = v[k % PF];
        s += f.x + f.y + f.z + f.w;
        if (k + PF < NR) v[k % PF] = p[(k + PF) * row_f4];
    }
    asm volatile("" ::"v"(s));
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { st[wid * 3] = t0; st[wid * 3 + 1] = t1; st[wid * 3 + 2] = t2; }
    out[wid * 64 + lane] = s;
}
template <int PF, int NR>
static void stream(const float4* src, long long row_f4)
{
    const int blocks = 255, threads = 1024, waves = blocks * 16;
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * waves * 64); hipMalloc(&st, 8 * 3 * waves);
    std::vector<unsigned long long> h(3 * waves);
    double best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {  // (warm: the plane was read by the launch before)
        hipLaunchKernelGGL((k_stream<PF, NR>), dim3(blocks), dim3(threads), 0, 0, src, out, st, row_f4, waves);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < waves; ++w) { tmin = std::min(tmin, h[3 * w]); tmax = std::max(tmax, h[3 * w + 2]); }
        if (rep) best = std::min(best, (tmax - tmin) / 100.0);
    }
    const double mb = (double)waves * NR * 1024 / 1e6;
    std::printf("stream, warm: 16 waves/CU x %2d rows (8 new per wave), %2d in flight per wave: %5.1f MB requested in %5.2f us\n", NR, PF, mb, best);
    hipFree(out); hipFree(st);
}

// the same kernel 24 times back to back on 8 different planes (no idle gap): does the rate of a short burst depend on what
// ran just before it?
static void chain(const float4* src, long long row_f4, size_t plane_f4)
{
    const int blocks = 255, threads = 1024, waves = blocks * 16, N = 24;
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * waves * 64); hipMalloc(&st, (size_t)8 * 3 * waves * N);
    for (int i = 0; i < N; ++i)
        hipLaunchKernelGGL((k_rows<8>), dim3(blocks), dim3(threads), 0, 0, src + (size_t)(i % 8) * plane_f4, out, st + (size_t)3 * waves * i, row_f4, waves);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)3 * waves * N);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::printf("24 launches back to back (33.4 MB each), us per launch:");
    for (int i = 0; i < N; ++i) {
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < waves; ++w) { tmin = std::min(tmin, h[(size_t)3 * (waves * i + w)]); tmax = std::max(tmax, h[(size_t)3 * (waves * i + w) + 2]); }
        std::printf(" %.1f", (tmax - tmin) / 100.0);
    }
    std::printf("\n");
}

int main()
{
    const long long row_f4 = 3840 / 4;  // rows of a 4K f32 plane: 15 KiB apart
    float4* src; hipMalloc(&src, (size_t)255 * 16 * 16 * 3840 * 4 + (1 << 20));
    hipMemset(src, 0, (size_t)255 * 16 * 16 * 3840 * 4);
    chain(src, row_f4, (size_t)2176 * 3840 / 4);
    chain(src, row_f4, (size_t)2176 * 3840 / 4);
    stream<2, 10>(src, row_f4); stream<4, 10>(src, row_f4); stream<6, 10>(src, row_f4); stream<10, 10>(src, row_f4);
    run<4>(1024, src, row_f4); run<8>(1024, src, row_f4); run<10>(1024, src, row_f4); run<16>(1024, src, row_f4);
    run<8>(512, src, row_f4); run<16>(512, src, row_f4); run<16>(256, src, row_f4);
    return 0;
}
