// dev micro-benchmark: how fast does a wave run through straight-line code it has never fetched (cold instruction cache)
// against the same code the second time?  One block per CU, 1 or 16 waves; the body is N VALU instructions without a loop.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/icache.hip -o tools/ubench/icache && tools/ubench/icache
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define BODY(N)                                                                                                   \
    asm volatile(".rept " #N "\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n .endr" \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(y))

template <int KB>
__global__ void k_code(float* out, unsigned long long* st, float y)
{
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    unsigned long long t[4];
    t[0] = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < 3; ++it) {
        if constexpr (KB == 4) BODY(256);     // 1024 instructions x 4 B = 4 KB
        if constexpr (KB == 16) BODY(1024);   // 16 KB
        if constexpr (KB == 48) BODY(3072);   // 48 KB
        asm volatile("s_nop 0" ::: "memory");
        t[it + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        for (int i = 0; i < 4; ++i) st[w * 4 + i] = t[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int KB>
static void run(int threads, bool flush = true)
{
    const int blocks = 256, waves = blocks * threads / 64;
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&st, sizeof(unsigned long long) * waves * 4);
    std::vector<unsigned long long> h(waves * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_code<KB>), dim3(blocks), dim3(threads), 0, 0, out, st, 1.0f);
        hipDeviceSynchronize();
        // evict: stream 512 MB through the caches so that the next launch finds the code in memory only
        static char* junk = nullptr; if (!junk) hipMalloc(&junk, 512u << 20);
        if (flush) hipMemset(junk, rep, 512u << 20);
        hipDeviceSynchronize();
    }
    hipLaunchKernelGGL((k_code<KB>), dim3(blocks), dim3(threads), 0, 0, out, st, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> p[3];
    for (int w = 0; w < waves; ++w)
        for (int i = 0; i < 3; ++i) p[i].push_back((double)(h[w * 4 + i + 1] - h[w * 4 + i]) / 100.0);
    std::printf("%2d KB straight-line, %2d waves/CU, %s:", KB, threads / 64, flush ? "caches flushed" : "previous launch just ran");
    for (int i = 0; i < 2; ++i) {
        std::sort(p[i].begin(), p[i].end());
        const size_t n = p[i].size();
        std::printf("  pass %d: p50 %.2f p90 %.2f p99 %.2f max %.2f us", i, p[i][n / 2], p[i][n * 9 / 10], p[i][n * 99 / 100], p[i].back());
    }
    std::printf("\n");
    if (threads == 1024 && KB == 16) {  // who is slow?  median of pass 0 per wave index, and each wave's start after the block's first
        std::printf("   pass 0 by wave index (median over blocks, us):");
        for (int wv = 0; wv < 16; ++wv) {
            std::vector<double> q, st0;
            for (int b = 0; b < blocks; ++b) {
                q.push_back((double)(h[(b * 16 + wv) * 4 + 1] - h[(b * 16 + wv) * 4]) / 100.0);
                unsigned long long m = ~0ull;
                for (int k = 0; k < 16; ++k) m = std::min(m, h[(b * 16 + k) * 4]);
                st0.push_back((double)(h[(b * 16 + wv) * 4] - m) / 100.0);
            }
            std::sort(q.begin(), q.end()); std::sort(st0.begin(), st0.end());
            std::printf(" w%d %.1f (start +%.2f)", wv, q[q.size() / 2], st0[st0.size() / 2]);
        }
        std::printf("\n");
    }
    hipFree(out); hipFree(st);
}

int main()
{
    run<4>(64); run<16>(64); run<48>(64);
    run<4>(1024); run<16>(1024); run<48>(1024);
    run<4>(1024, false); run<16>(1024, false); run<48>(1024, false);
    run<16>(256, false); run<16>(512, false);
    return 0;
}
