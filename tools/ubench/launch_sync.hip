// dev micro-benchmark: what one synchronous kernel call costs on the host side (launch + wait), three ways of waiting:
// hipStreamSynchronize, hipEventSynchronize, and spinning on a flag the kernel writes to device-mapped pinned memory.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/launch_sync.hip -o tools/ubench/launch_sync && tools/ubench/launch_sync
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void k_flag(volatile unsigned* flag, unsigned v, int spin)
{
    if (spin > 0) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {} }
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = v; }
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *h = nullptr, *d = nullptr;
    hipHostMalloc((void**)&h, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&d, h, 0);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    using clk = std::chrono::steady_clock;
    const int N = 2000;
    for (int spin : {0, 2000}) {  // kernel body of 0 / 20 us (100 MHz ticks)
        for (int mode = 0; mode < 5; ++mode) {
            *h = 0;
            for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_flag, dim3(255), dim3(1024), 0, s, d, 0u, spin); hipStreamSynchronize(s); }
            const auto t0 = clk::now();
            for (int i = 1; i <= N; ++i) {
                if (mode == 3) hipExtLaunchKernelGGL(k_flag, dim3(255), dim3(1024), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, (unsigned)i, spin);
                else if (mode == 4) hipExtLaunchKernelGGL(k_flag, dim3(255), dim3(1024), 0, s, nullptr, nullptr, 0, d, (unsigned)i, spin);
                else hipLaunchKernelGGL(k_flag, dim3(255), dim3(1024), 0, s, d, (unsigned)i, spin);
                if (mode == 0) hipStreamSynchronize(s);
                else if (mode == 1) { hipEventRecord(ev, s); hipEventSynchronize(ev); }
                else { while (*(volatile unsigned*)h != (unsigned)i) {} }
            }
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count() / N;
            std::printf("kernel body %2d us, wait by %-24s: %.2f us per call (host overhead %.2f)\n", spin / 100,
                        mode == 0 ? "hipStreamSynchronize" : (mode == 1 ? "hipEventSynchronize" : (mode == 2 ? "spin on mapped flag" : (mode == 3 ? "spin, any-order launch" : "spin, hipExtLaunch"))), us, us - spin / 100.0);
        }
    }
    // launch + spin floor by workgroup size (255 workgroups, empty body)
    for (int threads : {64, 256, 512, 1024}) {
        *h = 0;
        for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_flag, dim3(255), dim3(threads), 0, s, d, 0u, 0); hipStreamSynchronize(s); }
        const auto t0 = clk::now();
        for (int i = 1; i <= N; ++i) {
            hipLaunchKernelGGL(k_flag, dim3(255), dim3(threads), 0, s, d, (unsigned)i, 0);
            while (*(volatile unsigned*)h != (unsigned)i) {}
        }
        hipStreamSynchronize(s);
        std::printf("255 workgroups x %4d threads, empty body, spin on mapped flag: %.2f us per call\n", threads,
                    std::chrono::duration<double, std::micro>(clk::now() - t0).count() / N);
    }
    return 0;
}
