// dev micro-benchmark: 16-byte vector loads and stores at addresses that are only 4-byte aligned (dense f32 planes whose width is not a
// multiple of 4: every other row starts 8 bytes off a 16-byte boundary).  Checks the values and times a streaming read at
// offsets 0 / 4 / 8 / 12 bytes, as buffer loads (the sweeps' row loads) and as flat global loads.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/unaligned.hip -o tools/ubench/unaligned
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void k_flat(const float* __restrict__ src, float* __restrict__ dst, size_t n4, int off)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f4 v = *reinterpret_cast<const f4*>(src + off + 4 * i);
        *reinterpret_cast<f4*>(dst + off + 4 * i) = v + 1.0f;
    }
}
__global__ void k_buf(const float* __restrict__ src, float* __restrict__ dst, size_t n4, int off)
{
    typedef int v4i __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0xFFFFFFFF, 0x00020000);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        v4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((off + 4 * i) * 4), 0, 0);
        v.x = __float_as_int(__int_as_float(v.x) + 1.0f); v.y = __float_as_int(__int_as_float(v.y) + 1.0f);
        v.z = __float_as_int(__int_as_float(v.z) + 1.0f); v.w = __float_as_int(__int_as_float(v.w) + 1.0f);
        __builtin_amdgcn_raw_buffer_store_b128(v, rd, (unsigned)((off + 4 * i) * 4), 0, 2);
    }
}
int main()
{
    const size_t n4 = 64u << 20;  // 64 Mi float4 = 1 GiB
    const size_t n = 4 * n4 + 16;
    float *src, *dst;
    if (hipMalloc(&src, n * 4) != hipSuccess || hipMalloc(&dst, n * 4) != hipSuccess) return 1;
    std::vector<float> h(1 << 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    (void)hipMemset(src, 0, n * 4);
    (void)hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int kind = 0; kind < 2; ++kind)
        for (int off = 0; off < 4; ++off) {
            (void)hipMemset(dst, 0, n * 4);
            double best = 1e9;
            for (int r = 0; r < 4; ++r) {
                if (kind == 0) hipExtLaunchKernelGGL(k_flat, dim3(65536), dim3(256), 0, 0, a, b, 0, src, dst, n4 / 4, off);
                else hipExtLaunchKernelGGL(k_buf, dim3(65536), dim3(256), 0, 0, a, b, 0, src, dst, n4 / 4, off);
                if (hipDeviceSynchronize() != hipSuccess) { std::printf("kind %d off %d: FAILED\n", kind, off); return 2; }
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            std::vector<float> o(4096);
            (void)hipMemcpy(o.data(), dst + off, o.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (size_t i = 0; i < o.size(); ++i) bad += o[i] != (float)(i + off) + 1.0f;
            std::printf("%s 16-byte access at byte offset %2d: %s, copy of 256 MiB in %.1f us = %.2f TB/s (read + write)\n", kind ? "buffer" : "flat  ", 4 * off,
                        bad ? "WRONG VALUES" : "values ok", best * 1e3, 2.0 * (n4 / 4) * 16 / (best * 1e-3) / 1e12);
        }
    return 0;
}
