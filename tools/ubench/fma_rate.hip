// dev microbenchmark: VALU issue rate of f64 vs f32 FMA (and a few helpers) on gfx950.
// build: hipcc -O3 --offload-arch=gfx950 fma_rate.hip -o fma_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int NACC>
__global__ void k_fma(T* out, int iters, T a, T b)
{
    T acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (T)threadIdx.x + (T)i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NACC>
__global__ void k_fma32(float* out, int iters, float a, float b)
{
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_pkfma(float* out, int iters, float a, float b)
{
    v2f acc[NACC];
    const v2f a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v2f{(float)threadIdx.x + i, (float)threadIdx.x - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a2), "v"(b2));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cvt(double* out, int iters, float a)
{
    float v[8]; double acc = 0;
    for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc += (double)v[i]; v[i] += 1.0f; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// one instruction, 16 independent chains, issued back to back: cycles per wave instruction per SIMD
#define OP_KERNEL(NAME, DECL, INIT, ASM, CONSTR, SINK)                                                 \
    __global__ void NAME(double* out, int iters, float a)                                              \
    {                                                                                                  \
        DECL;                                                                                          \
        for (int i = 0; i < 16; ++i) { INIT; }                                                         \
        for (int it = 0; it < iters; ++it) {                                                           \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : CONSTR);                 \
        }                                                                                              \
        double s = 0;                                                                                  \
        for (int i = 0; i < 16; ++i) s += SINK;                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
    }
OP_KERNEL(k_op_cvt, float v[16]; double d[16], v[i] = a + threadIdx.x + i; d[i] = 0, "v_cvt_f64_f32 %0, %1", "=v"(d[i]) : "v"(v[i]), d[i])
OP_KERNEL(k_op_fma64, double d[16], d[i] = a + threadIdx.x + i, "v_fma_f64 %0, %0, %0, %0", "+v"(d[i]), d[i])
OP_KERNEL(k_op_fmac64, double d[16]; double e = a, d[i] = a + threadIdx.x + i, "v_fmac_f64 %0, %1, %1", "+v"(d[i]) : "v"(e), d[i])
OP_KERNEL(k_op_mul64, double d[16]; double e = a, d[i] = a + threadIdx.x + i, "v_mul_f64 %0, %0, %1", "+v"(d[i]) : "v"(e), d[i])
OP_KERNEL(k_op_add64, double d[16]; double e = a, d[i] = a + threadIdx.x + i, "v_add_f64 %0, %0, %1", "+v"(d[i]) : "v"(e), d[i])
OP_KERNEL(k_op_dpp, float v[16]; float d[16], v[i] = a + threadIdx.x + i; d[i] = 0, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(d[i]) : "v"(v[i]), d[i])
OP_KERNEL(k_op_mov, float v[16]; float d[16], v[i] = a + threadIdx.x + i; d[i] = 0, "v_mov_b32 %0, %1", "=v"(d[i]) : "v"(v[i]), d[i])
OP_KERNEL(k_op_fmac32, float d[16]; float e = a, d[i] = a + threadIdx.x + i, "v_fmac_f32 %0, %1, %1", "+v"(d[i]) : "v"(e), d[i])
OP_KERNEL(k_op_fmac32dpp, float d[16]; float e = a, d[i] = a + threadIdx.x + i, "v_fmac_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(d[i]) : "v"(e), d[i])
OP_KERNEL(k_op_dot4, unsigned v[16]; unsigned d[16], v[i] = threadIdx.x + i; d[i] = 0, "v_dot4_u32_u8 %0, %1, %1, %0", "+v"(d[i]) : "v"(v[i]), (double)d[i])
OP_KERNEL(k_op_rcp64, double d[16], d[i] = a + threadIdx.x + i, "v_rcp_f64 %0, %0", "+v"(d[i]), d[i])
OP_KERNEL(k_op_cvtu, float v[16]; unsigned d[16], v[i] = a + threadIdx.x + i; d[i] = 0, "v_cvt_u32_f32 %0, %1", "=v"(d[i]) : "v"(v[i]), (double)d[i])
template <typename F>
static double timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e-3;
}
int main()
{
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    void* buf; hipMalloc(&buf, (size_t)blocks * threads * 8);
    const double nwave = (double)blocks * threads / 64.0;
    {
        double t = timeit([&] { hipLaunchKernelGGL((k_fma<float, 16>), dim3(blocks), dim3(threads), 0, 0, (float*)buf, iters, 1.0001f, 0.5f); });
        double inst = nwave * iters * 16;
        printf("f32 fma: %.2f TFLOP/s, %.2f cycles per wave-instr per SIMD (2.4 GHz, 1024 SIMDs)\n", inst * 64 * 2 / t / 1e12, t * 2.4e9 * 1024 / inst);
    }
    {
        double t = timeit([&] { hipLaunchKernelGGL((k_fma<double, 16>), dim3(blocks), dim3(threads), 0, 0, (double*)buf, iters, 1.0001, 0.5); });
        double inst = nwave * iters * 16;
        printf("f64 fma: %.2f TFLOP/s, %.2f cycles per wave-instr per SIMD\n", inst * 64 * 2 / t / 1e12, t * 2.4e9 * 1024 / inst);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t = timeit([&] { hipLaunchKernelGGL((k_fma32<16>), dim3(blocks), dim3(threads), 0, 0, (float*)buf, iters, 1.0001f, 0.5f); });
        double inst = nwave * iters * 16;
        printf("v_fma_f32 (asm): %.2f TFLOP/s, %.2f cycles per wave-instr per SIMD\n", inst * 64 * 2 / t / 1e12, t * 2.4e9 * 1024 / inst);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t = timeit([&] { hipLaunchKernelGGL((k_pkfma<16>), dim3(blocks), dim3(threads), 0, 0, (float*)buf, iters, 1.0001f, 0.5f); });
        double inst = nwave * iters * 16;
        printf("pk_fma_f32: %.2f TFLOP/s, %.2f cycles per wave-instr per SIMD\n", inst * 64 * 4 / t / 1e12, t * 2.4e9 * 1024 / inst);
    }
    {
        double t = timeit([&] { hipLaunchKernelGGL(k_cvt, dim3(blocks), dim3(threads), 0, 0, (double*)buf, iters, 1.0f); });
        double inst = nwave * iters * 8;
        printf("cvt_f64_f32 + add_f64 + add_f32 triple: %.2f cycles per triple per SIMD\n", t * 2.4e9 * 1024 / inst);
    }
#define RUN_OP(K, LABEL) { double t = timeit([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, (double*)buf, iters, 1.5f); }); \
        printf("%-22s %.2f cycles per wave-instr per SIMD\n", LABEL, t * 2.4e9 * 1024 / (nwave * iters * 16)); }
    RUN_OP(k_op_cvt, "v_cvt_f64_f32") RUN_OP(k_op_fma64, "v_fma_f64") RUN_OP(k_op_fmac64, "v_fmac_f64") RUN_OP(k_op_mul64, "v_mul_f64")
    RUN_OP(k_op_add64, "v_add_f64") RUN_OP(k_op_dpp, "v_mov_b32_dpp") RUN_OP(k_op_mov, "v_mov_b32") RUN_OP(k_op_fmac32, "v_fmac_f32")
    RUN_OP(k_op_fmac32dpp, "v_fmac_f32_dpp") RUN_OP(k_op_dot4, "v_dot4_u32_u8") RUN_OP(k_op_rcp64, "v_rcp_f64") RUN_OP(k_op_cvtu, "v_cvt_u32_f32")
    return 0;
}
