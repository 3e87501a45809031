// dev micro-benchmark: VALU throughput of one SIMD by instruction form and by wavefronts per SIMD (1, 2, 4, 8).
// One block per CU; the number printed is cycles (at 2.4 GHz) per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue && tools/ubench/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define KERNEL(NAME, ASM)                                                                                         \
    __global__ void NAME(float* out, unsigned long long* st, float y, int iters)                                 \
    {                                                                                                             \
        float r[8]; float b = y * 0.5f, c = y + 1.0f;                                                             \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;                                                       \
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" ::"v"(r[0]), "v"(c) : "vcc", "s20", "s21");            \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                          \
        for (int it = 0; it < iters; ++it) {                                                                      \
            asm volatile(".rept 8\n" ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) ".endr"             \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                         : "v"(b), "v"(c));                                                                       \
        }                                                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();                                          \
        if ((threadIdx.x & 63) == 0) st[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;            \
        float s = 0; for (int i = 0; i < 8; ++i) s += r[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                           \
    }
#define A_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define A_FMA_ACC(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define A_FMA_MUL(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_SUB_ABS(i) "v_sub_f32 %" #i ", |%" #i "|, %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_MOVDPP(i) "v_mov_b32_dpp %" #i ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_CND_NEW(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define A_CND_SGPR(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define A_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MAXABS(i) "v_max_f32 %" #i ", |%" #i "|, %8\n"
#define A_MAX_E64(i) "v_max_f32_e64 %" #i ", %8, %" #i "\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_FMAC_DPP(i) "v_fmac_f32_dpp %" #i ", %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
KERNEL(k_cnd_new, A_CND_NEW) KERNEL(k_cnd_sgpr, A_CND_SGPR) KERNEL(k_min, A_MIN) KERNEL(k_med3, A_MED3) KERNEL(k_max3, A_MAX3)
KERNEL(k_maxabs, A_MAXABS) KERNEL(k_max64, A_MAX_E64) KERNEL(k_and, A_AND) KERNEL(k_cvt, A_CVT) KERNEL(k_mov, A_MOV)
KERNEL(k_fmacdpp, A_FMAC_DPP) KERNEL(k_rcp, A_RCP) KERNEL(k_lshl, A_LSHL)
KERNEL(k_fmac, A_FMAC) KERNEL(k_fma_acc, A_FMA_ACC) KERNEL(k_fma_mul, A_FMA_MUL) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD)
KERNEL(k_max, A_MAX) KERNEL(k_subabs, A_SUB_ABS) KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_movdpp, A_MOVDPP)

template <typename K>
static void run(const char* name, K kern)
{
    std::printf("%-34s", name);
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps, blocks = 256, iters = 256, waves = blocks * threads / 64;
        float* out; unsigned long long* st;
        hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&st, 8 * waves);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, st, 1.0f, iters); hipDeviceSynchronize(); }
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), st, 8 * waves, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        // the SIMD is busy until its slowest wave ends: cycles per instruction per SIMD = max wave time / (wps * instructions per wave)
        const double us = (double)h[waves * 99 / 100] / 100.0, inst = (double)iters * 64;
        std::printf("  %d/SIMD: %.2f (wave %.2f)", wps, us * 2400.0 / (inst * wps), (double)h[waves / 2] / 100.0 * 2400.0 / inst);
        hipFree(out); hipFree(st);
    }
    std::printf("\n");
}
int main()
{
    std::printf("cycles per wave-instruction per SIMD at 2.4 GHz (in brackets: of the median wave alone)\n");
    run("v_fmac_f32 d, a, b", k_fmac); run("v_fma_f32 d, a, b, d", k_fma_acc); run("v_fma_f32 d, d, a, b", k_fma_mul);
    run("v_mul_f32 d, d, a", k_mul); run("v_add_f32 d, d, a", k_add); run("v_max_f32 d, d, a", k_max);
    run("v_sub_f32 d, |d|, a", k_subabs); run("v_cndmask_b32 d, d, a, vcc", k_cndmask); run("v_mov_b32_dpp d, a wave_shr:1", k_movdpp);
    run("v_cndmask_b32 d, a, b, vcc", k_cnd_new); run("v_cndmask_b32_e64 d, d, a, s[20:21]", k_cnd_sgpr); run("v_min_f32 d, d, a", k_min);
    run("v_med3_f32 d, d, a, b", k_med3); run("v_max3_f32 d, d, a, b", k_max3); run("v_max_f32 d, |d|, a", k_maxabs);
    run("v_max_f32_e64 d, a, d", k_max64); run("v_and_b32 d, d, a", k_and); run("v_cvt_f32_u32 d, d", k_cvt); run("v_mov_b32 d, a", k_mov);
    run("v_fmac_f32_dpp d, a, b wave_shr:1", k_fmacdpp); run("v_rcp_f32 d, d", k_rcp); run("v_lshlrev_b32 d, 1, d", k_lshl);
    return 0;
}
