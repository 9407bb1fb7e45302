// op_rate.hip -- micro-benchmark: issue cost (shader cycles per wave64 instruction per SIMD) of the individual VALU
// instruction classes the Kolb kernels use on gfx950, timed in-kernel with s_memtime AND by wall clock, so the cost is
// independent of the clock the chip settles at.  4 independent chains per wave, 1/2/4/8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/op_rate.hip -o /tmp/op_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define OPS(X) \
    X(0, "v_fma_f32 vvv", "v_fma_f32 %0, %0, %2, %3") \
    X(1, "v_fma_f32 v,s,v", "v_fma_f32 %0, %0, %4, %3") \
    X(2, "v_mul_f32", "v_mul_f32 %0, %0, %2") \
    X(3, "v_add_f32", "v_add_f32 %0, %0, %3") \
    X(4, "v_mov_b32", "v_mov_b32 %0, %2") \
    X(5, "v_cndmask_b32 vcc", "v_cndmask_b32 %0, %0, %2, vcc") \
    X(6, "v_cmp_gt_f32 vcc", "v_cmp_gt_f32 vcc, %0, %2") \
    X(7, "v_cmp_gt_f32 s[20:21]", "v_cmp_gt_f32 s[20:21], %0, %2") \
    X(8, "v_xor_b32", "v_xor_b32 %0, %0, %2") \
    X(9, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0") \
    X(10, "v_add_u32", "v_add_u32 %0, %0, %2") \
    X(11, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %2") \
    X(12, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %2, %3") \
    X(13, "v_sqrt_f32", "v_sqrt_f32 %0, %0") \
    X(14, "v_rsq_f32", "v_rsq_f32 %0, %0") \
    X(15, "v_rcp_f32", "v_rcp_f32 %0, %0") \
    X(16, "v_max_f32", "v_max_f32 %0, %0, %2") \
    X(17, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0") \
    X(18, "v_fma_f64", "v_fma_f64 %1, %1, %5, %6") \
    X(19, "v_mul_f64", "v_mul_f64 %1, %1, %5") \
    X(20, "v_add_f64", "v_add_f64 %1, %1, %6") \
    X(21, "v_readfirstlane_b32", "v_readfirstlane_b32 s20, %0") \
    X(22, "v_fmaak_f32 literal", "v_fmaak_f32 %0, %0, %2, 0x3f7fbe77") \
    X(23, "v_pk_fma_f32", "v_pk_fma_f32 %1, %1, %5, %6") \
    X(24, "v_pk_mul_f32", "v_pk_mul_f32 %1, %1, %5") \
    X(25, "v_cndmask_b32 s[22:23]", "v_cndmask_b32 %0, %0, %2, s[22:23]") \
    X(26, "v_fma_f32 abs/neg mods", "v_fma_f32 %0, |%0|, -%2, %3") \
    X(27, "v_mul_f32 s,v", "v_mul_f32 %0, %4, %0") \
    X(28, "v_sub_f32", "v_sub_f32 %0, %2, %0") \
    X(29, "v_mad_u64_u32", "v_mad_u64_u32 %1, s[20:21], %2, %3, %1") \
    X(30, "v_rcp_f64", "v_rcp_f64 %1, %1") \
    X(31, "v_sqrt_f64", "v_sqrt_f64 %1, %1") \
    X(32, "v_cvt_f64_f32", "v_cvt_f64_f32 %1, %0") \
    X(33, "v_cvt_f32_f64", "v_cvt_f32_f64 %0, %1") \
    X(34, "v_and_or_b32", "v_and_or_b32 %0, %0, %2, %3") \
    X(35, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, s22, %0") \
    X(36, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8") \
    X(37, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %2") \
    X(38, "v_min3_f32", "v_min3_f32 %0, %0, %2, %3") \
    X(39, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %2") \
    X(40, "v_cmp_class/ v_cmp_lt_u32 vcc", "v_cmp_lt_u32 vcc, %0, %2") \
    X(41, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %2, %3") \
    X(42, "v_div_fmas_f32", "v_div_fmas_f32 %0, %0, %2, %3") \
    X(43, "v_div_scale_f32", "v_div_scale_f32 %0, vcc, %0, %2, %3") \
    X(44, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %2") \
    X(45, "v_frexp_mant_f32", "v_frexp_mant_f32 %0, %0") \
    X(46, "ds_bpermute_b32", "ds_bpermute_b32 %0, %2, %0\n s_waitcnt lgkmcnt(0)") \
    X(47, "MIX fma,cvt (x2)", "v_fma_f32 %0, %0, %2, %3\n v_cvt_f32_u32 %0, %0") \
    X(48, "MIX fma,fma,fma,cvt (x4)", "v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_cvt_f32_u32 %0, %0") \
    X(49, "MIX fma,cmp->s (x2)", "v_fma_f32 %0, %0, %2, %3\n v_cmp_gt_f32 s[20:21], %0, %2") \
    X(50, "MIX mul,add (x2)", "v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %3") \
    X(51, "MIX fma,sqrt (x2)", "v_fma_f32 %0, %0, %2, %3\n v_sqrt_f32 %0, %0") \
    X(52, "MIX 6 fp32,sqrt (x7)", "v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_sqrt_f32 %0, %0")

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float a0)
{
    float f0 = a0 + threadIdx.x * 1e-6f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    double x0 = f0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const double c = a0 * 0.999, d = a0 * 1e-3;
    const float cf = a0 * 0.999f, df = a0 * 1e-3f;
    float sc = a0 * 0.5f;
    asm volatile("s_mov_b64 s[22:23], exec" ::: "s22", "s23");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#define Y(F, D, ASM) asm volatile(ASM "\n" : "+v"(F), "+v"(D) : "v"(cf), "v"(df), "s"(sc), "v"(c), "v"(d) : "vcc", "s20", "s21");
#define X(ID, NAME, ASM)                                                                                               \
    if (OP == ID) { Y(f0, x0, ASM) Y(f1, x1, ASM) Y(f2, x2, ASM) Y(f3, x3, ASM) }
            OPS(X)
#undef X
#undef Y
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    const double s = x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3;
    if (s == 12345.678) out[0] = static_cast<float>(s);
}

template <int OP>
void run(const char *name, float *d, unsigned long long *dc)
{
    int perOp = 1;   // instructions per asm statement: "(xN)" in the name
    if (const char *x = strstr(name, "(x")) perOp = atoi(x + 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    printf("%-30s", name);
    for (int wavesPerSimd : {1, 4, 8}) {
        const int blocks = 256 * wavesPerSimd;  // 256 CUs x (4 waves per block = 1 per SIMD)
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, dc, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cyc = 0; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        const double instrPerWave = double(iters) * 16 * 4 * perOp;
        // s_memtime cycles per instruction per SIMD (the SIMD runs wavesPerSimd waves, each instrPerWave)
        const double cycPerInstr = double(cyc) / (instrPerWave * wavesPerSimd);
        const double wallRate = double(blocks) * 4 * instrPerWave / (ms * 1e-3) / 1e12;
        printf(" | w%d: %5.2f cyc  %.3f Twi/s (%.0f MHz eff)", wavesPerSimd, cycPerInstr, wallRate, double(cyc) / (ms * 1e-3) / 1e6);
    }
    printf("\n");
}

int main()
{
    float *d; hipMalloc(&d, 4);
    unsigned long long *dc; hipMalloc(&dc, 8);
#define X(ID, NAME, ASM) run<ID>(NAME, d, dc);
    OPS(X)
#undef X
    return 0;
}
