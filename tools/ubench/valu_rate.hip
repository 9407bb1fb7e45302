// valu_rate.hip -- micro-benchmark: issue rate of wave64 VALU instructions on gfx950, per SIMD, for the op mix of the
// Kolb trace (fma / mul / add / cmp+cndmask / sqrt / rcp), dependent vs independent, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int ILP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0)
{
    float x[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) x[j] = a0 + threadIdx.x * 1e-6f + j;
    const float c = a0 * 0.999f, d = a0 * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < ILP; ++j) {
                if (OP == 0) x[j] = __builtin_fmaf(x[j], c, d);
                if (OP == 1) x[j] = x[j] * c;
                if (OP == 2) x[j] = x[j] + d;
                if (OP == 3) x[j] = (x[j] > c) ? d : x[j] + 1.0f;           // cmp + cndmask (+add)
                if (OP == 4) x[j] = __builtin_amdgcn_sqrtf(x[j]);
                if (OP == 5) x[j] = __builtin_amdgcn_rcpf(x[j]);
                if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double *>(&x[j & ~1])) : "v"(*reinterpret_cast<const double *>(&c)), "v"(*reinterpret_cast<const double *>(&d)));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) s += x[j];
    if (s == 12345.678f) out[0] = s;
}

template <int OP, int ILP>
void run(const char *name, int instrPerOp, float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wavesPerSimd : {1, 2, 4, 8}) {
        const int blocks = 256 * wavesPerSimd;  // 256 CUs x (4 waves per block = 1 per SIMD)
        hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waveInstr = double(blocks) * 4 * iters * 16 * ILP * instrPerOp;
        const double perSimdPerCycle = waveInstr / (ms * 1e-3) / (1024.0 * 2.4e9);
        printf("%-14s ILP%d waves/SIMD %d : %.3f ms  %.1f Gwave-instr/s  cycles/instr/SIMD %.2f\n", name, ILP, wavesPerSimd, ms,
               waveInstr / (ms * 1e-3) / 1e9, 1.0 / perSimdPerCycle);
    }
}

int main()
{
    float *d; hipMalloc(&d, 4);
    run<0, 1>("fma dep", 1, d); run<0, 4>("fma indep", 1, d);
    run<1, 1>("mul dep", 1, d); run<2, 4>("add indep", 1, d);
    run<3, 1>("cmp+sel+add", 3, d); run<3, 4>("cmp+sel+add", 3, d);
    run<4, 1>("sqrt dep", 1, d); run<4, 4>("sqrt indep", 1, d);
    run<5, 4>("rcp indep", 1, d);
    return 0;
}
