// kolb_fast.hip -- the SIMPLE fast-mode kernel (one sample per lane, retry loop inside the lane).  Kept as the
// A/B baseline of the optimisation ladder in DESIGN.md (ZOIC_KOLB_VARIANT=simple); production is kolb_refill.hip.
// FAST-mode arithmetic (fast_optics.hpp): the same algorithm as the strict kernel (kernels.hip,
// zoic.cpp:1850-1964) with the arithmetic re-associated for the VALU:
//   * f32 only (the reference's f64 intermediates dropped), FMA contraction on;
//   * the ray direction is normalised once (v_rsq_f32) and then kept unit by construction -- the reference
//     re-normalises it at every surface and twice more inside Snell (zoic.cpp:974,1002,1009-1010);
//   * the surface normal is (centre - hit) * (1/R): |centre - hit| == |R| on the sphere, no sqrt;
//   * v_sqrt_f32 for the two remaining roots per surface.
// Accept/reject formulas are unchanged, so decisions flip only where the reference's own f32 rounding noise
// decides (measured by tests/test_parity_gpu.py: direction RMSE < 1e-5, flip fraction reported).
#include <hip/hip_runtime.h>

#include "device_search.hpp"
#include "fast_optics.hpp"
#include "kernels.hpp"
#include "ray_store.hpp"
#include "optics.hpp"

#pragma clang fp contract(fast)

namespace zoic {

constexpr int kBlockF = 256;

__device__ __forceinline__ V2 sample_lens_f32(bool useImage, const BokehTables &B, int bw, int bh, float u, float v)
{
    if (useImage) return bokeh_sample_device(B, bw, bh, u, v);
    return concentric_disk_f32(u, v);
}

__global__ __launch_bounds__(kBlockF) void kolb_rays_fast_kernel(const KolbTable T, const BokehTables B,
                                                                 const float4 *__restrict__ samples,
                                                                 const uint4 *__restrict__ rngStates, uint64_t rayBase, uint64_t n,
                                                                 RayRecord *__restrict__ out, DeviceCounters *counters)
{
    uint32_t succ = 0, vign = 0, tir = 0;
    const bool useImage = T.useImage != 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlockF;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlockF + threadIdx.x; i < n; i += stride) {
        const float4 s = samples[i];
        Rng rng;
        if (rngStates) { const uint4 r = rngStates[i]; rng = Rng{r.x, r.y, r.z, r.w}; }
        else rng = rng_for_ray(T.seed, rayBase + i);
        const V3 o0{s.x * T.halfSensor, s.y * T.halfSensor, T.originShift};
        V2 lens = sample_lens_f32(useImage, B, T.bokehW, T.bokehH, s.z, s.w);
        float maxScale = 0.0f, translation = 0.0f, sn = 0.0f, cs = 1.0f;
        uint32_t lutMiss = 0;
        V3 o = o0, d;
        if (!T.useLUT) {
            d = V3{(lens.x * T.rearAperture) - o.x, (lens.y * T.rearAperture) - o.y, T.dirZ};
        } else {
            const float dist = fsqrt_fast(o.x * o.x + o.y * o.y);
            const float theta = atan2f(o.y, o.x);
            sn = fast_sin_f32(theta);
            cs = fast_cos_f32(theta);
            lutMiss = lut_lookup(T, dist, maxScale, translation) ? 0u : 1u;
            lens.x *= maxScale; lens.y *= maxScale;
            lens.x += translation;
            const float rx = lens.x * cs - lens.y * sn, ry = lens.x * sn + lens.y * cs;
            d = V3{rx - o.x, ry - o.y, T.dirZ};
        }
        int tries = 0;
        while (!trace_lens_fast_rolled(T, o, d, tir) && tries <= kMaxTries) {
            o = o0;
            const float u = rng_unit(xor128(rng));
            const float v = rng_unit(xor128(rng));
            lens = sample_lens_f32(useImage, B, T.bokehW, T.bokehH, u, v);
            if (!T.useLUT) {
                d = V3{(lens.x * T.rearAperture) - o.x, (lens.y * T.rearAperture) - o.y, T.dirZ};
            } else {
                lens.x *= maxScale; lens.y *= maxScale;
                lens.x += translation; lens.y += translation;
                const float rx = lens.x * cs - lens.y * sn, ry = lens.x * sn + lens.y * cs;
                d = V3{rx - o.x, ry - o.y, T.dirZ};
            }
            ++tries;
        }
        float w = 1.0f;
        if (tries > kMaxTries) { w = 0.0f; ++vign; } else ++succ;
        if (T.exposureOn) w *= T.exposureMul;
        const uint32_t flags = (tries > 0 ? 1u : 0u) | (static_cast<uint32_t>(tries) << 1) | (lutMiss << 6);
        store_ray_record(out, i, -o.x, -o.y, -o.z, -d.x, -d.y, -d.z, w, flags);
    }
    // workgroup reduction of the three counters, one atomic each
    __shared__ uint32_t part[3][kBlockF / 64];
    for (int off = 32; off > 0; off >>= 1) {
        succ += __shfl_down(succ, off, 64);
        vign += __shfl_down(vign, off, 64);
        tir += __shfl_down(tir, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { part[0][wave] = succ; part[1][wave] = vign; part[2][wave] = tir; }
    __syncthreads();
    if (counters && threadIdx.x < 3) {
        unsigned long long sum = 0;
        for (int w = 0; w < kBlockF / 64; ++w) sum += part[threadIdx.x][w];
        DeviceCounters *cs = counter_set(counters);
        unsigned long long *dst = threadIdx.x == 0 ? &cs->succes : (threadIdx.x == 1 ? &cs->vignetted : &cs->tir);
        if (sum) atomicAdd(dst, sum);
    }
}

int launch_kolb_fast(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                     uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, void *stream)
{
    const uint64_t blocks = (n + kBlockF - 1) / kBlockF;
    const unsigned grid = static_cast<unsigned>(blocks < 2048 ? (blocks ? blocks : 1) : 2048);
    hipLaunchKernelGGL(kolb_rays_fast_kernel, dim3(grid), dim3(kBlockF), 0, static_cast<hipStream_t>(stream), table, bokeh,
                       reinterpret_cast<const float4 *>(d_samples), reinterpret_cast<const uint4 *>(d_rng), rayBase, n, out,
                       d_counters);
    return static_cast<int>(hipGetLastError());
}

}  // namespace zoic
