// kernels.hip -- the small hand-written gfx950 (CDNA4, wave64) kernels of zoic's per-sample lens hot path: the streaming
// thin-lens kernel, sample synthesis, the exit-pupil LUT probes and the Arnold AoS packing.  (The Kolb kernels:
// kolb_pool_body.hpp; the thin-lens kernel with optical vignetting: thin_refill.hip.)
//
// Mapping: one camera sample per lane, 256-lane workgroups (4 waves = one per SIMD), a grid capped at 8 workgroups per
// CU that strides over the sample buffer.  Tables arrive as by-value kernel arguments: wave-uniform, fetched by s_load
// through the scalar cache -- no LDS and no VGPRs spent on them.  Sample loads are one 16-byte global_load_dwordx4 per lane
// (1 KiB per wave instruction); results leave as one 32-byte record per ray.  The path is scalar FP32 per ray: no
// contraction to feed MFMA.  STRICT arithmetic (optics.hpp): bit-exact against the CPU oracle.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <cstring>

#include "device_search.hpp"
#include "kernels.hpp"
#include "ray_store.hpp"
#include "optics.hpp"
#include "thin_device.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

constexpr int kBlock = 256;

// sum three per-lane counters over the workgroup, one atomic per counter per workgroup
__device__ __forceinline__ void flush_counters(DeviceCounters *c, uint32_t succ, uint32_t vign, uint32_t tir)
{
    if (!c) return;
    c = counter_set(c);
    __shared__ uint32_t part[3][kBlock / 64];
    for (int off = 32; off > 0; off >>= 1) {
        succ += __shfl_down(succ, off, 64);
        vign += __shfl_down(vign, off, 64);
        tir += __shfl_down(tir, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { part[0][wave] = succ; part[1][wave] = vign; part[2][wave] = tir; }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += part[threadIdx.x][w];
        unsigned long long *dst = threadIdx.x == 0 ? &c->succes : (threadIdx.x == 1 ? &c->vignetted : &c->tir);
        if (s) atomicAdd(dst, s);
    }
}

extern __shared__ __align__(16) float thinDynLds[];   // bokeh row cell records (thin-lens kernel), after the static LDS

// ------------------------------------------------------------------------------------- THINLENS
// zoic.cpp:1771-1846 + empericalOpticalVignetting zoic.cpp:1297-1305 (thin_device.hpp).  All f32; ~60 flop / 44 B: HBM-bound.
__global__ __launch_bounds__(kBlock) void thin_rays_kernel(const ThinTable T, const BokehTables B,
                                                           const float4 *__restrict__ samples,
                                                           const uint4 *__restrict__ rngStates, uint64_t rayBase, uint64_t n,
                                                           RayRecord *__restrict__ out, DeviceCounters *counters, uint32_t ldsWords)
{
    uint32_t succ = 0, vign = 0;
    __shared__ float4 stage[kBlock / 64][128];   // per-wave transpose buffer for the coalesced record store
    const float *rowCells = nullptr;
    if (ldsWords > 0) {                          // bokeh row cell records, once per workgroup (tables.hpp)
        for (uint32_t i = threadIdx.x; i < ldsWords; i += kBlock) thinDynLds[i] = __builtin_bit_cast(float, B.rowCells[i]);
        rowCells = thinDynLds;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    // software pipeline: the sample of the wave's NEXT tile is requested before the current one is evaluated
    const uint64_t first = static_cast<uint64_t>(blockIdx.x) * kBlock + wave * 64u + lane;
    float4 sNext = nt_load(samples + (first < n ? first : n - 1));
    for (uint64_t tile = static_cast<uint64_t>(blockIdx.x) * kBlock; tile < n; tile += stride) {  // whole waves stay together
        const uint64_t waveBase = tile + wave * 64u;
        if (waveBase >= n) continue;
        const uint64_t i = waveBase + lane;
        const bool have = i < n;
        const float4 s = sNext;
        {
            const uint64_t j = i + stride;
            sNext = nt_load(samples + (j < n ? j : n - 1));
        }
        Rng rng{1u, 2u, 3u, 4u};
        const ThinRay r = thin_ray_strict(T, B, rowCells, s, rng, [&] {   // the private retry stream is seeded at the first retry only
            if (rngStates) { const uint4 q = rngStates[have ? i : n - 1]; rng = Rng{q.x, q.y, q.z, q.w}; }
            else rng = rng_for_ray(T.seed, rayBase + i);
        });
        if (T.useDof && have) { if (r.tries > static_cast<uint32_t>(kMaxTries)) ++vign; else ++succ; }  // zoic.cpp:1824-1830
        const uint64_t left = n - waveBase;
        store_ray_records_wave(out, waveBase, lane, left < 64 ? static_cast<uint32_t>(left) : 64u, stage[wave], r.origin.x, r.origin.y,
                               r.origin.z, r.dir.x, r.dir.y, r.dir.z, r.w, (r.tries > 0 ? 1u : 0u) | (r.tries << 1));
    }
    flush_counters(counters, succ, vign, 0u);
}

// ------------------------------------------------------------------------------------- synthetic samples
__device__ __forceinline__ float u01_24(uint32_t h) { return static_cast<float>(h >> 8) * 5.9604644775390625e-08f; }

__global__ __launch_bounds__(kBlock) void generate_samples_kernel(float4 *__restrict__ samples, uint64_t rayBase, uint64_t n,
                                                                  uint32_t W, uint32_t H, uint32_t spp, uint32_t seed)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    const float fW = static_cast<float>(W), fH = static_cast<float>(H);
    const float aspect = fW / fH;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t id = rayBase + i;
        const uint64_t pix = id / spp;
        const uint32_t px = static_cast<uint32_t>(pix % W), py = static_cast<uint32_t>((pix / W) % H);
        const uint32_t lo = static_cast<uint32_t>(id), hi = static_cast<uint32_t>(id >> 32);
        const uint32_t key = seed ^ pcg_hash(hi ^ 0x632BE5ABu);
        const float jx = u01_24(pcg_hash(key ^ (lo * 4u + 0u)));
        const float jy = u01_24(pcg_hash(key ^ (lo * 4u + 1u) ^ 0x85EBCA6Bu));
        const float lx = u01_24(pcg_hash(key ^ (lo * 4u + 2u) ^ 0xC2B2AE35u));
        const float ly = u01_24(pcg_hash(key ^ (lo * 4u + 3u) ^ 0x27D4EB2Fu));
        const float sx = 2.0f * (static_cast<float>(px) + jx) / fW - 1.0f;
        const float sy = (1.0f - 2.0f * (static_cast<float>(py) + jy) / fH) / aspect;
        samples[i] = make_float4(sx, sy, lx, ly);
    }
}

// ------------------------------------------------------------------------------------- exit-pupil LUT probes
__global__ __launch_bounds__(kBlock) void lut_probe_kernel(const KolbTable T, float originX, const float *__restrict__ lensU,
                                                           const float *__restrict__ lensV, uint64_t n,
                                                           uint8_t *__restrict__ accepted, unsigned int *tirOut)
{
    uint32_t tir = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        V3 o{originX, 0.0f, T.originShift};
        V3 d{(lensU[i] * T.rearAperture) - originX, (lensV[i] * T.rearAperture) - 0.0f, T.dirZ};
        accepted[i] = trace_lens_strict(T, o, d, tir) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) tir += __shfl_down(tir, off, 64);
    if ((threadIdx.x & 63) == 0 && tir && tirOut) atomicAdd(tirOut, tir);
}

// ------------------------------------------------------------------------------------- Arnold AoS packing
__global__ __launch_bounds__(kBlock) void pack_inputs_kernel(const float *__restrict__ in7, float4 *__restrict__ out4, uint64_t n)
{
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        const float *p = in7 + i * 7;  // sx sy dsx dsy lensx lensy relative_time
        out4[i] = make_float4(p[0], p[1], p[4], p[5]);
    }
}

// 32-byte records -> AtCameraOutput rows (84 bytes = 21 floats: origin, dir, dOdx, dOdy, dDdx, dDdy, weight[3]), the fields
// camera_create_ray writes (zoic.cpp:1960-1961 origin / dir, 1974-1977 dOdy = origin and dDdy = dir for retried rays,
// 1952 / 1981-1987 weight) and zeros in the ones it leaves alone.  One lane per output FLOAT: stores are fully coalesced
// (the 21 lanes of a ray read its one record through the L1).
__global__ __launch_bounds__(kBlock) void expand_outputs_kernel(const RayRecord *__restrict__ rays, float *__restrict__ out21, uint64_t n)
{
    const uint64_t total = n * 21u, stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t ray = t / 21u;
        const uint32_t f = static_cast<uint32_t>(t - ray * 21u);
        const float *r = reinterpret_cast<const float *>(rays + ray);
        const bool retried = (__builtin_bit_cast(uint32_t, r[7]) & 1u) != 0u;
        float v = 0.0f;                                  // dOdx (6-8), dDdx (12-14); dOdy / dDdy of first-try rays
        if (f < 6u) v = r[f];                            // origin, dir
        else if (f >= 18u) v = r[6];                     // weight r = g = b (the caller's initial weight is 1)
        else if (retried && f >= 9u && f < 12u) v = r[f - 9u];    // dOdy = origin
        else if (retried && f >= 15u) v = r[f - 12u];             // dDdy = dir
        out21[t] = v;
    }
}

// 32-byte records -> rows of 7 floats (origin, dir, weight): the 28-byte payload SURVEY 8(e) gathers on the root GPU.  One lane
// per output float: the stores are fully coalesced, the loads hit each record's sector once per wave.
__global__ __launch_bounds__(kBlock) void pack_payload_kernel(const RayRecord *__restrict__ rays, float *__restrict__ out7, uint64_t n)
{
    const uint64_t total = n * 7u, stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t ray = t / 7u;
        out7[t] = reinterpret_cast<const float *>(rays + ray)[static_cast<uint32_t>(t - ray * 7u)];
    }
}

// ------------------------------------------------------------------------------------- sparse payload (kernels.hpp)
// One workgroup per 256-ray tile: ballot per wave -> the tile's 256-bit mask, one atomicAdd for its row range, every live lane
// writes its 28-byte row at (range start + its rank).  SURVEY 8(e)'s gather ships 28 B for every ray; on a wide-open PETZVAL four
// fifths of them have weight 0 (zoic.cpp:1951-1953) and nothing downstream reads their origin / direction.
__global__ __launch_bounds__(kBlock) void pack_sparse_kernel(const RayRecord *__restrict__ rays, uint32_t *__restrict__ sparse, unsigned int *count, uint64_t m)
{
    __shared__ uint32_t waveCount[kBlock / 64];
    __shared__ uint32_t tileBase;
    const uint64_t tiles = (m + kSparseTileRays - 1) / kSparseTileRays;
    float *rows = reinterpret_cast<float *>(sparse + tiles * kSparseTileWords);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t i = tile * kSparseTileRays + threadIdx.x;
        float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (i < m) {
            const float4 a = reinterpret_cast<const float4 *>(rays + i)[0], b = reinterpret_cast<const float4 *>(rays + i)[1];
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z;
        }
        const bool live = i < m && r[6] != 0.0f;
        const unsigned long long mask = __ballot(live);
        uint32_t *hdr = sparse + tile * kSparseTileWords;
        if (lane == 0) { hdr[2 * wave] = static_cast<uint32_t>(mask); hdr[2 * wave + 1] = static_cast<uint32_t>(mask >> 32); waveCount[wave] = static_cast<uint32_t>(__popcll(mask)); }
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < kBlock / 64; ++w) { if (w < wave) before += waveCount[w]; total += waveCount[w]; }
        if (threadIdx.x == 0) { tileBase = atomicAdd(count, total); hdr[8] = tileBase; hdr[9] = total; hdr[10] = 0u; hdr[11] = 0u; }
        __syncthreads();
        if (live) {
            const uint32_t rank = before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
            float *dst = rows + static_cast<size_t>(tileBase + rank) * 7u;
#pragma unroll
            for (int f = 0; f < 7; ++f) dst[f] = r[f];
        }
        __syncthreads();   // waveCount / tileBase are reused by the next tile
    }
}

// one lane per output float (coalesced stores); a ray's row index = its tile's offset + the live rays before it in the tile
__global__ __launch_bounds__(kBlock) void expand_sparse_kernel(const uint32_t *__restrict__ sparse, float *__restrict__ out7, uint64_t m)
{
    const uint64_t tiles = (m + kSparseTileRays - 1) / kSparseTileRays;
    const float *rows = reinterpret_cast<const float *>(sparse + tiles * kSparseTileWords);
    const uint64_t total = m * 7u, stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t ray = t / 7u;
        const uint32_t f = static_cast<uint32_t>(t - ray * 7u);
        const uint32_t *hdr = sparse + (ray / kSparseTileRays) * kSparseTileWords;
        const uint32_t in = static_cast<uint32_t>(ray % kSparseTileRays), word = in >> 5, bit = in & 31u;
        float v = 0.0f;
        if ((hdr[word] >> bit) & 1u) {
            uint32_t rank = __builtin_popcount(hdr[word] & ((1u << bit) - 1u));
            for (uint32_t w = 0; w < word; ++w) rank += __builtin_popcount(hdr[w]);
            v = rows[static_cast<size_t>(hdr[8] + rank) * 7u + f];
        }
        out7[t] = v;
    }
}

__global__ __launch_bounds__(kBlock) void pack_payload_live_kernel(const RayRecord *__restrict__ rays, float *__restrict__ out7, uint64_t n)
{
    const uint64_t total = n * 7u, stride = static_cast<uint64_t>(gridDim.x) * kBlock;
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t ray = t / 7u;
        const float *r = reinterpret_cast<const float *>(rays + ray);
        out7[t] = r[6] != 0.0f ? r[static_cast<uint32_t>(t - ray * 7u)] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------- launchers
static inline unsigned grid_for(uint64_t n)
{
    // 256 CUs x 8 workgroups of 256 lanes = every wave slot of the chip (32 waves/CU); grid-stride the rest
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks < 2048 ? (blocks ? blocks : 1) : 2048);
}

int launch_thin_refill(const ThinTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng, uint64_t rayBase,
                       uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor, bool fast, void *stream);

int launch_thin_rays(const ThinTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                     uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor, bool fast,
                     void *stream)
{
    if (n == 0) return 0;
    if (table.useDof && table.ovDistance > 0.0f)   // the retry loop of zoic.cpp:1804-1819 can run
        return launch_thin_refill(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_workCursor, fast, stream);
    const uint32_t ldsWords = (table.useImage && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
    hipLaunchKernelGGL(thin_rays_kernel, dim3(grid_for(n)), dim3(kBlock), ldsWords * sizeof(float), static_cast<hipStream_t>(stream),
                       table, bokeh, reinterpret_cast<const float4 *>(d_samples), reinterpret_cast<const uint4 *>(d_rng), rayBase, n, out,
                       d_counters, ldsWords);
    return static_cast<int>(hipGetLastError());
}

int launch_generate_samples(float *d_samples, uint64_t rayBase, uint64_t n, uint32_t width, uint32_t height, uint32_t spp,
                            uint32_t seed, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(generate_samples_kernel, dim3(grid_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<float4 *>(d_samples), rayBase, n, width, height, spp, seed);
    return static_cast<int>(hipGetLastError());
}

int launch_lut_probes(const KolbTable &table, float originX, const float *d_lensU, const float *d_lensV, uint64_t n,
                      uint8_t *d_accepted, unsigned int *d_tir, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(lut_probe_kernel, dim3(grid_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), table, originX,
                       d_lensU, d_lensV, n, d_accepted, d_tir);
    return static_cast<int>(hipGetLastError());
}

int launch_pack_inputs(const float *d_inputs7, float *d_samples4, uint64_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_inputs_kernel, dim3(grid_for(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), d_inputs7,
                       reinterpret_cast<float4 *>(d_samples4), n);
    return static_cast<int>(hipGetLastError());
}

int launch_pack_payload(const RayRecord *d_rays, float *d_out7, uint64_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_payload_kernel, dim3(grid_for(n * 7u)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), d_rays, d_out7, n);
    return static_cast<int>(hipGetLastError());
}

int launch_pack_sparse(const RayRecord *d_rays, uint32_t *d_sparse, unsigned int *d_count, uint64_t m, void *stream)
{
    if (m == 0) return 0;
    const uint64_t tiles = (m + kSparseTileRays - 1) / kSparseTileRays;
    hipLaunchKernelGGL(pack_sparse_kernel, dim3(static_cast<unsigned>(tiles < 8192 ? tiles : 8192)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), d_rays, d_sparse, d_count, m);
    return static_cast<int>(hipGetLastError());
}

int launch_expand_sparse(const uint32_t *d_sparse, float *d_out7, uint64_t m, void *stream)
{
    if (m == 0) return 0;
    hipLaunchKernelGGL(expand_sparse_kernel, dim3(grid_for(m * 7u)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), d_sparse, d_out7, m);
    return static_cast<int>(hipGetLastError());
}

int launch_pack_payload_live(const RayRecord *d_rays, float *d_out7, uint64_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_payload_live_kernel, dim3(grid_for(n * 7u)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), d_rays, d_out7, n);
    return static_cast<int>(hipGetLastError());
}

int launch_expand_outputs(const RayRecord *d_rays, float *d_out21, uint64_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(expand_outputs_kernel, dim3(grid_for(n * 21u)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), d_rays, d_out21, n);
    return static_cast<int>(hipGetLastError());
}

}  // namespace zoic
