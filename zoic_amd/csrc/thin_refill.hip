// thin_refill.hip -- THINLENS with empirical optical vignetting (zoic.cpp:1771-1846, the retry loop 1804-1819) as a
// persistent-wave lane-refill kernel.
//
// Without opticalVignettingDistance a thin-lens sample needs no retry and the streaming kernel of kernels.hip is the
// right shape (HBM-bound, one sample per lane, coalesced records).  With it, up to 26 redraws per sample and a zero-weight
// tail make the in-lane loop run 27 iterations in practically every wave: opticalVignettingDistance 5 took 100 -> 20
// Grays/s.  Here a pass runs ONE test per lane (the reference's loop condition for the sample the lane holds) and lanes
// whose ray is finished are refilled from the wave's sample window, exactly like kolb_refill.hip: same work cursors
// (work_cursor.hpp), same LDS parking of finished records (ray_store.hpp), same per-ray retry streams -- so the result of
// a ray does not depend on lane, pass or wave and stays bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include "device_search.hpp"
#include "fast_optics.hpp"
#include "kernels.hpp"
#include "optics.hpp"
#include "ray_store.hpp"
#include "thin_device.hpp"
#include "work_cursor.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

namespace {

constexpr int kThinBlock = 256;
constexpr int kThinWaves = kThinBlock / 64;
constexpr uint32_t kMinTrying = 16;   // the redraw loop goes on while at least this many lanes of the wave are redrawing

extern __shared__ __align__(16) float thinRefillLds[];   // [bokeh row cell records (ldsWords)] [per wave: 128 pieces + 64 indices]

// FAST (zoic_camera_set_precision): f32 rsq normalisation, the f32 disk mapping and v_sqrt in the vignetting test instead
// of the reference's correctly rounded divides and square roots -- ~80 instead of ~140 instructions per redraw; direction
// error ~1e-7; a vignetting test within a few ulps of its limit is re-taken in the reference's arithmetic (decision-safe, like
// the Kolb kernels' guard band).  STRICT is bit-exact.
template <bool FAST>
__global__ __launch_bounds__(kThinBlock) void thin_refill_kernel(const ThinTable T, const BokehTables B, const float4 *__restrict__ samples,
                                                                 const uint4 *__restrict__ rngStates, uint64_t rayBase, uint32_t n,
                                                                 RayRecord *__restrict__ out, DeviceCounters *counters,
                                                                 unsigned int *__restrict__ workCursor, uint32_t ldsWords,
                                                                 uint32_t chunkRays, uint32_t chunksPerPart)
{
    const uint32_t lane = threadIdx.x & 63u;
    const bool useImage = T.useImage != 0;
    const float *rowCells = nullptr;
    if (ldsWords > 0) {                          // bokeh row cell records, once per workgroup (tables.hpp)
        for (uint32_t i = threadIdx.x; i < ldsWords; i += kThinBlock) thinRefillLds[i] = __builtin_bit_cast(float, B.rowCells[i]);
        rowCells = thinRefillLds;
        __syncthreads();
    }
    float4 *stage = reinterpret_cast<float4 *>(thinRefillLds + ldsWords) + (threadIdx.x >> 6) * 144u;
    uint32_t *stageIdx = reinterpret_cast<uint32_t *>(stage + 128);
    bool parked = false;

    uint32_t next = 0, end = 0, part = blockIdx.x % kCursorParts, partsTried = 0;
    bool exhausted = false;
    float4 win = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t winBase = 0xffffffffu;

    // per-lane ray state, alive across passes: the focus point (loop invariant of zoic.cpp:1804-1819) and the lens sample
    // whose ray the next pass tests
    bool active = false;
    uint32_t idx = 0, tries = 0;
    float fpx = 0, fpy = 0, fpz = 0, u = 0, v = 0;
    Rng rng{1, 2, 3, 4};
    uint32_t succ = 0, vign = 0;

    for (;;) {
        // ---- refill the free lanes from the sample window (ballot + prefix sum) -----------------------------------
        unsigned long long freeMask = __ballot(!active);
        while (freeMask != 0ull && !exhausted) {
            if (next >= end && !claim_chunk(workCursor, lane, part, partsTried, chunkRays, chunksPerPart, n, next, end)) { exhausted = true; break; }
            if (winBase != next) {   // first use of a chunk: fetched in line (once per chunk)
                const uint32_t wi = next + lane;
                win = samples[wi < n ? wi : n - 1];
                winBase = next;
            }
            const uint32_t avail = end - next;
            const uint32_t nfree = static_cast<uint32_t>(__popcll(freeMask));
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(freeMask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(freeMask), 0u));
            const float4 s = make_float4(__shfl(win.x, rank, 64), __shfl(win.y, rank, 64), __shfl(win.z, rank, 64), __shfl(win.w, rank, 64));
            if (!active && rank < avail) {
                idx = next + rank;
                const V3 p{s.x * T.tanFov, s.y * T.tanFov, 1.0f};           // zoic.cpp:1773-1777 (output.origin arrives as 0)
                const V3 dir0 = normalize3(V3{p.x - 0.0f, p.y - 0.0f, p.z - 0.0f});
                const float inter = fabsf(T.focalDistance / dir0.z);        // zoic.cpp:1796-1797
                fpx = dir0.x * inter; fpy = dir0.y * inter; fpz = dir0.z * inter;
                u = s.z; v = s.w;
                tries = 0;
                active = true;
            }
            next += (nfree < avail) ? nfree : avail;
            if (nfree <= avail) break;
            freeMask = __ballot(!active);
        }
        if (__ballot(active) == 0ull) break;

        // ---- re-base the window (consumed by the next pass's refill), write out what the last pass parked -----------
        if (next < end && winBase != next) {
            const uint32_t wi = next + lane;
            win = samples[wi < n ? wi : n - 1];
            winBase = next;
        }
        if (parked) { flush_parked_records(out, stage, stageIdx, lane); parked = false; }

        // ---- the reference's loop, zoic.cpp:1804-1819, for every active lane ------------------------------------------
        //     while (!empericalOpticalVignetting(origin, dir, ...) && tries <= 25) { redraw; ++tries; }
        // A lane evaluates the condition for the sample it holds and either finishes or redraws; the loop is wave-uniform
        // and goes on while at least kMinTrying lanes are still redrawing (heavy vignetting: the whole wave stays in this
        // tight loop; light vignetting: the few unlucky lanes carry their ray into the next pass next to fresh ones).
        uint32_t finishedIdx = 0xffffffffu;
        bool trying = active;
        for (;;) {
            if (trying) {
                V3 origin, dir;
                const bool clear = thin_vignet_try<FAST>(T, B, rowCells, useImage, fpx, fpy, fpz, u, v, origin, dir);   // thin_device.hpp
                const bool done = clear || tries > static_cast<uint32_t>(kMaxTries);
                if (done) {
                    float w = (tries > static_cast<uint32_t>(kMaxTries)) ? 0.0f : 1.0f;      // zoic.cpp:1824-1830
                    dir.z = dir.z * -1.0f;                                                   // zoic.cpp:1845
                    if (T.exposureOn) w *= T.exposureMul;
                    stage[2 * lane] = make_float4(origin.x, origin.y, origin.z, dir.x);
                    stage[2 * lane + 1] = make_float4(dir.y, dir.z, w, __builtin_bit_cast(float, (tries > 0 ? 1u : 0u) | (tries << 1)));
                    finishedIdx = idx;
                    active = false;
                    trying = false;
                } else {                                     // redraw from the ray's own stream, seeded at its first retry
                    if (tries == 0) {
                        if (rngStates) { const uint4 r = rngStates[idx]; rng = Rng{r.x, r.y, r.z, r.w}; }
                        else rng = rng_for_ray(T.seed, rayBase + idx);
                    }
                    u = rng_unit(xor128(rng));
                    v = rng_unit(xor128(rng));
                    ++tries;
                }
            }
            if (static_cast<uint32_t>(__popcll(__ballot(trying))) < kMinTrying) break;
        }
        {
            const bool fin = finishedIdx != 0xffffffffu;
            const uint32_t nv = static_cast<uint32_t>(__popcll(__ballot(fin && tries > static_cast<uint32_t>(kMaxTries))));
            vign += nv;
            succ += static_cast<uint32_t>(__popcll(__ballot(fin))) - nv;
        }
        stageIdx[lane] = finishedIdx;
        parked = true;
    }
    if (parked) flush_parked_records(out, stage, stageIdx, lane);
    if (counters && lane == 0) {
        DeviceCounters *cs = counter_set(counters);
        if (succ) atomicAdd(&cs->succes, static_cast<unsigned long long>(succ));
        if (vign) atomicAdd(&cs->vignetted, static_cast<unsigned long long>(vign));
    }
}

}  // namespace

int launch_thin_refill(const ThinTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng, uint64_t rayBase,
                       uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor, bool fast, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    constexpr uint64_t kMaxPerLaunch = 1ull << 31;   // 32-bit ray offsets inside the kernel; larger batches are split
    for (uint64_t done = 0; done < n; done += kMaxPerLaunch) {
        const uint64_t m = (n - done < kMaxPerLaunch) ? (n - done) : kMaxPerLaunch;
        hipError_t e = reset_work_cursors(d_workCursor, st);
        if (e != hipSuccess) return static_cast<int>(e);
        const WorkGrain grain = work_grain(m);
        const uint32_t ldsWords = (table.useImage && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
        const size_t ldsBytes = static_cast<size_t>(ldsWords) * sizeof(float) + kThinWaves * 144 * sizeof(float4);
#define ZOIC_LAUNCH_THIN(FAST_)                                                                                                  \
    hipLaunchKernelGGL(thin_refill_kernel<FAST_>, dim3(persistent_grid(m, kThinWaves)), dim3(kThinBlock), ldsBytes, st, table, bokeh,  \
                       reinterpret_cast<const float4 *>(d_samples) + done, d_rng ? reinterpret_cast<const uint4 *>(d_rng) + done : nullptr, \
                       rayBase + done, static_cast<uint32_t>(m), out + done, d_counters, d_workCursor, ldsWords, grain.chunkRays,      \
                       grain.chunksPerPart)
        if (fast) ZOIC_LAUNCH_THIN(true); else ZOIC_LAUNCH_THIN(false);
#undef ZOIC_LAUNCH_THIN
        e = hipGetLastError();
        if (e != hipSuccess) return static_cast<int>(e);
    }
    return 0;
}

}  // namespace zoic
