// kolb_refill_body.hpp -- the production Kolb kernel: persistent waves with ballot/prefix-sum lane refill.
// (compiled twice: kolb_refill.hip instantiates the kernels for cameras without retry-dead rays, kolb_refill_dead.hip those with)
//
// Why.  camera_create_ray retries a rejected sample up to 26 more times (zoic.cpp:1927-1947).  With one sample per
// lane and the retry loop inside the lane, a wave keeps iterating until its unluckiest lane is done: rocprof on the
// first kernel (profiles/r01_fast_v0) showed ~25 % VALU lane utilisation and 6000 lane-instructions per ray for a
// ~900-instruction first try.  Here a wave is persistent instead: every pass of the loop runs exactly ONE try for each
// of its 64 lanes; lanes whose ray finished (accepted, or out of tries) are refilled from the wave's sample cursor
// before the next pass:
//     freeMask = ballot(!active)             -- which lanes need work
//     rank     = mbcnt(freeMask)             -- exclusive prefix sum: my slot among the free lanes
//     mine     = cursor + rank               -- consecutive samples go to the free lanes (loads stay contiguous)
//     cursor  += popcount(freeMask)          -- wave-uniform, lives in an SGPR; no atomics, no LDS, no barriers
// so vignetted rays never hold finished lanes hostage, whatever the reject rate.  The cursor walks a chunk (64 ... 1024
// samples by batch size, 512 on a 4K x 16spp frame) claimed with one atomicAdd on one of eight partition cursors: waves
// that drew cheap image regions simply claim more chunks, so the frame's heavily vignetted corners cannot unbalance the
// chip and the grid size need not match the true residency.  Retry streams are per ray (keyed by the global ray index),
// hence the result of every ray is independent of which lane/pass/wave evaluated it -- the strict instantiation is
// bit-identical to the simple one-sample-per-lane kernel and to the CPU oracle.
//
// Tables: the lens prescription + LUT arrive by value (SGPRs via s_load, see tables.hpp); a bokeh lens sample is one LDS
// cell record + one global cell record (device_search.hpp; the 16-ary pyramid only for images without records).
// Samples: a 64-entry prefetch window per wave, one global_load_dwordx4 per lane per pass, handed to refilled lanes with
// ds_bpermute.  Rays: one 32-byte record per ray, parked in LDS by the finishing lane and written before the next trace
// as whole sectors (flush_parked_records) -- the scattered completion order causes no partial-line write-backs.
// Pass order (vmcnt is one in-order counter, see below): refill -> candidate search (draws, interface-0 test) ->
// window prefetch + record flush -> trace -> finish (park).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "kolb_device.hpp"
#include "work_cursor.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

// Register budgets.  SGPRs: a 256-lane workgroup is admitted per CU up to floor(800 / (ceil(sgpr/16)*16 + 16)) times
// (MI355X_MICROARCH.md): 106 SGPRs -> 6 workgroups, <= 96 -> 7; capping at 94 measured +4.5 % on C3.  VGPRs: the fast
// instantiations needed 73 (72 + the SGPR-spill register), one over the 7-waves-per-SIMD line (512 / 7 -> 72); with the
// per-lane counters moved to SGPRs they fit 71 and asking for 7 waves measured +2.6 % on C3.  The strict instantiations (127-137 VGPRs, f64
// intermediates) are held to 128 = 4 waves per SIMD (3 otherwise): +6-8 % (5 waves = 96 VGPRs spills too much: -12 %).
#ifndef ZOIC_REFILL_ATTR_STRICT
#define ZOIC_REFILL_ATTR_STRICT __attribute__((amdgpu_num_sgpr(94), amdgpu_waves_per_eu(4, 4)))
#endif
#ifndef ZOIC_REFILL_ATTR_FAST
#define ZOIC_REFILL_ATTR_FAST __attribute__((amdgpu_num_sgpr(94), amdgpu_waves_per_eu(7, 8)))
#endif

// Debug build only (-DZOIC_REGION_TIMERS, tools/region_times.py): per-wave s_memtime cycles spent in each region of the
// pass loop, summed over all waves.  Not part of the product build.
#ifdef ZOIC_REGION_TIMERS
static __device__ unsigned long long g_regionCycles[8];   // one copy per translation unit (summed by zoic_debug_*)
static __device__ unsigned long long g_waveLog[8192 * 4];   // per wave (blockIdx * 4 + wave): start, exhausted, end (s_memrealtime, 100 MHz), passes
static __device__ unsigned long long g_passStats[8];   // passes, sum active lanes, search iterations, sum looking lanes, trace passes, sum cand lanes, rays finished
#define ZOIC_WL_EXH if (wlExh == 0) wlExh = wall_clock64();
#define ZOIC_PS_DECL unsigned long long ps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ZOIC_PS_ADD(I, V) ps[I] += (V);
#define ZOIC_PS_FLUSH if (lane == 0) { for (int r = 0; r < 8; ++r) atomicAdd(&g_passStats[r], ps[r]); }
#define ZOIC_RT_DECL unsigned long long rtAcc[5] = {0, 0, 0, 0, 0}, rtLast = __builtin_readcyclecounter(), wlStart = wall_clock64(), wlExh = 0, wlPasses = 0;
#define ZOIC_RT_MARK(R) { const unsigned long long rtNow = __builtin_readcyclecounter(); rtAcc[R] += rtNow - rtLast; rtLast = rtNow; }
#define ZOIC_RT_FLUSH if (lane == 0) { for (int r = 0; r < 5; ++r) atomicAdd(&g_regionCycles[r], rtAcc[r]); atomicAdd(&g_regionCycles[7], 1ull); \
        const uint32_t wl = (blockIdx.x * 4u + (threadIdx.x >> 6)) & 8191u; g_waveLog[4 * wl] = wlStart; g_waveLog[4 * wl + 1] = wlExh; g_waveLog[4 * wl + 2] = wall_clock64(); g_waveLog[4 * wl + 3] = wlPasses; }
#else
#define ZOIC_WL_EXH
#define ZOIC_PS_DECL
#define ZOIC_PS_ADD(I, V)
#define ZOIC_PS_FLUSH
#define ZOIC_RT_DECL
#define ZOIC_RT_MARK(R)
#define ZOIC_RT_FLUSH
#endif

// Kernel arguments that only rare paths read (chunk claim, first retry, work-list flush, exit) are fetched from the kernarg
// segment where they are used instead of living in SGPRs for the whole kernel: the pass loop carries ~60 scalars, the
// budget is 94, and what does not fit is spilled to VGPR lanes and paid for with a v_readlane per use inside the trace.
// RefillArgs mirrors the kernels' parameter list (HIP lays kernel arguments out like a C struct; offsets checked against the
// code object's metadata, tools/isa_mix.py).
struct RefillArgs {
    KolbTable T; BokehTables B; const float4 *samples; const uint4 *rngStates; uint64_t rayBase; uint32_t n; RayRecord *out;
    DeviceCounters *counters; unsigned int *workCursor; uint32_t ldsWords, chunkRays, chunksPerPart, minSearching;
    uint32_t *redoList; unsigned int *redoCount;             // GUARD kernel: appends the rays it cannot decide; LISTED kernel: reads them
    uint8_t *deadMap;                                         // DEAD kernels: one byte per sample, set for the rays kolb_finish_kernel completes
};
template <class V, size_t OFFSET>
__device__ __forceinline__ V kernarg_field()
{
    typedef const char __attribute__((address_space(4))) *KernargBytes;
    KernargBytes base = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));   // loaded here, every time: not hoisted into a long-lived SGPR
    return *(const V __attribute__((address_space(4))) *)(base + OFFSET);
}
#define ZOIC_KARG(field) kernarg_field<decltype(RefillArgs::field), offsetof(RefillArgs, field)>()

// One body, four kernels.  A decision-safe FAST launch is a pipeline of two persistent kernels on the caller's stream:
//   GUARD (FAST only): every accept/reject decision of a try at a guarded interface (tables.hpp FastSurface::bandHousing:
//       in practice the stop) is checked against its guard band; a ray with a decision too close to call is dropped where
//       it stands -- no record, no counter -- and its index goes to the work list `redoList`;
//   LISTED (STRICT only) = the kernel that runs next on the stream and evaluates exactly the listed rays from scratch in
//       the reference's arithmetic (per-ray retry streams make that the same ray).
// Together: every ray's try count, weight and flags are the reference's; only the low-order bits of origin / direction of
// the FAST-evaluated rays differ.  Drops are staged in the wave's own LDS list, branch-free, every pass, and moved to the
// global list in whole batches OUTSIDE the pass loop (a wave-uniform rare block with an atomic inside the pass loop makes
// LLVM spill ~60 more SGPRs to VGPR lanes: 320 v_readlane against 90, +12-20 % kernel time).  TIR bumps are tallied per
// ray (above bit 0 of lutMiss) and reach the counters only when the ray finishes in this kernel.
// DEAD instantiations (cameras with KolbTable::retryOn) also drop "retry-dead" rays whose first try failed: staged in a
// second LDS list, marked in the launch's byte map `deadMap` outside the pass loop, completed by kolb_finish_kernel, which
// only has to step their retry stream (below).
// (A third role -- handing "search-heavy" rays, still without a candidate after a few draws, to a kernel where every lane
// is a searcher -- was built on the same lists and measured twice: it loses 10-40 %; removed.)
constexpr uint32_t kRetryDeadBit = 0x40000000u;   // lutMiss: bit 0 LUT miss, bits 1.. the ray's TIR tally, bit 30 retry-dead

// Per-ray constants of camera_create_ray (zoic.cpp:1853-1855, 1891-1911): the sensor point, the exit-pupil LUT's scale and
// translation, the (parabola) sine / cosine of the pupil rotation -- shared by the refill of the pass loop and by the
// finish kernel.  flags: bit 0 = outside the LUT (fenced UB), kRetryDeadBit = no retry of this ray can reach the rear element.
struct RaySetup { float o0x, o0y, maxScale, translation, sn, cs; uint32_t flags; bool dead, lutEdge; };
template <bool STRICT>
__device__ __forceinline__ RaySetup setup_ray(const KolbTable &T, const float2 *lutLds, float sx, float sy)
{
    RaySetup r;
    r.o0x = sx * T.halfSensor;  // zoic.cpp:1853-1854
    r.o0y = sy * T.halfSensor;
    r.maxScale = 0.0f; r.translation = 0.0f; r.sn = 0.0f; r.cs = 1.0f; r.flags = 0u; r.dead = false; r.lutEdge = false;
    if (T.useLUT) {            // zoic.cpp:1891-1911: per-sample constants of the exit-pupil transform
        float dist;
        if constexpr (STRICT) dist = fabsf(ZOIC_SQRT_RN(r.o0x * r.o0x + r.o0y * r.o0y));
        else dist = fsqrt_fast(r.o0x * r.o0x + r.o0y * r.o0y);
        r.flags = lut_lookup_lds(lutLds, T.lutSize, dist, r.maxScale, r.translation) ? 0u : 1u;
        // the only discontinuity of the lookup is the table's end (bin edges interpolate continuously)
        r.lutEdge = fabsf(dist * 8.0f - static_cast<float>(T.lutSize - 1)) < T.bandLutBin;
        // Outside the image circle the LUT entries are all zero (zoic.cpp:1403-1404 never grown): every try
        // then shoots lens = (0,0).  With o0x != 0 and o0y != 0 the direction (0 - o0x, 0 - o0y, dirZ) is
        // bit-identical for all 27 tries whatever the signs of the zeros, so one failed trace decides them all.
        r.dead = (r.maxScale == 0.0f) && (r.translation == 0.0f) && (r.o0x != 0.0f) && (r.o0y != 0.0f);
        if (!r.dead) {           // the rotation of (0,0) needs no angle: dead pixels skip atan2 + sin + cos
            if constexpr (STRICT) {
                const float theta = static_cast<float>(atan2(static_cast<double>(r.o0y), static_cast<double>(r.o0x)));
                r.sn = fast_sin(theta);
                r.cs = fast_cos(theta);
            } else {
                const float theta = atan2f(r.o0y, r.o0x);
                r.sn = fast_sin_f32(theta);
                r.cs = fast_cos_f32(theta);
            }
            if (T.retryOn) {
                // retry-dead test (tables.hpp): can ANY retry of this ray reach the rear element?  The retries sample
                // the disk of radius maxScale * |lens sample|max around the LUT centroid translated in BOTH components
                // and rotated by the ray's (parabola) cos/sin; 1 % + 1e-4 of margin dwarfs every rounding involved.
                const float k = T.useImage ? 1.4158f : 1.0023f;   // |lens sample| <= sqrt(2) (image) / 1.0011 (disk), x the rotation's 1.0011
                const float ccx = r.translation * (r.cs - r.sn) - r.o0x * T.retryK1, ccy = r.translation * (r.sn + r.cs) - r.o0y * T.retryK1;
                const float reach = (T.retryRho0 + dist * T.retrySpread + fabsf(r.maxScale) * k) * 1.01f + 1.0e-4f;
                // |d.xy| of any retry <= |rotated, translated lens point| + |o.xy|: below retryMaxD the opposite cap is out of reach
                const float dxyMax = fabsf(r.maxScale) * k + fabsf(r.translation) * 1.4158f + dist;
                if (ccx * ccx + ccy * ccy > reach * reach && dxyMax <= T.retryMaxD) r.flags |= kRetryDeadBit;
            }
        }
    }
    return r;
}

// direction of a RETRY's lens sample (zoic.cpp:1932-1943 with the LUT, 1882-1884 without): shared by the pass loop and the finish kernel
__device__ __forceinline__ V3 retry_direction(const KolbTable &T, V2 lens, float o0x, float o0y, float maxScale, float translation, float sn, float cs)
{
    if (!T.useLUT) return V3{(lens.x * T.rearAperture) - o0x, (lens.y * T.rearAperture) - o0y, T.dirZ};
    lens.x *= maxScale; lens.y *= maxScale;
    lens.x += translation;
    lens.y += translation;              // retries translate BOTH components (zoic.cpp:1933)
    const float rx = lens.x * cs - lens.y * sn, ry = lens.x * sn + lens.y * cs;
    return V3{rx - o0x, ry - o0y, T.dirZ};
}

// finish_dead_ray: completes a retry-dead ray whose first try failed (tables.hpp KolbTable::retry*).  All 26 retries of such
// a ray die at interface 0 -- they bump no counter and leave (o, d) untouched -- so the ray ends with weight 0, 26 tries and
// the untouched state of its LAST retry (zoic.cpp:1951-1961): step the ray's retry stream over 25 draws, evaluate the lens
// sample of the 26th, write the record.  ~800 instructions at full lane utilisation against 26 x ~95 in the draw loop at a
// third of the lanes.  A draw of exactly (0.5, 0.5) makes the concentric-disk sample NaN (zoic.cpp:697-699), and a NaN ray
// PASSES every comparison of the reference's trace: such a ray (probability 2e-15 per draw) is a success with NaN origin /
// direction at that try -- reproduced here; returns true in that case (the caller counts it as a success).
template <bool STRICT>
__device__ __forceinline__ bool finish_dead_ray(const KolbTable &T, const BokehTables &B, const float2 *lutLds, const float *bokehLds,
                                                const float4 *__restrict__ samples, const uint4 *__restrict__ states, uint64_t rayBase,
                                                RayRecord *__restrict__ out, uint32_t idx)
{
    const float4 s = samples[idx];
    const RaySetup rs = setup_ray<STRICT>(T, lutLds, s.x, s.y);
    Rng rng;
    if (states) { const uint4 r = states[idx]; rng = Rng{r.x, r.y, r.z, r.w}; }
    else rng = rng_for_ray(T.seed, rayBase + idx);
    // retries 1 ... 26 (zoic.cpp:1927-1947), branch-free and unrolled: one dependent chain of 52 xorshift steps that the
    // scheduler interleaves with the (independent) set-up arithmetic above; the first draw at the disk's centre, if any, is
    // remembered instead of leaving the loop (finish kernel 126 -> 113 us on C2; two rays per lane, to run two chains side
    // by side, cost two waves of occupancy and measured 122)
    uint32_t tries = static_cast<uint32_t>(kMaxTries) + 1u, a = 0, b = 0, hitTry = 0;
#pragma unroll
    for (uint32_t k = 1; k <= static_cast<uint32_t>(kMaxTries) + 1u; ++k) {
        a = xor128(rng); b = xor128(rng);
        // rng_unit(x) == 0.5f  <=>  x in [0x7fffffc0, 0x80000080]
        const bool centre = ((a - 0x7fffffc0u) <= 0xc0u) & ((b - 0x7fffffc0u) <= 0xc0u);
        hitTry = (centre && hitTry == 0u) ? k : hitTry;
    }
    const bool nanDraw = !T.useImage && hitTry != 0u;
    if (nanDraw) tries = hitTry;
    const float qnan = __builtin_bit_cast(float, 0x7fc00000u);
    float w = nanDraw ? 1.0f : 0.0f;
    if (T.exposureOn) w *= T.exposureMul;                                            // zoic.cpp:1981-1987
    V3 o{rs.o0x, rs.o0y, T.originShift}, d{qnan, qnan, qnan};
    if (nanDraw) o = V3{qnan, qnan, qnan};
    else d = retry_direction(T, lens_sample<STRICT>(T, B, bokehLds, rng_unit(a), rng_unit(b)), rs.o0x, rs.o0y, rs.maxScale, rs.translation, rs.sn, rs.cs);
    store_ray_record(out, idx, o.x * -1.0f, o.y * -1.0f, o.z * -1.0f, d.x * -1.0f, d.y * -1.0f, d.z * -1.0f, w,   // zoic.cpp:1960-1961
                     1u | (tries << 1) | ((rs.flags & 1u) << 6));
    return nanDraw;
}

template <bool STRICT, int NS, bool GUARD, bool LISTED, bool DEAD>
__device__ __forceinline__ void kolb_refill_body(const KolbTable &T, const BokehTables &B, const float4 *__restrict__ samples,
                                                 uint32_t n, RayRecord *__restrict__ out, uint32_t ldsWords, uint32_t minSearching)
{
    static_assert(!(GUARD && STRICT) && !(LISTED && !STRICT), "GUARD is a FAST mode, LISTED the STRICT kernel behind it");
    constexpr bool DEFER = GUARD || DEAD;   // the kernel hands rays on (GUARD: unsure rays to the STRICT kernel; DEAD: retry-dead rays to the finish kernel)
    constexpr uint32_t kDropWords = DEAD ? kDeadLdsWords : kGuardLdsWords;
    uint32_t redoChunk = 0, redoChunksPerPart = 0;
    if constexpr (LISTED) {   // the work list's length is only known on the device
        n = *ZOIC_KARG(redoCount);   // <= samples of the launch, which is what the list was sized for
        if (n == 0u) return;
        // 64-entry chunks while the list is short (every wave gets work), 256 once it could feed the chip several times
        // over: a claim costs three dependent round trips (cursor, list, samples) and is amortised over the chunk
        redoChunk = n > (1u << 20) ? 256u : 64u;
        const uint32_t totalChunks = (n + redoChunk - 1u) / redoChunk;
        if (blockIdx.x * kWavesPerBlock >= totalChunks) return;   // whole workgroup: nothing listed (the usual case for most of the grid)
        redoChunksPerPart = (totalChunks + kCursorParts - 1u) / kCursorParts;
    }
    const uint32_t lane = threadIdx.x & 63u;
    // LDS, once per workgroup: the 32 exit-pupil LUT pairs (maxScale, centroid.x) -- one ds_read_b128 fetches the
    // two entries a sample interpolates -- then (ldsWords > 0) the bokeh row cell records (tables.hpp)
    if (threadIdx.x < kLutEntries) {
        zoicDynLds[2 * threadIdx.x] = T.lutMaxScale[threadIdx.x];
        zoicDynLds[2 * threadIdx.x + 1] = T.lutCentroidX[threadIdx.x];
    }
    const float *bokehLds = nullptr;
    if (ldsWords > 0) {
        for (uint32_t i = threadIdx.x; i < ldsWords; i += kRefillBlock) zoicDynLds[kLutLdsWords + i] = __builtin_bit_cast(float, B.rowCells[i]);
        bokehLds = zoicDynLds + kLutLdsWords;
    }
    __syncthreads();
    const float2 *lutLds = reinterpret_cast<const float2 *>(zoicDynLds);
    // Deferred record stores.  vmcnt is ONE in-order counter for loads and stores: a store issued at the end of a pass
    // would be waited for (write-acknowledge latency) by the first vmcnt(0) of the next pass -- the refill's window wait
    // or the lens sampler's dependent cell load.  A finished ray is therefore parked in LDS (the wave's 64 record slots
    // of 32 bytes + their ray indices) and written to HBM just before the NEXT trace, under which the stores and the
    // window prefetch fly; every vmcnt wait then only meets operations that had a whole trace to complete.
    float4 *stage = reinterpret_cast<float4 *>(zoicDynLds + kLutLdsWords + ldsWords) + (threadIdx.x >> 6) * 144u;  // 128 pieces + 64 indices
    uint32_t *stageIdx = reinterpret_cast<uint32_t *>(stage + 128);
    bool parked = false;   // wave-uniform: records of the last pass wait in LDS
    const bool memoryPhasesFirst = T.useImage != 0;
    // wave-uniform work window [next, end): a chunk of chunkRays consecutive samples claimed from a partition cursor (work_cursor.hpp)
    uint32_t next = 0, end = 0;
    bool exhausted = false;
    uint32_t part = blockIdx.x % kCursorParts, partsTried = 0;   // the partition cursor this wave claims from (kernels.hpp)
    // sample prefetch window: lane l holds samples[winBase + l], loaded one pass ahead of its use so the HBM latency
    // hides under the trace; refilled lanes fetch their sample from lane `rank` with ds_bpermute (winBase == next)
    float4 win = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t winBase = 0xffffffffu;
    uint32_t winIdx = 0;   // LISTED: the ray index of the window's sample (read through the work list)
    // DEFER: the wave's LDS list of dropped rays (128 entries, tagged) + 64 per-lane TIR tallies of the rays it finished
    uint32_t *dropLds = reinterpret_cast<uint32_t *>(zoicDynLds + kLutLdsWords + ldsWords) + kWavesPerBlock * 576u + (threadIdx.x >> 6) * kDropWords;
    uint32_t *tirLds = dropLds + 128;
    uint32_t *finLds = dropLds + 192;   // DEAD: retry-dead rays whose first try failed, on their way to the byte map
    uint32_t dropCnt = 0, finCnt = 0;   // wave-uniform: entries staged in dropLds / finLds
    if constexpr (DEFER) tirLds[lane] = 0u;
    const auto fetch_window = [&](uint32_t base) {
        const uint32_t wi = base + lane;
        if constexpr (LISTED) { winIdx = ZOIC_KARG(redoList)[wi < n ? wi : n - 1]; win = samples[winIdx]; }
        else win = samples[wi < n ? wi : n - 1];
        winBase = base;
    };

    // per-lane ray state, alive across passes
    bool active = false, fresh = false, dead = false;
    uint32_t idx = 0, tries = 0, lutMiss = 0;
    float o0x = 0, o0y = 0, maxScale = 0, translation = 0, sn = 0, cs = 1, u = 0, v = 0;
    Rng rng{1, 2, 3, 4};
    uint32_t succ = 0, vign = 0, tir = 0;   // wave totals, wave-uniform (SGPRs: ballot + popcount, no per-lane counters)
    ZOIC_RT_DECL
    ZOIC_PS_DECL

    bool done = false;
#ifdef ZOIC_EXP_TAIL_CUT
    uint32_t tailPasses = 0;
#endif
    do {
    for (;;) {
        ZOIC_RT_MARK(4)
        // Wave priority (s_setprio; measured, same box): with the bokeh image on, waves wait half their cycles on the sampler's
        // LDS -> global chain, and letting the waves that are in their memory phases issue first gets those loads out
        // earlier (C3 +2 %); without it the launch is compute-dense and the waves inside the trace go first (C4 +3 %).
        if (memoryPhasesFirst) __builtin_amdgcn_s_setprio(1);
        bool unsure = false;   // GUARD: a decision of this pass lies inside its guard band -> the ray goes to the STRICT kernel
        FastSurfaceTable fsurf = nullptr;
        if constexpr (GUARD && (ZOIC_GUARD_PIN != 0)) fsurf = launder_table(kernarg_fast_surfaces());   // keeps the table's s_loads at their use (fast_optics.hpp)
        else if constexpr (!STRICT) fsurf = kernarg_fast_surfaces();
        (void)fsurf;
        // ---- refill the free lanes from the work window (ballot + prefix sum) ---------------------------------
        unsigned long long freeMask = __ballot(!active);
        while (freeMask != 0ull && !exhausted) {
            if (next >= end) {  // claim the next chunk: one atomic per chunkRays samples per wave (work_cursor.hpp)
                const uint32_t cr = LISTED ? redoChunk : ZOIC_KARG(chunkRays), cpp = LISTED ? redoChunksPerPart : ZOIC_KARG(chunksPerPart);
                if (!claim_chunk(ZOIC_KARG(workCursor), lane, part, partsTried, cr, cpp, n, next, end)) { exhausted = true; ZOIC_WL_EXH break; }
            }
            if (winBase != next) fetch_window(next);  // first use of a chunk: the window has to be fetched in line (once per chunk)
            const uint32_t avail = end - next;
            const uint32_t nfree = static_cast<uint32_t>(__popcll(freeMask));
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(freeMask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(freeMask), 0u));
            // my sample sits in lane `rank` of the window (all lanes take part in the permute)
            const float4 s = make_float4(__shfl(win.x, rank, 64), __shfl(win.y, rank, 64), __shfl(win.z, rank, 64),
                                         __shfl(win.w, rank, 64));  // (sx, sy, lensx, lensy)
            uint32_t listedIdx = 0;
            if constexpr (LISTED) listedIdx = __shfl(winIdx, rank, 64);
            if (!active && rank < avail) {
                idx = LISTED ? listedIdx : next + rank;   // the retry stream is seeded lazily, at the ray's first retry (most rays never need it)
                const RaySetup rs = setup_ray<STRICT>(T, lutLds, s.x, s.y);
                o0x = rs.o0x; o0y = rs.o0y; maxScale = rs.maxScale; translation = rs.translation; sn = rs.sn; cs = rs.cs;
                lutMiss = rs.flags; dead = rs.dead;
                if constexpr (GUARD) unsure = T.useLUT && rs.lutEdge;
                u = s.z; v = s.w;
                tries = 0;
                active = true; fresh = true;
            }
            next += (nfree < avail) ? nfree : avail;
            if (nfree <= avail) break;
            freeMask = __ballot(!active);
        }
        if (__ballot(active) == 0ull) { done = true; break; }
#ifdef ZOIC_EXP_TAIL_CUT   // experiment (WRONG results): how much of a launch is the tail of rays that keep failing inside the lens?
        if (exhausted && ++tailPasses > ZOIC_EXP_TAIL_CUT) { done = true; break; }
#endif
        ZOIC_RT_MARK(0)
        ZOIC_PS_ADD(0, 1) ZOIC_PS_ADD(1, __popcll(__ballot(active)))
#ifdef ZOIC_REGION_TIMERS
        ++wlPasses;
#endif

        // ---- one try for every active lane ---------------------------------------------------------------------
        // ---- candidate search: draw lens samples until one clears the rear element's housing -----------------------
        // Most rejected tries die at interface 0 (rear-element housing / first sphere miss): 94 % of TESSAR retries, 91 %
        // of wide-open PETZVAL retries, half of DOUBLE_GAUSS retries.  Testing interface 0 alone costs ~70 lane-
        // instructions against ~900 for a whole try, so a lane keeps drawing (tries and RNG draws advance exactly as in
        // the reference's loop, zoic.cpp:1927-1947) until its sample survives interface 0 or it runs out of tries; the
        // full trace then runs once for the survivors.  The search loop is wave-uniform: it goes on while at least
        // kMinSearching lanes are still looking, the rest simply carry their search into the next pass.
        V3 o{o0x, o0y, T.originShift}, d{0.0f, 0.0f, 1.0f};
        bool cand = false, finiteSample = true;
        bool searching = GUARD ? (active && !unsure) : active;
        bool toFinish = false;   // a retry-dead ray whose first try has failed: its 26 retries all die at interface 0 -> finish kernel
        for (;;) {
            if (searching) {
                const bool first = fresh;
                if (!first) {                       // retry: new lens sample from the ray's own stream, zoic.cpp:1930
                    if (tries == 0) {               // first retry of this ray: seed its private xorshift128 stream
                        const uint4 *states = ZOIC_KARG(rngStates);
                        if (states) { const uint4 r = states[idx]; rng = Rng{r.x, r.y, r.z, r.w}; }
                        else rng = rng_for_ray(kernarg_field<uint32_t, offsetof(RefillArgs, T) + offsetof(KolbTable, seed)>(), ZOIC_KARG(rayBase) + idx);
                    }
                    u = rng_unit(xor128(rng));
                    v = rng_unit(xor128(rng));
                    ++tries;
                }
                fresh = false;
                // A dead pixel's first sample is multiplied by maxScale = 0 and offset by translation = 0: whatever finite
                // point the sampler returns, the direction is (0 - o.x, 0 - o.y, dirZ) (o.x, o.y != 0).  Samples in [0,1)^2
                // always give a finite point -- except the disk mapping's 0/0 at its centre -- so those skip the sampler.
                const bool plainSample = (u >= 0.0f) & (u < 1.0f) & (v >= 0.0f) & (v < 1.0f) & !((u == 0.5f) & (v == 0.5f));
                const bool skipSampler = first && dead && plainSample;
                V2 lens{0.0f, 0.0f};
                if (!skipSampler) lens = lens_sample<STRICT>(T, B, bokehLds, u, v);
#ifdef ZOIC_EXP_DOUBLE_SAMPLE   // marginal-cost experiments (tools/ab_libs.sh): run a stage twice, time the difference
                { const V2 l2 = lens_sample<STRICT>(T, B, bokehLds, u + lens.x * 0.0f, v + lens.y * 0.0f); lens.x += l2.x * 0.0f; lens.y += l2.y * 0.0f; }
#endif
                // the dead-pixel shortcut needs a finite first sample (NaN*0 would differ from later tries)
                finiteSample = (fabsf(lens.x) <= 3.0e38f) && (fabsf(lens.y) <= 3.0e38f);
                if (!T.useLUT) {                    // zoic.cpp:1873-1877 / 1882-1884
                    d = V3{(lens.x * T.rearAperture) - o.x, (lens.y * T.rearAperture) - o.y, T.dirZ};
                } else {                            // zoic.cpp:1913-1924 / 1932-1943
                    lens.x *= maxScale; lens.y *= maxScale;
                    lens.x += translation;
                    if (!first) lens.y += translation;  // retries translate BOTH components (zoic.cpp:1933)
                    const float rx = lens.x * cs - lens.y * sn, ry = lens.x * sn + lens.y * cs;
                    d = V3{rx - o.x, ry - o.y, T.dirZ};
                }
                bool pass0, near0 = false;
                if constexpr (STRICT) {
                    bool inRange;
                    pass0 = interface0_clear_strict_lean(T, o, d, inRange);
                    if (__builtin_expect(!inRange, 0)) pass0 = interface0_clear_strict(T, o, d);   // never seen: guarded roots
                }
                else if constexpr (GUARD) pass0 = interface0_clear_fast_guard(load_surface<false>(fsurf, 0), o, d, near0);   // band 0 unless the rear interface is the stop
                else pass0 = interface0_clear_fast(load_surface<false>(fsurf, 0), o, d);
#ifdef ZOIC_EXP_DOUBLE_PRETEST
                if constexpr (!STRICT) { V3 d2 = d; d2.x += pass0 ? 0.0f : 1.0e-30f; pass0 = pass0 & interface0_clear_fast(load_surface(fsurf, 0), o, d2); }
#endif
                if (GUARD && near0) { unsure = true; searching = false; }   // too close to call: no decision is taken here
                else if (pass0) { cand = true; searching = false; }
                else {
                    // a clip at interface 0 bumps no TIR counter and leaves (o, d) untouched: with the dead-pixel
                    // shortcut all 27 tries are this one
                    if (first && dead && finiteSample) tries = static_cast<uint32_t>(kMaxTries) + 1u;
                    if (tries > static_cast<uint32_t>(kMaxTries)) searching = false;   // out of tries at interface 0
                    else if (DEAD && first && (lutMiss & kRetryDeadBit) != 0u) { toFinish = true; searching = false; }   // no retry can succeed
                }
            }
            // the loop spins while enough lanes are looking to be worth the others' wait; once the wave can no longer be
            // refilled nobody waits for anything else, and it spins while ANY lane is looking
            const uint32_t looking = static_cast<uint32_t>(__popcll(__ballot(searching)));
            ZOIC_PS_ADD(2, 1) ZOIC_PS_ADD(3, looking)
            if (looking < (exhausted ? 1u : minSearching)) break;
        }

        ZOIC_RT_MARK(1)
        // ---- re-base the sample window on the new cursor; consumed by the NEXT pass's refill ---------------------------
        // Issued here, not in the refill: vmcnt is one in-order counter, so the lens sampler's dependent global load
        // (cell record) waits for every older memory operation -- a window load issued before the search would be waited
        // for, at full HBM latency, inside the search instead of flying under the trace.
        if (next < end && winBase != next) fetch_window(next);
        if (parked) { flush_parked_records(out, stage, stageIdx, lane); parked = false; }
        // ---- one full trace for every lane that holds a candidate -----------------------------------------------------
        bool ok = false;
        const V3 oStart = o, dStart = d;
        const bool firstTry = tries == 0;
        if (__ballot(cand) != 0ull) {
            ZOIC_PS_ADD(4, 1) ZOIC_PS_ADD(5, __popcll(__ballot(cand)))
            if (memoryPhasesFirst) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
            uint32_t tirTry = 0;   // 0/1: this try ended in total internal reflection
            if constexpr (NS > 0) {
                if constexpr (STRICT) {
                    bool oor = false;
                    ok = trace_lens_strict_pred<NS>(T, o, d, tirTry, cand, oor);
                    if (__builtin_expect(__ballot(cand && oor) != 0ull, 0)) {   // never seen: a root left the lean sequences' verified range
                        if (cand && oor) { o = oStart; d = dStart; tirTry = 0; ok = trace_lens_strict(T, o, d, tirTry); }
                    }
                }
                else if constexpr (GUARD) { bool u2 = false; ok = trace_lens_fast_pred<NS, true>(fsurf, o, d, tirTry, cand, &u2); unsure |= cand && u2; }
                else ok = trace_lens_fast_pred<NS>(fsurf, o, d, tirTry, cand);
#ifdef ZOIC_EXP_DOUBLE_TRACE
                if constexpr (!STRICT) { V3 o2 = oStart, d2 = dStart; uint32_t t2 = 0; o2.x += o.x * 0.0f; const bool ok2 = trace_lens_fast_pred<NS>(fsurf, o2, d2, t2, cand); o.x += o2.x * 0.0f; ok = ok & (ok2 | !ok); }
#endif
            } else if (cand) {
                if constexpr (STRICT) ok = trace_lens_strict(T, o, d, tirTry);
                else if constexpr (GUARD) { bool u2 = false; ok = trace_lens_fast_rolled(T, o, d, tirTry, &u2); unsure |= u2; }
                else ok = trace_lens_fast_rolled(T, o, d, tirTry);
            }
            const bool shortcut = cand && !ok && firstTry && dead && finiteSample && !(GUARD && unsure);
            // the shortcut stands for 26 more identical failures: account for their TIR bumps as well
            if constexpr (DEFER) {
                // a dropped ray must leave no trace in the counters (the kernel that picks it up counts it): TIR bumps are
                // tallied per ray, above bit 0 of lutMiss, and reach the wave total only when the ray finishes here
                if (!unsure) lutMiss += (tirTry << 1) + (shortcut ? (tirTry * (static_cast<uint32_t>(kMaxTries) + 1u)) << 1 : 0u);
            } else {
                tir += static_cast<uint32_t>(__popcll(__ballot(tirTry != 0u))) +
                       (static_cast<uint32_t>(kMaxTries) + 1u) * static_cast<uint32_t>(__popcll(__ballot(shortcut && tirTry != 0u)));
            }
            if (shortcut) tries = static_cast<uint32_t>(kMaxTries) + 1u;   // ... then finish the ray as the reference would
            else if (DEAD && cand && !ok && firstTry && (lutMiss & kRetryDeadBit) != 0u) toFinish = true;   // first try failed inside the lens: same
            if constexpr (NS > 0) {
                // the predicated trace does not keep the partial state of a failed ray; a ray that FINISHES failed
                // (out of tries) gets it from the branchy trace, which stops at the failing interface
                if (cand && !ok && tries > static_cast<uint32_t>(kMaxTries) && !(GUARD && unsure)) {
                    uint32_t ignored = 0;
                    o = oStart; d = dStart;
                    if constexpr (STRICT) (void)trace_lens_strict(T, o, d, ignored);
                    else (void)trace_lens_fast_rolled(T, o, d, ignored);
                }
            }
        }
        if (!memoryPhasesFirst) __builtin_amdgcn_s_setprio(0);
        // a ray is finished when a try got through, or when it is out of tries (loop exit of zoic.cpp:1927); a lane that
        // ran out at interface 0 must hand out the untouched (o, d) of its last sample -- the reference's partial state
        // (the predicated trace scribbles over the registers of lanes that ride along)
        ZOIC_RT_MARK(2)
        if (!cand) { o = oStart; d = dStart; }
        uint32_t finishedIdx = 0xffffffffu;
        const bool finished = active && !searching && !toFinish && (ok || tries > static_cast<uint32_t>(kMaxTries)) && !(GUARD && unsure);
        if constexpr (DEFER) {
            // stage the dropped rays' indices, add the TIR tallies of the rays leaving the pass loop for good (LDS traffic only)
            const bool dropU = GUARD && active && unsure;                 // -> STRICT kernel, evaluated from scratch
            const bool dropF = DEAD && active && toFinish && !dropU;       // -> finish kernel: only the last retry's direction is missing
            const uint32_t tally = (finished || dropF) ? ((lutMiss & ~kRetryDeadBit) >> 1) : 0u;
            if (__ballot(dropU || dropF || tally != 0u) != 0ull) {
                if constexpr (GUARD) {
                    const unsigned long long m = __ballot(dropU);
                    const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                    if (dropU) { dropLds[dropCnt + r] = idx; active = false; }
                    dropCnt += static_cast<uint32_t>(__popcll(m));
                }
                if constexpr (DEAD) {
                    const unsigned long long mf = __ballot(dropF);
                    const uint32_t rf = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mf >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mf), 0u));
                    if (dropF) { finLds[finCnt + rf] = idx; active = false; }
                    finCnt += static_cast<uint32_t>(__popcll(mf));
                }
                if (tally != 0u) tirLds[lane] += tally;
            }
        }
        {
            const uint32_t nv = static_cast<uint32_t>(__popcll(__ballot(finished && tries > static_cast<uint32_t>(kMaxTries))));
            vign += nv;                                                                       // zoic.cpp:1951-1957
            succ += static_cast<uint32_t>(__popcll(__ballot(finished))) - nv;
        }
        if (finished) {
            float w = (tries > static_cast<uint32_t>(kMaxTries)) ? 0.0f : 1.0f;
            if (T.exposureOn) w *= T.exposureMul;                                            // zoic.cpp:1981-1987
            stage[2 * lane] = make_float4(o.x * -1.0f, o.y * -1.0f, o.z * -1.0f, d.x * -1.0f);                 // zoic.cpp:1960-1961
            stage[2 * lane + 1] = make_float4(d.y * -1.0f, d.z * -1.0f, w,
                                              __builtin_bit_cast(float, (tries > 0 ? 1u : 0u) | (tries << 1) | ((lutMiss & 1u) << 6)));
            finishedIdx = idx;
            active = false;
        }
        stageIdx[lane] = finishedIdx;
        parked = true;
        if constexpr (DEFER) { if (dropCnt > 64u || finCnt > 64u) break; }   // the LDS lists must keep room for a whole pass: flush below
    }
    if constexpr (DEFER) {
        // GUARD: move the staged indices to the STRICT kernel's work list: one atomic reserves exactly the entries written
        if constexpr (GUARD) {
            if (dropCnt != 0u) {
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(ZOIC_KARG(redoCount), dropCnt);
                at = __builtin_amdgcn_readfirstlane(at);
                uint32_t *list = ZOIC_KARG(redoList);
                for (uint32_t j = lane; j < dropCnt; j += 64u) list[at + j] = dropLds[j];
                dropCnt = 0;
            }
        }
        // DEAD: mark the staged rays in the launch's byte map (plain byte stores: no counter, no atomics -- handing them over
        // through ONE list counter ran into the L2's same-address atomic rate: 41 K flushes of a 16.6 M-ray TESSAR frame took
        // 0.5 ms)
        if constexpr (DEAD) {
            if (finCnt != 0u) {
                uint8_t *map = ZOIC_KARG(deadMap);
                for (uint32_t j = lane; j < finCnt; j += 64u) map[finLds[j]] = 1;
                finCnt = 0;
            }
        }
    }
    } while (!done);

    if (parked) flush_parked_records(out, stage, stageIdx, lane);   // records parked by the last pass
    if constexpr (DEFER) {   // TIR bumps of the rays this wave finished
        uint32_t t = tirLds[lane];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        tir += static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t)));
    }
    ZOIC_RT_FLUSH
    ZOIC_PS_FLUSH
    // ---- counters: the wave totals, one atomic per counter per wave ---------------------------------------------
    DeviceCounters *counters = counter_set(ZOIC_KARG(counters));
    if (counters) {
        if (lane == 0) {
            if (succ) atomicAdd(&counters->succes, static_cast<unsigned long long>(succ));
            if (vign) atomicAdd(&counters->vignetted, static_cast<unsigned long long>(vign));
            if (tir) atomicAdd(&counters->tir, static_cast<unsigned long long>(tir));
        }
    }
}

// kolb_finish_kernel (DEAD launches, last on the stream): completes the rays marked in the byte map.  A wave takes 1024
// consecutive map bytes (16 per lane), compacts the marked ones into its LDS queue (wave prefix sum of the lanes' counts)
// and runs finish_dead_ray for 64 of them at a time: full lanes whatever the frame's layout, nothing but a 1 byte/ray read
// for the regions without marked rays.
template <bool STRICT>
__global__ __launch_bounds__(kRefillBlock) void kolb_finish_kernel(const KolbTable T, const BokehTables B, const float4 *__restrict__ samples,
                                                                   const uint4 *__restrict__ rngStates, uint64_t rayBase, uint32_t n,
                                                                   RayRecord *__restrict__ out, DeviceCounters *counters, uint32_t ldsWords,
                                                                   const uint8_t *__restrict__ deadMap)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < kLutEntries) {
        zoicDynLds[2 * threadIdx.x] = T.lutMaxScale[threadIdx.x];
        zoicDynLds[2 * threadIdx.x + 1] = T.lutCentroidX[threadIdx.x];
    }
    const float *bokehLds = nullptr;
    if (ldsWords > 0) {
        for (uint32_t i = threadIdx.x; i < ldsWords; i += kRefillBlock) zoicDynLds[kLutLdsWords + i] = __builtin_bit_cast(float, B.rowCells[i]);
        bokehLds = zoicDynLds + kLutLdsWords;
    }
    __syncthreads();
    const float2 *lutLds = reinterpret_cast<const float2 *>(zoicDynLds);
    uint16_t *queue = reinterpret_cast<uint16_t *>(zoicDynLds + kLutLdsWords + ldsWords) + wave * kFinishBlock;   // the wave's own
    const uint32_t blocks = (n + kFinishBlock - 1u) / kFinishBlock;
    uint32_t succ = 0, vign = 0;
    for (uint32_t blk = blockIdx.x * kWavesPerBlock + wave; blk < blocks; blk += gridDim.x * kWavesPerBlock) {
        const uint4 m = reinterpret_cast<const uint4 *>(deadMap)[blk * 64u + lane];   // the map is padded to whole blocks
        if (__ballot((m.x | m.y | m.z | m.w) != 0u) == 0ull) continue;
        const uint32_t words[4] = {m.x, m.y, m.z, m.w};
        const uint32_t cnt = static_cast<uint32_t>(__popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w));   // bytes are 0 or 1
        uint32_t incl = cnt;
        for (uint32_t off = 1; off < 64u; off <<= 1) { const uint32_t t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        const uint32_t total = __shfl(incl, 63, 64);
        uint32_t at = incl - cnt;
#pragma unroll
        for (uint32_t b = 0; b < 16u; ++b)
            if (((words[b >> 2] >> (8u * (b & 3u))) & 0xffu) != 0u) queue[at++] = static_cast<uint16_t>(lane * 16u + b);
        __builtin_amdgcn_wave_barrier();   // LDS is in order per wave; this only pins the compiler's schedule
        for (uint32_t base = 0; base < total; base += 64u) {
            const bool mine = base + lane < total;
            bool nanDraw = false;
            if (mine) nanDraw = finish_dead_ray<STRICT>(T, B, lutLds, bokehLds, samples, rngStates, rayBase, out, blk * kFinishBlock + queue[base + lane]);
            const uint32_t ns = static_cast<uint32_t>(__popcll(__ballot(mine && nanDraw)));
            succ += ns;
            vign += static_cast<uint32_t>(__popcll(__ballot(mine))) - ns;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (counters && lane == 0) {
        DeviceCounters *cs = counter_set(counters);
        if (succ) atomicAdd(&cs->succes, static_cast<unsigned long long>(succ));
        if (vign) atomicAdd(&cs->vignetted, static_cast<unsigned long long>(vign));
    }
}

// the precisions / roles are separate kernels so that each can carry its own register-budget attributes
#define ZOIC_REFILL_PARAMS const KolbTable T, const BokehTables B, const float4 *__restrict__ samples, const uint4 *__restrict__ rngStates, \
        uint64_t rayBase, uint32_t n, RayRecord *__restrict__ out, DeviceCounters *counters, unsigned int *__restrict__ workCursor,          \
        uint32_t ldsWords, uint32_t chunkRays, uint32_t chunksPerPart, uint32_t minSearching, uint32_t *__restrict__ redoList,               \
        unsigned int *__restrict__ redoCount, uint8_t *__restrict__ deadMap
#define ZOIC_REFILL_ARGS T, B, samples, n, out, ldsWords, minSearching
#define ZOIC_REFILL_KERNEL(NAME_, ATTR_, STRICT_, GUARD_, LISTED_)                                                             \
    template <int NS, bool DEAD>                                                                                             \
    __global__ __launch_bounds__(kRefillBlock) ATTR_ void NAME_(ZOIC_REFILL_PARAMS)                                           \
    {                                                                                                                        \
        kolb_refill_body<STRICT_, NS, GUARD_, LISTED_, DEAD>(ZOIC_REFILL_ARGS);                                               \
    }
ZOIC_REFILL_KERNEL(kolb_refill_strict_kernel, ZOIC_REFILL_ATTR_STRICT, true, false, false)          // STRICT, whole batch
ZOIC_REFILL_KERNEL(kolb_refill_strict_listed_kernel, ZOIC_REFILL_ATTR_STRICT, true, false, true)    // STRICT over the work list of the GUARD kernel
ZOIC_REFILL_KERNEL(kolb_refill_fast_kernel, ZOIC_REFILL_ATTR_FAST, false, false, false)             // FAST unchecked (round 1's fast mode)
ZOIC_REFILL_KERNEL(kolb_refill_guard_kernel, ZOIC_REFILL_ATTR_FAST, false, true, false)             // FAST decision-safe
#undef ZOIC_REFILL_KERNEL
#undef ZOIC_REFILL_PARAMS
#undef ZOIC_REFILL_ARGS

// mode: 0 = STRICT, 1 = FAST decision-safe, 2 = FAST unchecked.  d_scratch: kolb_scratch_dwords() dwords (kernels.hpp):
// the work list of mode 1 (one dword per sample of a launch), then the byte map of a DEAD launch (padded to whole blocks)
template <bool DEAD>
int launch_kolb_refill_impl(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                            uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor,
                            int mode, uint32_t *d_scratch, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const bool noCounters = std::getenv("ZOIC_EXP_NO_COUNTERS") != nullptr;   // experiment: what do the per-wave counter atomics cost?
    if (noCounters) d_counters = nullptr;
    if ((mode == 1 || DEAD) && !d_scratch) return static_cast<int>(hipErrorInvalidValue);
    // one launch covers < 2^31 samples (32-bit ray offsets inside the kernel); larger batches are split
    constexpr uint64_t kMaxPerLaunch = 1ull << 31;
    const uint64_t perLaunch = n < kMaxPerLaunch ? n : kMaxPerLaunch;
    uint32_t *d_redoList = d_scratch;
    uint8_t *deadMap = DEAD ? reinterpret_cast<uint8_t *>(d_scratch + (mode == 1 ? perLaunch : 0)) : nullptr;
    for (uint64_t done = 0; done < n; done += kMaxPerLaunch) {
        const uint64_t m = (n - done < kMaxPerLaunch) ? (n - done) : kMaxPerLaunch;
        hipError_t e = reset_work_cursors(d_workCursor, st);   // both cursor sets and the work list's counter
        if (e != hipSuccess) return static_cast<int>(e);
        if constexpr (DEAD) {
            e = hipMemsetAsync(deadMap, 0, (m + kFinishBlock - 1) / kFinishBlock * kFinishBlock, st);
            if (e != hipSuccess) return static_cast<int>(e);
        }
        const unsigned grid = persistent_grid(m, kWavesPerBlock);
        const WorkGrain grain = work_grain(m, mode == 0 ? 256u : 512u);
        const uint32_t chunkRays = grain.chunkRays, chunksPerPart = grain.chunksPerPart;
        RayRecord *o = out + done;
        static const uint32_t minSearching = [] { const char *e = std::getenv("ZOIC_MIN_SEARCHING"); return e ? static_cast<uint32_t>(std::atoi(e)) : kMinSearching; }();
        const float4 *sp = reinterpret_cast<const float4 *>(d_samples) + done;
        const uint4 *rp = d_rng ? reinterpret_cast<const uint4 *>(d_rng) + done : nullptr;
        // bokeh row cell records in LDS when the image is on and has them (4 KB at 256 rows, 32 KB at the 2048-row limit)
        const uint32_t ldsWords = (table.useImage && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
        // ZOIC_LDS_PAD (bytes): occupancy experiments only -- extra dynamic LDS lowers the workgroups a CU admits
        static const size_t ldsPad = [] { const char *e = std::getenv("ZOIC_LDS_PAD"); return e ? static_cast<size_t>(std::atol(e)) : size_t(0); }();
        const size_t ldsBytes = static_cast<size_t>(ldsWords + kLutLdsWords) * sizeof(float) + kWavesPerBlock * 144 * sizeof(float4) + ldsPad +
                                kWavesPerBlock * (DEAD ? kDeadLdsWords : kGuardLdsWords) * sizeof(uint32_t);
        unsigned int *redoCount = d_workCursor + kRedoCountOffset, *redoCursor = d_workCursor + kRedoCursorOffset;
#define ZOIC_LAUNCH_REFILL(KERNEL_, NS_, CURSOR_)                                                                               \
    hipLaunchKernelGGL((KERNEL_<NS_, DEAD>), dim3(grid), dim3(kRefillBlock), ldsBytes, st, table, bokeh, sp, rp, rayBase + done, \
                       static_cast<uint32_t>(m), o, d_counters, CURSOR_, ldsWords, chunkRays, chunksPerPart, minSearching, d_redoList, redoCount, deadMap)
#define ZOIC_LAUNCH_BY_COUNT(KERNEL_, CURSOR_)                                                                                  \
    switch (table.lensCount) {  /* unrolled instantiations for the interface counts of real prescriptions */                   \
    case 7: ZOIC_LAUNCH_REFILL(KERNEL_, 7, CURSOR_); break;                                                                     \
    case 8: ZOIC_LAUNCH_REFILL(KERNEL_, 8, CURSOR_); break;                                                                     \
    case 9: ZOIC_LAUNCH_REFILL(KERNEL_, 9, CURSOR_); break;                                                                     \
    case 10: ZOIC_LAUNCH_REFILL(KERNEL_, 10, CURSOR_); break;                                                                   \
    case 11: ZOIC_LAUNCH_REFILL(KERNEL_, 11, CURSOR_); break;                                                                   \
    case 12: ZOIC_LAUNCH_REFILL(KERNEL_, 12, CURSOR_); break;                                                                   \
    default: ZOIC_LAUNCH_REFILL(KERNEL_, 0, CURSOR_); break;                                                                    \
    }
        if (mode == 0) { ZOIC_LAUNCH_BY_COUNT(kolb_refill_strict_kernel, d_workCursor) }
        else if (mode == 2) { ZOIC_LAUNCH_BY_COUNT(kolb_refill_fast_kernel, d_workCursor) }
        else {
            ZOIC_LAUNCH_BY_COUNT(kolb_refill_guard_kernel, d_workCursor)
            e = hipGetLastError();
            if (e != hipSuccess) return static_cast<int>(e);
            // the rays it listed, in the reference's arithmetic; workgroups beyond the list's length retire at once
            ZOIC_LAUNCH_BY_COUNT(kolb_refill_strict_listed_kernel, redoCursor)
            static const bool dbg = std::getenv("ZOIC_DEBUG_LISTS") != nullptr;   // experiments: how long is the work list?
            if (dbg) {
                unsigned int r = 0;
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(&r, redoCount, sizeof(r), hipMemcpyDeviceToHost);
                std::fprintf(stderr, "[zoic] %llu rays: %u handed to the strict kernel (%.3g)\n", static_cast<unsigned long long>(m), r, double(r) / double(m));
            }
        }
#undef ZOIC_LAUNCH_BY_COUNT
#undef ZOIC_LAUNCH_REFILL
        e = hipGetLastError();
        if (e != hipSuccess) return static_cast<int>(e);
        if constexpr (DEAD) {   // last: the rays the kernels above marked
            const uint32_t blocks = static_cast<uint32_t>((m + kFinishBlock - 1) / kFinishBlock);
            static const unsigned fcap = [] { const char *e = std::getenv("ZOIC_FINISH_BLOCKS"); return e ? static_cast<unsigned>(std::atoi(e)) : 2048u; }();   // experiments
            const unsigned fgrid = (blocks + kWavesPerBlock - 1) / kWavesPerBlock < fcap ? (blocks + kWavesPerBlock - 1) / kWavesPerBlock : fcap;
            const size_t fLds = static_cast<size_t>(ldsWords + kLutLdsWords) * sizeof(float) + kWavesPerBlock * kFinishBlock * sizeof(uint16_t);
            if (mode == 0) hipLaunchKernelGGL((kolb_finish_kernel<true>), dim3(fgrid), dim3(kRefillBlock), fLds, st, table, bokeh, sp, rp, rayBase + done,
                                              static_cast<uint32_t>(m), o, d_counters, ldsWords, deadMap);
            else hipLaunchKernelGGL((kolb_finish_kernel<false>), dim3(fgrid), dim3(kRefillBlock), fLds, st, table, bokeh, sp, rp, rayBase + done,
                                    static_cast<uint32_t>(m), o, d_counters, ldsWords, deadMap);
            e = hipGetLastError();
            if (e != hipSuccess) return static_cast<int>(e);
        }
    }
    return 0;
}

#ifdef ZOIC_REGION_TIMERS
static int read_wave_log(unsigned long long *out) { return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_waveLog), sizeof(unsigned long long) * 8192 * 4)); }
static int read_region_debug(unsigned long long *acc8, bool passStats, int reset)   // adds this translation unit's copy
{
    unsigned long long v[8];
    hipError_t e = passStats ? hipMemcpyFromSymbol(v, HIP_SYMBOL(g_passStats), sizeof(v)) : hipMemcpyFromSymbol(v, HIP_SYMBOL(g_regionCycles), sizeof(v));
    if (e != hipSuccess) return static_cast<int>(e);
    for (int i = 0; i < 8; ++i) acc8[i] += v[i];
    if (reset) {
        const unsigned long long z[8] = {};
        e = passStats ? hipMemcpyToSymbol(HIP_SYMBOL(g_passStats), z, sizeof(z)) : hipMemcpyToSymbol(HIP_SYMBOL(g_regionCycles), z, sizeof(z));
    }
    return static_cast<int>(e);
}
#endif

}  // namespace zoic
