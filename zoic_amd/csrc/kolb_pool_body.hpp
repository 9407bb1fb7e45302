// kolb_pool_body.hpp -- the Kolb kernel as whole-wave BATCHES plus a per-wave ray POOL in LDS.
// (compiled twice: kolb_pool.hip for cameras without retry-dead rays, kolb_pool_dead.hip for those with)
//
// Why.  camera_create_ray retries a rejected sample up to 26 more times (zoic.cpp:1927-1947).  Round 2's kernel
// (git history: kolb_refill_body.hpp) kept a ray in its lane until it was finished and refilled the free lanes of every pass from a
// sample window: ballot + prefix sum + four ds_bpermute + the per-ray set-up (atan2, parabola sin/cos, LUT lerp) executed
// for the handful of lanes that happened to be free, finished records parked in LDS, ~60 scalars and ~20 per-lane values
// alive across the pass loop (48 SGPR spills in the headline instantiation).  59 % of the instructions of the headline
// config were not optics.  Here NO per-ray state lives in registers from one pass to the next:
//   * phase A (fresh batch): the wave takes 64 CONSECUTIVE samples -- one coalesced global_load_dwordx4 per lane,
//     requested one pass ahead -- sets all 64 rays up at full lane width, runs the first try (candidate search +
//     predicated trace), writes the finished rays straight to their records (consecutive rays, consecutive lanes) and
//     PUSHES the unfinished ones -- ballot + mbcnt compaction -- onto the wave's pool in LDS (48 bytes per ray: index,
//     sensor point, exit-pupil constants, tries/flags, retry stream);
//   * phase B (retry pass): as soon as the pool holds 64 rays (or no fresh sample is left) the wave POPS 64 of them,
//     runs one more try for each -- same search loop, same trace -- writes the finished ones, pushes the rest back.
// Every pass therefore starts with (up to) 64 live lanes, set-up runs exactly once per ray at full width, and the pass
// loop carries a dozen wave-uniform scalars and the prefetched sample.  Per-ray retry streams (keyed by the global ray
// index) make the result independent of lane / pass / wave, so STRICT stays bit-identical to the oracle.
//
// Hand-overs (same contract as round 2): GUARD (decision-safe FAST) lists the rays with a decision inside its guard band
// for the STRICT kernel that follows on the stream; DEAD instantiations collect "retry-dead" rays whose first try failed
// (tables.hpp KolbTable::retry*) in a second LDS list and complete them 64 at a time INSIDE this kernel (finish_dead_ray
// at full lane width) -- the byte map, its memset and the separate finish kernel of round 2 are gone.
//
// Order of memory operations in a pass (vmcnt is ONE in-order counter): pool pop (LDS) -> candidate search (retries: the
// bokeh sampler's dependent global load) -> the fresh batches move up (IMAGE: the column cell record of the next batch's
// first lens sample is requested -- its LDS -> global chain flies under this pass's trace -- then the batch after it) ->
// trace -> record stores -> pool push (LDS).  The next pass waits for its probe and its samples with vmcnt(2+): the two
// record stores issued after them are never waited for.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "kolb_device.hpp"
#include "work_cursor.hpp"

#pragma STDC FP_CONTRACT OFF

#ifndef ZOIC_EXP_WHATIF
#define ZOIC_EXP_WHATIF 0   // TIMING-ONLY experiments (results are WRONG, never shipped; profiles/ab_r06/whatif.log; 5 / 6: the record stores, profiles/ab_r06/ab_store_whatif.log): 1 = the IMAGE kernels' retries sample the
                            // disk instead of the image (no retry gathers), 2 = their first try does (no probe gather), 3 = no listed kernel is launched,
                            // 4 = finish_dead_ray does not re-read its sample
#endif

#ifndef ZOIC_TWO_LEVEL_SEARCH
#define ZOIC_TWO_LEVEL_SEARCH 3   // draws per lane per round of the TWO-LEVEL retry search (DEAD kernels, disk sampler, cameras with KolbTable::twoLevel): a draw is
                                  // only evaluated (disk mapping, direction, interface 0) when a per-ray bound cannot reject it.  0 compiles it out.
                                  // [MI355X, profiles/ab_r06/ab_two_level.log] C2 decision-safe +3.9 % (3 draws; 2 draws +1.3 %), STRICT +2.4 %; C5 -2.5 % when forced on
                                  // (2.8 % of its draws are rejectable) -- hence per camera.
#endif
constexpr int kTwoLevelDraws = ZOIC_TWO_LEVEL_SEARCH;

namespace zoic {

// Kernel arguments that only rare paths read (chunk claim, first retry, work-list flush, exit) are fetched from the kernarg
// segment where they are used instead of living in SGPRs for the whole kernel: the pass loop carries ~60 scalars, the
// budget is 94, and what does not fit is spilled to VGPR lanes and paid for with a v_readlane per use inside the trace.
// KolbKernelArgs mirrors the kernels' parameter list (HIP lays kernel arguments out like a C struct; offsets checked against the
// code object's metadata, tools/isa_mix.py).
struct KolbKernelArgs {
    KolbTable T; BokehTables B; const float4 *samples; const uint4 *rngStates; uint64_t rayBase; uint32_t n; RayRecord *out;
    DeviceCounters *counters; unsigned int *workCursor; uint32_t ldsWords, chunkRays, chunksPerPart, minSearching;
    uint32_t *redoList; unsigned int *redoCount;             // GUARD kernel: appends the rays it cannot decide; LISTED kernel: reads them
    unsigned int *clearCursor;                                // the cursor block of the NEXT launch on this slot: zeroed by workgroup 0 (not LISTED)
};
template <class V, size_t OFFSET>
__device__ __forceinline__ V kernarg_field()
{
    typedef const char __attribute__((address_space(4))) *KernargBytes;
    KernargBytes base = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));   // loaded here, every time: not hoisted into a long-lived SGPR
    return *(const V __attribute__((address_space(4))) *)(base + OFFSET);
}
#define ZOIC_KARG(field) kernarg_field<decltype(KolbKernelArgs::field), offsetof(KolbKernelArgs, field)>()

// One body, four kernel families.  A decision-safe FAST launch is a pipeline of two kernels on the caller's stream:
//   GUARD (FAST only): every accept/reject decision of a try at a guarded interface (tables.hpp FastSurface::housingLo/Hi: in
//       practice the stop) is checked against its guard band; a ray with a decision too close to call is dropped where it
//       stands -- no record, no counter -- and its index goes to the work list `redoList`;
//   LISTED = kolb_listed_kernel (kolb_listed_body.hpp), next on the stream: evaluates exactly the listed rays from scratch, in the
//       reference's arithmetic wherever a decision is too close to call (per-ray retry streams make that the same ray).
// Together: every ray's try count, weight and flags are those of a STRICT evaluation unless FAST and STRICT disagree on a
// decision OUTSIDE the guarded interfaces (sphere miss, TIR, a clip at a well-conditioned housing: residual flips <= ~1e-6 of
// the rays, tests/test_parity_gpu.py); only the low-order bits of origin / direction of the FAST-evaluated rays differ.
// TIR bumps are tallied per ray (above bit 0 of lutMiss) and reach the counters only when the ray finishes in this kernel.
constexpr uint32_t kRetryDeadBit = 0x40000000u;   // lutMiss: bit 0 LUT miss, bits 1.. the ray's TIR tally, bit 30 retry-dead

// Per-ray constants of camera_create_ray (zoic.cpp:1853-1855, 1891-1911): the sensor point, the exit-pupil LUT's scale and
// translation, the (parabola) sine / cosine of the pupil rotation -- shared by phase A of the pass loop and by finish_dead_ray.  flags: bit 0 = outside the LUT (fenced UB), kRetryDeadBit = no retry of this ray can reach the rear element.
// rminq (two-level retry search): a RETRY whose unit-square draw has max(|2u-1|, |2v-1|) < rminq / 256 cannot reach the rear element (0: no bound)
struct RaySetup { float o0x, o0y, maxScale, translation, sn, cs; uint32_t flags; bool dead, lutEdge; uint32_t rminq; };
template <bool STRICT, bool RMIN = false>   // RMIN: also the per-draw reject bound of the two-level retry search (kolb_pool_two.hip's kernels)
__device__ __forceinline__ RaySetup setup_ray(const KolbTable &T, const float2 *lutLds, float sx, float sy)
{
    RaySetup r;
    r.o0x = sx * T.halfSensor;  // zoic.cpp:1853-1854
    r.o0y = sy * T.halfSensor;
    r.maxScale = 0.0f; r.translation = 0.0f; r.sn = 0.0f; r.cs = 1.0f; r.flags = 0u; r.dead = false; r.lutEdge = false; r.rminq = 0u;
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
                const float k = T.retryLensK;   // |lens sample| x the rotation's 1.0011: 1.0023 for the disk, the image's own bound otherwise (tables.hpp)
                const float ccx = r.translation * (r.cs - r.sn) - r.o0x * T.retryK1, ccy = r.translation * (r.sn + r.cs) - r.o0y * T.retryK1;
                const float reach = (T.retryRho0 + dist * T.retrySpread + fabsf(r.maxScale) * k) * 1.01f + 1.0e-4f;
                // |d.xy| of any retry <= |rotated, translated lens point| + |o.xy|: below retryMaxD the opposite cap is out of reach
                const float dxyMax = fabsf(r.maxScale) * k + fabsf(r.translation) * 1.4158f + dist;
                if (ccx * ccx + ccy * ccy > reach * reach && dxyMax <= T.retryMaxD) r.flags |= kRetryDeadBit;
                // The same bound per DRAW (the two-level retry search: cameras with KolbTable::twoLevel run kernels compiled with it).  A retry's lens point lies within |maxScale| k rho of the doubly translated centroid Q, rho =
                // max(|2u-1|, |2v-1|) (the concentric mapping's radius; k covers the parabola pair's 1.0011 twice), and can only pass
                // interface 0 from inside the disk of radius `pass` around cc: a draw with |Q - cc| - |maxScale| k rho > pass is rejected
                // whatever its angle.  rminq / 256 <= that rho bound, rounded DOWN, 1 % + 1e-4 of margin as above; disk sampler only.
                if (RMIN && dxyMax <= T.retryMaxD) {
                    const float pass = (T.retryRho0 + dist * T.retrySpread) * 1.01f + 1.0e-4f;
                    const float gap = fsqrt_fast(ccx * ccx + ccy * ccy) * 0.999f - pass;
                    const float rm = gap * frcp_fast(fabsf(r.maxScale) * k * 1.001f + 1.0e-30f);
                    r.rminq = rm > 0.0f ? static_cast<uint32_t>(fminf(rm, 1.0f) * 255.0f) : 0u;
                }
            }
        }
    }
    return r;
}

// direction of a RETRY's lens sample (zoic.cpp:1932-1943 with the LUT, 1882-1884 without): shared by the pass loop and finish_dead_ray
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
// (the arithmetic, apart from the sample fetch and the record store: shared with the resident tile workers, mailbox.hip)
struct DeadRayEnd { V3 o, d; float w; uint32_t tries; bool nanDraw; };
template <bool STRICT>
__device__ __forceinline__ DeadRayEnd dead_ray_end(const KolbTable &T, const BokehTables &B, const float *bokehLds, const RaySetup &rs, Rng rng)
{
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
    DeadRayEnd e;
    e.nanDraw = !T.useImage && hitTry != 0u;
    if (e.nanDraw) tries = hitTry;
    const float qnan = __builtin_bit_cast(float, 0x7fc00000u);
    e.w = e.nanDraw ? 1.0f : 0.0f;
    if (T.exposureOn) e.w *= T.exposureMul;                                            // zoic.cpp:1981-1987
    e.o = V3{rs.o0x, rs.o0y, T.originShift}; e.d = V3{qnan, qnan, qnan};
    if (e.nanDraw) e.o = V3{qnan, qnan, qnan};
    else e.d = retry_direction(T, lens_sample<STRICT>(T, B, bokehLds, rng_unit(a), rng_unit(b)), rs.o0x, rs.o0y, rs.maxScale, rs.translation, rs.sn, rs.cs);
    e.tries = tries;
    return e;
}

template <bool STRICT>
__device__ __forceinline__ bool finish_dead_ray(const KolbTable &T, const BokehTables &B, const float2 *lutLds, const float *bokehLds,
                                                const float4 *__restrict__ samples, const uint4 *__restrict__ states, uint64_t rayBase,
                                                RayRecord *__restrict__ out, uint32_t idx)
{
#if ZOIC_EXP_WHATIF == 4
    const float4 s = make_float4(__builtin_bit_cast(float, (idx & 0xffffu) | 0x3f000000u) - 0.75f, 0.3f, 0.0f, 0.0f);
#else
    const float4 s = samples[idx];
#endif
    const RaySetup rs = setup_ray<STRICT>(T, lutLds, s.x, s.y);
    Rng rng;
    if (states) { const uint4 r = states[idx]; rng = Rng{r.x, r.y, r.z, r.w}; }
    else rng = rng_for_ray(T.seed, rayBase + idx);
    const DeadRayEnd e = dead_ray_end<STRICT>(T, B, bokehLds, rs, rng);
    store_ray_record(out, idx, e.o.x * -1.0f, e.o.y * -1.0f, e.o.z * -1.0f, e.d.x * -1.0f, e.d.y * -1.0f, e.d.z * -1.0f, e.w,   // zoic.cpp:1960-1961
                     1u | (e.tries << 1) | ((rs.flags & 1u) << 6));
    return e.nanDraw;
}


// Debug build only (-DZOIC_PASS_STATS, tools/pass_stats.py): pass statistics summed over all waves.  Not part of the product build.
#ifdef ZOIC_PASS_STATS
static __device__ unsigned long long g_passStats[8];   // A passes, B passes, search iterations, sum looking lanes, traces, sum cand lanes, sum active lanes, finished
static __device__ unsigned long long g_regionCycles[16];   // s_memtime cycles per region of the pass loop, summed over all waves
#define ZOIC_PS_DECL unsigned long long ps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt[16] = {}, rtLast = __builtin_readcyclecounter();
#ifdef ZOIC_PS_LIST   // the work list instead: [0] rays listed, [1] sum of their tries when listed, [2] rays the LISTED kernel finished, [3] sum of their final tries
#define ZOIC_PS_ADD(I, V)
#define ZOIC_PS_LISTADD(I, V) ps[I] += (V);
#else
#define ZOIC_PS_ADD(I, V) ps[I] += (V);
#endif
#define ZOIC_MARK(N) { const unsigned long long rtNow = __builtin_readcyclecounter(); rt[N] += rtNow - rtLast; rtLast = rtNow; }
#define ZOIC_PS_FLUSH if (lane == 0) { for (int r = 0; r < 8; ++r) atomicAdd(&g_passStats[r], ps[r]); for (int r = 0; r < 16; ++r) atomicAdd(&g_regionCycles[r], rt[r]); }
#else
#define ZOIC_PS_DECL
#define ZOIC_PS_ADD(I, V)
#define ZOIC_MARK(N)
#define ZOIC_PS_FLUSH
#endif
#ifndef ZOIC_PS_LISTADD
#define ZOIC_PS_LISTADD(I, V)
#endif

// LDS per wave: the pool, 128 entries stored piece-major (arrays of 128 x 16 / 16 / 8 bytes: a push or pop is two
// ds_write_b128 / ds_read_b128 and one b64, conflict-free), then the hand-over lists (128 ray indices each).
// 128 entries: a pass starts with at most 63 pooled rays left behind and pushes at most 64.
#ifndef ZOIC_POOL_SLIM
#define ZOIC_POOL_SLIM 0   // 1: 40-byte entries: the exit-pupil scale / translation are looked up again when a ray is popped
#endif
#ifndef ZOIC_STORE_TRANSPOSED
#define ZOIC_STORE_TRANSPOSED 2   // a fresh batch's records leave through an LDS transpose (two contiguous-KB stores per wave): 0 never, 1 every kernel, 2 the FAST IMAGE kernels
                                  // [MI355X: -2 % on the fisheye (round 5: it is instructions on a pipe-bound kernel); on the image-sampler frame, whose waves wait for memory
                                  // behind the DRAM write stream, +0.3-1 % on a fast pair of buffers and +2.8 % on a slow one: profiles/ab_r06/ab_store_whatif.log]
                                  // [everywhere (=1), profiles/ab_r06/ab_transposed_configs.log: C2 -1.4 %, C4 -2.5 %, STRICT C3 -0.6 %, C5 +1.6 % (same kernel as C4's: it would need its own instantiation)]
#endif
#if ZOIC_STORE_TRANSPOSED && ZOIC_POOL_SLIM
#error "ZOIC_STORE_TRANSPOSED stages in the upper half of pool1's float4 array: not with ZOIC_POOL_SLIM"
#endif
#ifndef ZOIC_STORE_NT
#define ZOIC_STORE_NT 0   // experiments: 1 = the IMAGE kernels' records leave as non-temporal stores (they never come back; the bokeh table they evict does), 2 = every kernel's
#endif
#ifndef ZOIC_DEAD_WAVE_SAMPLER
#define ZOIC_DEAD_WAVE_SAMPLER 0   // 1: phase A samples the lens even in waves of nothing but dead pixels (rounds 3-5; A/B: profiles/ab_r06/ab_dead_wave.log)
#endif
#ifndef ZOIC_PHASE_A_TEST0
#define ZOIC_PHASE_A_TEST0 0   // 1: phase A always takes its own interface-0 test (rounds 3-5; A/B: profiles/ab_r06/ab_no_a0.log)
#endif
#ifndef ZOIC_SEARCH_DRAWS
#define ZOIC_SEARCH_DRAWS 2   // lens draws the retry search of the IMAGE kernels samples per round (one wait for all their records); measured on C3: 1 -> 38.8, 2 -> 41.1, 3 -> 40.4, 4 -> 39.6 Grays/s
#endif
constexpr uint32_t kPoolEntries = 128;
constexpr uint32_t kPoolWaveWords = kPoolEntries * (ZOIC_POOL_SLIM ? 10u : 12u);
constexpr uint32_t kPoolListWords = 128;
// packed word of a pooled ray: bit 0 outside the LUT, bits 1-6 the ray's TIR tally (DEFER kernels), bits 8-12 tries,
// bit 13 dead pixel, bit 14 retry-dead
constexpr uint32_t kPoolTriesShift = 8, kPoolDeadBit = 1u << 13, kPoolRetryDeadBit = 1u << 14;

// Register budgets: the FAST kernels are held to 80 VGPRs (6 waves per SIMD; the LDS pools admit 5-6 workgroups per CU), the
// STRICT ones (f64 intermediates) to 128 = 4 waves per SIMD.
#ifndef ZOIC_POOL_ATTR_FAST
#define ZOIC_POOL_ATTR_FAST __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
#ifndef ZOIC_POOL_ATTR_STRICT
#define ZOIC_POOL_ATTR_STRICT __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
#ifndef ZOIC_POOL_PROBE_STRICT
#define ZOIC_POOL_PROBE_STRICT 0   // the STRICT kernels spill when they also carry the probe of the next batch
#endif

__device__ __forceinline__ uint32_t mask_rank(unsigned long long m)   // exclusive prefix count of the lanes set in m
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}
__device__ __forceinline__ bool mask_bit(unsigned long long m, uint32_t lane) { return ((m >> lane) & 1ull) != 0ull; }

// work lists of kShortList rays or fewer are evaluated tries-in-parallel by the listed kernel (kolb_listed_body.hpp)
constexpr uint32_t kShortList = 1u << 17;
// IMAGE: the bokeh image is on AND its cell records are in LDS (tables.hpp; images up to 2048 rows x 4096 columns): every
// lens sample is one ds_read_b128 + one global_load_dwordx4.  IMAGE = false covers the concentric-disk sampler and images
// without records (16-ary pyramid / reference search through lens_sample's run-time branch).
// TWO: the two-level retry search (kTwoLevelDraws; kolb_pool_two.hip: DEAD kernels of the disk sampler, for cameras with KolbTable::twoLevel).
template <bool STRICT, int NS, bool GUARD, bool DEAD, bool IMAGE, bool TWO = false>
__device__ __forceinline__ void kolb_pool_body(const KolbTable &T, const BokehTables &B, const float4 *__restrict__ samples,
                                               uint32_t n, RayRecord *__restrict__ out, uint32_t ldsWords, uint32_t minSearching)
{
    static_assert(!(GUARD && STRICT), "GUARD is a FAST mode");
    static_assert(!TWO || (DEAD && !IMAGE && kTwoLevelDraws > 0), "the two-level search belongs to the DEAD kernels of the disk sampler");
    constexpr bool DEFER = GUARD || DEAD;   // rays may leave the pass loop unfinished: TIR bumps are tallied per ray
    constexpr bool PROBE = IMAGE && (!STRICT || ZOIC_POOL_PROBE_STRICT != 0);   // the next batch's first lens sample is requested a pass ahead
    constexpr uint32_t kOut = static_cast<uint32_t>(kMaxTries) + 1u;   // tries of a ray that ran out (zoic.cpp:1927: tries <= 25)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0) {   // the other cursor block of this launch slot, for the launch after this one (kernels.hpp)
        unsigned int *next = ZOIC_KARG(clearCursor);
        for (uint32_t i = threadIdx.x; i < kCursorBlockWords; i += kRefillBlock) next[i] = 0u;
    }
    // LDS, once per workgroup: the 32 exit-pupil LUT pairs (maxScale, centroid.x) -- one ds_read_b128 fetches the two entries
    // a sample interpolates -- then the bokeh row cell records (tables.hpp)
    if (threadIdx.x < kLutEntries) {
        zoicDynLds[2 * threadIdx.x] = T.lutMaxScale[threadIdx.x];
        zoicDynLds[2 * threadIdx.x + 1] = T.lutCentroidX[threadIdx.x];
    }
    const float *bokehLds = nullptr;
    if (IMAGE || ldsWords > 0) {
        for (uint32_t i = threadIdx.x; i < ldsWords; i += kRefillBlock) zoicDynLds[kLutLdsWords + i] = __builtin_bit_cast(float, B.rowCells[i]);
        bokehLds = zoicDynLds + kLutLdsWords;
    }
    __syncthreads();
    const float2 *lutLds = reinterpret_cast<const float2 *>(zoicDynLds);
    float4 *pool0 = reinterpret_cast<float4 *>(zoicDynLds + kLutLdsWords + ldsWords + wave * kPoolWaveWords);   // idx, o0x, o0y, packed
    uint4 *pool2 = reinterpret_cast<uint4 *>(pool0 + kPoolEntries);                                             // the ray's retry stream
#if ZOIC_POOL_SLIM
    float2 *pool1 = reinterpret_cast<float2 *>(pool2 + kPoolEntries);                                           // sn, cs
#else
    float4 *pool1 = reinterpret_cast<float4 *>(pool2 + kPoolEntries);                                           // maxScale, translation, sn, cs
#endif
    uint32_t *lists = reinterpret_cast<uint32_t *>(zoicDynLds + kLutLdsWords + ldsWords + kWavesPerBlock * kPoolWaveWords) +
                      wave * ((GUARD ? kPoolListWords : 0u) + (DEAD ? kPoolListWords : 0u));
    uint32_t *unsureLds = lists;                                  // GUARD: rays for the STRICT kernel
    uint32_t *deadLds = lists + (GUARD ? kPoolListWords : 0u);    // DEAD: retry-dead rays whose first try failed
    uint32_t poolCnt = 0, unsureCnt = 0, deadCnt = 0;             // wave-uniform

    const auto sample_lens = [&](float u, float v) {
        if constexpr (IMAGE) return bokeh_sample_cells<STRICT>(B, bokehLds, T.bokehW, T.bokehH, u, v);
        else return lens_sample<STRICT>(T, B, nullptr, u, v);
    };

    // fresh work: chunks of consecutive samples claimed from the partition cursors (work_cursor.hpp); [next, end) is what is
    // left of the wave's chunk.  Two batches are in flight: b1 runs next -- its samples have arrived and (PROBE) the column
    // cell record of its FIRST lens sample is already requested (zoic.cpp:1870) -- and b2, requested one pass before that.
    uint32_t next = 0, end = 0, part = blockIdx.x % kCursorParts, partsTried = 0;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    uint32_t base1 = 0, base2 = 0, cnt1 = 0, cnt2 = 0;
    bool have1 = false, have2 = false;
    CellProbe probe{make_uint4(0u, 0u, 0u, 0u), 0, 0u};
    const auto request_batch = [&]() {   // -> b2
        have2 = false;
        if (next >= end) {   // claim the next chunk: one atomic per chunkRays samples per wave
            if (!claim_chunk(ZOIC_KARG(workCursor), lane, part, partsTried, ZOIC_KARG(chunkRays), ZOIC_KARG(chunksPerPart), n, next, end)) return;
        }
        base2 = next;
        cnt2 = (end - next < 64u) ? end - next : 64u;
        const uint32_t wi = (lane < cnt2) ? next + lane : next;
        s2 = samples[wi];
        next += cnt2;
        have2 = true;
    };
    const auto advance_batches = [&]() {   // b1 <- b2 (+ its probe), b2 <- the next request
        s1 = s2; base1 = base2; cnt1 = cnt2; have1 = have2;
#if ZOIC_EXP_WHATIF != 2
        if constexpr (PROBE) { if (have1) probe = bokeh_cells_issue(B, bokehLds, T.bokehH, s1.z, s1.w); }
#endif
        if (have1) request_batch();
    };
    request_batch();
    advance_batches();

    uint32_t succ = 0, vign = 0, tir = 0;   // wave totals (SGPRs)
    uint32_t tirAcc = 0;                    // DEFER: per-lane sum of the TIR tallies of the rays this lane finished

    // Wave priority (s_setprio; measured, same box): with the bokeh image on, waves wait on the sampler's LDS -> global chain,
    // and letting the waves that are in their memory phases issue first gets those loads out earlier; without it the launch is
    // compute-dense and the waves inside the trace go first.
    constexpr bool memoryPhasesFirst = IMAGE;
    ZOIC_PS_DECL
    for (;;) {
        const bool drain = !have1;                                          // no fresh sample left for this wave
#ifdef ZOIC_PASS_STATS
        const unsigned long long drainT0 = __builtin_readcyclecounter();
#endif
        const bool fromPool = poolCnt >= 64u || (drain && poolCnt != 0u);
        if (!fromPool && drain) break;
        ZOIC_PS_ADD(fromPool ? 1 : 0, 1)
        if (memoryPhasesFirst) __builtin_amdgcn_s_setprio(1);
        FastSurfaceTable fsurf = nullptr;
        if constexpr (GUARD && (ZOIC_GUARD_PIN != 0)) fsurf = launder_table(kernarg_fast_surfaces());   // keeps the table's s_loads at their use (fast_optics.hpp)
        else if constexpr (!STRICT) fsurf = kernarg_fast_surfaces();
        (void)fsurf;

        // ---- the pass's 64 rays: a fresh batch (phase A) or 64 pooled rays (phase B) -------------------------------------
        bool active, dead, unsure = false;
        uint32_t idx, tries, lutMiss;   // lutMiss: bit 0 outside the LUT, bits 1.. the TIR tally, kRetryDeadBit
        float o0x, o0y, maxScale, translation, sn, cs;
        uint32_t rminq = 0u;   // two-level search: the ray's per-draw reject bound (setup_ray)
        (void)rminq;
        Rng rng{1, 2, 3, 4};
        V3 o, d{0.0f, 0.0f, 1.0f};
        bool cand = false, finiteSample = true, searching;
        bool toFinish = false;   // a retry-dead ray whose first try has failed -> the dead list
        // does (o, d) clear interface 0?  near0: too close to call (GUARD)
        const auto clears_rear = [&](const V3 &oo, const V3 &dd, bool &near0) {
            if constexpr (STRICT) {
                near0 = false;
                bool inRange;
                bool p = interface0_clear_strict_lean(T, oo, dd, inRange);
                if (__builtin_expect(!inRange, 0)) p = interface0_clear_strict(T, oo, dd);   // never seen: guarded roots
                return p;
            }
            else return interface0_clear_fast<GUARD>(load_surface<false>(fsurf, 0), oo, dd, near0);
        };
        ZOIC_MARK(0)   // loop top, flushes of the previous pass
        if (!fromPool) {
            // phase A: set 64 fresh rays up and run the search's FIRST step for all of them (zoic.cpp:1853-1925)
            active = lane < cnt1;
            idx = base1 + lane;
                const RaySetup rs = setup_ray<STRICT, TWO>(T, lutLds, s1.x, s1.y);
            o0x = rs.o0x; o0y = rs.o0y; maxScale = rs.maxScale; translation = rs.translation; sn = rs.sn; cs = rs.cs;
            lutMiss = rs.flags; dead = rs.dead; rminq = rs.rminq;
            if constexpr (GUARD) unsure = active && T.useLUT && rs.lutEdge;
                ZOIC_MARK(1)   // setup_ray
            tries = 0;
            o = V3{o0x, o0y, T.originShift};
            searching = GUARD ? (active && !unsure) : active;
            const float u = s1.z, v = s1.w;
            // dead pixel (outside the image circle, LUT entries zero): whatever finite point the sampler returns, the direction
            // is (0 - o.x, 0 - o.y, dirZ); samples in [0,1)^2 off the disk mapping's 0/0 centre need no sampler
            V2 lens;
            const bool anyDead = __ballot(dead) != 0ull;
            // plain: a sample in [0,1)^2 off the disk mapping's 0/0 centre -- for a dead pixel the sampler's (finite) point is never looked at
            bool plainSample = true;
            if (anyDead) plainSample = (u >= 0.0f) & (u < 1.0f) & (v >= 0.0f) & (v < 1.0f) & !((u == 0.5f) & (v == 0.5f));   // wave-uniform: most waves hold no dead pixel and skip these compares
#if ZOIC_EXP_WHATIF == 2
            if constexpr (PROBE) lens = concentric_disk_f32(u, v);
#else
            if constexpr (PROBE) lens = bokeh_cells_finish<STRICT>(B, T.bokehW, T.bokehH, v, probe);
#endif
            else {
                // a wave of nothing but dead pixels with plain samples (the corners of a frame wider than the image circle: three quarters of C5's) needs no
                // lens sample at all: ~35 instructions a ray of the ~250 such a ray costs [MI355X, profiles/ab_r06/ab_dead_wave.log]
                lens = V2{0.0f, 0.0f};
                if (ZOIC_DEAD_WAVE_SAMPLER != 0 || __ballot(active && !(dead && plainSample)) != 0ull) lens = sample_lens(u, v);
            }
            if (anyDead) {
                if (dead && plainSample) lens = V2{0.0f, 0.0f};
                finiteSample = (fabsf(lens.x) <= 3.0e38f) && (fabsf(lens.y) <= 3.0e38f);
            }
            if (!T.useLUT) {                    // zoic.cpp:1873-1877
                d = V3{(lens.x * T.rearAperture) - o.x, (lens.y * T.rearAperture) - o.y, T.dirZ};
            } else {                            // zoic.cpp:1913-1924: the first sample is translated in x only
                lens.x *= maxScale; lens.y *= maxScale;
                lens.x += translation;
                const float rx = lens.x * cs - lens.y * sn, ry = lens.x * sn + lens.y * cs;
                d = V3{rx - o.x, ry - o.y, T.dirZ};
            }
            // The first try's interface-0 test.  It is taken HERE only in waves that hold a dead pixel (whose 27 tries are decided by this one test: a wave
            // of them never enters the trace); everywhere else every fresh lane is a candidate and the predicated trace's own interface 0 -- FAST: the same
            // FastHit and guard band; STRICT: the same sequence of roundings -- takes the decision: a first try that is clipped there leaves the trace dead
            // like one clipped later and goes to the pool (or, retry-dead, to the dead list) from behind it.  Same rays; [MI355X, profiles/ab_r06/ab_no_a0.log]
            // the test was 30 (STRICT: 65) instructions a ray that the trace repeats: C4 +2.1 %, C3 +1.4 %, C2 +1.0 % decision-safe, STRICT +2.1 %; without the
            // dead-pixel exception C5 -3.5 %.
            bool near0 = false;
            bool pass0 = true;
            if (ZOIC_PHASE_A_TEST0 != 0 || anyDead) pass0 = clears_rear(o, d, near0);
            if (searching) {
                if (GUARD && near0) { unsure = true; searching = false; }   // too close to call: no decision is taken here
                else if (pass0) { cand = true; searching = false; }
                else {
                    // a clip at interface 0 bumps no TIR counter and leaves (o, d) untouched: for a dead pixel all 27 tries are this one
                    if (dead && finiteSample) { tries = kOut; searching = false; }
                    else if (DEAD && (lutMiss & kRetryDeadBit) != 0u) { toFinish = true; searching = false; }   // no retry can succeed
                }
            }
                ZOIC_MARK(2)   // first try: sampler finish, direction, interface 0
        } else {
            const uint32_t cnt = poolCnt < 64u ? poolCnt : 64u;
            poolCnt -= cnt;
            active = lane < cnt;
            const uint32_t slot = poolCnt + (active ? lane : 0u);
            const float4 e0 = pool0[slot];
            const uint4 e2 = pool2[slot];
            const uint32_t packed = __builtin_bit_cast(uint32_t, e0.w);
            idx = __builtin_bit_cast(uint32_t, e0.x); o0x = e0.y; o0y = e0.z;
#if ZOIC_POOL_SLIM
            const float2 e1 = pool1[slot];
            sn = e1.x; cs = e1.y;
            maxScale = 0.0f; translation = 0.0f;
            if (T.useLUT) {   // the same lookup as setup_ray's, on the same operands: the same two values
                float dist;
                if constexpr (STRICT) dist = fabsf(ZOIC_SQRT_RN(o0x * o0x + o0y * o0y));
                else dist = fsqrt_fast(o0x * o0x + o0y * o0y);
                (void)lut_lookup_lds(lutLds, T.lutSize, dist, maxScale, translation);
            }
#else
            const float4 e1 = pool1[slot];
            maxScale = e1.x; translation = e1.y; sn = e1.z; cs = e1.w;
#endif
            rng = Rng{e2.x, e2.y, e2.z, e2.w};
            tries = (packed >> kPoolTriesShift) & 31u;
            dead = (packed & kPoolDeadBit) != 0u;
            lutMiss = (packed & 0x7fu) | ((packed & kPoolRetryDeadBit) ? kRetryDeadBit : 0u);
            if constexpr (TWO) rminq = (packed >> 16) & 0xffu;
            o = V3{o0x, o0y, T.originShift};
            searching = active;
        }

        if (fromPool) ZOIC_MARK(3)   // pool pop
        // ---- candidate search: RETRIES draw lens samples until one clears the rear element's housing (zoic.cpp:1927-1947) ----
        // Most rejected tries die at interface 0 (94 % of TESSAR retries, 91 % of wide-open PETZVAL retries, half of DOUBLE_GAUSS
        // retries); testing it alone costs a tenth of a whole try, so a lane keeps drawing -- tries and the ray's retry stream
        // advance exactly as in the reference's loop -- until a sample survives or it runs out of tries; the full trace then runs
        // once for the survivors.  The loop is wave-uniform: it goes on while enough lanes are looking to be worth the others'
        // wait (any lane, once the wave has no fresh work left).
        for (;;) {
            const uint32_t looking = static_cast<uint32_t>(__popcll(__ballot(searching)));
            if (looking < (drain ? 1u : minSearching)) break;
            ZOIC_PS_ADD(2, 1) ZOIC_PS_ADD(3, looking)
            if (searching) {
                if (tries == 0) {               // first retry of this ray: seed its private xorshift128 stream
                    const uint4 *states = ZOIC_KARG(rngStates);
                    if (states) { const uint4 r = states[idx]; rng = Rng{r.x, r.y, r.z, r.w}; }
                    else rng = rng_for_ray(kernarg_field<uint32_t, offsetof(KolbKernelArgs, T) + offsetof(KolbTable, seed)>(), ZOIC_KARG(rayBase) + idx);
                }
                if constexpr (IMAGE && ZOIC_SEARCH_DRAWS > 1) {
                    // several draws per round: their column records are all requested before the first is read, so a round waits
                    // for memory once; a later draw counts (tries, retry stream) only when the ones before it were rejected with
                    // tries to spare -- exactly the reference's sequence
                    constexpr int kDraws = ZOIC_SEARCH_DRAWS;
                    Rng after[kDraws];
                    float vCol[kDraws];
#if ZOIC_EXP_WHATIF == 1
                    float uCol[kDraws];
#endif
                    CellProbe probes[kDraws];
                    Rng r = rng;
#pragma unroll
                    for (int j = 0; j < kDraws; ++j) {
                        const float uj = rng_unit(xor128(r));   // zoic.cpp:1930
                        vCol[j] = rng_unit(xor128(r));
                        after[j] = r;
#if ZOIC_EXP_WHATIF == 1
                        probes[j] = CellProbe{make_uint4(0u, 0u, 0u, 0u), 0, 0u}; uCol[j] = uj;
#else
                        probes[j] = bokeh_cells_issue(B, bokehLds, T.bokehH, uj, vCol[j]);
#endif
                    }
                    bool open = true;
#pragma unroll
                    for (int j = 0; j < kDraws; ++j) {
                        if (open) {
                            ++tries; rng = after[j];
#if ZOIC_EXP_WHATIF == 1
                            d = retry_direction(T, concentric_disk_f32(uCol[j], vCol[j]), o0x, o0y, maxScale, translation, sn, cs);
#else
                            d = retry_direction(T, bokeh_cells_finish<STRICT>(B, T.bokehW, T.bokehH, vCol[j], probes[j]), o0x, o0y, maxScale, translation, sn, cs);
#endif
                            bool near0;
                            const bool pass0 = clears_rear(o, d, near0);
                            if (GUARD && near0) { unsure = true; searching = false; open = false; }
                            else if (pass0) { cand = true; searching = false; open = false; }
                            else if (tries > static_cast<uint32_t>(kMaxTries)) { searching = false; open = false; }
                        }
                    }
                } else {
                if constexpr (TWO) { /* handled below, wave-wide */ } else
                {
                const float u = rng_unit(xor128(rng));   // zoic.cpp:1930
                const float v = rng_unit(xor128(rng));
                ++tries;
                d = retry_direction(T, sample_lens(u, v), o0x, o0y, maxScale, translation, sn, cs);   // zoic.cpp:1932-1943: BOTH components translated
                bool near0;
                const bool pass0 = clears_rear(o, d, near0);
                if (GUARD && near0) { unsure = true; searching = false; }
                else if (pass0) { cand = true; searching = false; }
                else if (tries > static_cast<uint32_t>(kMaxTries)) searching = false;   // out of tries at interface 0
                }
                }
            }
            if constexpr (TWO) {
                // Two levels.  Level 1: a lane draws -- tries and its retry stream advance exactly as in the reference's loop --
                // until it holds a draw the per-ray bound cannot reject (or its last try, whose state the ray hands out), at most
                // kTwoLevelDraws draws per round: ~25 instructions a draw.  Level 2: the disk mapping, the direction and the interface-0
                // test, once per round, for the lanes that hold one: ~70.  A draw rejected at level 1 bumps no counter and leaves (o, d)
                // untouched, like a clip at interface 0 (which is what it would have been).  A draw at the disk's centre (rho = 0: the
                // reference's 0/0 -> NaN ray, which passes everything) is never rejected here.
                bool pending = false;
                float pu = 0.0f, pv = 0.0f;
                const float rminf = static_cast<float>(rminq) * (1.0f / 256.0f);
#pragma unroll
                for (int k = 0; k < kTwoLevelDraws; ++k) {
                    if (k > 0 && __ballot(searching && !pending) == 0ull) break;
                    if (searching && !pending) {
                        pu = rng_unit(xor128(rng));   // zoic.cpp:1930
                        pv = rng_unit(xor128(rng));
                        ++tries;
                        const float rho = fmaxf(fabsf(ffma(2.0f, pu, -1.0f)), fabsf(ffma(2.0f, pv, -1.0f)));
                        const bool hopeless = (rho < rminf) & (rho > 0.0f);
                        pending = !hopeless | (tries > static_cast<uint32_t>(kMaxTries));
                    }
                }
                if (searching && pending) {
                    d = retry_direction(T, sample_lens(pu, pv), o0x, o0y, maxScale, translation, sn, cs);   // zoic.cpp:1932-1943: BOTH components translated
                    bool near0;
                    const bool pass0 = clears_rear(o, d, near0);
                    if (GUARD && near0) { unsure = true; searching = false; }
                    else if (pass0) { cand = true; searching = false; }
                    else if (tries > static_cast<uint32_t>(kMaxTries)) searching = false;   // out of tries at interface 0
                }
            }
        }

        ZOIC_MARK(4)   // retry search loop
        // ---- the fresh batches move up; issued here so that the sampler's dependent load above never waits for these loads ------
        if (!fromPool) advance_batches();

        ZOIC_MARK(5)   // advance batches: probe issue + sample request
        // ---- one full trace for every lane that holds a candidate -------------------------------------------------------------
        bool ok = false;
        const V3 oStart = o, dStart = d;
        const bool firstTry = tries == 0;
        const unsigned long long candMask = __ballot(cand);
        ZOIC_PS_ADD(6, __popcll(__ballot(active)))
        if (candMask != 0ull) {
            ZOIC_PS_ADD(4, 1) ZOIC_PS_ADD(5, __popcll(candMask))
            if (memoryPhasesFirst) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
            uint32_t tirTry = 0;   // 0/1: this try ended in total internal reflection
            if constexpr (NS > 0) {
                if constexpr (STRICT) {
                    bool oor = false;
                    ok = trace_lens_strict_pred<NS>(T, o, d, tirTry, cand, oor);
                    if (__builtin_expect(__ballot(cand && oor) != 0ull, 0)) {   // never seen: a root left the lean sequences' verified range
                        if (cand && oor) { o = oStart; d = dStart; tirTry = 0; ok = trace_lens_strict(T, o, d, tirTry); }
                    }
                } else {
                    unsigned long long tirMask, unsureMask;
                    const unsigned long long alive = trace_lens_fast_pred<NS, GUARD>(fsurf, o, d, candMask, tirMask, unsureMask);
                    ok = mask_bit(alive, lane);
                    tirTry = mask_bit(tirMask, lane) ? 1u : 0u;
                    if constexpr (GUARD) unsure |= mask_bit(unsureMask, lane);
                }
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
                if (!unsure) lutMiss += (tirTry << 1) + (shortcut ? (tirTry * kOut) << 1 : 0u);
            } else {
                tir += static_cast<uint32_t>(__popcll(__ballot(tirTry != 0u))) + kOut * static_cast<uint32_t>(__popcll(__ballot(shortcut && tirTry != 0u)));
            }
            if (shortcut) tries = kOut;   // ... then finish the ray as the reference would
            else if (DEAD && cand && !ok && firstTry && (lutMiss & kRetryDeadBit) != 0u) toFinish = true;   // first try failed inside the lens: same
            if constexpr (NS > 0) {
                // the predicated trace does not keep the partial state of a failed ray; a ray that FINISHES failed (out of
                // tries) gets it from the branchy trace, which stops at the failing interface
                if (cand && !ok && tries > static_cast<uint32_t>(kMaxTries) && !(GUARD && unsure)) {
                    uint32_t ignored = 0;
                    o = oStart; d = dStart;
                    if constexpr (STRICT) (void)trace_lens_strict(T, o, d, ignored);
                    else (void)trace_lens_fast_rolled(T, o, d, ignored);
                }
            }
        }
        ZOIC_MARK(6)   // trace
        if (!memoryPhasesFirst) __builtin_amdgcn_s_setprio(0);
        // a lane that ran out at interface 0 hands out the untouched (o, d) of its last sample -- the reference's partial state
        // (the predicated trace scribbles over the registers of lanes that ride along)
        if (!cand) { o = oStart; d = dStart; }

        // ---- finished rays: counters + record; hand-overs; everything else goes (back) to the pool --------------------------
        const bool dropU = GUARD && active && unsure;                 // -> STRICT kernel, evaluated from scratch
        const bool dropF = DEAD && active && toFinish && !dropU;       // -> completed below, 64 at a time
        const bool finished = active && !searching && !toFinish && (ok || tries > static_cast<uint32_t>(kMaxTries)) && !dropU;
        if constexpr (DEFER) {
            if (finished || dropF) tirAcc += (lutMiss & ~kRetryDeadBit) >> 1;
        }
        {
            const uint32_t nv = static_cast<uint32_t>(__popcll(__ballot(finished && tries > static_cast<uint32_t>(kMaxTries))));
            vign += nv;                                                                       // zoic.cpp:1951-1957
            succ += static_cast<uint32_t>(__popcll(__ballot(finished))) - nv;
            ZOIC_PS_ADD(7, __popcll(__ballot(finished)))
#ifdef ZOIC_PS_LIST
            if constexpr (GUARD) {
                ZOIC_PS_LISTADD(0, __popcll(__ballot(dropU)))
                for (uint32_t b = 0; b < 5u; ++b) ZOIC_PS_LISTADD(1, static_cast<unsigned long long>(__popcll(__ballot(dropU && ((tries >> b) & 1u)))) << b)
            }
#endif
        }
        {
            float w = (tries > static_cast<uint32_t>(kMaxTries)) ? 0.0f : 1.0f;
            if (T.exposureOn) w *= T.exposureMul;                                            // zoic.cpp:1981-1987
            const uint32_t flags = (tries > 0 ? 1u : 0u) | (tries << 1) | ((lutMiss & 1u) << 6);
            constexpr bool kTransposed = (ZOIC_STORE_TRANSPOSED == 1) || (ZOIC_STORE_TRANSPOSED == 2 && IMAGE && !STRICT);
            // Phase A holds 64 CONSECUTIVE rays in lane order and most of them finish here: their records are one contiguous 2 KB
            // block.  store_ray_record writes it as 2 x 64 half-sectors at a 32-byte stride per instruction; transposed through LDS --
            // the upper halves of pool0 / pool1 are free during a fresh batch (poolCnt < 64 and the push comes after this) -- each of
            // the two store instructions writes one contiguous KB (lane j: 16-byte piece j, then 64 + j), with the lanes of unfinished
            // rays masked off.  Phase B's rays are scattered: they keep the per-lane store.
            if (kTransposed && !fromPool) {
                float4 *stA = pool0 + 64, *stB = reinterpret_cast<float4 *>(pool1) + 64;
                stA[lane] = make_float4(o.x * -1.0f, o.y * -1.0f, o.z * -1.0f, d.x * -1.0f);   // zoic.cpp:1960-1961
                stB[lane] = make_float4(d.y * -1.0f, d.z * -1.0f, w, __builtin_bit_cast(float, flags));
                __builtin_amdgcn_wave_barrier();
                const unsigned long long fin = __ballot(finished);
                const float4 *half = (lane & 1u) ? stB : stA;
                float4 *dst = reinterpret_cast<float4 *>(out + (idx - lane));   // idx = batch base + lane (advance_batches has moved base1 on)
                const float4 p0 = half[lane >> 1], p1 = half[32u + (lane >> 1)];
                if ((fin >> (lane >> 1)) & 1ull) dst[lane] = p0;
                if ((fin >> (32u + (lane >> 1))) & 1ull) dst[64u + lane] = p1;
                __builtin_amdgcn_wave_barrier();
            } else
            if (finished) {
#if ZOIC_EXP_WHATIF == 5
                const uint32_t at = idx & 0xffffu;   // timing only: every record into one 2 MB window (the write stream never reaches DRAM)
#else
                const uint32_t at = idx;
#endif
                if constexpr (ZOIC_STORE_NT == 2 || (ZOIC_STORE_NT == 1 && IMAGE))
                    store_ray_record_nt(out, at, o.x * -1.0f, o.y * -1.0f, o.z * -1.0f, d.x * -1.0f, d.y * -1.0f, d.z * -1.0f, w, flags);
                else store_ray_record(out, at, o.x * -1.0f, o.y * -1.0f, o.z * -1.0f, d.x * -1.0f, d.y * -1.0f, d.z * -1.0f, w, flags);   // zoic.cpp:1960-1961
            }
#if ZOIC_EXP_WHATIF == 6
            __builtin_amdgcn_s_waitcnt(0x0f70);      // timing only: vmcnt(0) right behind the record stores -- are their acknowledgements what a later wait pays for?
#endif
        }
        ZOIC_MARK(7)   // finish: counters + record store
        if constexpr (GUARD) {
            const unsigned long long m = __ballot(dropU);
            if (m != 0ull) {
                if (dropU) unsureLds[unsureCnt + mask_rank(m)] = idx;
                unsureCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
        if constexpr (DEAD) {
            const unsigned long long m = __ballot(dropF);
            if (m != 0ull) {
                if (dropF) deadLds[deadCnt + mask_rank(m)] = idx;
                deadCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
        {
            const bool keep = active && !finished && !dropU && !dropF;
            const unsigned long long m = __ballot(keep);
            if (m != 0ull) {
                if (keep) {
                    const uint32_t slot = poolCnt + mask_rank(m);
                    uint32_t packed = (lutMiss & 0x7fu) | (tries << kPoolTriesShift) | (dead ? kPoolDeadBit : 0u) |
                                      ((lutMiss & kRetryDeadBit) ? kPoolRetryDeadBit : 0u);
                    if constexpr (TWO) packed |= rminq << 16;
                    pool0[slot] = make_float4(__builtin_bit_cast(float, idx), o0x, o0y, __builtin_bit_cast(float, packed));
#if ZOIC_POOL_SLIM
                    pool1[slot] = make_float2(sn, cs);
#else
                    pool1[slot] = make_float4(maxScale, translation, sn, cs);
#endif
                }
                // the retry stream of a ray that has not drawn yet is seeded when it is popped (tries == 0): nothing to store
                if (__ballot(keep && tries != 0u) != 0ull) { if (keep) pool2[poolCnt + mask_rank(m)] = make_uint4(rng.x, rng.y, rng.z, rng.w); }
                poolCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
        ZOIC_MARK(8)   // hand-overs + pool push
#ifdef ZOIC_PASS_STATS
        if (drain) rt[9] += __builtin_readcyclecounter() - drainT0;   // [9]: the whole of every pass run without fresh work (the wave's tail)
#endif
        // ---- hand-over lists: emptied in whole batches ---------------------------------------------------------------------
        if constexpr (GUARD) {
            if (unsureCnt >= 64u) {   // one atomic reserves exactly the entries written: no holes in the work list
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(ZOIC_KARG(redoCount), unsureCnt);
                at = __builtin_amdgcn_readfirstlane(at);
                uint32_t *list = ZOIC_KARG(redoList);
                for (uint32_t j = lane; j < unsureCnt; j += 64u) list[at + j] = unsureLds[j];
                unsureCnt = 0;
            }
        }
        if constexpr (DEAD) {
            while (deadCnt >= 64u) {
                deadCnt -= 64u;
                const bool nanDraw = finish_dead_ray<STRICT>(T, B, lutLds, bokehLds, samples, ZOIC_KARG(rngStates), ZOIC_KARG(rayBase), out, deadLds[deadCnt + lane]);
                const uint32_t ns = static_cast<uint32_t>(__popcll(__ballot(nanDraw)));
                succ += ns; vign += 64u - ns;
            }
        }
    }

    // ---- what is left in the hand-over lists ---------------------------------------------------------------------------------
    if constexpr (GUARD) {
        if (unsureCnt != 0u) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(ZOIC_KARG(redoCount), unsureCnt);
            at = __builtin_amdgcn_readfirstlane(at);
            uint32_t *list = ZOIC_KARG(redoList);
            for (uint32_t j = lane; j < unsureCnt; j += 64u) list[at + j] = unsureLds[j];
        }
    }
    if constexpr (DEAD) {
        if (deadCnt != 0u) {
            const bool mine = lane < deadCnt;
            bool nanDraw = false;
            if (mine) nanDraw = finish_dead_ray<STRICT>(T, B, lutLds, bokehLds, samples, ZOIC_KARG(rngStates), ZOIC_KARG(rayBase), out, deadLds[lane]);
            const uint32_t ns = static_cast<uint32_t>(__popcll(__ballot(mine && nanDraw)));
            succ += ns; vign += deadCnt - ns;
        }
    }
    if constexpr (DEFER) {   // TIR bumps of the rays this wave finished
        uint32_t t = tirAcc;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        tir += static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t)));
    }
    ZOIC_PS_FLUSH
    // ---- counters: the wave totals, one atomic per counter per wave -------------------------------------------------------
    DeviceCounters *counters = counter_set(ZOIC_KARG(counters));
    if (counters) {
        if (lane == 0) {
            if (succ) atomicAdd(&counters->succes, static_cast<unsigned long long>(succ));
            if (vign) atomicAdd(&counters->vignetted, static_cast<unsigned long long>(vign));
            if (tir) atomicAdd(&counters->tir, static_cast<unsigned long long>(tir));
        }
    }
}

// the precisions / roles are separate kernels so that each can carry its own register-budget attributes
#define ZOIC_POOL_PARAMS const KolbTable T, const BokehTables B, const float4 *__restrict__ samples, const uint4 *__restrict__ rngStates, \
        uint64_t rayBase, uint32_t n, RayRecord *__restrict__ out, DeviceCounters *counters, unsigned int *__restrict__ workCursor,        \
        uint32_t ldsWords, uint32_t chunkRays, uint32_t chunksPerPart, uint32_t minSearching, uint32_t *__restrict__ redoList,             \
        unsigned int *__restrict__ redoCount, unsigned int *__restrict__ clearCursor
#define ZOIC_POOL_KERNEL(NAME_, ATTR_, STRICT_, GUARD_)                                                                        \
    template <int NS, bool DEAD, bool IMAGE, bool TWO = false>                                                               \
    __global__ __launch_bounds__(kRefillBlock) ATTR_ void NAME_(ZOIC_POOL_PARAMS)                                             \
    {                                                                                                                        \
        kolb_pool_body<STRICT_, NS, GUARD_, DEAD, IMAGE, TWO>(T, B, samples, n, out, ldsWords, minSearching);                 \
    }
ZOIC_POOL_KERNEL(kolb_pool_strict_kernel, ZOIC_POOL_ATTR_STRICT, true, false)          // STRICT, whole batch
ZOIC_POOL_KERNEL(kolb_pool_fast_kernel, ZOIC_POOL_ATTR_FAST, false, false)             // FAST unchecked
ZOIC_POOL_KERNEL(kolb_pool_guard_kernel, ZOIC_POOL_ATTR_FAST, false, true)             // FAST decision-safe: lists what it cannot decide for kolb_listed_kernel
#undef ZOIC_POOL_KERNEL
#undef ZOIC_POOL_PARAMS

int launch_kolb_listed(const KolbTable &table, const BokehTables &bokeh, const float4 *d_samples, const uint4 *d_rng, uint64_t rayBase, uint32_t m,
                       RayRecord *out, DeviceCounters *d_counters, unsigned int *d_redoCursor, uint32_t *d_redoList, unsigned int *d_redoCount,
                       unsigned grid, void *stream);   // kolb_listed.hip

// mode: 0 = STRICT, 1 = FAST decision-safe, 2 = FAST unchecked.  d_scratch: the work list of mode 1 (kolb_scratch_dwords():
// one dword per sample of a launch)
template <bool DEAD, bool IMAGE, bool TWO = false>
int launch_kolb_pool_impl(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                          uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_cursorPair, unsigned *parity,
                          int mode, uint32_t *d_scratch, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == 1 && !d_scratch) return static_cast<int>(hipErrorInvalidValue);
    constexpr uint64_t kMaxPerLaunch = 1ull << 31;   // 32-bit ray offsets inside the kernel; larger batches are split
    uint32_t *d_redoList = d_scratch;
    for (uint64_t done = 0; done < n; done += kMaxPerLaunch) {
        const uint64_t m = (n - done < kMaxPerLaunch) ? (n - done) : kMaxPerLaunch;
        hipError_t e = hipSuccess;
        unsigned int *d_workCursor = d_cursorPair + (*parity & 1u) * kCursorBlockWords;          // zero: cleared by the previous launch's workgroup 0
        unsigned int *d_clearCursor = d_cursorPair + ((*parity & 1u) ^ 1u) * kCursorBlockWords;
        *parity ^= 1u;
        const unsigned grid = persistent_grid(m, kWavesPerBlock);
        const WorkGrain grain = work_grain(m, mode == 0 ? 256u : 512u);
        const uint32_t chunkRays = grain.chunkRays, chunksPerPart = grain.chunksPerPart;
        RayRecord *o = out + done;
        const float4 *sp = reinterpret_cast<const float4 *>(d_samples) + done;
        const uint4 *rp = d_rng ? reinterpret_cast<const uint4 *>(d_rng) + done : nullptr;
        const uint32_t ldsWords = IMAGE ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;   // bokeh row cell records: 4 KB at 256 rows, 32 KB at the 2048-row limit
        const auto lds_bytes = [&](bool guard) {
            return static_cast<size_t>(ldsWords + kLutLdsWords + kWavesPerBlock * (kPoolWaveWords + (guard ? kPoolListWords : 0u) + (DEAD ? kPoolListWords : 0u))) * sizeof(float);
        };
        unsigned int *redoCount = d_workCursor + kRedoCountOffset, *redoCursor = d_workCursor + kRedoCursorOffset;
#define ZOIC_LAUNCH_POOL(KERNEL_, NS_, CURSOR_, GUARD_)                                                                          \
    hipLaunchKernelGGL((KERNEL_<NS_, DEAD, IMAGE, TWO>), dim3(grid), dim3(kRefillBlock), lds_bytes(GUARD_), st, table, bokeh, sp, rp, rayBase + done, \
                       static_cast<uint32_t>(m), o, d_counters, CURSOR_, ldsWords, chunkRays, chunksPerPart, kMinSearching, d_redoList, redoCount, d_clearCursor)
#define ZOIC_LAUNCH_POOL_BY_COUNT(KERNEL_, CURSOR_, GUARD_)                                                                      \
    switch (table.lensCount) {  /* unrolled instantiations for the interface counts of real prescriptions */                   \
    case 7: ZOIC_LAUNCH_POOL(KERNEL_, 7, CURSOR_, GUARD_); break;                                                               \
    case 8: ZOIC_LAUNCH_POOL(KERNEL_, 8, CURSOR_, GUARD_); break;                                                               \
    case 9: ZOIC_LAUNCH_POOL(KERNEL_, 9, CURSOR_, GUARD_); break;                                                               \
    case 10: ZOIC_LAUNCH_POOL(KERNEL_, 10, CURSOR_, GUARD_); break;                                                             \
    case 11: ZOIC_LAUNCH_POOL(KERNEL_, 11, CURSOR_, GUARD_); break;                                                             \
    case 12: ZOIC_LAUNCH_POOL(KERNEL_, 12, CURSOR_, GUARD_); break;                                                             \
    default: ZOIC_LAUNCH_POOL(KERNEL_, 0, CURSOR_, GUARD_); break;                                                              \
    }
        if (mode == 0) { ZOIC_LAUNCH_POOL_BY_COUNT(kolb_pool_strict_kernel, d_workCursor, false) }
        else if (mode == 2) { ZOIC_LAUNCH_POOL_BY_COUNT(kolb_pool_fast_kernel, d_workCursor, false) }
        else {
            ZOIC_LAUNCH_POOL_BY_COUNT(kolb_pool_guard_kernel, d_workCursor, true)
            e = hipGetLastError();
            if (e != hipSuccess) return static_cast<int>(e);
            // the rays it listed (kolb_listed_body.hpp): the reference's arithmetic where a decision is too close to call;
            // workgroups beyond the list's length retire at once
#if ZOIC_EXP_WHATIF != 3
            e = static_cast<hipError_t>(launch_kolb_listed(table, bokeh, sp, rp, rayBase + done, static_cast<uint32_t>(m), o, d_counters, redoCursor,
                                                           d_redoList, redoCount, grid, stream));
#endif
        }
#undef ZOIC_LAUNCH_POOL_BY_COUNT
#undef ZOIC_LAUNCH_POOL
        e = hipGetLastError();
        if (e != hipSuccess) return static_cast<int>(e);
    }
    return 0;
}

#ifdef ZOIC_PASS_STATS
static int read_pass_stats(unsigned long long *acc8, int reset)   // adds this translation unit's copy
{
    unsigned long long v[8];
    hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_passStats), sizeof(v));
    if (e != hipSuccess) return static_cast<int>(e);
    for (int i = 0; i < 8; ++i) acc8[i] += v[i];
    if (reset) { const unsigned long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_passStats), z, sizeof(z)); }
    return static_cast<int>(e);
}
static int read_region_cycles(unsigned long long *acc16, int reset)
{
    unsigned long long v[16];
    hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_regionCycles), sizeof(v));
    if (e != hipSuccess) return static_cast<int>(e);
    for (int i = 0; i < 16; ++i) acc16[i] += v[i];
    if (reset) { const unsigned long long z[16] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_regionCycles), z, sizeof(z)); }
    return static_cast<int>(e);
}
#endif

// cell records usable by the IMAGE kernels: present and small enough for LDS (tables.hpp)
inline bool kolb_image_cells(const KolbTable &table, const BokehTables &bokeh)
{
    return table.useImage && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240;
}

}  // namespace zoic
