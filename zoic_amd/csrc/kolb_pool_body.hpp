// kolb_pool_body.hpp -- the Kolb kernel as whole-wave BATCHES plus a per-wave ray POOL in LDS.
// (compiled twice: kolb_pool.hip for cameras without retry-dead rays, kolb_pool_dead.hip for those with)
//
// Why.  camera_create_ray retries a rejected sample up to 26 more times (zoic.cpp:1927-1947).  Round 2's kernel
// (kolb_refill_body.hpp) kept a ray in its lane until it was finished and refilled the free lanes of every pass from a
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
// Order of memory operations in a pass (vmcnt is ONE in-order counter): pool pop (LDS) -> candidate search (the bokeh
// sampler's dependent global load) -> request the next fresh batch -> trace -> record stores -> pool push (LDS).  The next
// pass waits for its batch with vmcnt(2): the two record stores issued after the request are never waited for.
#pragma once
#include "kolb_refill_body.hpp"   // RefillArgs / ZOIC_KARG, setup_ray, retry_direction, finish_dead_ray

#pragma STDC FP_CONTRACT OFF

namespace zoic {

// LDS per wave: the pool, 128 entries of three 16-byte pieces each, stored piece-major (three arrays of 128 float4: a
// push or pop is three conflict-free ds_write_b128 / ds_read_b128), then the hand-over lists (128 ray indices each).
// 128 entries: a pass starts with at most 63 pooled rays left behind and pushes at most 64.
constexpr uint32_t kPoolEntries = 128;
constexpr uint32_t kPoolWaveWords = kPoolEntries * 12u;
constexpr uint32_t kPoolListWords = 128;
// packed word of a pooled ray: bit 0 outside the LUT, bits 1-6 the ray's TIR tally (DEFER kernels), bits 8-12 tries,
// bit 13 dead pixel, bit 14 retry-dead
constexpr uint32_t kPoolTriesShift = 8, kPoolDeadBit = 1u << 13, kPoolRetryDeadBit = 1u << 14;

#ifndef ZOIC_POOL_ATTR_FAST
#define ZOIC_POOL_ATTR_FAST __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
#ifndef ZOIC_POOL_ATTR_STRICT
#define ZOIC_POOL_ATTR_STRICT __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif

template <bool STRICT, int NS, bool GUARD, bool LISTED, bool DEAD>
__device__ __forceinline__ void kolb_pool_body(const KolbTable &T, const BokehTables &B, const float4 *__restrict__ samples,
                                               uint32_t n, RayRecord *__restrict__ out, uint32_t ldsWords, uint32_t minSearching)
{
    static_assert(!(GUARD && STRICT) && !(LISTED && !STRICT), "GUARD is a FAST mode, LISTED the STRICT kernel behind it");
    constexpr bool DEFER = GUARD || DEAD;   // rays may leave this kernel unfinished: TIR bumps are tallied per ray
    constexpr uint32_t kOut = static_cast<uint32_t>(kMaxTries) + 1u;   // tries of a ray that ran out (zoic.cpp:1927: tries <= 25)
    uint32_t redoChunk = 0, redoChunksPerPart = 0;
    if constexpr (LISTED) {   // the work list's length is only known on the device
        n = *ZOIC_KARG(redoCount);
        if (n == 0u) return;
        redoChunk = n > (1u << 20) ? 256u : 64u;
        const uint32_t totalChunks = (n + redoChunk - 1u) / redoChunk;
        if (blockIdx.x * kWavesPerBlock >= totalChunks) return;   // whole workgroup: nothing listed for it
        redoChunksPerPart = (totalChunks + kCursorParts - 1u) / kCursorParts;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // LDS, once per workgroup: the 32 exit-pupil LUT pairs, then the bokeh row cell records (tables.hpp)
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
    float4 *pool0 = reinterpret_cast<float4 *>(zoicDynLds + kLutLdsWords + ldsWords + wave * kPoolWaveWords);   // idx, o0x, o0y, packed
    float4 *pool1 = pool0 + kPoolEntries;                                                                       // maxScale, translation, sn, cs
    uint4 *pool2 = reinterpret_cast<uint4 *>(pool1 + kPoolEntries);                                             // the ray's retry stream
    uint32_t *lists = reinterpret_cast<uint32_t *>(zoicDynLds + kLutLdsWords + ldsWords + kWavesPerBlock * kPoolWaveWords) +
                      wave * ((GUARD ? kPoolListWords : 0u) + (DEAD ? kPoolListWords : 0u));
    uint32_t *unsureLds = lists;                                  // GUARD: rays for the STRICT kernel
    uint32_t *deadLds = lists + (GUARD ? kPoolListWords : 0u);    // DEAD: retry-dead rays whose first try failed
    uint32_t poolCnt = 0, unsureCnt = 0, deadCnt = 0;             // wave-uniform

    // fresh work: chunks of consecutive samples claimed from the partition cursors (work_cursor.hpp); [next, end) is what is
    // left of the wave's chunk, `pre` the batch requested one pass ahead
    uint32_t next = 0, end = 0, part = blockIdx.x % kCursorParts, partsTried = 0;
    float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t preIdx = 0, preBase = 0, preCnt = 0;
    bool havePre = false;
    const auto request_batch = [&]() {
        havePre = false;
        if (next >= end) {
            const uint32_t cr = LISTED ? redoChunk : ZOIC_KARG(chunkRays), cpp = LISTED ? redoChunksPerPart : ZOIC_KARG(chunksPerPart);
            if (!claim_chunk(ZOIC_KARG(workCursor), lane, part, partsTried, cr, cpp, n, next, end)) return;
        }
        preBase = next;
        preCnt = (end - next < 64u) ? end - next : 64u;
        const uint32_t wi = (lane < preCnt) ? next + lane : next;
        if constexpr (LISTED) { preIdx = ZOIC_KARG(redoList)[wi]; pre = samples[preIdx]; }
        else pre = samples[wi];
        next += preCnt;
        havePre = true;
    };
    request_batch();

    uint32_t succ = 0, vign = 0, tir = 0;   // wave totals (SGPRs)
    uint32_t tirAcc = 0;                    // DEFER: per-lane sum of the TIR tallies of the rays this lane finished

    const bool memoryPhasesFirst = T.useImage != 0;
    for (;;) {
        const bool drain = !havePre;                                        // no fresh sample left for this wave
        const bool fromPool = poolCnt >= 64u || (drain && poolCnt != 0u);
        if (!fromPool && drain) break;
        if (memoryPhasesFirst) __builtin_amdgcn_s_setprio(1);
        FastSurfaceTable fsurf = nullptr;
        if constexpr (GUARD && (ZOIC_GUARD_PIN != 0)) fsurf = launder_table(kernarg_fast_surfaces());
        else if constexpr (!STRICT) fsurf = kernarg_fast_surfaces();
        (void)fsurf;

        // ---- the pass's 64 rays: a fresh batch (phase A) or 64 pooled rays (phase B) -------------------------------------
        bool active, fresh, dead, unsure = false;
        uint32_t idx, tries, lutMiss;   // lutMiss: bit 0 outside the LUT, bits 1.. the TIR tally, kRetryDeadBit
        float o0x, o0y, maxScale, translation, sn, cs, u, v;
        Rng rng{1, 2, 3, 4};
        if (!fromPool) {
            active = lane < preCnt;
            idx = LISTED ? preIdx : preBase + lane;
            const RaySetup rs = setup_ray<STRICT>(T, lutLds, pre.x, pre.y);
            o0x = rs.o0x; o0y = rs.o0y; maxScale = rs.maxScale; translation = rs.translation; sn = rs.sn; cs = rs.cs;
            lutMiss = rs.flags; dead = rs.dead;
            if constexpr (GUARD) unsure = active && T.useLUT && rs.lutEdge;
            u = pre.z; v = pre.w;
            tries = 0; fresh = true;
        } else {
            const uint32_t cnt = poolCnt < 64u ? poolCnt : 64u;
            poolCnt -= cnt;
            active = lane < cnt;
            const uint32_t slot = poolCnt + (active ? lane : 0u);
            const float4 e0 = pool0[slot], e1 = pool1[slot];
            const uint4 e2 = pool2[slot];
            const uint32_t packed = __builtin_bit_cast(uint32_t, e0.w);
            idx = __builtin_bit_cast(uint32_t, e0.x); o0x = e0.y; o0y = e0.z;
            maxScale = e1.x; translation = e1.y; sn = e1.z; cs = e1.w;
            rng = Rng{e2.x, e2.y, e2.z, e2.w};
            tries = (packed >> kPoolTriesShift) & 31u;
            dead = (packed & kPoolDeadBit) != 0u;
            lutMiss = (packed & 0x7fu) | ((packed & kPoolRetryDeadBit) ? kRetryDeadBit : 0u);
            u = 0.0f; v = 0.0f; fresh = false;
        }

        // ---- candidate search: draw lens samples until one clears the rear element's housing ---------------------------------
        // (zoic.cpp:1870-1925 first sample, 1927-1947 retries; tries and the retry stream advance exactly as in the reference's
        // loop; the loop is wave-uniform and goes on while enough lanes are looking to be worth the others' wait)
        V3 o{o0x, o0y, T.originShift}, d{0.0f, 0.0f, 1.0f};
        bool cand = false, finiteSample = true;
        bool searching = GUARD ? (active && !unsure) : active;
        bool toFinish = false;   // a retry-dead ray whose first try has failed -> the dead list
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
                // dead pixel (outside the image circle, LUT entries zero): whatever finite point the sampler returns, the
                // direction is (0 - o.x, 0 - o.y, dirZ); samples in [0,1)^2 off the disk mapping's 0/0 centre skip the sampler
                const bool plainSample = (u >= 0.0f) & (u < 1.0f) & (v >= 0.0f) & (v < 1.0f) & !((u == 0.5f) & (v == 0.5f));
                const bool skipSampler = first && dead && plainSample;
                V2 lens{0.0f, 0.0f};
                if (!skipSampler) lens = lens_sample<STRICT>(T, B, bokehLds, u, v);
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
                else if constexpr (GUARD) pass0 = interface0_clear_fast_guard(load_surface<false>(fsurf, 0), o, d, near0);
                else pass0 = interface0_clear_fast(load_surface<false>(fsurf, 0), o, d);
                if (GUARD && near0) { unsure = true; searching = false; }   // too close to call: no decision is taken here
                else if (pass0) { cand = true; searching = false; }
                else {
                    // a clip at interface 0 bumps no TIR counter and leaves (o, d) untouched: for a dead pixel all 27 tries are this one
                    if (first && dead && finiteSample) tries = kOut;
                    if (tries > static_cast<uint32_t>(kMaxTries)) searching = false;   // out of tries at interface 0
                    else if (DEAD && first && (lutMiss & kRetryDeadBit) != 0u) { toFinish = true; searching = false; }   // no retry can succeed
                }
            }
            const uint32_t looking = static_cast<uint32_t>(__popcll(__ballot(searching)));
            if (looking < (drain ? 1u : minSearching)) break;
        }

        // ---- request the batch after this one: issued here so that the sampler's dependent load above never waits for it ----
        if (!fromPool) request_batch();

        // ---- one full trace for every lane that holds a candidate -------------------------------------------------------------
        bool ok = false;
        const V3 oStart = o, dStart = d;
        const bool firstTry = tries == 0;
        if (__ballot(cand) != 0ull) {
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
            } else if (cand) {
                if constexpr (STRICT) ok = trace_lens_strict(T, o, d, tirTry);
                else if constexpr (GUARD) { bool u2 = false; ok = trace_lens_fast_rolled(T, o, d, tirTry, &u2); unsure |= u2; }
                else ok = trace_lens_fast_rolled(T, o, d, tirTry);
            }
            const bool shortcut = cand && !ok && firstTry && dead && finiteSample && !(GUARD && unsure);
            // the shortcut stands for 26 more identical failures: account for their TIR bumps as well
            if constexpr (DEFER) {
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
        if (!memoryPhasesFirst) __builtin_amdgcn_s_setprio(0);
        // a lane that ran out at interface 0 hands out the untouched (o, d) of its last sample -- the reference's partial state
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
        }
        if (finished) {
            float w = (tries > static_cast<uint32_t>(kMaxTries)) ? 0.0f : 1.0f;
            if (T.exposureOn) w *= T.exposureMul;                                            // zoic.cpp:1981-1987
            store_ray_record(out, idx, o.x * -1.0f, o.y * -1.0f, o.z * -1.0f, d.x * -1.0f, d.y * -1.0f, d.z * -1.0f, w,   // zoic.cpp:1960-1961
                             (tries > 0 ? 1u : 0u) | (tries << 1) | ((lutMiss & 1u) << 6));
        }
        if constexpr (GUARD) {
            const unsigned long long m = __ballot(dropU);
            if (m != 0ull) {
                const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                if (dropU) unsureLds[unsureCnt + r] = idx;
                unsureCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
        if constexpr (DEAD) {
            const unsigned long long m = __ballot(dropF);
            if (m != 0ull) {
                const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                if (dropF) deadLds[deadCnt + r] = idx;
                deadCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
        {
            const bool keep = active && !finished && !dropU && !dropF;
            const unsigned long long m = __ballot(keep);
            if (m != 0ull) {
                const uint32_t r = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
                if (keep) {
                    const uint32_t slot = poolCnt + r;
                    const uint32_t packed = (lutMiss & 0x7fu) | (tries << kPoolTriesShift) | (dead ? kPoolDeadBit : 0u) |
                                            ((lutMiss & kRetryDeadBit) ? kPoolRetryDeadBit : 0u);
                    pool0[slot] = make_float4(__builtin_bit_cast(float, idx), o0x, o0y, __builtin_bit_cast(float, packed));
                    pool1[slot] = make_float4(maxScale, translation, sn, cs);
                    pool2[slot] = make_uint4(rng.x, rng.y, rng.z, rng.w);
                }
                poolCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
        // ---- hand-over lists: flushed in whole batches ---------------------------------------------------------------------
        if constexpr (GUARD) {
            if (unsureCnt >= 64u) {   // one atomic reserves exactly the entries written
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

#define ZOIC_POOL_PARAMS const KolbTable T, const BokehTables B, const float4 *__restrict__ samples, const uint4 *__restrict__ rngStates, \
        uint64_t rayBase, uint32_t n, RayRecord *__restrict__ out, DeviceCounters *counters, unsigned int *__restrict__ workCursor,        \
        uint32_t ldsWords, uint32_t chunkRays, uint32_t chunksPerPart, uint32_t minSearching, uint32_t *__restrict__ redoList,             \
        unsigned int *__restrict__ redoCount, uint8_t *__restrict__ deadMap
#define ZOIC_POOL_KERNEL(NAME_, ATTR_, STRICT_, GUARD_, LISTED_)                                                               \
    template <int NS, bool DEAD>                                                                                             \
    __global__ __launch_bounds__(kRefillBlock) ATTR_ void NAME_(ZOIC_POOL_PARAMS)                                             \
    {                                                                                                                        \
        kolb_pool_body<STRICT_, NS, GUARD_, LISTED_, DEAD>(T, B, samples, n, out, ldsWords, minSearching);                    \
    }
ZOIC_POOL_KERNEL(kolb_pool_strict_kernel, ZOIC_POOL_ATTR_STRICT, true, false, false)          // STRICT, whole batch
ZOIC_POOL_KERNEL(kolb_pool_strict_listed_kernel, ZOIC_POOL_ATTR_STRICT, true, false, true)    // STRICT over the work list of the GUARD kernel
ZOIC_POOL_KERNEL(kolb_pool_fast_kernel, ZOIC_POOL_ATTR_FAST, false, false, false)             // FAST unchecked
ZOIC_POOL_KERNEL(kolb_pool_guard_kernel, ZOIC_POOL_ATTR_FAST, false, true, false)             // FAST decision-safe
#undef ZOIC_POOL_KERNEL
#undef ZOIC_POOL_PARAMS

// mode: 0 = STRICT, 1 = FAST decision-safe, 2 = FAST unchecked.  d_scratch: the work list of mode 1 (one dword per sample of a launch)
template <bool DEAD>
int launch_kolb_pool_impl(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                          uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor,
                          int mode, uint32_t *d_scratch, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == 1 && !d_scratch) return static_cast<int>(hipErrorInvalidValue);
    constexpr uint64_t kMaxPerLaunch = 1ull << 31;   // 32-bit ray offsets inside the kernel; larger batches are split
    uint32_t *d_redoList = d_scratch;
    for (uint64_t done = 0; done < n; done += kMaxPerLaunch) {
        const uint64_t m = (n - done < kMaxPerLaunch) ? (n - done) : kMaxPerLaunch;
        hipError_t e = reset_work_cursors(d_workCursor, st);   // both cursor sets and the work list's counter
        if (e != hipSuccess) return static_cast<int>(e);
        const unsigned grid = persistent_grid(m, kWavesPerBlock);
        const WorkGrain grain = work_grain(m, mode == 0 ? 256u : 512u);
        const uint32_t chunkRays = grain.chunkRays, chunksPerPart = grain.chunksPerPart;
        RayRecord *o = out + done;
        static const uint32_t minSearching = [] { const char *e = std::getenv("ZOIC_MIN_SEARCHING"); return e ? static_cast<uint32_t>(std::atoi(e)) : kMinSearching; }();
        const float4 *sp = reinterpret_cast<const float4 *>(d_samples) + done;
        const uint4 *rp = d_rng ? reinterpret_cast<const uint4 *>(d_rng) + done : nullptr;
        const uint32_t ldsWords = (table.useImage && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
        const auto lds_bytes = [&](bool guard) {
            return static_cast<size_t>(ldsWords + kLutLdsWords + kWavesPerBlock * (kPoolWaveWords + (guard ? kPoolListWords : 0u) + (DEAD ? kPoolListWords : 0u))) * sizeof(float);
        };
        unsigned int *redoCount = d_workCursor + kRedoCountOffset, *redoCursor = d_workCursor + kRedoCursorOffset;
#define ZOIC_LAUNCH_POOL(KERNEL_, NS_, CURSOR_, GUARD_)                                                                          \
    hipLaunchKernelGGL((KERNEL_<NS_, DEAD>), dim3(grid), dim3(kRefillBlock), lds_bytes(GUARD_), st, table, bokeh, sp, rp, rayBase + done, \
                       static_cast<uint32_t>(m), o, d_counters, CURSOR_, ldsWords, chunkRays, chunksPerPart, minSearching, d_redoList, redoCount, \
                       static_cast<uint8_t *>(nullptr))
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
            // the rays it listed, in the reference's arithmetic; workgroups beyond the list's length retire at once
            ZOIC_LAUNCH_POOL_BY_COUNT(kolb_pool_strict_listed_kernel, redoCursor, false)
        }
#undef ZOIC_LAUNCH_POOL_BY_COUNT
#undef ZOIC_LAUNCH_POOL
        e = hipGetLastError();
        if (e != hipSuccess) return static_cast<int>(e);
    }
    return 0;
}

}  // namespace zoic
