// kolb_listed_body.hpp -- the second kernel of a decision-safe FAST launch: the rays the GUARD kernel could not decide
// (kolb_pool_body.hpp: a clip inside an interface's guard band, in practice at the stop, which the reference traces as a sphere
// of |R| ~ 1e4 cm -- zoic.cpp:933, 1111-1117) evaluated so that every decision is the reference's.
//
// Round 3 ran these rays through the STRICT pool kernel from scratch.  On the fisheye (C4) that was 14.5 % of the launch: 4.9 M
// listed rays x 5335 lane-instructions -- a STRICT set-up, a STRICT first try, and on average 1.1 MORE tries in STRICT (a ray
// that grazes the stop fails half of the time), although a RETRY is a fresh lens sample that lands in a guard band no more often
// than any other (1.85 %).  The rule here -- the same for every listed ray whatever the length of the list, so that a ray's
// bits do not depend on how a frame is cut into launches (a sharded frame equals the one-GPU frame):
//     set-up            : the reference's arithmetic (setup_ray<true>)
//     try 0             : a STRICT try (83 % of the listed rays were listed AT their first try)
//     try k >= 1        : FAST arithmetic with its guard bands on the STRICT set-up constants; a try with a decision inside a
//                         guard band is evaluated again as a STRICT try (same draws) and THAT result stands.
//     a STRICT try      : lens sample and direction in the reference's arithmetic, the trace in the reference's arithmetic UP TO AND
//                         INCLUDING THE STOP (listed_strict_try).  The stop is where the reference's own rounding decides (a sphere
//                         of |R| ~ 1e4 cm: the hit point is good to ~ulp(|R|)) and why the ray was listed; behind it the trace goes on
//                         in FAST arithmetic with its guard bands from the STRICT state at the stop, and only if one of THOSE clips is
//                         too close to call (bands of a few ulps: ~1e-5 of the tries) the rest is traced in the reference's
//                         arithmetic from the stop on instead.  Half of a 12-interface STRICT trace (160 instructions per
//                         interface against 29) is saved.
// Every accept / reject decision is therefore either a FAST decision outside every guard band or a STRICT one -- what
// "decision-safe" means (include/zoic_amd.h, ZOIC_PRECISION_FAST).
//
// Long lists (> kShortList rays; the fisheye): batches + a per-wave pool in LDS like the main kernel, with THREE kinds of pass:
//     A  64 fresh listed rays : STRICT set-up + STRICT first try; the failures go to the pool
//     B  64 pooled rays       : one more FAST-guarded try each (candidate search + predicated trace); a try too close to call
//                               goes to the STRICT stack with its draws rewound
//     C  64 rays of the STRICT stack : that try again, in the reference's arithmetic; the failures return to the pool
// The pool and the STRICT stack share one double-ended LDS array of 192 entries per wave (B is taken first while the pool holds
// 64, then C, then A: pool + stack never exceed 190).
// Short lists (<= kShortList rays: every camera but the fisheye): their time is not work but ONE ray's chain of tries at one
// wave per SIMD, so the tries of a ray run side by side in the G lanes of a group (round 3's listed_short), same rule per try.
// G = 16 / 8 / 4 by the length of the list, so that all its waves are resident at once (4 per SIMD at 128 VGPRs = 4096 waves):
// the chain that costs is try 0's STRICT trace, and a second generation of waves would pay it a second time.
#pragma once
#include "kolb_pool_body.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

constexpr uint32_t kListedEntries = 192;                       // pool (from the bottom) + STRICT stack (from the top)
constexpr uint32_t kListedWaveWords = kListedEntries * 12u;    // 48-byte entries, piece-major like the main kernel's pool

// one try of a listed ray, k >= 1, in the reference's arithmetic: (o, d) = the state the reference leaves (zoic.cpp:1927-1947)
__device__ __forceinline__ V3 listed_retry_direction_strict(const KolbTable &T, const BokehTables &B, const float *bokehLds, float u, float v,
                                                            float o0x, float o0y, float maxScale, float translation, float sn, float cs)
{
    return retry_direction(T, lens_sample<true>(T, B, bokehLds, u, v), o0x, o0y, maxScale, translation, sn, cs);
}

// A STRICT try's trace (see the rule above), branchy: leaves (o, d) as the arithmetic that decided leaves them.
__device__ __forceinline__ bool listed_strict_try(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount)
{
    const int n = T.lensCount, stop = T.apertureElement;
    if (!trace_lens_strict_range(T, o, d, tirCount, 0, stop)) return false;
    if (stop + 1 >= n) return true;
    const V3 os = o, ds = d;
    uint32_t tirFast = 0;
    bool unsure = false;
    const bool ok = trace_lens_fast_rolled_from(kernarg_fast_surfaces(), n, stop + 1, o, d, tirFast, &unsure);
    if (!unsure) { tirCount += tirFast; return ok; }
    o = os; d = ds;
    return trace_lens_strict_range(T, o, d, tirCount, stop + 1, n - 1);
}

// the same, predicated and unrolled for NS interfaces (alive lanes bit-identical to listed_strict_try; rays that FINISH failed
// get their partial state from it).  The FAST tail has no early-out bookkeeping of its own: half a trace.
template <int NS>
__device__ __forceinline__ bool listed_strict_try_pred(const KolbTable &T, V3 &o, V3 &d, uint32_t &tirCount, bool cand, bool &outOfRange)
{
    const int stop = T.apertureElement;
    bool alive = trace_lens_strict_pred<NS, true>(T, o, d, tirCount, cand, outOfRange, 0, stop);
    if (stop + 1 >= NS) return alive;
    const unsigned long long aliveMask = __ballot(alive);
    if (aliveMask == 0ull) return false;
    const V3 os = o, ds = d;
    // FAST tail from the STRICT state at the stop (fast_optics.hpp: the arithmetic of trace_lens_fast_pred / _rolled)
    const FastSurfaceTable surf = kernarg_fast_surfaces();
    const float inv = frsq_fast(fast_norm2(d));
    V3 u{d.x * inv, d.y * inv, d.z * inv};
    float oAxis2 = fast_axis2(o);
    unsigned long long al = aliveMask, tirSeen = 0ull, unsure = 0ull, nanRays = 0ull;
    bool firstTail = true;
#pragma unroll
    for (int i = 1; i < NS; ++i) {
        if (i <= stop) continue;
        const FastSurface S = load_surface<true>(surf, i);
        const FastHit h = fast_hit(S, o, oAxis2, u);
        if (firstTail) { nanRays = __ballot(is_nan_ray(h)); firstTail = false; }   // a NaN ray "passes" every comparison of the reference
        const unsigned long long clipped = ~__ballot(h.h2 <= S.housingLo);
        unsure |= al & clipped & __ballot(h.h2 <= S.housingHi);
        o = h.hit;
        oAxis2 = h.h2;
        const unsigned long long tirHere = __ballot(fast_refract(S, h, u) < 0.0f);
        al &= ~clipped;
        tirSeen |= al & tirHere;
        al &= ~tirHere;
    }
    d = u;
    const uint32_t lane = threadIdx.x & 63u;
    bool ok = mask_bit(al | (aliveMask & nanRays), lane);
    if (alive && !mask_bit(unsure, lane)) tirCount += mask_bit(tirSeen, lane) ? 1u : 0u;
    if (__builtin_expect(unsure != 0ull, 0)) {   // a clip behind the stop too close to call: the rest in the reference's arithmetic
        const bool redo = alive && mask_bit(unsure, lane);
        V3 o2 = os, d2 = ds;
        bool oor2 = false;
        const bool ok2 = trace_lens_strict_pred<NS, true>(T, o2, d2, tirCount, redo, oor2, stop + 1, NS - 1);
        if (redo) { o = o2; d = d2; ok = ok2; outOfRange |= oor2; }
    }
    return ok;
}

// The tries of ONE listed ray side by side in the G lanes of a group (round 3's listed_short; the rule per try is the one above):
// lane j of the group evaluates try k = G * round + j.  The tries of a ray are independent given its retry stream (try k >= 1 uses
// draws 2(k-1), 2(k-1)+1); the first success in try order wins and TIR bumps count for the tries before it only (try 26 hands out its
// state with weight 0 whether it got through or not, zoic.cpp:1927 / 1951).  Shared by the listed kernel's short lists (G fixed per
// launch) and by the resident tile workers (mailbox.hip: G = the lanes a wave can spare per listed ray of its batch).
//   rng      : this lane's copy of the ray's retry stream AT ITS TRY'S DRAWS (the caller steps lane j over j - 1 pairs before round 0;
//              stepped over G - 1 pairs here for the next round)
//   shift    : the group's first lane (its lanes are shift ... shift + G - 1)
//   done     : the ray is finished (or the group holds none); returns the new value
//   emit     : this lane holds the ray's final state -> (oOut, dOut, wOut, flagsOut) = the record's fields (negated, zoic.cpp:1960-1961)
template <int NS>
__device__ __forceinline__ bool listed_group_round(const KolbTable &T, const BokehTables &B, const float *bokehLds, const float4 &s, const RaySetup &rs,
                                                   bool deadPixel, Rng &rng, uint32_t G, uint32_t j, uint32_t shift, uint32_t round, bool done,
                                                   bool &emit, V3 &oOut, V3 &dOut, float &wOut, uint32_t &flagsOut, uint32_t &succ, uint32_t &vign, uint32_t &tir)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t kOutTries = static_cast<uint32_t>(kMaxTries) + 1u;
    const V3 o0{rs.o0x, rs.o0y, T.originShift};
    const uint32_t k = G * round + j;                                // this lane's try: 0 = the sample's own lens point, k = tries
    const bool valid = !done && k <= kOutTries;
    V3 o = o0, d{0.0f, 0.0f, 1.0f};
    uint32_t tirTry = 0;
    bool ok = false, unsure = false;
    float u = 0.0f, v = 0.0f;
    emit = false;
    // every lane's direction first; then ONE FAST-guarded trace for the tries k >= 1 of the whole wave -- the predicated trace of
    // the long-list path, so that a ray's bits do not depend on the path (one arithmetic, written with explicit FMAs: fast_optics.hpp) --
    // and the reference's arithmetic for try 0 and for the tries that were too close to call
    if (valid) {
        if (k == 0u) {
            V2 lens = lens_sample<true>(T, B, bokehLds, s.z, s.w);   // zoic.cpp:1870
            if (!T.useLUT) d = V3{(lens.x * T.rearAperture) - o.x, (lens.y * T.rearAperture) - o.y, T.dirZ};
            else {                                                    // zoic.cpp:1913-1924: x-only translation
                lens.x *= rs.maxScale; lens.y *= rs.maxScale;
                lens.x += rs.translation;
                const float rx = lens.x * rs.cs - lens.y * rs.sn, ry = lens.x * rs.sn + lens.y * rs.cs;
                d = V3{rx - o.x, ry - o.y, T.dirZ};
            }
        } else {
            u = rng_unit(xor128(rng));                                // zoic.cpp:1930
            v = rng_unit(xor128(rng));
            d = retry_direction(T, lens_sample<false>(T, B, bokehLds, u, v), rs.o0x, rs.o0y, rs.maxScale, rs.translation, rs.sn, rs.cs);
        }
    }
    const bool fastTry = valid && k != 0u;
    if constexpr (NS > 0) {
        const unsigned long long fastMask = __ballot(fastTry);
        if (fastMask != 0ull) {
            // the search's interface-0 test first, exactly as a B pass takes it (a try that dies there leaves (o, d) untouched)
            bool near0 = false;
            const bool pass0 = interface0_clear_fast<true>(load_surface<false>(kernarg_fast_surfaces(), 0), o, d, near0);
            const bool cand = fastTry && pass0 && !near0;
            unsure = fastTry && near0;
            unsigned long long tirMask, unsureMask;
            V3 ot = o, dt = d;
            const unsigned long long alive = trace_lens_fast_pred<NS, true>(kernarg_fast_surfaces(), ot, dt, __ballot(cand), tirMask, unsureMask);
            if (cand) {
                ok = mask_bit(alive, lane);
                tirTry = mask_bit(tirMask, lane) ? 1u : 0u;
                unsure |= mask_bit(unsureMask, lane);
                if (ok) { o = ot; d = dt; }
                else if (!unsure && k == kOutTries) {   // only try 26's partial state is ever handed out
                    uint32_t ignored = 0;
                    (void)trace_lens_fast_rolled(T, o, d, ignored);
                }
            }
        }
    } else if (fastTry) ok = trace_lens_fast_rolled(T, o, d, tirTry, &unsure);
    if (valid && (k == 0u || unsure)) {   // the reference's arithmetic: try 0, and a try too close to call (same draws)
        o = o0; tirTry = 0;
        if (k != 0u) d = listed_retry_direction_strict(T, B, bokehLds, u, v, rs.o0x, rs.o0y, rs.maxScale, rs.translation, rs.sn, rs.cs);
        ok = listed_strict_try(T, o, d, tirTry);
    }
    // the next try of this lane, k + G, starts at draw 2 (k + G - 1): G - 1 pairs past where this try ended (2 k; try 0 drew nothing)
    for (uint32_t a = 0; a + 1u < G; ++a) { (void)xor128(rng); (void)xor128(rng); }
    // the group's decision, in try order
    const unsigned long long okAll = __ballot(valid && ok);
    const uint32_t groupBits = G >= 32u ? 0xffffffffu : ((1u << G) - 1u);
    const uint32_t okGroup = static_cast<uint32_t>(okAll >> shift) & groupBits;
    const uint32_t winner = okGroup ? static_cast<uint32_t>(__builtin_ctz(okGroup)) : G;   // lowest try that got through
    // a dead pixel whose try 0 failed: 26 identical failures follow -- ITS (STRICT) state with weight 0 and 27 x its TIR bump, as pass
    // A of the long lists hands it out (the other lanes' speculative tries of this ray count for nothing)
    const bool deadEnd = round == 0u && deadPixel && (okGroup & 1u) == 0u && !done;
    if (deadEnd) {
        if (valid && j == 0u) {
            tir += (kOutTries + 1u) * tirTry;
            wOut = 0.0f;
            if (T.exposureOn) wOut *= T.exposureMul;                      // zoic.cpp:1981-1987
            flagsOut = 1u | (kOutTries << 1) | ((rs.flags & 1u) << 6);
            emit = true;
            ++vign;
        }
    } else {
        if (valid && j < winner) tir += tirTry;                            // only the tries the reference actually ran
        const bool last = k == kOutTries;                                  // try 26 failed as well: weight 0, ITS partial state
        if (valid && (j == winner || (winner == G && last))) {
            const bool okRay = j == winner && !last;
            wOut = okRay ? 1.0f : 0.0f;
            if (T.exposureOn) wOut *= T.exposureMul;                       // zoic.cpp:1981-1987
            flagsOut = (k > 0u ? 1u : 0u) | (k << 1) | ((rs.flags & 1u) << 6);
            emit = true;
            succ += okRay ? 1u : 0u; vign += okRay ? 0u : 1u;
        }
    }
    oOut = V3{o.x * -1.0f, o.y * -1.0f, o.z * -1.0f}; dOut = V3{d.x * -1.0f, d.y * -1.0f, d.z * -1.0f};   // zoic.cpp:1960-1961
    return done || deadEnd || winner != G || G * (round + 1u) > kOutTries;
}

// Short lists: kShortGroup tries of a ray side by side, 64 / kShortGroup rays per wave.
__device__ __forceinline__ uint32_t short_group_for(uint32_t n) { return n <= 16384u ? 16u : (n <= 32768u ? 8u : 4u); }
// a dead pixel (outside the image circle) whose try 0 fails: its 26 retries are that try again (kolb_pool_body.hpp) -- unless the sample
// is one of the few the sampler maps to a non-finite lens point
__device__ __forceinline__ bool listed_dead_pixel(const KolbTable &T, const BokehTables &B, const float *bokehLds, const RaySetup &rs, const float4 &s)
{
    bool deadPixel = rs.dead;
    if (deadPixel) {
        const bool plainSample = (s.z >= 0.0f) & (s.z < 1.0f) & (s.w >= 0.0f) & (s.w < 1.0f) & !((s.z == 0.5f) & (s.w == 0.5f));
        if (!plainSample) { const V2 l0 = lens_sample<true>(T, B, bokehLds, s.z, s.w); deadPixel = (fabsf(l0.x) <= 3.0e38f) && (fabsf(l0.y) <= 3.0e38f); }
    }
    return deadPixel;
}
template <int NS, uint32_t kShortGroup>
__device__ __forceinline__ void listed_short_hybrid(const KolbTable &T, const BokehTables &B, const float2 *lutLds, const float *bokehLds,
                                                    const float4 *__restrict__ samples, uint32_t n, RayRecord *__restrict__ out)
{
    constexpr uint32_t kShortRaysPerWave = 64u / kShortGroup;
    const uint32_t lane = threadIdx.x & 63u, j = lane % kShortGroup, g = lane / kShortGroup;
    const uint32_t wavesTotal = gridDim.x * kWavesPerBlock, waveId = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint32_t *list = ZOIC_KARG(redoList);
    const uint4 *states = ZOIC_KARG(rngStates);
    const uint64_t rayBase = ZOIC_KARG(rayBase);
    uint32_t succ = 0, vign = 0, tir = 0;   // per lane; reduced at the end
    for (uint32_t first = waveId * kShortRaysPerWave; first < n; first += wavesTotal * kShortRaysPerWave) {
        const uint32_t li = first + g;
        const bool have = li < n;
        const uint32_t idx = list[have ? li : n - 1u];
        const float4 s = samples[idx];
        const RaySetup rs = setup_ray<true>(T, lutLds, s.x, s.y);
        Rng rng;
        if (states) { const uint4 q = states[idx]; rng = Rng{q.x, q.y, q.z, q.w}; }
        else rng = rng_for_ray(T.seed, rayBase + idx);
        for (uint32_t a = 1; a < j; ++a) { (void)xor128(rng); (void)xor128(rng); }   // lane j >= 1 starts at draw 2 (j - 1)
        bool done = !have;
        const bool deadPixel = listed_dead_pixel(T, B, bokehLds, rs, s);
        for (uint32_t round = 0; round * kShortGroup <= static_cast<uint32_t>(kMaxTries) + 1u; ++round) {
            bool emit; V3 o, d; float w; uint32_t flags;
            done = listed_group_round<NS>(T, B, bokehLds, s, rs, deadPixel, rng, kShortGroup, j, kShortGroup * g, round, done, emit, o, d, w, flags, succ, vign, tir);
            if (emit) store_ray_record(out, idx, o.x, o.y, o.z, d.x, d.y, d.z, w, flags);
            if (__ballot(!done) == 0ull) break;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { succ += __shfl_xor(succ, off, 64); vign += __shfl_xor(vign, off, 64); tir += __shfl_xor(tir, off, 64); }
    DeviceCounters *counters = counter_set(ZOIC_KARG(counters));
    if (counters && lane == 0) {
        if (succ) atomicAdd(&counters->succes, static_cast<unsigned long long>(succ));
        if (vign) atomicAdd(&counters->vignetted, static_cast<unsigned long long>(vign));
        if (tir) atomicAdd(&counters->tir, static_cast<unsigned long long>(tir));
    }
}

// (Round 5 tried the short lists the other way round -- the reference's set-up and try 0 at one ray per LANE, then rounds sharing the wave's
// lanes among the rays still open -- parity green and not faster: with 64 tries in a wave's round the rare per-lane paths (a try too close
// to call, a failed try 26's partial state) run in most rounds.  profiles/ab_r05/ab_listed_short_waves.log; the code is commit d9afcf9.)
__device__ __forceinline__ void wave_lds_fence()   // the wave's LDS writes have landed before any lane reads another lane's words (mailbox.hip)
{
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

template <int NS>
__device__ __forceinline__ void kolb_listed_body(const KolbTable &T, const BokehTables &B, const float4 *__restrict__ samples,
                                                 RayRecord *__restrict__ out, uint32_t ldsWords, uint32_t minSearching)
{
    constexpr uint32_t kOut = static_cast<uint32_t>(kMaxTries) + 1u;   // tries of a ray that ran out (zoic.cpp:1927: tries <= 25)
    const uint32_t n = *ZOIC_KARG(redoCount);   // the work list's length is only known on the device
    if (n == 0u) return;
    // 64-entry chunks while the list is short (every wave gets work), 256 once it could feed the chip several times over
    const uint32_t redoChunk = n > (1u << 20) ? 256u : 64u;
    const uint32_t shortGroup = short_group_for(n), shortRaysPerWave = 64u / shortGroup;
    const uint32_t totalChunks = n <= kShortList ? (n + shortRaysPerWave - 1u) / shortRaysPerWave : (n + redoChunk - 1u) / redoChunk;
    if (blockIdx.x * kWavesPerBlock >= totalChunks) return;   // whole workgroup: nothing listed for it
    const uint32_t redoChunksPerPart = (totalChunks + kCursorParts - 1u) / kCursorParts;
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
    if (n <= kShortList) {
        if (shortGroup == 16u) listed_short_hybrid<NS, 16u>(T, B, lutLds, bokehLds, samples, n, out);
        else if (shortGroup == 8u) listed_short_hybrid<NS, 8u>(T, B, lutLds, bokehLds, samples, n, out);
        else listed_short_hybrid<NS, 4u>(T, B, lutLds, bokehLds, samples, n, out);
        return;
    }

    // the double-ended array: pool entries 0 .. poolCnt-1, STRICT stack entries kListedEntries-1 downwards
    float4 *pool0 = reinterpret_cast<float4 *>(zoicDynLds + kLutLdsWords + ldsWords + wave * kListedWaveWords);   // idx, o0x, o0y, packed
    uint4 *pool2 = reinterpret_cast<uint4 *>(pool0 + kListedEntries);                                             // the ray's retry stream
    float4 *pool1 = reinterpret_cast<float4 *>(pool2 + kListedEntries);                                           // maxScale, translation, sn, cs
    uint32_t poolCnt = 0, stackCnt = 0;   // wave-uniform

    uint32_t next = 0, end = 0, part = blockIdx.x % kCursorParts, partsTried = 0;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    uint32_t idx1 = 0, idx2 = 0, cnt1 = 0, cnt2 = 0;
    bool have1 = false, have2 = false;
    const auto request_batch = [&]() {   // -> b2
        have2 = false;
        if (next >= end) {
            if (!claim_chunk(ZOIC_KARG(workCursor), lane, part, partsTried, redoChunk, redoChunksPerPart, n, next, end)) return;
        }
        cnt2 = (end - next < 64u) ? end - next : 64u;
        const uint32_t wi = (lane < cnt2) ? next + lane : next;
        idx2 = ZOIC_KARG(redoList)[wi];
        s2 = samples[idx2];
        next += cnt2;
        have2 = true;
    };
    const auto advance_batches = [&]() {
        s1 = s2; idx1 = idx2; cnt1 = cnt2; have1 = have2;
        if (have1) request_batch();
    };
    request_batch();
    advance_batches();

    uint32_t succ = 0, vign = 0, tir = 0;   // wave totals (SGPRs): no ray leaves this kernel unfinished, TIR bumps count as they happen
    for (;;) {
        const bool drain = !have1;
        // B while the pool holds a full batch, then C, then fresh work; without fresh work whatever is left, pool first
        const bool passB = poolCnt >= 64u || (drain && poolCnt != 0u);
        const bool passC = !passB && (stackCnt >= 64u || (drain && stackCnt != 0u));
        if (!passB && !passC && drain) break;
        const bool strictPass = !passB;   // A and C run the reference's arithmetic

        bool active, dead = false, unsure = false, cand = false, searching = false, finiteSample = true;
        uint32_t idx, tries, lutMiss;
        float o0x, o0y, maxScale, translation, sn, cs;
        Rng rng{1, 2, 3, 4}, rngBefore{1, 2, 3, 4};
        V3 o, d{0.0f, 0.0f, 1.0f};
        const auto clears_rear_strict = [&](const V3 &oo, const V3 &dd) {
            bool inRange;
            bool p = interface0_clear_strict_lean(T, oo, dd, inRange);
            if (__builtin_expect(!inRange, 0)) p = interface0_clear_strict(T, oo, dd);   // never seen: guarded roots
            return p;
        };
        if (!passB && !passC) {
            // ---- A: 64 fresh listed rays, the reference's set-up and first try (zoic.cpp:1853-1925) ----------------------------
            active = lane < cnt1;
            idx = idx1;
            const RaySetup rs = setup_ray<true>(T, lutLds, s1.x, s1.y);
            o0x = rs.o0x; o0y = rs.o0y; maxScale = rs.maxScale; translation = rs.translation; sn = rs.sn; cs = rs.cs;
            lutMiss = rs.flags & 1u; dead = rs.dead;
            tries = 0;
            o = V3{o0x, o0y, T.originShift};
            const float u = s1.z, v = s1.w;
            V2 lens = lens_sample<true>(T, B, bokehLds, u, v);
            if (__ballot(dead) != 0ull) {   // dead pixel (outside the image circle, LUT entries zero): all 27 tries are this one (kolb_pool_body.hpp)
                const bool plainSample = (u >= 0.0f) & (u < 1.0f) & (v >= 0.0f) & (v < 1.0f) & !((u == 0.5f) & (v == 0.5f));
                if (dead && plainSample) lens = V2{0.0f, 0.0f};
                finiteSample = (fabsf(lens.x) <= 3.0e38f) && (fabsf(lens.y) <= 3.0e38f);
            }
            if (!T.useLUT) d = V3{(lens.x * T.rearAperture) - o.x, (lens.y * T.rearAperture) - o.y, T.dirZ};
            else {
                lens.x *= maxScale; lens.y *= maxScale;
                lens.x += translation;
                const float rx = lens.x * cs - lens.y * sn, ry = lens.x * sn + lens.y * cs;
                d = V3{rx - o.x, ry - o.y, T.dirZ};
            }
            cand = active && clears_rear_strict(o, d);
            advance_batches();
        } else {
            // ---- B / C: 64 rays from the pool / the STRICT stack -------------------------------------------------------------
            uint32_t cnt, slotBase;
            if (passB) { cnt = poolCnt < 64u ? poolCnt : 64u; poolCnt -= cnt; slotBase = poolCnt; }
            else { cnt = stackCnt < 64u ? stackCnt : 64u; stackCnt -= cnt; slotBase = kListedEntries - stackCnt - cnt; }
            active = lane < cnt;
            const uint32_t slot = slotBase + (active ? lane : 0u);
            const float4 e0 = pool0[slot];
            const uint4 e2 = pool2[slot];
            const float4 e1 = pool1[slot];
            const uint32_t packed = __builtin_bit_cast(uint32_t, e0.w);
            idx = __builtin_bit_cast(uint32_t, e0.x); o0x = e0.y; o0y = e0.z;
            maxScale = e1.x; translation = e1.y; sn = e1.z; cs = e1.w;
            rng = Rng{e2.x, e2.y, e2.z, e2.w};
            tries = (packed >> kPoolTriesShift) & 31u;
            lutMiss = packed & 1u;
            o = V3{o0x, o0y, T.originShift};
            if (passC) {
                // the try that was too close to call, again, in the reference's arithmetic: same draws (the stack holds the stream
                // BEFORE them and the try count before its increment)
                const float u = rng_unit(xor128(rng));                        // zoic.cpp:1930
                const float v = rng_unit(xor128(rng));
                ++tries;
                d = listed_retry_direction_strict(T, B, bokehLds, u, v, o0x, o0y, maxScale, translation, sn, cs);
                cand = active && clears_rear_strict(o, d);
            } else {
                // candidate search in FAST arithmetic with the guard band of interface 0 (kolb_pool_body.hpp): a lane keeps drawing
                // while enough lanes are looking; tries and the retry stream advance exactly as in the reference's loop
                const FastSurfaceTable fsurf = kernarg_fast_surfaces();
                searching = active;
                for (;;) {
                    const uint32_t looking = static_cast<uint32_t>(__popcll(__ballot(searching)));
                    if (looking < (drain ? 1u : minSearching)) break;
                    if (searching) {
                        if (tries == 0) {               // first retry of this ray: seed its private xorshift128 stream
                            const uint4 *states = ZOIC_KARG(rngStates);
                            if (states) { const uint4 r = states[idx]; rng = Rng{r.x, r.y, r.z, r.w}; }
                            else rng = rng_for_ray(kernarg_field<uint32_t, offsetof(KolbKernelArgs, T) + offsetof(KolbTable, seed)>(), ZOIC_KARG(rayBase) + idx);
                        }
                        rngBefore = rng;
                        const float u = rng_unit(xor128(rng));   // zoic.cpp:1930
                        const float v = rng_unit(xor128(rng));
                        ++tries;
                        d = retry_direction(T, lens_sample<false>(T, B, bokehLds, u, v), o0x, o0y, maxScale, translation, sn, cs);
                        bool near0;
                        const bool pass0 = interface0_clear_fast<true>(load_surface<false>(fsurf, 0), o, d, near0);
                        if (near0) { unsure = true; searching = false; }
                        else if (pass0) { cand = true; searching = false; }
                        else if (tries > static_cast<uint32_t>(kMaxTries)) searching = false;   // out of tries at interface 0
                    }
                }
            }
        }

        // ---- one full trace for every lane that holds a candidate --------------------------------------------------------------
        bool ok = false;
        const V3 oStart = o, dStart = d;
        const unsigned long long candMask = __ballot(cand);
        if (candMask != 0ull) {
            uint32_t tirTry = 0;
            if (strictPass) {
                if constexpr (NS > 0) {
                    bool oor = false;
                    ok = listed_strict_try_pred<NS>(T, o, d, tirTry, cand, oor);
                    if (__builtin_expect(__ballot(cand && oor) != 0ull, 0)) {   // never seen: a root left the lean sequences' verified range
                        if (cand && oor) { o = oStart; d = dStart; tirTry = 0; ok = listed_strict_try(T, o, d, tirTry); }
                    }
                } else if (cand) ok = listed_strict_try(T, o, d, tirTry);
            } else {
                if constexpr (NS > 0) {
                    unsigned long long tirMask, unsureMask;
                    const unsigned long long alive = trace_lens_fast_pred<NS, true>(kernarg_fast_surfaces(), o, d, candMask, tirMask, unsureMask);
                    ok = mask_bit(alive, lane);
                    tirTry = mask_bit(tirMask, lane) ? 1u : 0u;
                    unsure |= mask_bit(unsureMask, lane);
                } else if (cand) { bool u2 = false; ok = trace_lens_fast_rolled(T, o, d, tirTry, &u2); unsure |= u2; }
            }
            // a dead pixel's 26 retries repeat its first try bit for bit (the retry direction of lens = (0,0) is the first try's)
            const bool shortcut = cand && !ok && tries == 0u && dead && finiteSample;
            // TIR bumps count for the tries this pass DECIDED (a try on its way to the STRICT stack is counted there)
            tir += static_cast<uint32_t>(__popcll(__ballot(tirTry != 0u && !unsure))) + kOut * static_cast<uint32_t>(__popcll(__ballot(shortcut && tirTry != 0u)));
            if (shortcut) tries = kOut;
            if constexpr (NS > 0) {
                // the predicated traces do not keep the partial state of a failed ray; a ray that FINISHES failed gets it from the
                // branchy trace of the arithmetic that decided it
                if (cand && !ok && tries > static_cast<uint32_t>(kMaxTries) && !unsure) {
                    uint32_t ignored = 0;
                    o = oStart; d = dStart;
                    if (strictPass) (void)listed_strict_try(T, o, d, ignored);
                    else (void)trace_lens_fast_rolled(T, o, d, ignored);
                }
            }
        }
        // A: a dead pixel whose first try died at interface 0 (no trace): all 27 tries are that one
        if (!passB && !passC && active && !cand && dead && finiteSample) tries = kOut;
        if (!cand) { o = oStart; d = dStart; }   // a lane that ran out at interface 0 hands out the untouched (o, d) of its last sample

        // ---- finished rays; the rest goes to the pool, a try too close to call to the STRICT stack -----------------------------
        const bool toStack = active && unsure;
        const bool finished = active && !searching && !unsure && (ok || tries > static_cast<uint32_t>(kMaxTries));
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
        {
            const unsigned long long m = __ballot(toStack);
            if (m != 0ull) {
                if (toStack) {   // rewound: the stream before this try's draws, the try count before its increment
                    const uint32_t slot = kListedEntries - 1u - (stackCnt + mask_rank(m));
                    const uint32_t packed = (lutMiss & 1u) | ((tries - 1u) << kPoolTriesShift);
                    pool0[slot] = make_float4(__builtin_bit_cast(float, idx), o0x, o0y, __builtin_bit_cast(float, packed));
                    pool1[slot] = make_float4(maxScale, translation, sn, cs);
                    pool2[slot] = make_uint4(rngBefore.x, rngBefore.y, rngBefore.z, rngBefore.w);
                }
                stackCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
        {
            const bool keep = active && !finished && !toStack;
            const unsigned long long m = __ballot(keep);
            if (m != 0ull) {
                if (keep) {
                    const uint32_t slot = poolCnt + mask_rank(m);
                    const uint32_t packed = (lutMiss & 1u) | (tries << kPoolTriesShift);
                    pool0[slot] = make_float4(__builtin_bit_cast(float, idx), o0x, o0y, __builtin_bit_cast(float, packed));
                    pool1[slot] = make_float4(maxScale, translation, sn, cs);
                    if (tries != 0u) pool2[slot] = make_uint4(rng.x, rng.y, rng.z, rng.w);   // a ray that has not drawn yet is seeded when it is popped
                }
                poolCnt += static_cast<uint32_t>(__popcll(m));
            }
        }
    }
    DeviceCounters *counters = counter_set(ZOIC_KARG(counters));
    if (counters && lane == 0) {
        if (succ) atomicAdd(&counters->succes, static_cast<unsigned long long>(succ));
        if (vign) atomicAdd(&counters->vignetted, static_cast<unsigned long long>(vign));
        if (tir) atomicAdd(&counters->tir, static_cast<unsigned long long>(tir));
    }
}

template <int NS>
__global__ __launch_bounds__(kRefillBlock) ZOIC_POOL_ATTR_STRICT void kolb_listed_kernel(
    const KolbTable T, const BokehTables B, const float4 *__restrict__ samples, const uint4 *__restrict__ rngStates, uint64_t rayBase, uint32_t n,
    RayRecord *__restrict__ out, DeviceCounters *counters, unsigned int *__restrict__ workCursor, uint32_t ldsWords, uint32_t chunkRays,
    uint32_t chunksPerPart, uint32_t minSearching, uint32_t *__restrict__ redoList, unsigned int *__restrict__ redoCount,
    unsigned int *__restrict__ clearCursor)
{
    kolb_listed_body<NS>(T, B, samples, out, ldsWords, minSearching);
}

}  // namespace zoic
