// mailbox.hip -- camera_create_ray(node, input, output, tid) (zoic.cpp:1752) at CALL latency, one sample or one TILE at a time:
// a resident kernel per camera that render threads talk to through mapped, page-locked host memory.
//
// Why.  Arnold calls camera_create_ray once per camera sample from every render thread and waits for the ray.  Answering a
// call with a kernel launch + stream synchronise costs 28-33 us per sample (round 2) and 52-76 us per batch call whatever the
// batch carries (round 4): a bucket of 64 x 64 pixels served by launches runs 30-80x below what the kernels do.  Here nothing is
// launched per call:
//   * SLOT waves.  Slot w (= tid mod 64) of the mailbox is owned by wave w of the resident launch.  A render thread writes its
//     request into the slot's 64-byte REQUEST line (three 16-byte chunks, each ending in the call's sequence number, written last)
//     and spins on the answer.  The wave polls the line with ONE load across PCIe (lanes 0-3 fetch its four chunks; host memory
//     is mapped uncached on the GPU: every poll sees memory); when the three chunks carry one NEW sequence number the request is
//     complete.  A chunk is one PCIe transaction: torn reads are impossible within a chunk and detected across chunks, so no
//     fences or doorbells are needed in either direction;
//   * ONE SAMPLE (kind 0): the slot's wave evaluates the ray with one lane -- the reference's own loop, STRICT or FAST arithmetic
//     (optics.hpp / fast_optics.hpp; a FAST ray with a decision inside a guard band is re-evaluated on the spot by the listed
//     kernel's rule, kolb_listed_body.hpp) -- and writes the REPLY line: three 16-byte chunks, sequence number last;
//   * A TILE (kind 1: n <= 65536 AtCameraInput rows in mapped host memory -> n AtCameraOutput rows in mapped host memory): the
//     slot's wave POSTS the tile as a job in device memory -- descriptor, a ticket counter (generation << 32 | next batch), a bit
//     in the launch's 64-bit work mask -- and the WORKER waves of the launch (kTileWorkerGroups x 4, started with the first tile a
//     camera sees) take 64-sample batches off it with one atomicAdd each, the slot's wave among them.  A batch is evaluated at
//     FULL LANE WIDTH by the same device functions as a single sample (one ray per lane; ray i draws its retries from the stream
//     keyed by base + i exactly as the batch kernels do, so a tile equals zoic_create_rays_arnold bit for bit): the 28-byte input
//     rows arrive as 7 coalesced dword loads per lane across PCIe and are transposed through LDS, the 84-byte output rows leave
//     as 21 coalesced dword stores per lane.  Every wave releases its rows at system scope before it counts its batch done; the
//     wave that counts the last one writes the slot's TILE-DONE line.  No launch, no stream, no synchronise: a 4096-sample tile
//     is answered in ~15 us where a launch-based call took 76;
//   * the kernel retires by itself after 1 ms without a call and after 50 ms in any case (a resident kernel would stall the
//     application's hipDeviceSynchronize / hipFree for ever); the next call finds `alive == 0` and launches it again
//     (~20 us, once).  node_update / node_finish / the counter getters stop it first.  A wave never leaves with a batch in hand,
//     and a slot's wave hands out every batch of its own tile before it looks at its exit flag.
// Results are those of the batch kernels for RAYTRACED, bit for bit in STRICT and in FAST: every Kolb kernel of the library
// evaluates a ray with the same device functions, the FAST arithmetic is written with explicit FMAs (fast_optics.hpp: the branchy
// trace here and the unrolled trace of the batch kernels round alike) and a ray too close to call follows the listed kernel's rule
// (tests/test_boundary_gpu.py: a fresh tid's first call == the one-ray batch launch on the same stream; tests/test_tile_gpu.py: a
// tile == zoic_create_rays_arnold).  THINLENS: a single sample is evaluated in the reference's arithmetic (thin_ray_strict) in EVERY
// precision mode -- one lane has nothing to gain from the fast variant -- so under ZOIC_PRECISION_FAST with optical vignetting on,
// where the batch path runs thin_refill.hip's fast arithmetic, a per-sample ray and a batch ray of the same sample can differ in
// low-order bits (never in a decision: the fast vignetting test is decision-safe); a TILE takes the batch path's arithmetic
// (thin_vignet_try<true>) and equals it.  include/zoic_amd.h states this at zoic_camera_create_ray.
#include "kolb_listed_body.hpp"   // listed_one_ray; kolb_pool_body.hpp: setup_ray, retry_direction (+ kolb_device.hpp: lens_sample, the traces, zoicDynLds)
#include "mailbox.hpp"
#include "thin_device.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// One poll = ONE PCIe transaction: lanes 0..3 of the wave read the four 16-byte chunks of the slot's 64-byte request line
// with one load instruction (the other lanes repeat lane 3's address: no further request), the launch's control block in
// device memory rides along, and both are waited for together.  sc0 sc1: system scope -- never served from a GPU cache; asm
// volatile: never from a register.  (Three loads per poll from 16 resident waves made every poll slower: 7.7 us per call with
// one calling thread, 19 us with sixteen.)
__device__ __forceinline__ void poll_uncached(const uint4 *myChunk, const uint32_t *control, u32x4 &line, u32x4 &ctl)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %3, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(line), "=&v"(ctl) : "v"(myChunk), "v"(control) : "memory");
}
__device__ __forceinline__ uint32_t lane_word(uint32_t v, int lane) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane)); }
__device__ __forceinline__ void store_uncached(uint4 *p, uint4 q)
{
    const u32x4 v = {q.x, q.y, q.z, q.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

struct LaneRay { V3 o, d; float w; uint32_t tries, lutMiss, tir; bool vignetted; };

// camera_create_ray's RAYTRACED branch (zoic.cpp:1850-1964) for the (up to) 64 samples a wave holds, one ray per lane, in the batch
// kernels' ORDER OF WORK (kolb_pool_body.hpp) without their pool -- a tile's batch stays in its lanes until its last ray is done:
//   * set-up once at full width; then rounds of { candidate search: a lane draws lens samples until one clears interface 0 (a tenth of
//     a try: most rejected tries die there) or it runs out of tries; ONE trace for the lanes that hold a candidate }.  The wave's time is
//     the number of ROUNDS -- the deepest chain of tries that got past the rear element -- not the largest try count of its 64 rays
//     (the reference's loop as it stands, one ray per lane, measured 46 us per batch: every ray waits for the unluckiest one's 26
//     whole traces);
//   * dead pixels (outside the image circle all 27 tries are one) and retry-dead rays (dead_ray_end: no retry can reach the rear
//     element) end at their first failure as they do in the batch kernels;
//   * GUARD (decision-safe FAST): a ray with a decision inside a guard band stops where it stands and is evaluated by the listed
//     kernel's rule (listed_one_ray), on the spot.
// Every try is evaluated by the same device functions as everywhere else in the library: same bits as the batch kernels and as the
// reference's loop order (tests/test_tile_gpu.py, tests/test_boundary_gpu.py).  rng: the ray's retry stream at its first draw.
template <bool STRICT>
__device__ __forceinline__ LaneRay kolb_wave_rays(const KolbTable &T, const BokehTables &B, const float2 *lutLds, const float *bokehLds, float4 s,
                                                  Rng rng, bool active, bool guard)
{
    constexpr uint32_t kOut = static_cast<uint32_t>(kMaxTries) + 1u;   // tries of a ray that ran out (zoic.cpp:1927: tries <= 25)
    LaneRay r;
    const Rng rng0 = rng;
    const RaySetup rs = setup_ray<STRICT>(T, lutLds, s.x, s.y);
    r.lutMiss = rs.flags & 1u; r.tir = 0; r.tries = 0; r.vignetted = false;
    bool unsure = guard && T.useLUT && rs.lutEdge;
    const V3 o0{rs.o0x, rs.o0y, T.originShift};
    V2 lens = lens_sample<STRICT>(T, B, bokehLds, s.z, s.w);   // zoic.cpp:1870
    bool finiteSample = true;
    if (rs.dead) {   // whatever finite point the sampler returns, the direction is (0 - o.x, 0 - o.y, dirZ) (kolb_pool_body.hpp)
        const bool plainSample = (s.z >= 0.0f) & (s.z < 1.0f) & (s.w >= 0.0f) & (s.w < 1.0f) & !((s.z == 0.5f) & (s.w == 0.5f));
        if (plainSample) lens = V2{0.0f, 0.0f};
        finiteSample = (fabsf(lens.x) <= 3.0e38f) && (fabsf(lens.y) <= 3.0e38f);
    }
    V3 d;
    if (!T.useLUT) {                                           // zoic.cpp:1873-1877
        d = V3{(lens.x * T.rearAperture) - o0.x, (lens.y * T.rearAperture) - o0.y, T.dirZ};
    } else {                                                   // zoic.cpp:1913-1924: x-only translation on the first sample
        lens.x *= rs.maxScale; lens.y *= rs.maxScale;
        lens.x += rs.translation;
        const float rx = lens.x * rs.cs - lens.y * rs.sn, ry = lens.x * rs.sn + lens.y * rs.cs;
        d = V3{rx - o0.x, ry - o0.y, T.dirZ};
    }
    V3 o = o0;
    const bool deadPixel = rs.dead && finiteSample, retryDead = (rs.flags & kRetryDeadBit) != 0u;
    bool live = active && !unsure, endDead = false;
    const auto clears_rear = [&](const V3 &dd, bool &near0) {
        if constexpr (STRICT) {
            near0 = false;
            bool inRange;
            bool p = interface0_clear_strict_lean(T, o0, dd, inRange);
            if (__builtin_expect(!inRange, 0)) p = interface0_clear_strict(T, o0, dd);   // never seen: guarded roots
            return p;
        } else {
            bool p;
            if (guard) p = interface0_clear_fast<true>(T.fsurf[0], o0, dd, near0);
            else p = interface0_clear_fast<false>(T.fsurf[0], o0, dd, near0);
            return p;
        }
    };
    const auto draw = [&]() {                                   // zoic.cpp:1930-1943
        const float u = rng_unit(xor128(rng));
        const float v = rng_unit(xor128(rng));
        ++r.tries;
        d = retry_direction(T, lens_sample<STRICT>(T, B, bokehLds, u, v), rs.o0x, rs.o0y, rs.maxScale, rs.translation, rs.sn, rs.cs);
    };
    // what a FAILED try does next: the shortcuts of a ray's first failure, out of tries, or the next draw.  Returns false when the ray ends.
    const auto after_failure = [&](uint32_t tirTry) {
        if (r.tries == 0u && deadPixel) { r.tir += kOut * tirTry; r.tries = kOut; return false; }   // 26 more identical failures
        if (r.tries == 0u && retryDead) { endDead = true; return false; }
        if (r.tries > static_cast<uint32_t>(kMaxTries)) return false;
        draw();
        return true;
    };
    while (__ballot(live) != 0ull) {
        bool cand = false, searching = live;
        while (__ballot(searching) != 0ull) {
            if (searching) {
                bool near0 = false;
                const bool pass0 = clears_rear(d, near0);
                if (near0) { unsure = true; live = false; searching = false; }
                else if (pass0) { cand = true; searching = false; }
                else if (!after_failure(0u)) { o = o0; live = false; searching = false; }   // a clip at interface 0 leaves (o, d) untouched
            }
        }
        if (cand) {
            V3 ot = o0, dt = d;
            uint32_t tirTry = 0;
            bool ok, near = false;
            if constexpr (STRICT) ok = trace_lens_strict(T, ot, dt, tirTry);
            else ok = trace_lens_fast_rolled(T, ot, dt, tirTry, guard ? &near : nullptr);
            if (near) { unsure = true; live = false; }
            else {
                r.tir += tirTry;
                if (ok) { o = ot; d = dt; live = false; }
                else {
                    if (!after_failure(tirTry)) { if (!endDead) { o = ot; d = dt; } live = false; }   // the reference's partial state (zoic.cpp:1951-1961)
                }
            }
        }
    }
    if (endDead) {   // a retry-dead ray whose first try failed: tries 1 ... 26 die at interface 0; the state of the last draw
        const DeadRayEnd e = dead_ray_end<STRICT>(T, B, bokehLds, rs, rng0);
        r.o = e.o; r.d = e.d; r.w = e.w; r.tries = e.tries; r.vignetted = !e.nanDraw;
        return r;
    }
    if (unsure) {    // a decision too close to call: the ray is evaluated as the batch path's listed kernel does it (same rule, same bits)
        const ListedRay q = listed_one_ray(T, B, lutLds, bokehLds, s, rng0);
        r.o = q.o; r.d = q.d; r.w = q.w; r.tries = q.tries; r.lutMiss = q.lutMiss; r.tir = q.tir;
        r.vignetted = r.tries > static_cast<uint32_t>(kMaxTries);
        return r;
    }
    r.o = o; r.d = d;
    r.vignetted = r.tries > static_cast<uint32_t>(kMaxTries);
    r.w = r.vignetted ? 0.0f : 1.0f;                                    // zoic.cpp:1951-1957
    if (T.exposureOn) r.w *= T.exposureMul;                             // zoic.cpp:1981-1987
    return r;
}

// system scope (mapped host memory: never from / into a GPU cache) and agent scope (the job table: coherent across the XCDs' L2s)
// (the addresses arrive as integers: say "global" so that these are global_load / global_store, not flat_*)
typedef uint32_t __attribute__((address_space(1))) GlobalWord;
__device__ __forceinline__ uint32_t load_sys(const uint32_t *p) { return __hip_atomic_load((const GlobalWord *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_sys(uint32_t *p, uint32_t v) { __hip_atomic_store((GlobalWord *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <class V> __device__ __forceinline__ V load_dev(const V *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class V> __device__ __forceinline__ void store_dev(V *p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes of the job table in one instruction, past the non-coherent caches (sc1: device scope)
__device__ __forceinline__ void load_dev4x3(const uint4 *p, uint4 &a, uint4 &b, uint4 &c)   // p[0], p[1], p[2]: ONE wait for the three
{
    u32x4 x, y, z;
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:32 sc1\n\t"
                 "s_waitcnt vmcnt(0)" : "=&v"(x), "=&v"(y), "=&v"(z) : "v"(p) : "memory");
    a = make_uint4(x.x, x.y, x.z, x.w); b = make_uint4(y.x, y.y, y.z, y.w); c = make_uint4(z.x, z.y, z.z, z.w);
}
__device__ __forceinline__ void store_dev4(uint4 *p, uint4 q)
{
    const u32x4 v = {q.x, q.y, q.z, q.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t first_lane(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ unsigned long long first_lane64(unsigned long long v)
{
    return (static_cast<unsigned long long>(first_lane(static_cast<uint32_t>(v >> 32))) << 32) | first_lane(static_cast<uint32_t>(v));
}
__device__ __forceinline__ void wave_lds_fence()   // the wave's LDS writes have landed before any lane reads another lane's words
{
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

constexpr uint32_t kTileStageWords = 512;   // per wave: 64 x 7 input dwords, then 64 x 8 record dwords

// model: ZOIC_THINLENS 0 / ZOIC_RAYTRACED 1 (zoic_amd.h); mode: 0 STRICT, 1 FAST decision-safe, 2 FAST unchecked.
// Waves 0 .. 63 of the launch are the SLOT waves (wave w owns slot w: a render thread never waits for another thread's ray --
// one wave for all slots measured 8 us per call with one calling thread, 29 us with four), the waves behind them the tile WORKERS.
// st->control (device memory): [0] exit flag (set by wave 0: stop request / 1 ms without a call / 50 ms of life), [1] waves that
// have left, [2..3] wall-clock time of the last call any wave answered, [4..5] the work mask.
__global__ __launch_bounds__(kMailBlock) void mailbox_kernel(const KolbTable T, const ThinTable Th, const BokehTables B, int model, int mode,
                                                             char *mapped, MailDeviceState *st, DeviceCounters *counters, uint32_t ldsWords,
                                                             uint32_t totalWaves)
{
    if (threadIdx.x < kLutEntries) {
        zoicDynLds[2 * threadIdx.x] = T.lutMaxScale[threadIdx.x];
        zoicDynLds[2 * threadIdx.x + 1] = T.lutCentroidX[threadIdx.x];
    }
    const float *bokehLds = nullptr;
    if (ldsWords > 0) {
        for (uint32_t i = threadIdx.x; i < ldsWords; i += kMailBlock) zoicDynLds[kLutLdsWords + i] = __builtin_bit_cast(float, B.rowCells[i]);
        bokehLds = zoicDynLds + kLutLdsWords;
    }
    __syncthreads();
    const float2 *lutLds = reinterpret_cast<const float2 *>(zoicDynLds);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waveId = blockIdx.x * (kMailBlock / 64u) + (threadIdx.x >> 6);
    const bool slotRole = waveId < kMailSlots;
    const uint32_t slot = slotRole ? waveId : 0u;
    float *stage = zoicDynLds + kLutLdsWords + ldsWords + (threadIdx.x >> 6) * kTileStageWords;
    MailHeader *header = reinterpret_cast<MailHeader *>(mapped);
    const MailRequest *requests = reinterpret_cast<const MailRequest *>(mapped + kMailRequestsOffset);
    MailReply *replies = reinterpret_cast<MailReply *>(mapped + kMailRepliesOffset);
    uint32_t *tileFlags = reinterpret_cast<uint32_t *>(mapped + kMailTileFlagsOffset);   // [slot][batch]: the tile's sequence number when the batch's rows are complete
    uint32_t *control = st->control;
    unsigned long long *workMask = reinterpret_cast<unsigned long long *>(control + 4);
    volatile unsigned long long *lastCall = reinterpret_cast<volatile unsigned long long *>(control + 2);
    // All control flow below is wave-uniform (the polled words are broadcast to SGPRs).
    uint32_t mine = slotRole ? st->served[slot] : 0u;   // sequence number of the last call this slot answered
    const unsigned long long start = wall_clock64();   // 100 MHz
    if (waveId == 0 && lane == 0) *lastCall = start;
    uint32_t succ = 0, vign = 0, tir = 0;               // per lane; summed when the wave retires
    // slots no render thread has used yet are not watched at all: their waves leave at once (the host starts the launch again
    // when a new tid shows up -- 63 waves reading host memory for nothing slowed every poll of the busy ones down)
    bool watched = true;
    if (slotRole) {
        u32x4 line, ctl;
        poll_uncached(reinterpret_cast<const uint4 *>(header), control, line, ctl);
        watched = slot < lane_word(line.z, 0);
    }
    const uint4 *myChunk = reinterpret_cast<const uint4 *>(requests + slot) + (lane < 3u ? lane : 3u);
    bool ownJob = false;        // slot role: this slot's tile still has batches to hand out
    uint32_t idlePolls = 0;     // worker role
    while (watched) {
        if (wall_clock64() - start > kMailHardLifeTicks) break;   // safety net (never seen): the host reports a tile that is never answered
        uint32_t work = 0;      // 1: one sample (slot role), 2: one 64-sample batch of a tile
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        Rng rng{1u, 2u, 3u, 4u};
        uint32_t seq = 0, jobSlot = 0, batch = 0, jobN = 0, jobSeq = 0;
        unsigned long long jobIn = 0, jobOut = 0, jobBase = 0;
        if (slotRole && !ownJob) {
            u32x4 line, ctl;   // ctl: the launch's control block = {exit flag, waves out, time of the last call (lo, hi)}
            poll_uncached(myChunk, control, line, ctl);
            // chunk k of the request sits in lane k: {sx, sy, lensx, seq} {lensy, rng.x, rng.y, seq} {rng.z, rng.w, kind, seq} {stop (slot 0), ...}
            seq = lane_word(line.w, 0);
            const bool fresh = seq != mine && seq == lane_word(line.w, 1) && seq == lane_word(line.w, 2);
            const uint32_t stop = lane_word(line.x, 3);
            const uint32_t leave = lane_word(ctl.x, 0);
            const unsigned long long lastSeen = (static_cast<unsigned long long>(lane_word(ctl.w, 0)) << 32) | lane_word(ctl.z, 0);
            const unsigned long long now = wall_clock64();
            if (!fresh) {
                if (slot == 0 && (stop != 0u || (now > lastSeen && now - lastSeen > kMailIdleTicks) || now - start > kMailLifeTicks)) {
                    if (lane == 0) __hip_atomic_store(control, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // everybody out
                    break;
                }
                if (leave != 0u) break;
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            if (lane == 0) *lastCall = now;
            mine = seq;
            if (lane_word(line.z, 2) == 0u) {   // ---- one sample
                work = 1;
                s = make_float4(__builtin_bit_cast(float, lane_word(line.x, 0)), __builtin_bit_cast(float, lane_word(line.y, 0)),
                                __builtin_bit_cast(float, lane_word(line.z, 0)), __builtin_bit_cast(float, lane_word(line.x, 1)));
                rng = Rng{lane_word(line.y, 1), lane_word(line.z, 1), lane_word(line.x, 2), lane_word(line.y, 2)};
            } else {                            // ---- a tile: {inLo, inHi, n, seq} {outLo, outHi, baseLo, seq} {baseHi, -, kind, seq}
                jobN = lane_word(line.z, 0);
                jobIn = (static_cast<unsigned long long>(lane_word(line.y, 0)) << 32) | lane_word(line.x, 0);
                jobOut = (static_cast<unsigned long long>(lane_word(line.y, 1)) << 32) | lane_word(line.x, 1);
                jobBase = (static_cast<unsigned long long>(lane_word(line.x, 2)) << 32) | lane_word(line.z, 1);
                jobSeq = seq; jobSlot = slot; batch = 0;
                if (jobN == 0u) continue;       // (the host never posts an empty tile)
                work = 2;                       // batch 0 is this wave's, straight from the request: a tile of up to 64 samples involves nobody else
                if (jobN > 64u) {
                    // the rest is POSTED for the workers: descriptor (three 16-byte chunks, each ending in the tile's number, like a
                    // request line), then the ticket counter at batch 1, then the slot's bit.  Whoever draws a ticket of generation
                    // `seq` finds this descriptor; the ticket and the bit need no order between them (a worker that sees the bit first
                    // draws a stale ticket and comes back).
                    if (lane == 0) {
                        uint4 *J = reinterpret_cast<uint4 *>(st->jobs + slot);
                        store_dev4(J, make_uint4(lane_word(line.x, 0), lane_word(line.y, 0), jobN, seq));
                        store_dev4(J + 1, make_uint4(lane_word(line.x, 1), lane_word(line.y, 1), lane_word(line.z, 1), seq));
                        store_dev4(J + 2, make_uint4(lane_word(line.x, 2), (jobN + 63u) >> 6, 0u, seq));
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        store_dev(&st->tickets[slot].next, (static_cast<unsigned long long>(seq) << 32) | 1ull);
                        (void)__hip_atomic_fetch_or(workMask, 1ull << slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ownJob = true;
                }
            }
        }
        if (work == 0u) {   // ---- draw a batch: a slot's wave from its own tile, a worker from any slot with its bit set
            if (slotRole) jobSlot = slot;
            else {
                const unsigned long long mask = first_lane64(load_dev(workMask));
                if (mask == 0ull) {
                    if ((++idlePolls & 7u) == 0u && first_lane(load_dev(control)) != 0u) break;   // exit flag: only with nothing left to hand out
                    __builtin_amdgcn_s_sleep(4);   // the launch only lives while calls keep coming (1 ms): no point in polling slowly
                    continue;
                }
                const uint32_t r = waveId & 63u;   // every worker starts its search at another slot
                const unsigned long long rot = r ? ((mask >> r) | (mask << (64u - r))) : mask;
                jobSlot = (static_cast<uint32_t>(__builtin_ctzll(rot)) + r) & 63u;
            }
            // the ticket and -- speculatively, in the same round trip -- the descriptor
            const uint4 *J = reinterpret_cast<const uint4 *>(st->jobs + jobSlot);
            unsigned long long t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(&st->tickets[jobSlot].next, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint4 c0, c1, c2;
            load_dev4x3(J, c0, c1, c2);
            t = first_lane64(t);
            const uint32_t gen = static_cast<uint32_t>(t >> 32);
            if (first_lane(c0.w) != gen || first_lane(c1.w) != gen || first_lane(c2.w) != gen) {
                // read before the poster's stores were visible -- or a ticket of a tile that is long done: once more, behind the ticket
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                load_dev4x3(J, c0, c1, c2);
            }
            batch = static_cast<uint32_t>(t);
            const uint32_t batches = first_lane(c2.y);
            if (first_lane(c0.w) != gen || first_lane(c1.w) != gen || first_lane(c2.w) != gen || batch >= batches) {   // nothing left of that tile
                if (slotRole) ownJob = false; else __builtin_amdgcn_s_sleep(1);
                continue;
            }
            // the LAST valid ticket clears the slot's bit: the tile cannot complete (and the slot post another) before this wave is done
            if (batch + 1u == batches && lane == 0) (void)__hip_atomic_fetch_and(workMask, ~(1ull << jobSlot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            jobSeq = gen;
            jobN = first_lane(c0.z);
            jobIn = (static_cast<unsigned long long>(first_lane(c0.y)) << 32) | first_lane(c0.x);
            jobOut = (static_cast<unsigned long long>(first_lane(c1.y)) << 32) | first_lane(c1.x);
            jobBase = (static_cast<unsigned long long>(first_lane(c2.x)) << 32) | first_lane(c1.z);
            idlePolls = 0;
            work = 2;
        }

        // ---- the pass's samples: the slot's one sample in lane 0, or 64 consecutive rows of the tile --------------------------
        bool active = lane == 0;
        uint32_t first = 0, cnt = 1;
        unsigned long long rayIndex = 0;
        if (work == 2u) {
            first = batch * 64u;
            cnt = jobN - first < 64u ? jobN - first : 64u;
            // AtCameraInput rows are 7 dwords (sx sy dsx dsy lensx lensy relative_time): lane l fetches dword k * 64 + l of the batch's
            // 7 * cnt -- whole 256-byte runs per instruction across PCIe -- and picks its row out of LDS
            const uint32_t *src = reinterpret_cast<const uint32_t *>(jobIn) + static_cast<size_t>(first) * 7u;
            const uint32_t total = cnt * 7u;
            uint32_t w7[7];
#pragma unroll
            for (uint32_t k = 0; k < 7u; ++k) { const uint32_t j = k * 64u + lane; w7[k] = load_sys(src + (j < total ? j : total - 1u)); }
#pragma unroll
            for (uint32_t k = 0; k < 7u; ++k) stage[k * 64u + lane] = __builtin_bit_cast(float, w7[k]);
            wave_lds_fence();
            active = lane < cnt;
            const uint32_t row = (active ? lane : 0u) * 7u;
            s = make_float4(stage[row], stage[row + 1u], stage[row + 4u], stage[row + 5u]);
            wave_lds_fence();   // ... before the records go into the same words
            rayIndex = jobBase + first + lane;
        }

        // ---- the rays (ONE site for both kinds of work) -------------------------------------------------------------------------
        V3 o{0.0f, 0.0f, 0.0f}, d{0.0f, 0.0f, 0.0f};
        float w = 0.0f;
        uint32_t tries = 0, lutMiss = 0;
        if (model == 0) {   // THINLENS, zoic.cpp:1771-1846
            if (active) {
                ThinRay r;
                if (work == 2u) {   // a tile's ray = the batch kernels' ray: its own stream, seeded at its first redraw; FAST where they run FAST
                    Rng q{1u, 2u, 3u, 4u};
                    if (mode != 0 && Th.useDof && Th.ovDistance > 0.0f) r = thin_ray_fast_vignet(Th, B, bokehLds, s, q, [&] { q = rng_for_ray(Th.seed, rayIndex); });
                    else r = thin_ray_strict(Th, B, bokehLds, s, q, [&] { q = rng_for_ray(Th.seed, rayIndex); });
                } else {            // one sample: the calling tid's stream, the reference's arithmetic in every precision mode
                    Rng q = rng;
                    r = thin_ray_strict(Th, B, bokehLds, s, q, [] {});
                }
                o = r.origin; d = r.dir; w = r.w; tries = r.tries;
                if (Th.useDof) { if (tries > static_cast<uint32_t>(kMaxTries)) ++vign; else ++succ; }
            }
        } else {            // RAYTRACED: every lane takes part in the wave's rounds (kolb_wave_rays); idle lanes ride along
            if (work == 2u) rng = rng_for_ray(T.seed, rayIndex);
            LaneRay r;
            if (mode == 0) r = kolb_wave_rays<true>(T, B, lutLds, bokehLds, s, rng, active, false);
            else r = kolb_wave_rays<false>(T, B, lutLds, bokehLds, s, rng, active, mode == 1);
            o = V3{r.o.x * -1.0f, r.o.y * -1.0f, r.o.z * -1.0f}; d = V3{r.d.x * -1.0f, r.d.y * -1.0f, r.d.z * -1.0f};   // zoic.cpp:1960-1961
            w = r.w; tries = r.tries; lutMiss = r.lutMiss;
            if (active) { tir += r.tir; if (r.vignetted) ++vign; else ++succ; }
        }
        const uint32_t flags = (tries > 0 ? 1u : 0u) | (tries << 1) | (lutMiss << 6);

        // ---- the answer -----------------------------------------------------------------------------------------------------------
        if (work == 1u) {
            if (lane == 0) {
                uint4 *a = reinterpret_cast<uint4 *>(replies + slot);
                store_uncached(a, make_uint4(__builtin_bit_cast(uint32_t, o.x), __builtin_bit_cast(uint32_t, o.y), __builtin_bit_cast(uint32_t, o.z), seq));
                store_uncached(a + 1, make_uint4(__builtin_bit_cast(uint32_t, d.x), __builtin_bit_cast(uint32_t, d.y), __builtin_bit_cast(uint32_t, d.z), seq));
                store_uncached(a + 2, make_uint4(__builtin_bit_cast(uint32_t, w), flags, 0u, seq));
            }
            continue;
        }
        // AtCameraOutput rows (84 bytes = 21 floats: origin, dir, dOdx, dOdy, dDdx, dDdy, weight[3]) as zoic_create_rays_arnold
        // writes them (kernels.hip expand_outputs_kernel): origin / dir, dOdy = origin and dDdy = dir for retried rays
        // (zoic.cpp:1974-1977), weight r = g = b, zeros in what camera_create_ray leaves alone.  One lane per output FLOAT.
        {
            float4 *rec = reinterpret_cast<float4 *>(stage) + 2u * lane;
            rec[0] = make_float4(o.x, o.y, o.z, d.x);
            rec[1] = make_float4(d.y, d.z, w, __builtin_bit_cast(float, flags));
            wave_lds_fence();
            uint32_t *dst = reinterpret_cast<uint32_t *>(jobOut) + static_cast<size_t>(first) * 21u;
            const uint32_t total = cnt * 21u;
#pragma unroll
            for (uint32_t k = 0; k < 21u; ++k) {
                const uint32_t j = k * 64u + lane;
                if (j < total) {
                    const uint32_t ray = j / 21u, f = j - ray * 21u;
                    const float *r = stage + ray * 8u;
                    const bool retried = (__builtin_bit_cast(uint32_t, r[7]) & 1u) != 0u;
                    float v = 0.0f;                                  // dOdx (6-8), dDdx (12-14); dOdy / dDdy of first-try rays
                    if (f < 6u) v = r[f];                            // origin, dir
                    else if (f >= 18u) v = r[6];                     // weight r = g = b (the caller's initial weight is 1)
                    else if (retried && f >= 9u && f < 12u) v = r[f - 9u];    // dOdy = origin
                    else if (retried && f >= 15u) v = r[f - 12u];             // dDdy = dir
                    store_sys(dst + j, __builtin_bit_cast(uint32_t, v));
                }
            }
            wave_lds_fence();
        }
        // the batch's rows are released at system scope; then its flag says so to the render thread (which waits for every flag of
        // its tile: no counter, no last wave, nobody waits for anybody on the device)
        __threadfence_system();
        if (lane == 0) store_sys(tileFlags + jobSlot * kTileMaxBatches + batch, jobSeq);
    }
    for (int off = 32; off > 0; off >>= 1) { succ += __shfl_xor(succ, off, 64); vign += __shfl_xor(vign, off, 64); tir += __shfl_xor(tir, off, 64); }
    if (lane == 0) {
        if (slotRole) st->served[slot] = mine;
        // the counters of node_finish (zoic.cpp:1729-1732): one atomic per counter for the whole stay
        if (counters) {
            DeviceCounters *cs = counter_set(counters);
            if (succ) atomicAdd(&cs->succes, static_cast<unsigned long long>(succ));
            if (vign) atomicAdd(&cs->vignetted, static_cast<unsigned long long>(vign));
            if (tir) atomicAdd(&cs->tir, static_cast<unsigned long long>(tir));
        }
        __threadfence_system();
        if (atomicAdd(control + 1, 1u) == totalWaves - 1u) {   // the last wave out resets the control block and clears `alive`
            control[1] = 0u;
            __hip_atomic_store(control, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            store_uncached(reinterpret_cast<uint4 *>(header) + 1, make_uint4(0u, 0u, 0u, 0u));   // the last thing this launch does
        }
    }
}

}  // namespace

int launch_mailbox(const KolbTable &kolb, const ThinTable &thin, const BokehTables &bokeh, int model, int mode, void *d_mapped,
                   MailDeviceState *d_state, DeviceCounters *d_counters, uint32_t workerGroups, void *stream)
{
    const bool image = (model == 0 ? thin.useImage : kolb.useImage) != 0;
    const uint32_t ldsWords = (image && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
    const uint32_t groups = kMailSlotGroups + workerGroups;
    hipLaunchKernelGGL(mailbox_kernel, dim3(groups), dim3(kMailBlock), (kLutLdsWords + ldsWords + (kMailBlock / 64u) * kTileStageWords) * sizeof(float),
                       static_cast<hipStream_t>(stream), kolb, thin, bokeh, model, mode, static_cast<char *>(d_mapped), d_state, d_counters, ldsWords,
                       groups * (kMailBlock / 64u));
    return static_cast<int>(hipGetLastError());
}

}  // namespace zoic
