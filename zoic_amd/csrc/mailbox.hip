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
//   * A TILE (kind 1: n <= 65536 AtCameraInput rows in mapped host memory -> n AtCameraOutput rows in mapped host memory) is cut
//     into BATCHES (RAYTRACED 16 samples, THINLENS 64).  The slot's wave takes batch 0 itself, straight from the request -- a tile of
//     one batch involves nobody else -- and POSTS the rest as a job in device memory: a descriptor (three 16-byte chunks, each ending
//     in the tile's number, like a request line), ticket counters (generation << 32 | next batch; up to 32 partitions of about eight
//     batches: same-address atomics across the XCDs queue up at ~0.1 us each), a bit in the launch's 64-bit work mask; then it WAKES
//     as many WORKER waves of the launch (kTileWorkerGroups x 4, started with the first tile a camera sees) as there are batches
//     left, each through its own line in device memory with the partition it should start on (a thousand waves polling one word
//     slowed every poll down).  A woken worker draws a ticket and -- in the same round trip -- reads the descriptor; a descriptor
//     whose three numbers equal the ticket's generation is that tile's.  Lost wake-ups only cost time: the slot's wave keeps drawing
//     tickets of its own tile until every batch is handed out, and a worker that runs out of batches looks at the work mask for
//     other tiles.  A batch's 28-byte input rows arrive as coalesced dword loads across PCIe and are transposed through LDS, its
//     84-byte output rows leave as coalesced dword stores (whole rows, as zoic_create_rays_arnold writes them); the wave releases
//     them at system scope and then writes the batch's FLAG -- the tile's number -- in mapped host memory.  The render thread waits
//     for all flags of its tile: no counter, no last wave, nobody waits for anybody on the device.  No launch, no stream, no
//     synchronise: a 4096-sample tile is answered in 22-28 us (thin lens: 18 us) where a launch-based call took 76-122, and 16
//     render threads with 65536-sample tiles run at the PCIe rate of the rows (bench.py host_path.tile);
//   * the rays of a batch: kolb_wave_rays below (the tries of a ray side by side: a resident wave is alone on its SIMD and waits
//     for its longest dependent chain, not for its instruction count);
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
#include "kolb_listed_body.hpp"   // listed_group_round; kolb_pool_body.hpp: setup_ray, retry_direction (+ kolb_device.hpp: lens_sample, the traces, zoicDynLds)
#include "mailbox.hpp"
#include "thin_device.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

#ifdef ZOIC_TILE_TIMING
// timing builds only: [0..7] batches by number of rounds (7 = seven or more), [8] batches with a listed ray, [9] ticks in the listed part, [10] ticks in
// the rounds, [12] sum of rays open after round 0, [13] batches with more than 8 open after round 0, [14] their ticks in the rounds
__device__ unsigned long long g_tileDbg[16];
#endif

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

// (wave_lds_fence: kolb_listed_body.hpp)

// ---- RAYTRACED, zoic.cpp:1850-1964, for the rays a wave holds: the TRIES of a ray side by side ------------------------------------
// A resident wave is alone on its SIMD: what a call waits for is the instructions on its longest PATH -- every one of them, every taken
// branch, every scalar-load and LDS wait: ~10 cycles apiece -- not the number of lanes that execute them.  The reference's loop -- trace,
// and while the trace fails and tries <= 25 draw the next lens sample -- is such a path, up to 27 traces long; with one ray per lane all 64 rays of a wave waited for the
// unluckiest (measured: 39 us for 64 samples of a double Gauss at f/2, 10 us for ONE sample), and a tile waits for its unluckiest
// wave.  The tries of a ray are independent given its retry stream (try k >= 1 uses draws 2(k-1), 2(k-1)+1), so a wave holds
// kKolbBatch = 16 rays and runs ROUNDS:
//   * round 0: four lanes per ray evaluate tries 0 ... 3 side by side; the first success in try order wins (round 3's
//     listed_short, kolb_listed_body.hpp): ~90 % of the rays are done;
//   * every further round shares ALL 64 lanes among the rays still open (L = 64 / #open lanes each, up to 32): one or two open
//     rays have their remaining 23 tries evaluated AT ONCE.  A batch is through after two rounds, three at the outside -- whatever
//     its rays' try counts -- instead of up to seven;
//   * ray state travels through the wave's LDS stage (the set-up constants, the stream at the next try, the next try's number), so
//     a lane can take any ray in any round.
// Per try: the sample's own lens point (try 0: x-only translation, zoic.cpp:1914) or a draw from the ray's stream; interface 0
// first (most rejected tries die there), then the trace.  TIR bumps count for the tries before the winner only; try 26 hands out
// its (partial) state with weight 0 (zoic.cpp:1927 / 1951); dead pixels (outside the image circle all 27 tries are one) and
// retry-dead rays (dead_ray_end) end at try 0's failure, as in the batch kernels; GUARD (decision-safe FAST): a try with a decision
// inside a guard band, at or before the winner, sends the ray to the listed kernel's rule (listed_group_round: its tries side by
// side as well) once the rounds are over.  Every try is evaluated by the same device functions as everywhere else in the library: same bits as
// the batch kernels and as the reference's loop order (tests/test_tile_gpu.py, tests/test_boundary_gpu.py).
constexpr uint32_t kTileStageWords = 576;   // LDS stage per wave (THINLENS: 64 x 7 input dwords, then 64 x 8 record dwords)
// RAYTRACED waves carry a second area behind it for WIDE batches (64 rays, mailbox.hpp kTileWideSamples): [576, 1088) the batch's 64 records (its 64 x 7 input
// dwords on arrival), [1088, 1344) the 64 samples, [1344, 1408) the order in which the rays the first pass did not settle go through the 16-ray rounds
constexpr uint32_t kWideRecs = kTileStageWords, kWideSamp = kWideRecs + 512u, kWideOrder = kWideSamp + 256u, kTileStageWordsKolb = kWideOrder + 64u;
__host__ __device__ constexpr uint32_t stage_words(int model) { return model == 0 ? kTileStageWords : kTileStageWordsKolb; }
constexpr uint32_t kKolbBatch = kTileRaysRaytraced;   // rays per wave pass (mailbox.hpp)
// LDS stage of a wave (kTileStageWords dwords): [0, 128) the finished records, 8 dwords per ray (what the output stage reads);
// [128, 512) the rays' state, 24 dwords each; [384, 496) the batch's input rows on arrival (dead before the state is written);
// [512, 516) a round's outcome masks
constexpr uint32_t kStageState = 128, kStageInput = 384, kStageMasks = 512;
struct RayState {     // 24 dwords
    float sx, sy, lensx, lensy;                         // the sample
    uint32_t rng[4];                                    // the ray's retry stream at its first draw
    float o0x, o0y, maxScale, translation, sn, cs;      // RaySetup
    uint32_t flags;                                     // RaySetup::flags | dead pixel << 8 | lutEdge << 9 | try 0's sample finite << 10
    uint32_t nextTry, tirTally, pad[3];
    uint32_t rngNext[4];                                // the stream at the draws of try max(nextTry, 1): a round's lanes step on from here
};
static_assert(sizeof(RayState) == 96 && kStageState + kKolbBatch * 24u <= kStageMasks && kStageMasks + 4u <= kTileStageWords, "stage layout");
// (the state overlaps the input rows: they are in registers before the first state word is written)

// Evaluates the rays 0 ... cnt-1 whose samples sit in `samples` (one per ray, lane r < cnt holds ray r's) and leaves their records
// (ox oy oz dx dy dz weight flags, already negated, zoic.cpp:1960-1961) in stage[8 r ...].  rngOf(r): ray r's retry stream at its first
// draw.  succ / vign / tir: the calling lane's counters.
template <bool STRICT, bool GUARD, int NS, class RngOf>
__device__ __forceinline__ void kolb_wave_rays(const KolbTable &T, const BokehTables &B, const float2 *lutLds, const float *bokehLds, float *stage,
                                               float4 sample, uint32_t cnt, uint32_t lane, RngOf rngOf, uint32_t &succ, uint32_t &vign, uint32_t &tir)
{
    static_assert(!(STRICT && GUARD), "GUARD is a FAST mode");
    constexpr uint32_t kOut = static_cast<uint32_t>(kMaxTries) + 1u;   // tries of a ray that ran out (zoic.cpp:1927: tries <= 25)
    RayState *state = reinterpret_cast<RayState *>(stage + kStageState);
    float4 *records = reinterpret_cast<float4 *>(stage);
    // ---- set-up, once per ray, by the lane that holds its sample ------------------------------------------------------------------
    uint32_t listedMask = 0, deadEndMask = 0;   // wave-uniform: rays for the listed rule / retry-dead rays whose try 0 failed
    {
        const RaySetup rs = setup_ray<STRICT>(T, lutLds, sample.x, sample.y);
        bool finiteSample = true;
        if (rs.dead) {   // try 0 of a dead pixel shoots lens = (0, 0) whatever finite point the sampler returns (kolb_pool_body.hpp)
            const bool plainSample = (sample.z >= 0.0f) & (sample.z < 1.0f) & (sample.w >= 0.0f) & (sample.w < 1.0f) & !((sample.z == 0.5f) & (sample.w == 0.5f));
            if (!plainSample) { const V2 l = lens_sample<STRICT>(T, B, bokehLds, sample.z, sample.w); finiteSample = (fabsf(l.x) <= 3.0e38f) && (fabsf(l.y) <= 3.0e38f); }
        }
        if (lane < cnt) {
            RayState &q = state[lane];
            q.sx = sample.x; q.sy = sample.y; q.lensx = sample.z; q.lensy = sample.w;
            q.o0x = rs.o0x; q.o0y = rs.o0y; q.maxScale = rs.maxScale; q.translation = rs.translation; q.sn = rs.sn; q.cs = rs.cs;
            q.flags = rs.flags | (rs.dead ? 0x100u : 0u) | (rs.lutEdge ? 0x200u : 0u) | (finiteSample ? 0x400u : 0u);
            q.nextTry = 0u; q.tirTally = 0u;
            const Rng r0 = rngOf(lane);
            q.rng[0] = r0.x; q.rng[1] = r0.y; q.rng[2] = r0.z; q.rng[3] = r0.w;
            q.rngNext[0] = r0.x; q.rngNext[1] = r0.y; q.rngNext[2] = r0.z; q.rngNext[3] = r0.w;
        }
        if constexpr (GUARD) listedMask = static_cast<uint32_t>(__ballot(lane < cnt && T.useLUT && rs.lutEdge));   // the exit-pupil LUT's only discontinuity: the table's end
    }
    wave_lds_fence();
    uint32_t open = (cnt >= 32u ? 0xffffffffu : ((1u << cnt) - 1u)) & ~listedMask;   // wave-uniform: rays with tries to run
#ifdef ZOIC_TILE_TIMING
    const unsigned long long dbgT1 = wall_clock64();
    uint32_t dbgRounds = 0, dbgOpen1 = 0;
#endif
    // R = tries per LANE.  Measured with R = 2 where the unrolled FAST trace exists (lane t of a block: tries nextTry + t and nextTry + L + t,
    // round 0 covering tries 0 ... 7) [MI355X, profiles/ab_r05/tile_latency_v8.txt, _v9.txt]: a 4096-sample tile 29.1 -> 28.5 us, but a
    // 64-sample tile 17.3 -> 19.6 and the per-sample call 7.1 -> 8.6 us.  A lone wave is not waiting for its dependent instructions -- a wave64
    // VALU instruction issues every 4 cycles whether or not it depends on the one before -- it is paying for every instruction it issues
    // (and for every taken branch and scalar-load wait), so a second try in the same lane costs what it would cost in another round.  R = 1.
    // (R = 2 only in a later round that finds more than eight rays open -- four lanes a ray -- bought nothing either: tile_latency_v11.txt.)
    constexpr int R = 1;
    bool firstRound = true;
    while (open != 0u) {
#ifdef ZOIC_TILE_TIMING
        if (dbgRounds == 1u) dbgOpen1 = static_cast<uint32_t>(__builtin_popcount(open));
        ++dbgRounds;
#endif
        // ---- this round's lanes: L per open ray (4 in round 0, up to 32 / R for the stragglers) ----------------------------------------
        // (round 0 stays at four lanes per ray however few rays there are: a lane steps the ray's stream over the draws of the tries in
        // front of its own first, and 25 of those in front of a single sample's first round cost its median call 1.2 us for tries 95 % of
        // the rays never need)
        const uint32_t nOpen = static_cast<uint32_t>(__builtin_popcount(open));
        uint32_t L = firstRound ? 4u : 32u / static_cast<uint32_t>(R);
        while (L * nOpen > 64u) L >>= 1;
        firstRound = false;
        const uint32_t pos = lane / L, t = lane & (L - 1u), blockBase = lane & ~(L - 1u);
        // the pos-th open ray
        uint32_t ray = pos;
        if ((open & (open + 1u)) != 0u) {   // (holes in the mask: rays 0 ... nOpen-1 otherwise, round 0's case)
            uint32_t m = open;
            for (uint32_t i = 0; i < pos && m != 0u; ++i) m &= m - 1u;
            ray = m ? static_cast<uint32_t>(__builtin_ctz(m)) : 0u;
        }
        const bool mine = pos < nOpen;
        const RayState q = state[mine ? ray : static_cast<uint32_t>(__builtin_ctz(open))];
        const bool deadPixel = (q.flags & 0x500u) == 0x500u, retryDead = (q.flags & kRetryDeadBit) != 0u;
        const V3 o0{q.o0x, q.o0y, T.originShift};
        uint32_t k[R];                                  // this lane's tries
        bool valid[R], ok[R], near[R], pass0[R];
        uint32_t tirTry[R];
        float su[R], sv[R];                             // the unit-square point each try's lens sample is made from
        V3 o[R], d[R];
        // (1) the draws, one try after the other: the ray's stream is sequential (try k >= 1 uses draws 2 (k - 1), 2 (k - 1) + 1, zoic.cpp:1930)
        Rng rng{q.rngNext[0], q.rngNext[1], q.rngNext[2], q.rngNext[3]};
        uint32_t at = q.nextTry > 1u ? q.nextTry : 1u;  // the try whose draws the stream stands at
#pragma unroll
        for (int r = 0; r < R; ++r) {
            k[r] = q.nextTry + static_cast<uint32_t>(r) * L + t;
            valid[r] = mine && k[r] <= kOut;
            su[r] = q.lensx; sv[r] = q.lensy;           // try 0: the sample's own point (zoic.cpp:1870)
            if (valid[r] && k[r] != 0u) {
                for (; at < k[r]; ++at) { (void)xor128(rng); (void)xor128(rng); }
                su[r] = rng_unit(xor128(rng));
                sv[r] = rng_unit(xor128(rng));
                ++at;
            }
        }
        // (2) lens sample, direction, interface 0 of every try: straight-line code, so that the R tries of a lane overlap (what is not
        // valid is computed on whatever the registers hold and never looked at)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ok[r] = false; near[r] = false; tirTry[r] = 0u;
            o[r] = o0;
            const bool first = k[r] == 0u;
            V2 lens = lens_sample<STRICT>(T, B, bokehLds, su[r], sv[r]);
            if (!T.useLUT) d[r] = V3{(lens.x * T.rearAperture) - o0.x, (lens.y * T.rearAperture) - o0.y, T.dirZ};   // zoic.cpp:1873-1877 / 1882-1884
            else {
                // a dead pixel's try 0 shoots lens = (0, 0) whatever finite point the sampler returns (kolb_pool_body.hpp)
                const bool plainSample = (q.lensx >= 0.0f) & (q.lensx < 1.0f) & (q.lensy >= 0.0f) & (q.lensy < 1.0f) & !((q.lensx == 0.5f) & (q.lensy == 0.5f));
                if (first && deadPixel && plainSample) lens = V2{0.0f, 0.0f};
                lens.x *= q.maxScale; lens.y *= q.maxScale;
                lens.x += q.translation;
                const float ly = lens.y + q.translation;
                lens.y = first ? lens.y : ly;           // zoic.cpp:1914 translates the first sample in x only, 1933 the retries in both
                const float rx = lens.x * q.cs - lens.y * q.sn, ry = lens.x * q.sn + lens.y * q.cs;
                d[r] = V3{rx - o0.x, ry - o0.y, T.dirZ};
            }
            // interface 0 first (most rejected tries die there: a clip leaves (o, d) untouched and bumps nothing), then the trace
            if constexpr (STRICT) {
                bool inRange;
                pass0[r] = interface0_clear_strict_lean(T, o0, d[r], inRange);
                if (__builtin_expect(!inRange, 0)) pass0[r] = interface0_clear_strict(T, o0, d[r]);   // never seen: guarded roots
            } else pass0[r] = interface0_clear_fast<GUARD>(T.fsurf[0], o0, d[r], near[r]);
            pass0[r] = pass0[r] && valid[r]; near[r] = near[r] && valid[r];
            if (!valid[r]) d[r] = V3{0.0f, 0.0f, 1.0f};
            if constexpr (STRICT || NS == 0) {
                if (pass0[r] && !near[r]) {
                    if constexpr (STRICT) ok[r] = trace_lens_strict(T, o[r], d[r], tirTry[r]);
                    else ok[r] = trace_lens_fast_rolled(T, o[r], d[r], tirTry[r], GUARD ? &near[r] : nullptr);
                }
            }
        }
        if constexpr (!STRICT && NS > 0) {
            // FAST with a known interface count: ONE predicated, unrolled trace for all the wave's tries (27 instructions per interface and
            // the table words requested an interface ahead, against ~45 and three scalar-cache round trips per interface in the branchy
            // loop: a lone wave waits for every one of them).  A try that fails is left with the branchy trace's partial state -- what try
            // 26 hands out (zoic.cpp:1951-1961) -- so no second trace is ever needed.  Same FastHit / fast_refract: same bits.
            unsigned long long alive[R], tirMask[R], unsureMask[R];
            unsigned long long anyCand = 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) { alive[r] = __ballot(valid[r] && pass0[r] && !near[r]); anyCand |= alive[r]; }
            if (anyCand != 0ull) {
                unsigned long long cand[R];
#pragma unroll
                for (int r = 0; r < R; ++r) cand[r] = alive[r];
                trace_lens_fast_pred_keep<NS, GUARD, R>(kernarg_fast_surfaces(), o, d, alive, tirMask, unsureMask);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    ok[r] = mask_bit(alive[r], lane);
                    tirTry[r] = mask_bit(tirMask[r], lane) ? 1u : 0u;
                    if constexpr (GUARD) near[r] = near[r] || mask_bit(unsureMask[r] & cand[r], lane);
                }
            }
        }
        // ---- the ray's decision, in try order, by the L lanes of its block: try nextTry + i sits in lane i % L, slot i / L --------------------
        const unsigned long long blockMask = (L >= 32u ? 0xffffffffull : ((1ull << L) - 1ull));
        uint32_t okW = 0, nearW = 0, tirW = 0;   // bit i: try nextTry + i (R x L <= 32)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            okW |= static_cast<uint32_t>((__ballot(valid[r] && ok[r] && !near[r]) >> blockBase) & blockMask) << (static_cast<uint32_t>(r) * L);
            nearW |= static_cast<uint32_t>((__ballot(valid[r] && near[r]) >> blockBase) & blockMask) << (static_cast<uint32_t>(r) * L);
            tirW |= static_cast<uint32_t>((__ballot(valid[r] && tirTry[r] != 0u) >> blockBase) & blockMask) << (static_cast<uint32_t>(r) * L);
        }
        const uint32_t winner = okW ? static_cast<uint32_t>(__builtin_ctz(okW)) : 32u;         // lowest try that got through
        const uint32_t firstNear = nearW ? static_cast<uint32_t>(__builtin_ctz(nearW)) : 32u;
        const bool firstFailed = q.nextTry == 0u && (okW & 1u) == 0u && (nearW & 1u) == 0u;       // try 0 failed, decided
        // 0 open still, 1 finished by try nextTry + holder, 2 listed, 3 retry-dead end
        uint32_t outcome = 0, holder = 0, tirAdd = 0;
        if (GUARD && firstNear < 32u && firstNear < winner) outcome = 2;          // too close to call before anything got through
        else if (firstFailed && deadPixel) { outcome = 1; holder = 0; tirAdd = (1u + kOut) * (tirW & 1u); }    // 26 more identical failures
        else if (firstFailed && retryDead) { outcome = 3; tirAdd = tirW & 1u; }    // tries 1 ... 26 die at interface 0
        else if (winner < 32u) { outcome = 1; holder = winner; tirAdd = static_cast<uint32_t>(__builtin_popcount(tirW & ((1u << winner) - 1u))); }   // only the tries the reference ran
        else {
            tirAdd = static_cast<uint32_t>(__builtin_popcount(tirW));
            if (q.nextTry + static_cast<uint32_t>(R) * L > kOut) { outcome = 1; holder = kOut - q.nextTry; }   // try 26 failed as well: ITS partial state, weight 0
        }
        const uint32_t tally = q.tirTally + tirAdd;
        if (mine && outcome == 1u && t == (holder & (L - 1u))) {
            V3 oh = o[0], dh = d[0];
            uint32_t kh = k[0];
#pragma unroll
            for (int r = 1; r < R; ++r) { if (holder / L == static_cast<uint32_t>(r)) { oh = o[r]; dh = d[r]; kh = k[r]; } }
            const uint32_t tries = (firstFailed && deadPixel) ? kOut : kh;
            const bool vignetted = tries > static_cast<uint32_t>(kMaxTries);
            float w = vignetted ? 0.0f : 1.0f;                                  // zoic.cpp:1951-1957 (try 26 that got through as well)
            if (T.exposureOn) w *= T.exposureMul;                               // zoic.cpp:1981-1987
            records[2u * ray] = make_float4(oh.x * -1.0f, oh.y * -1.0f, oh.z * -1.0f, dh.x * -1.0f);   // zoic.cpp:1960-1961
            records[2u * ray + 1u] = make_float4(dh.y * -1.0f, dh.z * -1.0f, w, __builtin_bit_cast(float, (tries > 0 ? 1u : 0u) | (tries << 1) | ((q.flags & 1u) << 6)));
            tir += tally;
            vign += vignetted ? 1u : 0u; succ += vignetted ? 0u : 1u;   // (written as selects: `if ... ++a; else ++b;` became an indexed counter in scratch)
        }
        if (mine && outcome == 0u && t == 0u) { state[ray].nextTry = q.nextTry + static_cast<uint32_t>(R) * L; state[ray].tirTally = tally; }
        // ... and the block's last lane stands behind the draws of the round's last try: where the next round's lanes step on from
        if (mine && outcome == 0u && t == L - 1u) { state[ray].rngNext[0] = rng.x; state[ray].rngNext[1] = rng.y; state[ray].rngNext[2] = rng.z; state[ray].rngNext[3] = rng.w; }
        if (mine && outcome == 3u && t == 0u) state[ray].tirTally = tally;
        // the wave's view of who is still open: every block's first lane ORs its ray's bit into the round's masks
        uint32_t *masks = reinterpret_cast<uint32_t *>(stage + kStageMasks);
        if (lane < 3u) masks[lane] = 0u;
        wave_lds_fence();
        if (mine && t == 0u && (outcome == 0u || outcome >= 2u)) atomicOr(masks + (outcome == 0u ? 0 : outcome - 1u), 1u << ray);
        wave_lds_fence();
        const uint32_t stillOpen = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(masks[0]))), nowListed = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(masks[1]))),
                       nowDeadEnd = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(masks[2])));
        listedMask |= nowListed; deadEndMask |= nowDeadEnd;
        open = stillOpen;
        wave_lds_fence();
    }
#ifdef ZOIC_TILE_TIMING
    const unsigned long long dbgT2 = wall_clock64();
#endif
    // ---- retry-dead rays whose try 0 failed: the state of the last draw, one lane each ------------------------------------------------------
    if (deadEndMask != 0u) {
        uint32_t m = deadEndMask;
        for (uint32_t i = 0; i < lane && m != 0u; ++i) m &= m - 1u;
        if (lane < static_cast<uint32_t>(__builtin_popcount(deadEndMask)) && m != 0u) {
            const uint32_t ray = static_cast<uint32_t>(__builtin_ctz(m));
            const RayState q = state[ray];
            RaySetup rs;
            rs.o0x = q.o0x; rs.o0y = q.o0y; rs.maxScale = q.maxScale; rs.translation = q.translation; rs.sn = q.sn; rs.cs = q.cs; rs.flags = q.flags & (1u | kRetryDeadBit);
            rs.dead = false; rs.lutEdge = false;
            const DeadRayEnd e = dead_ray_end<STRICT>(T, B, bokehLds, rs, Rng{q.rng[0], q.rng[1], q.rng[2], q.rng[3]});
            records[2u * ray] = make_float4(e.o.x * -1.0f, e.o.y * -1.0f, e.o.z * -1.0f, e.d.x * -1.0f);
            records[2u * ray + 1u] = make_float4(e.d.y * -1.0f, e.d.z * -1.0f, e.w, __builtin_bit_cast(float, (e.tries > 0 ? 1u : 0u) | (e.tries << 1) | ((q.flags & 1u) << 6)));
            tir += q.tirTally;
            vign += e.nanDraw ? 0u : 1u; succ += e.nanDraw ? 1u : 0u;
        }
    }
    // ---- rays with a decision too close to call: the listed kernel's rule (kolb_listed_body.hpp), the tries of a ray side by side ----------
    // in the G lanes the wave can spare per listed ray (one listed ray: all 27 tries at once).  What a batch with such a ray waits for is
    // then ONE chain -- the reference's set-up, then try 0 in the reference's arithmetic up to the stop next to the FAST-guarded retries --
    // where one lane used to run set-up, try 0 and every retry one after the other (+12 us on its batch, measured: the batches
    // a 4096-sample tile waited for).  Same device function as the listed kernel's short lists: same bits whatever evaluates the ray.
    if constexpr (GUARD) {
        if (listedMask != 0u) {
            const uint32_t nListed = static_cast<uint32_t>(__builtin_popcount(listedMask));
            uint32_t G = 32u;
            while (G * nListed > 64u) G >>= 1;
            const uint32_t pos = lane / G, j = lane & (G - 1u), shift = lane & ~(G - 1u);
            uint32_t m = listedMask;
            for (uint32_t i = 0; i < pos && m != 0u; ++i) m &= m - 1u;
            const bool have = pos < nListed && m != 0u;
            const uint32_t ray = have ? static_cast<uint32_t>(__builtin_ctz(m)) : static_cast<uint32_t>(__builtin_ctz(listedMask));
            const RayState q = state[ray];
            const float4 s4 = make_float4(q.sx, q.sy, q.lensx, q.lensy);
            const RaySetup rs = setup_ray<true>(T, lutLds, s4.x, s4.y);
            Rng rng{q.rng[0], q.rng[1], q.rng[2], q.rng[3]};
            for (uint32_t a = 1; a < j; ++a) { (void)xor128(rng); (void)xor128(rng); }   // lane j >= 1 starts at draw 2 (j - 1)
            const bool deadPixel = listed_dead_pixel(T, B, bokehLds, rs, s4);
            bool done = !have;
            for (uint32_t round = 0; round * G <= kOut; ++round) {
                bool emit; V3 o, d; float w; uint32_t flags;
                done = listed_group_round<NS>(T, B, bokehLds, s4, rs, deadPixel, rng, G, j, shift, round, done, emit, o, d, w, flags, succ, vign, tir);
                if (emit) {
                    records[2u * ray] = make_float4(o.x, o.y, o.z, d.x);
                    records[2u * ray + 1u] = make_float4(d.y, d.z, w, __builtin_bit_cast(float, flags));
                }
                if (__ballot(!done) == 0ull) break;
            }
        }
    }
#ifdef ZOIC_TILE_TIMING
    if (lane == 0u && cnt > 1u) {
        const unsigned long long dbgT3 = wall_clock64();
        atomicAdd(&g_tileDbg[dbgRounds < 7u ? dbgRounds : 7u], 1ull);
        if (listedMask != 0u) { atomicAdd(&g_tileDbg[8], 1ull); atomicAdd(&g_tileDbg[9], dbgT3 - dbgT2); }
        atomicAdd(&g_tileDbg[10], dbgT2 - dbgT1);
        atomicAdd(&g_tileDbg[12], static_cast<unsigned long long>(dbgOpen1));
        if (dbgOpen1 > 8u) { atomicAdd(&g_tileDbg[13], 1ull); atomicAdd(&g_tileDbg[14], dbgT2 - dbgT1); }
    }
#endif
    wave_lds_fence();
}

// ---- WIDE batches: 64 rays of a large tile (mailbox.hpp kTileWideSamples) -----------------------------------------------------------------
// Pass 1, one ray per lane: the reference's first try (set-up, the sample's own lens point, x-only translation, the trace) for every ray that is
// nothing special -- inside the LUT, not a dead pixel, not at the LUT's end (GUARD), a plain sample in [0,1)^2 off the disk mapping's centre.  A
// ray whose first try gets through (GUARD: outside every guard band) is FINISHED: origin, direction, weight, flag word 0 -- the bits
// kolb_wave_rays' round 0 gives it (same set-up, sampler and trace arithmetic; a first success wins whatever the speculative tries 1-3 did) and
// the batch kernels' phase A.  Everything else -- failed or undecided first tries, dead pixels, hostile samples -- goes through kolb_wave_rays
// 16 rays at a time, FROM SCRATCH (try 0 again: the same failure, the same TIR bump, counted there), so that no rule exists twice: the caller's
// group loop (ONE call site of kolb_wave_rays for both batch shapes: inlined twice, the FAST kernels needed 262 VGPRs -- more than two waves per SIMD
// have -- where they need 226 with one).
template <bool STRICT, bool GUARD, int NS>
__device__ __forceinline__ uint32_t kolb_wide_first_pass(const KolbTable &T, const BokehTables &B, const float2 *lutLds, const float *bokehLds, float *stage,
                                                         float4 sample, uint32_t cnt, uint32_t lane, uint32_t &succ)
{
    float4 *recs = reinterpret_cast<float4 *>(stage + kWideRecs);
    float4 *samp = reinterpret_cast<float4 *>(stage + kWideSamp);
    uint32_t *order = reinterpret_cast<uint32_t *>(stage + kWideOrder);
    const bool have = lane < cnt;
    const RaySetup rs = setup_ray<STRICT>(T, lutLds, sample.x, sample.y);
    const float u = sample.z, v = sample.w;
    const bool plain = (u >= 0.0f) & (u < 1.0f) & (v >= 0.0f) & (v < 1.0f) & !((u == 0.5f) & (v == 0.5f));
    const bool cand = have && plain && !rs.dead && (rs.flags & 1u) == 0u && !(GUARD && T.useLUT && rs.lutEdge);
    V2 lens = lens_sample<STRICT>(T, B, bokehLds, u, v);                     // zoic.cpp:1870
    V3 o{rs.o0x, rs.o0y, T.originShift}, d;
    if (!T.useLUT) d = V3{(lens.x * T.rearAperture) - o.x, (lens.y * T.rearAperture) - o.y, T.dirZ};   // zoic.cpp:1873-1877
    else {                                                                   // zoic.cpp:1913-1924: the first sample is translated in x only
        lens.x *= rs.maxScale; lens.y *= rs.maxScale;
        lens.x += rs.translation;
        const float rx = lens.x * rs.cs - lens.y * rs.sn, ry = lens.x * rs.sn + lens.y * rs.cs;
        d = V3{rx - o.x, ry - o.y, T.dirZ};
    }
    bool ok = false, unsure = false;
    if constexpr (!STRICT && NS > 0) {
        unsigned long long tirMask, unsureMask;
        const unsigned long long alive = trace_lens_fast_pred<NS, GUARD>(kernarg_fast_surfaces(), o, d, __ballot(cand), tirMask, unsureMask);
        ok = mask_bit(alive, lane);
        if constexpr (GUARD) unsure = mask_bit(unsureMask, lane);
    } else if (cand) {
        uint32_t tirTry = 0;   // (a first try that is totally reflected is not finished here: its bump is counted where the ray ends)
        if constexpr (STRICT) ok = trace_lens_strict(T, o, d, tirTry);
        else ok = trace_lens_fast_rolled(T, o, d, tirTry, GUARD ? &unsure : nullptr);
    }
    const bool done = cand && ok && !unsure;
    if (done) {
        float w = 1.0f;
        if (T.exposureOn) w *= T.exposureMul;                               // zoic.cpp:1981-1987
        recs[2u * lane] = make_float4(o.x * -1.0f, o.y * -1.0f, o.z * -1.0f, d.x * -1.0f);   // zoic.cpp:1960-1961
        recs[2u * lane + 1u] = make_float4(d.y * -1.0f, d.z * -1.0f, w, __builtin_bit_cast(float, 0u));
        succ += 1u;
    }
    samp[lane] = sample;
    const unsigned long long pending = __ballot(have && !done);
    if (have && !done) order[mask_rank(pending)] = lane;
    wave_lds_fence();
    return static_cast<uint32_t>(__popcll(pending));   // the rays left for kolb_wave_rays: order[0 .. n), their samples in samp[]
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
__device__ __forceinline__ uint4 load_dev4(const uint4 *p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_dev4(uint4 *p, uint4 q)
{
    const u32x4 v = {q.x, q.y, q.z, q.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
// 16 bytes of mapped host memory in one instruction, system scope (never from a GPU cache)
__device__ __forceinline__ uint4 load_sys4(const uint4 *p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t first_lane(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ unsigned long long first_lane64(unsigned long long v)
{
    return (static_cast<unsigned long long>(first_lane(static_cast<uint32_t>(v >> 32))) << 32) | first_lane(static_cast<uint32_t>(v));
}


// model: ZOIC_THINLENS 0 / ZOIC_RAYTRACED 1 (zoic_amd.h); mode: 0 STRICT, 1 FAST decision-safe, 2 FAST unchecked.
// Waves 0 .. 63 of the launch are the SLOT waves (wave w owns slot w: a render thread never waits for another thread's ray --
// one wave for all slots measured 8 us per call with one calling thread, 29 us with four), the waves behind them the tile WORKERS.
// st->control (device memory): [0] exit flag (set by wave 0: stop request / 1 ms without a call / 50 ms of life), [1] waves that
// have left, [2..3] wall-clock time of the last call any wave answered, [4..5] the work mask.
// One kernel per (lens model, precision mode): a resident wave is alone on its SIMD and waits for its instruction fetches like for
// everything else -- with all six combinations in one kernel (138 KB of code against a 64 KB instruction cache) every pass missed.
// NS: the lens's interface count where the FAST modes have an unrolled trace for it (7 ... 12, as the batch kernels), 0 otherwise.
template <int MODEL, int MODE, int NS>
__global__ __launch_bounds__(kMailBlock) void mailbox_kernel(const KolbTable T, const ThinTable Th, const BokehTables B, char *mapped,
                                                             MailDeviceState *st, DeviceCounters *counters, uint32_t ldsWords, uint32_t totalWaves)
{
    constexpr uint32_t kRays = 64u;   // the LARGEST batch of a tile (tile_rays_per_batch, mailbox.hpp: THINLENS always, RAYTRACED from kTileWideSamples samples on; 16 otherwise)
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
    float *stage = zoicDynLds + kLutLdsWords + ldsWords + (threadIdx.x >> 6) * stage_words(MODEL);
    MailHeader *header = reinterpret_cast<MailHeader *>(mapped);
    const MailRequest *requests = reinterpret_cast<const MailRequest *>(mapped + kMailRequestsOffset);
    MailReply *replies = reinterpret_cast<MailReply *>(mapped + kMailRepliesOffset);
    uint32_t *tileFlags = reinterpret_cast<uint32_t *>(mapped + kMailTileFlagsOffset);   // [slot][batch]: the tile's sequence number when the batch's rows are complete
    uint32_t *control = st->control;
    unsigned long long *workMask = reinterpret_cast<unsigned long long *>(control + 4);   // control[4..7] = {mask lo, mask hi, turn, exit}: ONE load per worker poll
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
    constexpr uint32_t kNoSlot = 0xffu;
    const uint32_t workerWaves = totalWaves - kMailSlots;
    uint32_t curSlot = kNoSlot, curPart = 0, partsTried = 0;   // the tile / ticket partition this wave draws from
    bool scanMask = false, leaving = false;                    // worker role: look at the work mask; the exit flag has been seen
    uint32_t idlePolls = 0;
#ifdef ZOIC_TILE_TIMING
    uint32_t lastWasHint = 0;
#endif
    unsigned long long lastWake = 0;                           // worker role: the wake word last acted on (a posted one is never 0)
    while (watched) {
        if (wall_clock64() - start > kMailHardLifeTicks) break;   // safety net (never seen): the host reports a tile that is never answered
        uint32_t work = 0;      // 1: one sample (slot role), 2: one 64-sample batch of a tile
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        Rng rng{1u, 2u, 3u, 4u};
        uint32_t seq = 0, jobSlot = 0, batch = 0, jobN = 0, jobSeq = 0, jobRays = kRays, jobRows = 0, jobIns = 0;
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
                    // everybody out: the slot waves read [0] with their request line, every worker its own wake line
                    if (lane == 0) __hip_atomic_store(control, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    for (uint32_t i = lane; i < workerWaves; i += 64u) store_dev(&st->wake[i].word, (static_cast<unsigned long long>(static_cast<uint32_t>(now) | 1u) << 32) | 0x10000ull);
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
                jobRows = lane_word(line.y, 2) & 1u;   // what to write at `out`: AtCameraOutput rows / zoic_ray records (mailbox.hpp)
                jobIns = (lane_word(line.y, 2) >> 1) & 1u;   // what stands at `in`: AtCameraInput rows / (sx, sy, lensx, lensy) samples
                if (jobN == 0u) continue;       // (the host never posts an empty tile)
                work = 2;                       // batch 0 is this wave's, straight from the request: a tile of one batch involves nobody else
                jobRays = tile_rays_per_batch(MODEL == 0, jobN);
                if (jobN > jobRays) {
                    // the rest is POSTED for the workers: descriptor (three 16-byte chunks, each ending in the tile's number, like a
                    // request line), the ticket counters (partition 0 starts behind this wave's batch), the slot's bit; then as many
                    // workers are woken as there are batches left, each with the partition it should start on.  Whoever draws a ticket
                    // of generation `seq` finds this descriptor.
                    const uint32_t batches = (jobN + jobRays - 1u) / jobRays;
                    uint32_t parts = batches / 8u;   // about eight batches per counter: same-address atomics queue up (~0.1 us each across the XCDs)
                    parts = parts < 1u ? 1u : (parts > kTileParts ? kTileParts : parts);
                    // the first worker to wake: the next stretch of the wake lines, so that tiles of different render threads wake different
                    // waves (the returning atomic is issued HERE and comes back behind the fence below: rounds 5's first versions waited for it
                    // where the wake lines are written, ~1 us of every tile's hand-off; a stretch computed from the slot and the tile's number
                    // instead cost nothing and made the tiles of four threads share their workers)
                    const uint32_t wakeN = workerWaves == 0u ? 0u : (batches - 1u < workerWaves ? batches - 1u : workerWaves);
                    uint32_t wakeAt = 0;
                    if (lane == 0 && wakeN != 0u) wakeAt = __hip_atomic_fetch_add(control + 8, wakeN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0) {
                        uint4 *J = reinterpret_cast<uint4 *>(st->jobs + slot);
                        store_dev4(J, make_uint4(lane_word(line.x, 0), lane_word(line.y, 0), jobN, seq));
                        store_dev4(J + 1, make_uint4(lane_word(line.x, 1), lane_word(line.y, 1), lane_word(line.z, 1), seq));
                        store_dev4(J + 2, make_uint4(lane_word(line.x, 2), batches, parts | (jobRows << 8) | (jobIns << 9) | (jobRays << 16), seq));
                        {   // the partitions that hold a batch at all (ceil(batches / parts) per partition can leave the last ones empty)
                            const uint32_t per0 = (batches + parts - 1u) / parts, live = (batches + per0 - 1u) / per0;
                            store_dev(&st->tickets[slot].partMask, live >= 32u ? 0xffffffffu : ((1u << live) - 1u));
                        }
#ifdef ZOIC_TILE_TIMING
                        store_dev(reinterpret_cast<unsigned long long *>(st->jobs + slot) + 6, static_cast<unsigned long long>(now));
#endif
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    }
                    if (lane < parts) store_dev(&st->tickets[slot].part[lane].next, (static_cast<unsigned long long>(seq) << 32) | (lane == 0u ? 1ull : 0ull));
                    // (the descriptor is fenced in front of the counters: whoever draws a ticket of this generation MUST find it, or the
                    // ticket -- a batch -- is lost.  The counters and the bit are NOT fenced in front of the wake lines: a woken worker acts on
                    // them a poll and an atomic's round trip later, and one that still draws a stale ticket only goes back to sleep -- a lost
                    // wake-up costs time, the slot's wave hands its tile's batches out itself in the end.  The fence that used to stand here cost
                    // every tile ~0.5 us of hand-off.)
                    if (lane == 0) (void)__hip_atomic_fetch_or(workMask, 1ull << slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (wakeN != 0u) {
                        const uint32_t at = first_lane(wakeAt);
                        const unsigned long long tag = (static_cast<unsigned long long>(static_cast<uint32_t>(now) | 1u) << 32) | slot;
                        // the i-th woken worker is meant for batch i + 1: it starts on that batch's partition (partition 0 is one batch
                        // short -- this wave's -- and every partition gets exactly as many workers as it has batches)
                        const uint32_t per = (batches + parts - 1u) / parts;
                        for (uint32_t i = lane; i < wakeN; i += 64u) store_dev(&st->wake[(at + i) % workerWaves].word, tag | (static_cast<unsigned long long>((i + 1u) / per) << 8));
                    }
                    ownJob = true; curPart = 0; partsTried = 0;
                }
            }
        }
        if (work == 0u) {   // ---- draw a batch: a slot's wave from its own tile, a worker from the tile it was woken for (then any posted one)
            if (slotRole) jobSlot = slot;
            else if (curSlot != kNoSlot) jobSlot = curSlot;
            else {
                if (scanMask) {   // nothing left of the tile this wave was on: any other posted tile?
                    // A slot's bit says "batches left to hand out"; its partition word says WHERE.  Both are loads: a wave that runs dry costs the
                    // tiles still at work no ticket atomic until it knows a partition that has a batch for it, and the waves that run dry
                    // together (a tile's 255 workers, all at once) spread over the slots and partitions by their own number -- they used to
                    // fall on partition 0 of the first posted slot one same-address atomic after the other, in front of that tile's own workers.
                    unsigned long long mask = first_lane64(load_dev(workMask));
                    bool found = false;
                    for (uint32_t tries = 0; mask != 0ull && tries < 4u && !found; ++tries) {
                        // the k-th posted slot, k by the wave's own number: every posted tile gets its share of the waves that run dry
                        // ("the first set bit at or behind waveId mod 64" sent three quarters of them to the lowest slot of sixteen)
                        uint32_t k = (waveId + tries * 7u) % static_cast<uint32_t>(__builtin_popcountll(mask));
                        unsigned long long mm = mask;
                        for (; k != 0u; --k) mm &= mm - 1ull;
                        const uint32_t cand = static_cast<uint32_t>(__builtin_ctzll(mm));
                        const uint32_t live = first_lane(load_dev(&st->tickets[cand].partMask));
                        if (live != 0u) {
                            const uint32_t q = (waveId >> 6) & 31u;
                            const uint32_t rotp = q ? ((live >> q) | (live << (32u - q))) : live;
                            curSlot = cand; curPart = (static_cast<uint32_t>(__builtin_ctz(rotp)) + q) & 31u; partsTried = 0;
                            found = true;
                        } else mask &= ~(1ull << cand);   // (its last batches are being drawn: nothing for this wave there)
                    }
                    if (found) continue;
                    scanMask = false;
                    if (leaving) break;   // exit flag: only with nothing left to hand out
                }
                const unsigned long long wk = first_lane64(load_dev(&st->wake[waveId - kMailSlots].word));
                if (wk == lastWake) {
                    // (a hint written behind the exit word hides it: every 16th idle poll looks at the launch's exit flag itself)
                    if ((++idlePolls & 15u) == 0u && first_lane(load_dev(control)) != 0u) { leaving = true; scanMask = true; continue; }
                    __builtin_amdgcn_s_sleep(6);   // ~0.15 us: every worker polls its own line, but a thousand of them add up on the fabric
                    continue;
                }
                lastWake = wk;
                if ((wk >> 16) & 1ull) { leaving = true; scanMask = true; continue; }
                curSlot = static_cast<uint32_t>(wk) & 63u; curPart = static_cast<uint32_t>(wk >> 8) & (kTileParts - 1u); partsTried = 0;
#ifdef ZOIC_TILE_TIMING
                lastWasHint = 2u;
#endif
                continue;
            }
            // the ticket and -- speculatively, in the same round trip -- the descriptor
            const uint4 *J = reinterpret_cast<const uint4 *>(st->jobs + jobSlot);
            unsigned long long t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(&st->tickets[jobSlot].part[curPart].next, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint4 c0, c1, c2;
            load_dev4x3(J, c0, c1, c2);
            t = first_lane64(t);
            const uint32_t gen = static_cast<uint32_t>(t >> 32);
            if (first_lane(c0.w) != gen || first_lane(c1.w) != gen || first_lane(c2.w) != gen) {
                // read before the poster's stores were visible -- or a ticket of a tile that is long done: once more, behind the ticket
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                load_dev4x3(J, c0, c1, c2);
            }
            const uint32_t batches = first_lane(c2.y), parts = first_lane(c2.z) & 0xffu;
            jobRays = first_lane(c2.z) >> 16; jobRows = (first_lane(c2.z) >> 8) & 1u; jobIns = (first_lane(c2.z) >> 9) & 1u;
            const bool sameTile = first_lane(c0.w) == gen && first_lane(c1.w) == gen && first_lane(c2.w) == gen;
            const uint32_t per = sameTile ? (batches + parts - 1u) / parts : 1u;
            const uint32_t lo = curPart * per, hi = lo + per < batches ? lo + per : batches;   // the partition's batches
            batch = lo + static_cast<uint32_t>(t);
            if (!sameTile || curPart >= parts || batch >= hi) {
                // a stale ticket (the tile has been handed out: its counters belong to nobody, or to the next tile -- whose descriptor
                // then differs) or a partition that is empty: on to the next partition; after all of them, the tile is done with
                // (measured with four render threads: a worker used to walk the partitions one ticket atomic and one look at the work mask at a
                // time -- ~1.7 us each, up to 31 of them behind every batch it had finished -- and a new tile's batches started 83 us late)
                ++partsTried;
                bool gone = !sameTile || partsTried > parts + 2u;
                uint32_t live = 0;
                if (!gone) { live = first_lane(load_dev(&st->tickets[jobSlot].partMask)); gone = live == 0u; }
                if (gone) {
                    if (slotRole) ownJob = false; else { curSlot = kNoSlot; scanMask = true; }
                } else {   // a partition that still has batches (one load: no ticket atomic is spent on an empty one)
                    const uint32_t q = (waveId + partsTried) & 31u;   // (spread by the wave's own number: the waves that find a partition empty find it together)
                    const uint32_t rotp = q ? ((live >> q) | (live << (32u - q))) : live;
                    curPart = (static_cast<uint32_t>(__builtin_ctz(rotp)) + q) & 31u;
                }
                continue;
            }
            // the LAST batch of a partition clears the partition's bit, the last partition the slot's bit in the work mask: both by waves that hold
            // a valid batch, so the tile cannot complete (and the slot post another) before they are done
            if (batch + 1u == hi && lane == 0) {
                const uint32_t others = ~(1u << curPart);
                if ((__hip_atomic_fetch_and(&st->tickets[jobSlot].partMask, others, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & others) == 0u)
                    (void)__hip_atomic_fetch_and(workMask, ~(1ull << jobSlot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            jobSeq = gen;
            jobN = first_lane(c0.z);
            jobIn = (static_cast<unsigned long long>(first_lane(c0.y)) << 32) | first_lane(c0.x);
            jobOut = (static_cast<unsigned long long>(first_lane(c1.y)) << 32) | first_lane(c1.x);
            jobBase = (static_cast<unsigned long long>(first_lane(c2.x)) << 32) | first_lane(c1.z);
            work = 2;
        }
#ifdef ZOIC_TILE_TIMING
        const unsigned long long tt0 = wall_clock64();
        if (lastWasHint) --lastWasHint;   // 1: this batch is the first after a wake-up
#endif

        // ---- the pass's samples: the slot's one sample in lane 0, or 64 consecutive rows of the tile --------------------------
        // (lane r holds ray r's sample; RAYTRACED shares the wave's lanes among the rays' tries afterwards, kolb_wave_rays)
        // where the input rows land in the wave's stage: THINLENS at its start, a 16-ray RAYTRACED batch in the state area, a WIDE one (64 rays) in its record area
        const bool wide = MODEL != 0 && work == 2u && jobRays > kKolbBatch;
        const uint32_t kIn = MODEL == 0 ? 0u : (wide ? kWideRecs : kStageInput);
        bool active = lane == 0u;
        uint32_t first = 0, cnt = 1;
        unsigned long long rayBase = 0;
        if (work == 2u) {
            first = batch * jobRays;
            cnt = jobN - first < jobRays ? jobN - first : jobRays;
            if (jobIns != 0u) {   // ZOIC_TILE_INPUTS_SAMPLES: 16 bytes a sample, the lane's own -- one instruction, no transpose
                active = lane < cnt;
                const uint4 q = load_sys4(reinterpret_cast<const uint4 *>(jobIn) + first + (active ? lane : 0u));
                s = make_float4(__builtin_bit_cast(float, q.x), __builtin_bit_cast(float, q.y), __builtin_bit_cast(float, q.z), __builtin_bit_cast(float, q.w));
            } else {
            // AtCameraInput rows are 7 dwords (sx sy dsx dsy lensx lensy relative_time): lane l fetches dword k * 64 + l of the batch's
            // 7 * cnt -- whole 256-byte runs per instruction across PCIe -- and picks its row out of LDS
            const uint32_t *src = reinterpret_cast<const uint32_t *>(jobIn) + static_cast<size_t>(first) * 7u;
            const uint32_t total = cnt * 7u;
            constexpr uint32_t kLoads = (kRays * 7u + 63u) / 64u;
            uint32_t w7[kLoads];
            // (wave-uniform guards: a 16-ray batch issues two loads, not seven -- every one of them is a PCIe read)
#pragma unroll
            for (uint32_t k = 0; k < kLoads; ++k) { const uint32_t j = k * 64u + lane; w7[k] = 0u; if (k * 64u < total) w7[k] = load_sys(src + (j < total ? j : total - 1u)); }
#pragma unroll
            for (uint32_t k = 0; k < kLoads; ++k) { const uint32_t j = k * 64u + lane; if (k * 64u < total && j < total) stage[kIn + j] = __builtin_bit_cast(float, w7[k]); }
            wave_lds_fence();
            active = lane < cnt;
            const uint32_t row = kIn + (active ? lane : 0u) * 7u;
            s = make_float4(stage[row], stage[row + 1u], stage[row + 4u], stage[row + 5u]);
            wave_lds_fence();   // ... before the records go into the same words
            }
            rayBase = jobBase + first;
        }

#ifdef ZOIC_TILE_TIMING
        const unsigned long long tt1 = wall_clock64();
#endif
        // ---- the rays (ONE site for both kinds of work) -------------------------------------------------------------------------
        if constexpr (MODEL == 0) {   // THINLENS, zoic.cpp:1771-1846: one ray per lane
            V3 o{0.0f, 0.0f, 0.0f}, d{0.0f, 0.0f, 0.0f};
            float w = 0.0f;
            uint32_t tries = 0;
            if (active) {
                ThinRay r;
                if (work == 2u) {   // a tile's ray = the batch kernels' ray: its own stream, seeded at its first redraw; FAST where they run FAST
                    Rng q{1u, 2u, 3u, 4u};
                    if (MODE != 0 && Th.useDof && Th.ovDistance > 0.0f) r = thin_ray_fast_vignet(Th, B, bokehLds, s, q, [&] { q = rng_for_ray(Th.seed, rayBase + lane); });
                    else r = thin_ray_strict(Th, B, bokehLds, s, q, [&] { q = rng_for_ray(Th.seed, rayBase + lane); });
                } else {            // one sample: the calling tid's stream, the reference's arithmetic in every precision mode
                    Rng q = rng;
                    r = thin_ray_strict(Th, B, bokehLds, s, q, [] {});
                }
                o = r.origin; d = r.dir; w = r.w; tries = r.tries;
                if (Th.useDof) { const bool out = tries > static_cast<uint32_t>(kMaxTries); vign += out ? 1u : 0u; succ += out ? 0u : 1u; }
                float4 *rec = reinterpret_cast<float4 *>(stage) + 2u * lane;
                rec[0] = make_float4(o.x, o.y, o.z, d.x);
                rec[1] = make_float4(d.y, d.z, w, __builtin_bit_cast(float, (tries > 0 ? 1u : 0u) | (tries << 1)));
            }
            wave_lds_fence();
        } else {                      // RAYTRACED: the tries of a ray side by side (kolb_wave_rays); every lane takes part in the wave's rounds
            const bool tileWork = work == 2u;
            const uint32_t seed = T.seed;
            const auto rngOf = [&](uint32_t ray) { return tileWork ? rng_for_ray(seed, rayBase + ray) : rng; };
            // a 16-ray batch is ONE group: its own rays in order; a wide batch: the first pass, then the rays it left, 16 at a time, their records
            // copied to the batch's 64-record area
            uint32_t nLeft = cnt;
            const uint32_t *order = reinterpret_cast<const uint32_t *>(stage + kWideOrder);
            if (wide) nLeft = kolb_wide_first_pass<MODE == 0, MODE == 1, NS>(T, B, lutLds, bokehLds, stage, s, cnt, lane, succ);
            for (uint32_t g0 = 0; g0 < nLeft; g0 += kKolbBatch) {
                const uint32_t cntG = nLeft - g0 < kKolbBatch ? nLeft - g0 : kKolbBatch;
                float4 sG = s;
                if (wide) sG = reinterpret_cast<const float4 *>(stage + kWideSamp)[order[g0 + (lane < cntG ? lane : 0u)]];
                kolb_wave_rays<MODE == 0, MODE == 1, NS>(T, B, lutLds, bokehLds, stage, sG, cntG, lane, [&](uint32_t r) { return rngOf(wide ? order[g0 + r] : r); }, succ, vign, tir);
                if (wide) {   // (kolb_wave_rays ends behind an LDS fence: the group's records stand in stage[0 .. 8 cntG))
                    float *rec64 = stage + kWideRecs;
                    for (uint32_t j = lane; j < cntG * 8u; j += 64u) rec64[order[g0 + (j >> 3)] * 8u + (j & 7u)] = stage[j];
                    wave_lds_fence();
                }
            }
        }
        const float *recStage = stage + ((MODEL != 0 && wide) ? kWideRecs : 0u);   // where the batch's records stand

#ifdef ZOIC_TILE_TIMING
        const unsigned long long tt2 = wall_clock64();
#endif
        // ---- the answer -----------------------------------------------------------------------------------------------------------
        if (work == 1u) {
            if (lane == 0) {   // the one ray's record: ox oy oz dx | dy dz w flags
                const float4 r0 = reinterpret_cast<const float4 *>(stage)[0], r1 = reinterpret_cast<const float4 *>(stage)[1];
                uint4 *a = reinterpret_cast<uint4 *>(replies + slot);
                store_uncached(a, make_uint4(__builtin_bit_cast(uint32_t, r0.x), __builtin_bit_cast(uint32_t, r0.y), __builtin_bit_cast(uint32_t, r0.z), seq));
                store_uncached(a + 1, make_uint4(__builtin_bit_cast(uint32_t, r0.w), __builtin_bit_cast(uint32_t, r1.x), __builtin_bit_cast(uint32_t, r1.y), seq));
                store_uncached(a + 2, make_uint4(__builtin_bit_cast(uint32_t, r1.z), __builtin_bit_cast(uint32_t, r1.w), 0u, seq));
            }
            continue;
        }
        // AtCameraOutput rows (84 bytes = 21 floats: origin, dir, dOdx, dOdy, dDdx, dDdy, weight[3]) as zoic_create_rays_arnold
        // writes them (kernels.hip expand_outputs_kernel): origin / dir, dOdy = origin and dDdy = dir for retried rays
        // (zoic.cpp:1974-1977), weight r = g = b, zeros in what camera_create_ray leaves alone.  One lane per output FLOAT.
        if (jobRows != 0u) {   // ZOIC_TILE_ROWS_RAYS: the records as they stand in the stage -- 32 bytes a ray instead of 84 across PCIe
            uint32_t *dst = reinterpret_cast<uint32_t *>(jobOut) + static_cast<size_t>(first) * 8u;
            const uint32_t total = cnt * 8u;
#pragma unroll
            for (uint32_t k = 0; k < (kRays * 8u + 63u) / 64u; ++k) {
                const uint32_t j = k * 64u + lane;
                if (k * 64u < total && j < total) store_sys(dst + j, __builtin_bit_cast(uint32_t, recStage[j]));
            }
            wave_lds_fence();
        } else {
            uint32_t *dst = reinterpret_cast<uint32_t *>(jobOut) + static_cast<size_t>(first) * 21u;
            const uint32_t total = cnt * 21u;
#pragma unroll
            for (uint32_t k = 0; k < (kRays * 21u + 63u) / 64u; ++k) {
                const uint32_t j = k * 64u + lane;
                if (k * 64u < total && j < total) {
                    const uint32_t ray = j / 21u, f = j - ray * 21u;
                    const float *r = recStage + ray * 8u;
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
        if (lane == 0) store_sys(tileFlags + static_cast<size_t>(jobSlot) * kTileMaxBatches + batch, jobSeq);
#ifdef ZOIC_TILE_TIMING
        if (lane == 0) {
            const unsigned long long tt3 = wall_clock64();
            const int o = slotRole ? 0 : 4;
            if (!slotRole) {
                const unsigned long long tp = load_dev(reinterpret_cast<unsigned long long *>(st->jobs + jobSlot) + 6);
                atomicAdd(&st->timing[8], tt0 - tp); atomicMax(&st->timing[9], tt0 - tp); atomicAdd(&st->timing[10], tt3 - tp); atomicMax(&st->timing[11], tt3 - tp);
                if (tt0 - tp > 800ull) { atomicAdd(&st->timing[12], 1ull); atomicAdd(&st->timing[13], static_cast<unsigned long long>(batch)); if (lastWasHint == 0u) atomicAdd(&st->timing[14], 1ull); }
                if (tt3 - tp > 2500ull) { atomicAdd(&st->timing[15], 1ull); }
                const unsigned long long rt = tt2 - tt1, stt = tt0 - tp;   // histograms: the rays' time, the start's delay
                atomicAdd(&st->timing[16 + (rt < 800ull ? 0 : rt < 1200ull ? 1 : rt < 1600ull ? 2 : rt < 2400ull ? 3 : 4)], 1ull);
                atomicAdd(&st->timing[24 + (stt < 600ull ? 0 : stt < 800ull ? 1 : stt < 1200ull ? 2 : 3)], 1ull);
            }
            atomicAdd(&st->timing[o], 1ull); atomicAdd(&st->timing[o + 1], tt1 - tt0); atomicAdd(&st->timing[o + 2], tt2 - tt1); atomicAdd(&st->timing[o + 3], tt3 - tt2);
        }
#endif
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
            for (uint32_t i = 0; i < workerWaves; ++i) st->wake[i].word = 0ull;   // the exit words: the next launch's workers start on clean lines
            __hip_atomic_store(control, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            store_uncached(reinterpret_cast<uint4 *>(header) + 1, make_uint4(0u, 0u, 0u, 0u));   // the last thing this launch does
        }
    }
}

}  // namespace

#ifdef ZOIC_TILE_TIMING
int read_tile_dbg(unsigned long long *out16)
{
    return static_cast<int>(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tileDbg), 16 * sizeof(unsigned long long)));
}
#endif

int launch_mailbox(const KolbTable &kolb, const ThinTable &thin, const BokehTables &bokeh, int model, int mode, void *d_mapped,
                   MailDeviceState *d_state, DeviceCounters *d_counters, uint32_t workerGroups, void *stream)
{
    const bool image = (model == 0 ? thin.useImage : kolb.useImage) != 0;
    const uint32_t ldsWords = (image && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
    const uint32_t groups = kMailSlotGroups + workerGroups;
    const size_t lds = (kLutLdsWords + ldsWords + (kMailBlock / 64u) * stage_words(model)) * sizeof(float);
#define ZOIC_LAUNCH_MAILBOX(MODEL_, MODE_, NS_)                                                                                  \
    hipLaunchKernelGGL((mailbox_kernel<MODEL_, MODE_, NS_>), dim3(groups), dim3(kMailBlock), lds, static_cast<hipStream_t>(stream), kolb, thin, bokeh, \
                       static_cast<char *>(d_mapped), d_state, d_counters, ldsWords, groups * (kMailBlock / 64u))
#define ZOIC_LAUNCH_MAILBOX_BY_COUNT(MODE_)                                                                                      \
    switch (kolb.lensCount) {  /* unrolled traces for the interface counts of real prescriptions (kolb_pool_body.hpp) */        \
    case 7: ZOIC_LAUNCH_MAILBOX(1, MODE_, 7); break;                                                                             \
    case 8: ZOIC_LAUNCH_MAILBOX(1, MODE_, 8); break;                                                                             \
    case 9: ZOIC_LAUNCH_MAILBOX(1, MODE_, 9); break;                                                                             \
    case 10: ZOIC_LAUNCH_MAILBOX(1, MODE_, 10); break;                                                                           \
    case 11: ZOIC_LAUNCH_MAILBOX(1, MODE_, 11); break;                                                                           \
    case 12: ZOIC_LAUNCH_MAILBOX(1, MODE_, 12); break;                                                                           \
    default: ZOIC_LAUNCH_MAILBOX(1, MODE_, 0); break;                                                                            \
    }
    if (model == 0) {   // (THINLENS: MODE only tells the vignetting loop's arithmetic apart)
        if (mode == 0) ZOIC_LAUNCH_MAILBOX(0, 0, 0); else ZOIC_LAUNCH_MAILBOX(0, 1, 0);
    } else if (mode == 0) ZOIC_LAUNCH_MAILBOX(1, 0, 0);
    else if (mode == 1) { ZOIC_LAUNCH_MAILBOX_BY_COUNT(1) }
    else { ZOIC_LAUNCH_MAILBOX_BY_COUNT(2) }
#undef ZOIC_LAUNCH_MAILBOX_BY_COUNT
#undef ZOIC_LAUNCH_MAILBOX
    return static_cast<int>(hipGetLastError());
}

}  // namespace zoic
