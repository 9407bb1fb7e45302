// mailbox.hip -- camera_create_ray(node, input, output, tid) (zoic.cpp:1752) at CALL latency: a persistent one-wave
// kernel per camera that render threads talk to through mapped, page-locked host memory.
//
// Why.  Arnold calls camera_create_ray once per camera sample from every render thread and waits for the ray.  Answering a
// call with a kernel launch + stream synchronise costs 28-33 us (round 2) -- 30x slower than the CPU plug-in it replaces,
// whatever the kernel does.  Here nothing is launched per call:
//   * slot w (= tid mod 64) of the mailbox is owned by wave w of the resident launch (64 waves).  A render thread writes its sample and
//     its retry-stream state into the slot's REQUEST (three 16-byte chunks, each carrying the call's sequence number in
//     its first word, written last) and spins on the slot's REPLY;
//   * the wave polls its request with three global_load_dwordx4 across PCIe (host memory is mapped uncached on the GPU:
//     every poll sees memory); when the three chunks carry one NEW sequence number it evaluates the ray -- the reference's own loop, one ray per lane, STRICT or FAST arithmetic (optics.hpp /
//     fast_optics.hpp; a FAST ray with a decision inside a guard band is re-evaluated on the spot by the listed kernel's rule, kolb_listed_body.hpp) -- bumps the
//     camera's counters and writes the REPLY: three 16-byte chunks, sequence number last.  A chunk is one PCIe
//     transaction: torn reads are impossible within a chunk and detected across chunks (all three numbers must agree), so
//     no fences or doorbells are needed in either direction;
//   * the kernel retires by itself after 1 ms without a call and after 50 ms in any case (a resident kernel would stall the
//     application's hipDeviceSynchronize / hipFree for ever); the next call finds `alive == 0` and launches it again
//     (~20 us, once).  node_update / node_finish / the counter getters stop it first.
// Results are those of the batch kernels for RAYTRACED, bit for bit in STRICT and in FAST: every Kolb kernel of the library
// evaluates a ray with the same device functions, the FAST arithmetic is written with explicit FMAs (fast_optics.hpp: the branchy
// trace here and the unrolled trace of the batch kernels round alike) and a ray too close to call follows the listed kernel's rule
// (tests/test_boundary_gpu.py: a fresh tid's first call == the one-ray batch launch on the same stream).  THINLENS is evaluated in the reference's arithmetic (thin_ray_strict) in EVERY precision mode -- one lane has
// nothing to gain from the fast variant -- so under ZOIC_PRECISION_FAST with optical vignetting on, where the batch path runs
// thin_refill.hip's fast arithmetic, a per-sample ray and a batch ray of the same sample can differ in low-order bits (never in
// a decision: the fast vignetting test is decision-safe).  include/zoic_amd.h states this at zoic_camera_create_ray.
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

struct OneRay { V3 o, d; float w; uint32_t tries, lutMiss, tir; bool unsure; };

// camera_create_ray's RAYTRACED branch for ONE sample (zoic.cpp:1850-1964), the reference's loop as it stands: trace, and
// while the trace fails and tries <= 25 draw the next lens sample.  GUARD: `unsure` is set when a decision lay inside its
// guard band (fast_optics.hpp) -- the caller then evaluates the ray again in STRICT.
template <bool STRICT>
__device__ __forceinline__ OneRay kolb_one_ray(const KolbTable &T, const BokehTables &B, const float2 *lutLds, const float *bokehLds,
                                               float4 s, Rng rng, bool guard)
{
    OneRay r;
    const RaySetup rs = setup_ray<STRICT>(T, lutLds, s.x, s.y);
    r.lutMiss = rs.flags & 1u; r.tir = 0; r.tries = 0;
    r.unsure = guard && T.useLUT && rs.lutEdge;
    const V3 o0{rs.o0x, rs.o0y, T.originShift};
    V2 lens = lens_sample<STRICT>(T, B, bokehLds, s.z, s.w);   // zoic.cpp:1870
    if (!T.useLUT) {                                           // zoic.cpp:1873-1877
        r.d = V3{(lens.x * T.rearAperture) - o0.x, (lens.y * T.rearAperture) - o0.y, T.dirZ};
    } else {                                                   // zoic.cpp:1913-1924: x-only translation on the first sample
        lens.x *= rs.maxScale; lens.y *= rs.maxScale;
        lens.x += rs.translation;
        const float rx = lens.x * rs.cs - lens.y * rs.sn, ry = lens.x * rs.sn + lens.y * rs.cs;
        r.d = V3{rx - o0.x, ry - o0.y, T.dirZ};
    }
    r.o = o0;
    for (;;) {
        bool ok, near = false;
        if constexpr (STRICT) ok = trace_lens_strict(T, r.o, r.d, r.tir);
        else ok = trace_lens_fast_rolled(T, r.o, r.d, r.tir, guard ? &near : nullptr);
        r.unsure |= near;
        if (ok || r.tries > static_cast<uint32_t>(kMaxTries)) break;   // zoic.cpp:1879 / 1927
        r.o = o0;
        const float u = rng_unit(xor128(rng));                  // zoic.cpp:1930
        const float v = rng_unit(xor128(rng));
        r.d = retry_direction(T, lens_sample<STRICT>(T, B, bokehLds, u, v), rs.o0x, rs.o0y, rs.maxScale, rs.translation, rs.sn, rs.cs);
        ++r.tries;
    }
    r.w = (r.tries > static_cast<uint32_t>(kMaxTries)) ? 0.0f : 1.0f;   // zoic.cpp:1951-1957
    if (T.exposureOn) r.w *= T.exposureMul;                             // zoic.cpp:1981-1987
    return r;
}

// model: ZOIC_THINLENS 0 / ZOIC_RAYTRACED 1 (zoic_amd.h); mode: 0 STRICT, 1 FAST decision-safe, 2 FAST unchecked.
// 4 workgroups x 16 waves: wave w of the launch owns slot w and works with ONE lane -- a render thread never waits for
// another thread's ray (one wave for all slots measured 8 us per call with one calling thread, 29 us with four).
// control (device memory): [0] exit flag (set by wave 0: stop request / 1 ms without a call / 50 ms of life),
// [1] waves that have left, [2..3] wall-clock time of the last call any wave answered.
__global__ __launch_bounds__(kMailBlock) void mailbox_kernel(const KolbTable T, const ThinTable Th, const BokehTables B, int model, int mode,
                                                             MailHeader *header, const MailRequest *requests, MailReply *replies,
                                                             uint32_t *served, uint32_t *control, DeviceCounters *counters, uint32_t ldsWords)
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
    const uint32_t slot = blockIdx.x * (kMailBlock / 64u) + (threadIdx.x >> 6);
    volatile unsigned long long *lastCall = reinterpret_cast<volatile unsigned long long *>(control + 2);
    // All control flow below is wave-uniform (the polled words are broadcast to SGPRs); lane 0 evaluates the ray.
    uint32_t mine = served[slot];                      // sequence number of the last call this slot answered
    const unsigned long long start = wall_clock64();   // 100 MHz
    if (slot == 0 && lane == 0) *lastCall = start;
    uint32_t succ = 0, vign = 0, tir = 0;
    // slots no render thread has used yet are not watched at all: their waves leave at once (the host starts the launch again
    // when a new tid shows up -- 63 waves reading host memory for nothing slowed every poll of the busy ones down)
    bool watched;
    {
        u32x4 line, ctl;
        poll_uncached(reinterpret_cast<const uint4 *>(header), control, line, ctl);
        watched = slot < lane_word(line.z, 0);
    }
    const uint4 *myChunk = reinterpret_cast<const uint4 *>(requests + slot) + (lane < 3u ? lane : 3u);
    while (watched) {
        u32x4 line, ctl;   // ctl: the launch's control block = {exit flag, waves out, time of the last call (lo, hi)}
        poll_uncached(myChunk, control, line, ctl);
        // chunk k of the request sits in lane k: {sx, sy, lensx, seq} {lensy, rng.x, rng.y, seq} {rng.z, rng.w, -, seq} {stop (slot 0), ...}
        const uint32_t seq = lane_word(line.w, 0);
        const bool work = seq != mine && seq == lane_word(line.w, 1) && seq == lane_word(line.w, 2);
        const uint32_t stop = lane_word(line.x, 3);
        const uint32_t leave = lane_word(ctl.x, 0);
        const unsigned long long lastSeen = (static_cast<unsigned long long>(lane_word(ctl.w, 0)) << 32) | lane_word(ctl.z, 0);
        const unsigned long long now = wall_clock64();
        if (work) {
            if (lane == 0) {
                *lastCall = now;
                const float4 s = make_float4(__builtin_bit_cast(float, lane_word(line.x, 0)), __builtin_bit_cast(float, lane_word(line.y, 0)),
                                             __builtin_bit_cast(float, lane_word(line.z, 0)), __builtin_bit_cast(float, lane_word(line.x, 1)));
                const Rng rng{lane_word(line.y, 1), lane_word(line.z, 1), lane_word(line.x, 2), lane_word(line.y, 2)};
                V3 o, d; float w; uint32_t tries, lutMiss = 0;
                if (model == 0) {   // THINLENS, zoic.cpp:1771-1846 (reference arithmetic in every precision mode)
                    Rng q = rng;
                    const ThinRay r = thin_ray_strict(Th, B, bokehLds, s, q, [] {});
                    o = r.origin; d = r.dir; w = r.w; tries = r.tries;
                    if (Th.useDof) { if (tries > static_cast<uint32_t>(kMaxTries)) ++vign; else ++succ; }
                } else {
                    OneRay r;
                    if (mode == 0) r = kolb_one_ray<true>(T, B, lutLds, bokehLds, s, rng, false);
                    else {
                        r = kolb_one_ray<false>(T, B, lutLds, bokehLds, s, rng, mode == 1);
                        if (r.unsure) {   // a decision too close to call: the ray is evaluated as the batch path's listed kernel does it
                            const ListedRay q = listed_one_ray(T, B, lutLds, bokehLds, s, rng);   // (kolb_listed_body.hpp: same rule, same bits)
                            r.o = q.o; r.d = q.d; r.w = q.w; r.tries = q.tries; r.lutMiss = q.lutMiss; r.tir = q.tir;
                        }
                    }
                    o = V3{r.o.x * -1.0f, r.o.y * -1.0f, r.o.z * -1.0f}; d = V3{r.d.x * -1.0f, r.d.y * -1.0f, r.d.z * -1.0f};   // zoic.cpp:1960-1961
                    w = r.w; tries = r.tries; lutMiss = r.lutMiss; tir += r.tir;
                    if (tries > static_cast<uint32_t>(kMaxTries)) ++vign; else ++succ;
                }
                const uint32_t flags = (tries > 0 ? 1u : 0u) | (tries << 1) | (lutMiss << 6);
                uint4 *a = reinterpret_cast<uint4 *>(replies + slot);
                store_uncached(a, make_uint4(__builtin_bit_cast(uint32_t, o.x), __builtin_bit_cast(uint32_t, o.y), __builtin_bit_cast(uint32_t, o.z), seq));
                store_uncached(a + 1, make_uint4(__builtin_bit_cast(uint32_t, d.x), __builtin_bit_cast(uint32_t, d.y), __builtin_bit_cast(uint32_t, d.z), seq));
                store_uncached(a + 2, make_uint4(__builtin_bit_cast(uint32_t, w), flags, 0u, seq));
            }
            mine = seq;
            continue;
        }
        if (slot == 0 && (stop != 0u || (now > lastSeen && now - lastSeen > kMailIdleTicks) || now - start > kMailLifeTicks)) {
            if (lane == 0) __hip_atomic_store(control, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // everybody out
            break;
        }
        if (leave != 0u) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (lane == 0) {
        served[slot] = mine;
        // the counters of node_finish (zoic.cpp:1729-1732): one atomic per counter for the whole stay
        if (counters) {
            DeviceCounters *cs = counter_set(counters);
            if (succ) atomicAdd(&cs->succes, static_cast<unsigned long long>(succ));
            if (vign) atomicAdd(&cs->vignetted, static_cast<unsigned long long>(vign));
            if (tir) atomicAdd(&cs->tir, static_cast<unsigned long long>(tir));
        }
        __threadfence_system();
        if (atomicAdd(control + 1, 1u) == kMailSlots - 1u) {   // the last wave out resets the control block and clears `alive`
            control[1] = 0u;
            __hip_atomic_store(control, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            store_uncached(reinterpret_cast<uint4 *>(header) + 1, make_uint4(0u, 0u, 0u, 0u));   // the last thing this launch does
        }
    }
}

}  // namespace

int launch_mailbox(const KolbTable &kolb, const ThinTable &thin, const BokehTables &bokeh, int model, int mode, MailHeader *d_header,
                   const MailRequest *d_requests, MailReply *d_replies, uint32_t *d_served, uint32_t *d_control, DeviceCounters *d_counters,
                   void *stream)
{
    const bool image = (model == 0 ? thin.useImage : kolb.useImage) != 0;
    const uint32_t ldsWords = (image && bokeh.ldsWords > 0 && bokeh.ldsWords <= 10240) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
    hipLaunchKernelGGL(mailbox_kernel, dim3(kMailSlots * 64u / kMailBlock), dim3(kMailBlock), (kLutLdsWords + ldsWords) * sizeof(float),
                       static_cast<hipStream_t>(stream), kolb, thin, bokeh, model, mode, d_header, d_requests, d_replies, d_served, d_control,
                       d_counters, ldsWords);
    return static_cast<int>(hipGetLastError());
}

}  // namespace zoic
