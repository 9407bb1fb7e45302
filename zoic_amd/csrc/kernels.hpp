// kernels.hpp -- host-callable launchers of the HIP kernels (kernels.hip).  Plain C++ types only.
#pragma once
#include <cstddef>
#include <cstdint>

#include "tables.hpp"

namespace zoic {

// One finished camera ray: a 32-byte record, so the lane that finishes a ray writes exactly one aligned 32-byte
// sector with two 16-byte stores.  (The first kernels wrote seven 4-byte planes; with persistent waves the finishing
// lanes hold non-consecutive rays and rocprof measured 2.6x write amplification from partially written lines --
// profiles/r01_fast_v1.  Same layout as zoic_ray in include/zoic_amd.h.)
struct alignas(16) RayRecord {
    float ox, oy, oz, dx, dy, dz, weight;
    uint32_t flags;  // bit0 retried, bits1-5 tries, bit6 outside the exit-pupil LUT
};
static_assert(sizeof(RayRecord) == 32, "ray record is one 32-byte sector");


// zoic.cpp:533-534: succesRays, vignettedRays, totalInternalReflection.  The camera holds kCounterSets copies, one 128-byte
// line each: a wave adds its totals to the copy of its workgroup (blockIdx % kCounterSets) and the host sums them.  With
// ONE copy the 8192 waves of a launch queue three atomics each on one L2 line at ~12 ns apiece while they retire: 36 us of
// a 0.72 ms TESSAR 1080p x 8 launch, 80 us of a 3.9 ms double-Gauss 4K x 16 launch (ZOIC_EXP_NO_COUNTERS A/B).
struct DeviceCounters {
    unsigned long long succes, vignetted, tir;
    unsigned long long pad[13];
};
static_assert(sizeof(DeviceCounters) == 128, "one counter set per 128-byte line");
constexpr unsigned kCounterSets = 64;
#if defined(__HIPCC__)
__device__ __forceinline__ DeviceCounters *counter_set(DeviceCounters *base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return base ? base + (__builtin_amdgcn_workgroup_id_x() % kCounterSets) : nullptr;
#else
    return base;
#endif
}
#endif

// camera_create_ray, RAYTRACED branch (zoic.cpp:1850-1964) over n samples.  fast=false: strict arithmetic.
// d_workCursor: kCursorParts device words, kCursorPartStride dwords apart, the persistent kernel uses as its chunk cursors
// (zeroed on the stream per launch).  One cursor per eighth of the batch: same-address atomics are served one per ~12 ns
// by the L2, different addresses in parallel; a wave starts on its workgroup's home partition and moves on when it is empty.
#ifndef ZOIC_CURSOR_PARTS
#define ZOIC_CURSOR_PARTS 8
#endif
constexpr unsigned kCursorParts = ZOIC_CURSOR_PARTS;   // compile-time A/B: tools/build_variant.sh p4 -DZOIC_CURSOR_PARTS=4
constexpr unsigned kCursorPartStride = 64;   // 256 bytes apart
// The same 2 KB block also holds what the decision-safe FAST mode needs (kolb_refill.hip), each on its own 64-byte line;
// the one memset per launch zeroes all of it.  Dword offsets into the block:
constexpr unsigned kRedoCountOffset = 32;     // length of the STRICT kernel's work list
constexpr unsigned kRedoCursorOffset = 16;    // partition p's cursor of that kernel: block[16 + p * kCursorPartStride]
// precision modes of the Kolb launch (zoic_precision): 0 strict, 1 fast decision-safe, 2 fast unchecked.
// d_scratch: kolb_scratch_dwords() dwords -- the work list of mode 1 (one dword per sample of a launch, launches cover up to
// 2^31 samples); may be null when that is 0
inline size_t kolb_scratch_dwords(const KolbTable &, uint64_t n, int mode)
{
    return mode == 1 ? static_cast<size_t>(n < (1ull << 31) ? n : (1ull << 31)) : 0;
}
// d_cursorPair: TWO cursor blocks (2 x kCursorBlockWords dwords, zero when the camera is created); a launch works on block
// *parity and its first workgroup clears the other one for the launch after it (which the slot's event chain orders behind
// this one) -- no memset node per launch.  *parity is flipped per kernel pipeline launched.
constexpr unsigned kCursorBlockWords = kCursorParts * kCursorPartStride;
int launch_kolb_rays(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                     uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_cursorPair, unsigned *parity,
                     int mode, uint32_t *d_scratch, void *stream);

// camera_create_ray, THINLENS branch (zoic.cpp:1771-1846).  With optical vignetting on (retries possible) the
// persistent-wave refill kernel of thin_refill.hip runs, otherwise the streaming kernel.
int launch_thin_rays(const ThinTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                     uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor, bool fast,
                     void *stream);   // fast: only the refill kernel has a fast arithmetic variant (the streaming kernel is HBM-bound)

// synthetic camera samples: id=(py*W+px)*spp+s, pcg-hashed jitter and lens samples
int launch_generate_samples(float *d_samples, uint64_t rayBase, uint64_t n, uint32_t width, uint32_t height, uint32_t spp,
                            uint32_t seed, void *stream);

// exit-pupil LUT probes (traceThroughLensElementsForApertureSize, zoic.cpp:1309-1350) for one film position
int launch_lut_probes(const KolbTable &table, float originX, const float *d_lensU, const float *d_lensV, uint64_t n,
                      uint8_t *d_accepted, unsigned int *d_tir, void *stream);

// Arnold AoS <-> planes (AtCameraInput 28 B -> sample 16 B; planes -> AtCameraOutput 84 B)
int launch_pack_inputs(const float *d_inputs7, float *d_samples4, uint64_t n, void *stream);
int launch_expand_outputs(const RayRecord *d_rays, float *d_out21, uint64_t n, void *stream);
// records -> 28-byte payload rows (ox oy oz dx dy dz weight): what a multi-device frame moves to its root (frame.cpp)
int launch_pack_payload(const RayRecord *d_rays, float *d_out7, uint64_t n, void *stream);

// ---- sparse payload (frame.cpp, ZOIC_FRAME_PAYLOAD_SPARSE): a chunk of m records as [per 256-ray tile: 256-bit live mask, offset, count]
// [the 28-byte rows of the rays with weight != 0, compacted].  d_sparse: kSparseTileWords dwords per tile, then the rows; d_count: one
// dword, zero on entry, the number of live rays on exit.  Tiles take their row ranges with one atomicAdd each: the ORDER of the tiles'
// ranges is whatever the hardware made it, their content is not (a tile's header says where its rows are).
constexpr unsigned kSparseTileRays = 256, kSparseTileWords = 12;   // mask[8], row offset, live count, 2 x pad: 48 bytes per tile
inline size_t sparse_header_bytes(uint64_t m) { return static_cast<size_t>((m + kSparseTileRays - 1) / kSparseTileRays) * kSparseTileWords * 4u; }
int launch_pack_sparse(const RayRecord *d_rays, uint32_t *d_sparse, unsigned int *d_count, uint64_t m, void *stream);
// the inverse on the root: m rows of 7 floats, the rows of weight-0 rays all zero
int launch_expand_sparse(const uint32_t *d_sparse, float *d_out7, uint64_t m, void *stream);
// the root's own slab in the same convention: payload rows with the rows of weight-0 rays zeroed
int launch_pack_payload_live(const RayRecord *d_rays, float *d_out7, uint64_t n, void *stream);

}  // namespace zoic
