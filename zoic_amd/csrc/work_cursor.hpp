// work_cursor.hpp -- how the persistent kernels (kolb_refill.hip, thin_refill.hip) share out a batch of samples.
//
// A wave consumes chunks of `chunkRays` consecutive samples; a chunk is claimed with one atomicAdd.  Claims on ONE
// address are served by the L2 at about one per 12 ns (measured: 518 K claims of 256 samples took 6.3 ms on a frame that
// otherwise takes 4.0; 130 K claims of 64 took 1.75 ms on an 8.3 M-sample batch that takes 0.63 ms with 256), so the
// batch is cut into kCursorParts partitions with a cursor each (different addresses are served in parallel) and a launch
// gets a budget of ~32 K claims per cursor.  A wave starts on its workgroup's home partition and moves on, for good, when
// a partition is used up: expensive image regions cannot unbalance the chip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "kernels.hpp"

#ifndef ZOIC_CHUNK_RAYS
#define ZOIC_CHUNK_RAYS 0u
#endif
#ifndef ZOIC_GRID_BLOCKS
#define ZOIC_GRID_BLOCKS 0ull
#endif
#ifndef ZOIC_INTERLEAVED_PARTS
#define ZOIC_INTERLEAVED_PARTS 0
#endif

namespace zoic {

constexpr uint32_t kMaxChunkRays = 1024;   // 16 passes of fresh work
constexpr bool kInterleavedParts = ZOIC_INTERLEAVED_PARTS != 0;

struct WorkGrain { uint32_t chunkRays, chunksPerPart; };

// chunk size for a batch of m samples: 64-sample tiles on small batches; from 4 M samples at least `floorRays` (a wave
// changes chunk -- an exposed atomic + window fetch -- every few passes otherwise): 512 for the FAST Kolb kernels (TESSAR
// 1080p x 8, 16.6 M rays: 256 -> 21.5, 384 -> 22.4, 512 -> 23.1, 768 -> 21.9, 1024 -> 21.0 Grays/s unchecked), 256 for the
// STRICT ones, whose passes are 2.5x longer (14.9 at 256, 14.5 at 512) and for the thin lens; 512 on a 4K x 16spp frame by
// the claim budget, 1024 at most.  -DZOIC_CHUNK_RAYS=n overrides the rule (experiments).
inline WorkGrain work_grain(uint64_t m, uint32_t floorRays = 256)
{
    uint64_t chunk = (m / (32768ull * kCursorParts) + 63) / 64 * 64;
    if (chunk < floorRays && m >= (4ull << 20)) chunk = floorRays;
    constexpr uint32_t chunkOverride = ZOIC_CHUNK_RAYS;
    WorkGrain g;
    g.chunkRays = chunkOverride ? chunkOverride : static_cast<uint32_t>(chunk < 64 ? 64 : (chunk > kMaxChunkRays ? kMaxChunkRays : chunk));
    const uint64_t totalChunks = (m + g.chunkRays - 1) / g.chunkRays;
    g.chunksPerPart = static_cast<uint32_t>((totalChunks + kCursorParts - 1) / kCursorParts);
    return g;
}

inline hipError_t reset_work_cursors(unsigned int *d_workCursor, hipStream_t st)
{
    return hipMemsetAsync(d_workCursor, 0, kCursorParts * kCursorPartStride * sizeof(unsigned int), st);   // same stream as the kernel: ordered
}

// persistent grid.  Large batches: enough 256-lane workgroups to fill every wave slot of 256 CUs (late or surplus workgroups
// find the cursors exhausted and retire at once, so residency need not be known exactly).  Small batches: sqrt(m) / 2
// workgroups -- every wave pays its start (LDS tables, first claim, first window: three dependent round trips) and its
// tail (the last rays' remaining tries at a few lanes per pass) once, so fewer waves with more rays each win until the
// chip runs short of waves to hide latency with.  Measured optimum on TESSAR unchecked / double Gauss decision-safe /
// TESSAR strict alike (tools/exp_grid.py): 256 K rays -> 256 workgroups, 1 M -> 512, 4 M -> 1024, 16 M -> 2048; against
// "one wave per 64 rays, 2048 at most": 1 M rays 293 -> 167 us, 2 M 337 -> 235, 4 M 303 -> 272 (TESSAR unchecked).
// -DZOIC_GRID_BLOCKS=n overrides the rule (experiments).
inline unsigned persistent_grid(uint64_t m, unsigned wavesPerBlock)
{
    const uint64_t tiles = (m + 63) / 64;
    const uint64_t wantBlocks = (tiles + wavesPerBlock - 1) / wavesPerBlock;
    constexpr uint64_t capOverride = ZOIC_GRID_BLOCKS;
    uint64_t cap = 2048;
    if (capOverride) cap = capOverride;
    else {
        uint64_t r = 1;
        while (4 * r * r < m && r < 2048) ++r;   // ceil(sqrt(m) / 2)
        cap = r < 2048 ? r : 2048;
    }
    return static_cast<unsigned>(wantBlocks < cap ? (wantBlocks ? wantBlocks : 1) : cap);
}

#if defined(__HIPCC__)
// Wave-uniform claim of the next chunk [next, end) of a batch of n samples; false when every partition is used up.
// `part` / `partsTried` are the wave's persistent cursor state (part starts at blockIdx.x % kCursorParts).
__device__ __forceinline__ bool claim_chunk(unsigned int *__restrict__ workCursor, uint32_t lane, uint32_t &part, uint32_t &partsTried,
                                            uint32_t chunkRays, uint32_t chunksPerPart, uint32_t n, uint32_t &next, uint32_t &end)
{
    uint64_t begin = n;
    while (partsTried < kCursorParts) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(workCursor + part * kCursorPartStride, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        begin = kInterleavedParts ? (static_cast<uint64_t>(c) * kCursorParts + part) * chunkRays
                                  : (static_cast<uint64_t>(part) * chunksPerPart + c) * chunkRays;
        if (c < chunksPerPart && begin < n) break;
        begin = n;                                   // this partition is used up: on to the next one, for good
        part = (part + 1u) % kCursorParts;
        ++partsTried;
    }
    if (begin >= n) return false;
    next = static_cast<uint32_t>(begin);
    end = (begin + chunkRays < n) ? static_cast<uint32_t>(begin + chunkRays) : n;
    return true;
}
#endif

}  // namespace zoic
