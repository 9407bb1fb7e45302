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
#include <cstring>

#include "kernels.hpp"

#ifndef ZOIC_CHUNK_RAYS
#define ZOIC_CHUNK_RAYS 0u
#endif
#ifndef ZOIC_GRID_BLOCKS
#define ZOIC_GRID_BLOCKS 0ull
#endif

namespace zoic {

constexpr uint32_t kMaxChunkRays = 1024;   // 16 passes of fresh work

struct WorkGrain { uint32_t chunkRays, chunksPerPart; };   // chunksPerPart: PACKED, see below

// The tail of a launch.  Chunks are claimed dynamically, so the waves of a launch run dry within one chunk's duration of each other --
// on a 4K x 16spp frame (3 ms) that is 2 % of the launch, on a 1080p x 8spp frame (0.5 ms, ~4 chunks of 512 rays per wave at 120 us each)
// a fifth: the PMC run of C2 shows 3.9 of 5 wave slots occupied on average.  "Guided" chunk sizes that shrink towards the end were tried in
// round 3 inside the pass loop and lost to their own extra scalars.  Here the LAST `fineEighths` eighths of every partition are handed out in
// chunks of chunkRays >> fineShift rays from a second cursor per partition, and only to waves that found all coarse partitions used up
// (partsTried counts on to 2 x kCursorParts): everything new lives inside claim_chunk, which runs once per chunk, and in two bit fields of
// the chunks-per-partition word the kernels already fetch there:
//   bits 0-23 chunks per partition (of chunkRays rays), bits 24-27 fineEighths (0 = off: the old behaviour, bit for bit), bits 28-31 fineShift
constexpr uint32_t kChunksPerPartMask = 0xffffffu;
constexpr unsigned kFineCursorOffset = 48;    // dword offset of a partition's fine cursor inside its 64-dword stride (0: coarse, 16: listed kernel, 32: list length)
inline uint32_t pack_chunks_per_part(uint32_t chunksPerPart, uint32_t fineEighths, uint32_t fineShift)
{
    return (chunksPerPart & kChunksPerPartMask) | ((fineEighths & 15u) << 24) | ((fineShift & 15u) << 28);
}

// chunk size for a batch of m samples: 64-sample tiles on small batches; from 4 M samples at least `floorRays` (a wave
// changes chunk -- an exposed atomic + window fetch -- every few passes otherwise): 512 for the FAST Kolb kernels (TESSAR
// 1080p x 8, 16.6 M rays: 256 -> 21.5, 384 -> 22.4, 512 -> 23.1, 768 -> 21.9, 1024 -> 21.0 Grays/s unchecked), 256 for the
// STRICT ones, whose passes are 2.5x longer (14.9 at 256, 14.5 at 512) and for the thin lens; 512 on a 4K x 16spp frame by
// the claim budget, 1024 at most.  -DZOIC_CHUNK_RAYS=n overrides the rule (experiments).
#ifndef ZOIC_FINE_TAIL_EIGHTHS
#define ZOIC_FINE_TAIL_EIGHTHS 0   // build default of the fine tail (0 = off); ZOIC_FINE_TAIL=e,s overrides at run time
#endif
#ifndef ZOIC_FINE_TAIL_SHIFT
#define ZOIC_FINE_TAIL_SHIFT 2
#endif
inline WorkGrain work_grain(uint64_t m, uint32_t floorRays = 256, bool allowFineTail = false)
{
    uint64_t chunk = (m / (32768ull * kCursorParts) + 63) / 64 * 64;
    if (chunk < floorRays && m >= (4ull << 20)) chunk = floorRays;
    constexpr uint32_t chunkOverride = ZOIC_CHUNK_RAYS;
    WorkGrain g;
    g.chunkRays = chunkOverride ? chunkOverride : static_cast<uint32_t>(chunk < 64 ? 64 : (chunk > kMaxChunkRays ? kMaxChunkRays : chunk));
    const uint64_t totalChunks = (m + g.chunkRays - 1) / g.chunkRays;
    g.chunksPerPart = static_cast<uint32_t>((totalChunks + kCursorParts - 1) / kCursorParts);
    // fine tail (above): ZOIC_FINE_TAIL="eighths,shift" (run time, experiments); needs whole 64-sample batches per fine chunk and a
    // chunk count that fits its field
    static const unsigned fineCfg = [] {
        const char *e = std::getenv("ZOIC_FINE_TAIL");
        unsigned f = ZOIC_FINE_TAIL_EIGHTHS, sh = ZOIC_FINE_TAIL_SHIFT;
        if (e) { f = static_cast<unsigned>(std::atoi(e)); const char *c = std::strchr(e, ','); sh = c ? static_cast<unsigned>(std::atoi(c + 1)) : 2u; }
        return (f & 7u) | ((sh & 7u) << 8);
    }();
    const uint32_t f = fineCfg & 0xffu, sh = fineCfg >> 8;
    if (f != 0u && sh != 0u && allowFineTail && g.chunkRays % (64u << sh) == 0u && g.chunksPerPart >= 16u && g.chunksPerPart <= kChunksPerPartMask)
        g.chunksPerPart = pack_chunks_per_part(g.chunksPerPart, f, sh);
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
                                            uint32_t chunkRays, uint32_t chunksPerPartPacked, uint32_t n, uint32_t &next, uint32_t &end)
{
    const uint32_t chunksPerPart = chunksPerPartPacked & kChunksPerPartMask, fineEighths = (chunksPerPartPacked >> 24) & 15u, fineShift = chunksPerPartPacked >> 28;
    const uint32_t coarse = chunksPerPart - ((chunksPerPart * fineEighths) >> 3);     // chunks of a partition handed out whole (all of them when the fine tail is off)
    const uint32_t phases = fineEighths != 0u ? 2u * kCursorParts : kCursorParts;
    uint64_t begin = n;
    uint32_t size = chunkRays;
    while (partsTried < phases) {
        const bool fine = partsTried >= kCursorParts;      // every coarse partition is used up: the partitions' tails, in small chunks
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(workCursor + part * kCursorPartStride + (fine ? kFineCursorOffset : 0u), 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        size = fine ? chunkRays >> fineShift : chunkRays;
        const uint32_t limit = fine ? (chunksPerPart - coarse) << fineShift : coarse;
        begin = (static_cast<uint64_t>(part) * chunksPerPart + (fine ? coarse : 0u)) * chunkRays + static_cast<uint64_t>(c) * size;
        if (c < limit && begin < n) break;
        begin = n;                                   // this partition is used up: on to the next one, for good
        part = (part + 1u) % kCursorParts;
        ++partsTried;
    }
    if (begin >= n) return false;
    next = static_cast<uint32_t>(begin);
    end = (begin + size < n) ? static_cast<uint32_t>(begin + size) : n;
    return true;
}
#endif

}  // namespace zoic
