// mailbox.hpp -- the memory layout the resident kernel (mailbox.hip) and the host (capi.cpp) share: per-sample calls
// (zoic_camera_create_ray) and TILE requests (zoic_tile_submit / zoic_camera_create_rays_tile) of one camera.
#pragma once
#include <cstdint>

#include "kernels.hpp"

namespace zoic {

constexpr uint32_t kMailSlots = 64;                       // one slot per SLOT wave of the resident launch; tid -> slot tid % 64
constexpr uint32_t kMailBlock = 256;                      // 4 waves per workgroup: 16 slot workgroups, then the tile workers'
constexpr uint32_t kMailSlotGroups = kMailSlots * 64u / kMailBlock;
constexpr uint32_t kTileWorkerGroups = 256;               // default number of worker workgroups (x 4 waves) once a tile has been submitted
constexpr uint32_t kTileMaxSamples = 65536;               // samples per tile request (ZOIC_TILE_MAX_SAMPLES)
constexpr unsigned long long kMailIdleTicks = 100000;     // 1 ms of the 100 MHz wall clock without a call: the kernel retires
constexpr unsigned long long kMailLifeTicks = 5000000;    // 50 ms: ... and in any case, so that a device-wide synchronise ends
constexpr unsigned long long kMailHardLifeTicks = 100000000;   // 1 s: no wave of a launch outlives this whatever it is waiting for (a safety net: never seen)

// Every 16-byte chunk is written and read as ONE PCIe transaction and ends with the call's sequence number, written last
// by the host: a chunk whose number is new carries new data, and a message is complete when its three numbers agree.
// kind (the third chunk's third word): 0 = one sample, 1 = a tile.
struct alignas(64) MailRequest {              // the number is the LAST word of every chunk
    float sx, sy, lensx; uint32_t seq0;       // AtCameraInput fields zoic reads (zoic.cpp:1853-1854, 1870)
    float lensy; uint32_t rngX, rngY, seq1;   // the calling tid's xorshift128 retry stream (zoic.cpp:647-652)
    uint32_t rngZ, rngW, kind, seq2;
    uint32_t stop, fill[3];                   // slot 0 only: the host asks the launch to retire (the wave polls ONE line)
};
// the same line carrying a tile: n AtCameraInput rows at `in` -> n AtCameraOutput rows at `out` (device addresses of mapped,
// page-locked host memory), ray i drawing its retries from the stream keyed by base + i
struct alignas(64) MailTileRequest {
    uint32_t inLo, inHi, n, seq0;
    uint32_t outLo, outHi, baseLo, seq1;
    uint32_t baseHi, pad, kind, seq2;         // pad: bit 0 what to write at `out` -- AtCameraOutput rows (84 B) / zoic_ray records (32 B); bit 1 what stands at `in` -- AtCameraInput rows (28 B) / samples (16 B)
    uint32_t stop, fill[3];
};
struct alignas(64) MailReply {                // the number is the LAST word of a reply chunk: whatever order a chunk's bytes land in
    float ox, oy, oz; uint32_t seq0;          // output.origin
    float dx, dy, dz; uint32_t seq1;          // output.dir
    float weight; uint32_t flags, pad2, seq2; // zoic_ray::weight / flags
    uint32_t fill[4];
};
struct alignas(64) MailHeader {
    uint32_t pad0[2], slotsInUse, workerGroups;   // chunk 0: written by the host only, read once by every wave of a launch
    uint32_t alive, pad2[3];                      // chunk 1: set by the host before a launch, cleared by the kernel as its last act
    uint32_t fill[8];
};
static_assert(sizeof(MailRequest) == 64 && sizeof(MailTileRequest) == 64 && sizeof(MailReply) == 64 && sizeof(MailHeader) == 64, "mailbox layout");
// byte offsets into the mapped allocation
// (tile flags: one word per 64-sample batch of a slot's tile -- the tile's sequence number once the batch's rows are complete, written
// by the wave that made them behind a system-scope release; the render thread waits for all of its tile's)
#ifndef ZOIC_TILE_RAYS
#define ZOIC_TILE_RAYS 16   // rays a resident wave holds at once (RAYTRACED): 64 / this lanes per ray in the first round
#endif
constexpr uint32_t kTileRaysRaytraced = ZOIC_TILE_RAYS, kTileRaysThin = 64;
// LARGE RAYTRACED tiles (round 6): from kTileWideSamples samples on a batch is 64 rays -- a first try at ONE RAY PER LANE (what 84 % of a double Gauss's rays
// need), then the rays it did not settle in groups of kTileRaysRaytraced through the tries-side-by-side rounds (mailbox.hip kolb_wave_wide).  Four lanes per
// ray are a bucket's LATENCY (a 4096-sample tile waits for its slowest 16-ray batch); a 65 536-sample request is THROUGHPUT: at 16 rays per wave pass the tile
// workers top out at ~0.6 Grays/s for one request whatever its rows cross (a device-resident one: 109 us, VERDICT r5 #7 asked for 25).
#ifndef ZOIC_TILE_WIDE_SAMPLES
#define ZOIC_TILE_WIDE_SAMPLES 16384
#endif
constexpr uint32_t kTileRaysWide = 64, kTileWideSamples = ZOIC_TILE_WIDE_SAMPLES;
// Samples per batch of an n-sample tile (host and kernel agree on this; the descriptor carries it).  A RAYTRACED wave shares its 64 lanes
// among the rays it holds, so half a batch of rays finishes in fewer rounds -- and serves half the rays per wave pass.  Measured
// [MI355X, profiles/ab_r05/tile_latency_v6.txt]: 8 instead of 16 rays for tiles up to 8192 samples: double Gauss 4096 samples 36.7 ->
// 34.2 us (STRICT 41 -> 35), but TESSAR / fisheye / PETZVAL 1-2 us SLOWER, a 16-sample tile 14 -> 16.4 us (two batches), 16 threads x
// 4096 samples 212 -> 141 Mrays/s; 8 rays throughout: 16 threads x 65536 samples 479 -> 280 Mrays/s.  One size, 16.
#if defined(__HIPCC__) || defined(__cplusplus)
inline
#if defined(__HIPCC__)
__host__ __device__
#endif
uint32_t tile_rays_per_batch(bool thinLens, uint32_t n) { return thinLens ? kTileRaysThin : (n >= kTileWideSamples ? kTileRaysWide : kTileRaysRaytraced); }
#endif   // samples per batch of a tile (mailbox.hip: RAYTRACED spends four lanes on a ray)
constexpr uint32_t kTileMaxBatches = kTileMaxSamples / kTileRaysRaytraced;   // (>= 8192 / (kTileRaysRaytraced / 2))
constexpr size_t kMailRequestsOffset = 64, kMailRepliesOffset = kMailRequestsOffset + 64 * kMailSlots,
                 kMailTileFlagsOffset = kMailRepliesOffset + 64 * kMailSlots, kMailBytes = kMailTileFlagsOffset + 4u * kTileMaxBatches * kMailSlots;

// Device-memory state of the resident launch (survives its retirements): what a slot has answered, the launch's control block,
// and the tile jobs the slot waves post for the workers.
constexpr uint32_t kTileParts = 32;           // ticket partitions of a tile: same-address atomics are served one at a time
constexpr uint32_t kTileMaxWorkerWaves = 1024;   // wake lines (kTileWorkerGroups x 4 waves by default; ZOIC_TILE_WORKER_GROUPS <= 256)
struct alignas(64) TileJob {                  // three 16-byte chunks, each ending in the tile's sequence number (like a request line):
    uint32_t inLo, inHi, n, seq0;             // written by the slot's wave, read -- in the same round trip as the ticket -- by whoever
    uint32_t outLo, outHi, baseLo, seq1;      // draws one; a descriptor whose three numbers equal the ticket's generation is that tile's
    uint32_t baseHi, batches, parts, seq2;    // parts: ticket partitions in use (1 ... kTileParts) | rows (0 / 1) << 8 | inputs (0 / 1) << 9 | samples per batch << 16; partition p hands out batches
    uint32_t fill[4];                         //        [p * per, min((p + 1) * per, batches)), per = ceil(batches / parts)
};
struct alignas(64) TileCounter { unsigned long long next; uint32_t pad[14]; };   // (generation << 32) | next batch of the partition
struct alignas(64) TileTickets {
    TileCounter part[kTileParts];             // one atomicAdd hands a batch out
    uint32_t partMask, pad[15];               // bit p: partition p has batches left.  The wave that draws a partition's last batch clears its bit, the
};                                            // one that clears the last bit clears the slot's bit in the work mask; a wave that finds its partition
                                              // empty reads this word and goes straight to one that is not (up to 31 wasted ticket atomics before)
struct alignas(64) TileWake { unsigned long long word; uint32_t pad[14]; };   // per WORKER wave: (changing number << 32) | exit << 16 | partition << 8 | slot (6 bits)
struct MailDeviceState {
    uint32_t served[kMailSlots];              // sequence number of the last call each slot answered
    alignas(64) uint32_t control[16];         // [0] exit flag, [1] waves that have left, [2..3] wall-clock time of the last call,
                                              // [4..5] 64-bit mask of the slots whose tile still has batches to hand out,
                                              // [8] next worker wave to wake (rotates: tiles of different slots wake different waves)
    TileJob jobs[kMailSlots];
    TileTickets tickets[kMailSlots];
    TileWake wake[kTileMaxWorkerWaves];
    unsigned long long timing[32];            // -DZOIC_TILE_TIMING builds only: 10 ns ticks per region of a batch, summed over all waves (tools/)
};

#ifdef ZOIC_TILE_TIMING
int read_tile_dbg(unsigned long long *out16);   // timing builds: mailbox.hip's g_tileDbg
#endif
int launch_mailbox(const KolbTable &kolb, const ThinTable &thin, const BokehTables &bokeh, int model, int mode, void *d_mapped,
                   MailDeviceState *d_state, DeviceCounters *d_counters, uint32_t workerGroups, void *stream);

}  // namespace zoic
