// mailbox.hpp -- the memory layout the per-sample mailbox kernel (mailbox.hip) and the host (capi.cpp) share.
#pragma once
#include <cstdint>

#include "kernels.hpp"

namespace zoic {

constexpr uint32_t kMailSlots = 64;                       // one slot per wave of the resident launch; tid -> slot tid % 64
constexpr uint32_t kMailBlock = 1024;                     // 16 waves per workgroup, 4 workgroups
constexpr unsigned long long kMailIdleTicks = 100000;     // 1 ms of the 100 MHz wall clock without a call: the kernel retires
constexpr unsigned long long kMailLifeTicks = 5000000;    // 50 ms: ... and in any case, so that a device-wide synchronise ends

// Every 16-byte chunk is written and read as ONE PCIe transaction and starts with the call's sequence number, written last
// by the host: a chunk whose number is new carries new data, and a message is complete when its three numbers agree.
struct alignas(64) MailRequest {              // the number is the LAST word of every chunk
    float sx, sy, lensx; uint32_t seq0;       // AtCameraInput fields zoic reads (zoic.cpp:1853-1854, 1870)
    float lensy; uint32_t rngX, rngY, seq1;   // the calling tid's xorshift128 retry stream (zoic.cpp:647-652)
    uint32_t rngZ, rngW, pad2, seq2;
    uint32_t stop, fill[3];                   // slot 0 only: the host asks the launch to retire (the wave polls ONE line)
};
struct alignas(64) MailReply {                // the number is the LAST word of a reply chunk: whatever order a chunk's bytes land in
    float ox, oy, oz; uint32_t seq0;          // output.origin
    float dx, dy, dz; uint32_t seq1;          // output.dir
    float weight; uint32_t flags, pad2, seq2; // zoic_ray::weight / flags
    uint32_t fill[4];
};
struct alignas(64) MailHeader {
    uint32_t pad0[2], slotsInUse, pad1;       // chunk 0: written by the host only, read once by every wave of a launch
    uint32_t alive, pad2[3];                  // chunk 1: set by the host before a launch, cleared by the kernel as its last act
    uint32_t fill[8];
};
static_assert(sizeof(MailRequest) == 64 && sizeof(MailReply) == 64 && sizeof(MailHeader) == 64, "mailbox layout");

int launch_mailbox(const KolbTable &kolb, const ThinTable &thin, const BokehTables &bokeh, int model, int mode, MailHeader *d_header,
                   const MailRequest *d_requests, MailReply *d_replies, uint32_t *d_served, uint32_t *d_control, DeviceCounters *d_counters,
                   void *stream);

}  // namespace zoic
