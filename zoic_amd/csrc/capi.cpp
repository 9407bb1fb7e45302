// capi.cpp -- the C-ABI of libzoic_amd.so (include/zoic_amd.h): zoic's Arnold node methods restated over plain
// pointers.  Host logic only; every ray is produced by the HIP kernels (kernels.hip / kolb_fast.hip).
//
//   zoic_camera_create   <- node_initialize  zoic.cpp:1565-1572
//   zoic_camera_update   <- node_update      zoic.cpp:1575-1720
//   zoic_create_rays_*   <- camera_create_ray zoic.cpp:1752-1990
//   zoic_camera_destroy  <- node_finish      zoic.cpp:1723-1749
//
// Threading (the reference's contract): node_initialize / node_update / node_finish run on one thread with no
// ray call in flight; camera_create_ray is called concurrently from every render thread.  Accordingly every
// zoic_create_rays_* / zoic_camera_create_ray entry point may be called from any number of host threads on ONE camera:
//   * a kernel launch takes a LaunchSlot (its own set of work cursors, guarded by an event: a slot is handed to the next
//     launch only behind the previous user's completion, device side, without blocking the host);
//   * the host-buffer entry points lease a CallContext (two private HIP streams, private device + pinned scratch) for the
//     duration of the call, synchronise only their own streams, and never touch camera-wide scratch;
//   * the per-sample adapter keeps one retry stream per Arnold thread id (tid 0 = the reference's own global xor128 state).
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/zoic_amd.h"
#include "kernels.hpp"
#include "mailbox.hpp"
#include "lens_system.hpp"
#include "host_util.hpp"

#pragma STDC FP_CONTRACT OFF

using namespace zoic;

struct zoic_tile;

namespace {

thread_local std::string g_lastError;

constexpr unsigned kLaunchSlots = 64;   // launches of one camera that may be in flight at once without waiting on each other
constexpr unsigned kCursorStride = kCursorParts * kCursorPartStride;  // one launch's set of partition cursors (kernels.hpp)
// (a >2^31-sample call splits into several launches that reuse ONE slot: they are ordered on the caller's stream)

zoic_status fail(zoic_status s, const std::string &msg)
{
    g_lastError = msg;
    return s;
}
}  // namespace
namespace zoic {
zoic_status fail_status(zoic_status s, const std::string &msg) { return fail(s, msg); }   // frame.cpp reports through the same thread-local text
}
namespace {

#define ZOIC_HIP(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ZOIC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

struct OwnedParams {  // struct cameraParams, zoic.cpp:544-612
    zoic_params p{};
    std::string bokehPath, lensDataPath;
    bool valid = false;
};

// page-locked host memory, mapped into the device's address space (zero-copy for the per-sample adapter, async D2H target
// for the Arnold-layout batch path)
struct PinnedBuffer {
    void *host = nullptr, *dev = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        hipError_t e = hipHostMalloc(&host, bytes, hipHostMallocMapped);
        if (e != hipSuccess) { host = nullptr; return e; }
        e = hipHostGetDevicePointer(&dev, host, 0);
        if (e != hipSuccess) { (void)hipHostFree(host); host = dev = nullptr; return e; }
        cap = bytes;
        return hipSuccess;
    }
    void release() { if (host) (void)hipHostFree(host); host = dev = nullptr; cap = 0; }
};

// One launch's work cursors.  `done` is recorded behind the launch that used the slot last; the next launch to draw the
// slot makes its stream wait on it (hipStreamWaitEvent), so cursors are never reset under a running kernel however many
// launches are in flight -- with up to kLaunchSlots of them no launch waits at all.
struct LaunchSlot {
    std::mutex m;                 // held only while a launch is being enqueued
    unsigned int *cursor = nullptr;     // TWO cursor blocks: a Kolb launch works on block `parity` and clears the other (kernels.hpp)
    unsigned parity = 0;
    bool cursorsDirty = false;          // a thin-lens launch or a failed launch left a block non-zero: memset both before the next Kolb launch
    hipEvent_t done = nullptr;
    bool recorded = false;
    hipStream_t lastStream = nullptr;   // a launch on the same stream is ordered behind the previous one: no event wait needed
    DeviceBuffer<uint32_t> redo;        // the Kolb launch's scratch (kolb_scratch_dwords): work list of decision-safe FAST
};

// Private scratch of ONE host-buffer call in flight: leased from the camera's pool for the duration of the call.
// Two streams / two buffer sets: piece k+1 is copied in while piece k is traced and copied out.
struct CallContext {
    // Three streams: copy-in, kernels, copy-out.  PCIe is full duplex and the copy engines run beside the compute units,
    // so with the three stages of consecutive pieces chained only through events (and two buffer sets) the call runs at
    // the rate of its slowest stage -- normally the 32 B/ray copy-out.
    hipStream_t sIn = nullptr, sRun = nullptr, sOut = nullptr;
    hipEvent_t inDone[2] = {nullptr, nullptr};    // piece's samples are in dSamples[b]
    hipEvent_t runDone[2] = {nullptr, nullptr};   // piece's kernels have finished (dSamples[b] free, dRays[b] full)
    hipEvent_t outDone[2] = {nullptr, nullptr};   // piece's records have left dRays[b] (landed in the caller's / hRays[b] memory)
    DeviceBuffer<float> dSamples[2], dInputs7[2];
    DeviceBuffer<RayRecord> dRays[2];
    DeviceBuffer<uint32_t> dRng[2];
    DeviceBuffer<float> dOut21[2];               // Arnold-layout path: the records expanded to AtCameraOutput rows (21 floats)
    hipError_t init()
    {
        hipError_t e = hipStreamCreateWithFlags(&sIn, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&sRun, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&sOut, hipStreamNonBlocking);
        for (int b = 0; b < 2 && e == hipSuccess; ++b) {
            e = hipEventCreateWithFlags(&inDone[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&runDone[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&outDone[b], hipEventDisableTiming);
        }
        return e;
    }
    hipError_t sync_all()
    {
        hipError_t e = hipSuccess;
        for (hipStream_t st : {sIn, sRun, sOut}) {
            const hipError_t x = st ? hipStreamSynchronize(st) : hipSuccess;
            if (x != hipSuccess && e == hipSuccess) e = x;
        }
        return e;
    }
    void release()
    {
        (void)sync_all();
        for (hipStream_t *st : {&sIn, &sRun, &sOut}) if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
        for (int b = 0; b < 2; ++b) {
            for (hipEvent_t *ev : {&inDone[b], &runDone[b], &outDone[b]}) if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; }
            dSamples[b].release(); dInputs7[b].release(); dRays[b].release(); dRng[b].release(); dOut21[b].release();
        }
    }
};

// The mailbox of the resident kernel (mailbox.hip): header + 64 request lines + 64 reply lines + the tiles' per-batch completion flags in ONE mapped,
// page-locked allocation; the launch's device-side state (what each slot has answered, the control block, the tile jobs: they
// survive its retirements); a private stream.
struct Mailbox {
    std::mutex launchM;                        // launch / stop of the resident kernel
    PinnedBuffer mem;
    MailDeviceState *dState = nullptr;
    hipStream_t stream = nullptr;
    std::mutex slotM[kMailSlots];              // one call per slot at a time (tids 64 apart share a slot)
    uint32_t seq[kMailSlots] = {};
    uint32_t tileSeq[kMailSlots] = {};         // the tile in flight on the slot (0: none); under slotM
    uint32_t tileBatches[kMailSlots] = {}, tileSeen[kMailSlots] = {};   // ... its 64-sample batches, and how many of their flags have been seen
    std::atomic<uint32_t> slotsInUse{0};       // slots the resident launch watches; written under launchM (0: not initialised)
    std::atomic<uint32_t> workerGroups{0};     // tile worker workgroups of the resident launch (0 until the camera sees its first tile)
    zoic_tile *ownTile[kMailSlots][2] = {};    // zoic_camera_create_rays_tile's staging for callers' pageable arrays (under slotM)
    volatile MailHeader *header() const { return static_cast<volatile MailHeader *>(mem.host); }
    volatile MailRequest *request(unsigned slot) const { return reinterpret_cast<volatile MailRequest *>(static_cast<char *>(mem.host) + kMailRequestsOffset) + slot; }
    volatile MailReply *reply(unsigned slot) const { return reinterpret_cast<volatile MailReply *>(static_cast<char *>(mem.host) + kMailRepliesOffset) + slot; }
    volatile uint32_t *tile_flags(unsigned slot) const { return reinterpret_cast<volatile uint32_t *>(static_cast<char *>(mem.host) + kMailTileFlagsOffset) + static_cast<size_t>(slot) * kTileMaxBatches; }
    // every batch of the slot's tile has reported (non-blocking; remembers how far it got)
    bool tile_complete(unsigned slot)
    {
        volatile uint32_t *f = tile_flags(slot);
        uint32_t &seen = tileSeen[slot];
        while (seen < tileBatches[slot] && f[seen] == tileSeq[slot]) ++seen;
        return seen == tileBatches[slot];
    }
    hipError_t init()
    {
        if (mem.host && dState && stream) return hipSuccess;
        // all or nothing: a half-built mailbox (no device state, the null stream) must never reach a launch -- the next call retries
        hipError_t e = mem.reserve(kMailBytes);
        if (e == hipSuccess) {
            std::memset(mem.host, 0, mem.cap);
            if (!dState) e = hipMalloc(reinterpret_cast<void **>(&dState), sizeof(MailDeviceState));
        }
        if (e == hipSuccess && !stream) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        // The state is zeroed ON THE RESIDENT KERNEL'S OWN STREAM and waited for.  (hipMemset on device memory returns before the fill has run, and it ran
        // on the null stream, which a hipStreamNonBlocking stream does not wait for: round 5's order -- memset, then the stream, then the launch -- let the
        // fill land AFTER the first tile's descriptor, ticket counters and wake lines had been posted; the workers then drew tickets of generation 0, the
        // tile's batches were never handed out and the render thread waited its 20 s.  One full `-m gpu` run in five: found with the time-out's dump,
        // "tile 1 with 256 batches of which 1 flagged".)
        if (e == hipSuccess) e = hipMemsetAsync(dState, 0, sizeof(MailDeviceState), stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
            if (dState) { (void)hipFree(dState); dState = nullptr; }
            mem.release();
        }
        return e;
    }
    // the resident kernel retires (adding its ray counters to the camera's) and nothing of it is left in flight; a tile it was
    // working on is finished first (mailbox.hip: no wave leaves with a batch in hand)
    hipError_t stop()
    {
        std::lock_guard<std::mutex> lk(launchM);
        if (!mem.host || !stream) return hipSuccess;
        request(0)->stop = 1u;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        const hipError_t e = hipStreamSynchronize(stream);
        request(0)->stop = 0u;
        header()->alive = 0u;
        return e;
    }
    void release();
};

// camera_create_ray's `tid` (zoic.cpp:1752): one retry stream per render thread, advanced by every call that retries.
struct TidState {
    std::mutex m;   // Arnold never runs two samples of one tid at once; a caller that does is serialised, not corrupted
    Rng rng{};
};
constexpr unsigned kTidStates = 65536;   // uint16_t tid

}  // namespace

struct zoic_camera {  // struct cameraData, zoic.cpp:627-643
    int device = 0;
    OwnedParams params;            // camera->params (last applied)
    LensSystem lens;               // camera->lens
    BokehCdf image;                // camera->image
    float fov = 0, tanFov = 0, apertureRadius = 0;
    Rng stream{};                  // the xor128 function-static state (zoic.cpp:648): LUT build draws from it
    zoic_precision precision = ZOIC_PRECISION_STRICT;
    bool fastVerdict = false, fastVerdictValid = false;   // fast_self_check's answer for the tables the camera holds now
    float frameMaxSy = 1.0f;  // zoic_camera_set_frame_aspect: the self-check's probe lattice covers sy in [-frameMaxSy, frameMaxSy]
    bool fastDomain = true;   // RAYTRACED: the lens is inside the FAST modes' domain (include/zoic_amd.h, zoic_precision); else every mode runs STRICT
    // kernel mode of a launch: 0 = STRICT, 1 = FAST decision-safe, 2 = FAST unchecked
    int kernel_mode() const
    {
        if (precision == ZOIC_PRECISION_STRICT || (params.p.lensModel == ZOIC_RAYTRACED && !fastDomain)) return 0;
        return precision == ZOIC_PRECISION_FAST ? 1 : 2;
    }
    uint32_t seed = 1;
    uint32_t tirInCounters = 0;    // the precompute's TIR bumps the counters currently include (zoic_lens_info::precomputeTIR)
    bool updated = false;
    bool lutOnHost = false, lutHostDraws = false;
    // pending inputs; the dirty flags force the rebuild on the next update even under an unchanged path
    std::vector<float> pendingPixels; int pendW = 0, pendH = 0, pendC = 0;
    std::string lensText; bool haveLensText = false;
    bool bokehDirty = false, lensDirty = false;
    // flattened tables
    KolbTable kolb{};
    ThinTable thin{};
    // device state
    DeviceBuffer<float> dCdfRow, dCdfColumn, dPyramid;
    DeviceBuffer<uint32_t> dBokehCells;
    DeviceBuffer<int32_t> dRowIdx, dColIdx;
    BokehTables bokehDev{};
    DeviceCounters *dCounters = nullptr;
    DeviceBuffer<float> dProbeU, dProbeV;   // node_update scratch (single-threaded by contract)
    DeviceBuffer<float> dFastProbe;         // fast_self_check: its samples ...
    DeviceBuffer<RayRecord> dFastProbeRays; // ... and the STRICT + FAST records of them
    bool fastProbeReady = false;
    DeviceBuffer<uint8_t> dProbeOk;
    unsigned int *dProbeTir = nullptr;
    // ---- concurrent ray calls (see the threading note at the top of this file)
    unsigned int *dWorkCursor = nullptr;    // kLaunchSlots sets of partition cursors
    LaunchSlot slots[kLaunchSlots];
    std::atomic<unsigned> nextSlot{0};
    std::mutex poolM;
    std::vector<CallContext *> freeContexts;
    std::vector<std::unique_ptr<CallContext>> contexts;
    Mailbox mail;                           // camera_create_ray per sample (mailbox.hip)
    std::unique_ptr<std::atomic<TidState *>[]> tidStates;   // kTidStates entries, created on first use
    std::mutex tidCreateM;
    // the caller's tiles (zoic_tile_create): zoic_camera_destroy settles and DETACHES the ones still alive -- their page-locked arrays go
    // with the camera, the handles stay valid for zoic_tile_destroy and answer every other call with an error
    std::mutex tilesM;
    std::vector<zoic_tile *> liveTiles;
    std::atomic<int> waitMode{ZOIC_WAIT_SPIN};   // zoic_camera_set_wait_mode: how a render thread waits for the resident kernel

    TidState *tid_state(uint16_t tid);
    CallContext *lease_context(hipError_t &err);
    void return_context(CallContext *c);
};

TidState *zoic_camera::tid_state(uint16_t tid)
{
    std::atomic<TidState *> &slot = tidStates[tid];
    TidState *t = slot.load(std::memory_order_acquire);
    if (t) return t;
    std::lock_guard<std::mutex> lk(tidCreateM);
    t = slot.load(std::memory_order_relaxed);
    if (!t) {
        t = new TidState();
        // tid 0 is never seeded here: it IS the camera's `stream` (the reference's global xor128 state), so a single
        // render thread reproduces the reference's sequential output; the other threads get streams of their own
        t->rng = rng_for_ray(seed, (0xA7100000ull | tid) << 32);
        slot.store(t, std::memory_order_release);
    }
    return t;
}

CallContext *zoic_camera::lease_context(hipError_t &err)
{
    err = hipSuccess;
    {
        std::lock_guard<std::mutex> lk(poolM);
        if (!freeContexts.empty()) { CallContext *c = freeContexts.back(); freeContexts.pop_back(); return c; }
    }
    std::unique_ptr<CallContext> fresh(new CallContext());   // as many contexts as calls were ever in flight at once
    err = fresh->init();
    if (err != hipSuccess) { fresh->release(); return nullptr; }
    CallContext *c = fresh.get();
    std::lock_guard<std::mutex> lk(poolM);
    contexts.push_back(std::move(fresh));
    return c;
}

void zoic_camera::return_context(CallContext *c)
{
    std::lock_guard<std::mutex> lk(poolM);
    freeContexts.push_back(c);
}

// zoic_tile (include/zoic_amd.h): a render thread's bucket of samples -- page-locked AtCameraInput / AtCameraOutput arrays the GPU
// reads and writes in place -- bound to the mailbox slot of its tid.
struct zoic_tile {
    zoic_camera *cam = nullptr;
    uint16_t tid = 0;
    unsigned slot = 0;
    uint32_t capacity = 0;
    PinnedBuffer mem;                 // [capacity x 28 B inputs][pad to 64][capacity x 84 B outputs]
    zoic_camera_input *inputs = nullptr;
    zoic_camera_output *outputs = nullptr;
    uint64_t dIn = 0, dOut = 0;       // the same arrays as the device sees them
    uint32_t seq = 0;                 // the submit not waited for yet (0: none); under the slot's mutex
    uint32_t polls = 0;               // zoic_tile_done calls since the submit (every 1024th asks the stream whether the kernel still lives)
    bool registered = false;          // listed in cam->liveTiles (the caller's tiles; the slots' own staging tiles are not)
    int rows = ZOIC_TILE_ROWS_ARNOLD; // what the kernel writes at dOut: AtCameraOutput rows or zoic_ray records
    int ins = ZOIC_TILE_INPUTS_ARNOLD; // what the caller writes at dIn: AtCameraInput rows or (sx, sy, lensx, lensy) samples
};

void Mailbox::release()
{
    (void)stop();
#ifdef ZOIC_TILE_TIMING   // (tools/: -DZOIC_TILE_TIMING builds print where a tile batch's time went when the camera is destroyed)
    if (dState) {
        unsigned long long t[32];
        if (hipMemcpy(t, dState->timing, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
            if (t[4]) std::fprintf(stderr, "rays per batch < 8 / 12 / 16 / 24 / more us: %llu %llu %llu %llu %llu; start < 6 / 8 / 12 / more us after the request: %llu %llu %llu %llu\n", t[16], t[17], t[18], t[19], t[20], t[24], t[25], t[26], t[27]);
            if (t[4])
                std::fprintf(stderr, "worker batches since the slot wave saw the request: start avg %.2f us (max %.2f), flag written avg %.2f us (max %.2f)\n",
                             t[8] * 0.01 / t[4], t[9] * 0.01, t[10] * 0.01 / t[4], t[11] * 0.01);
            if (t[4]) std::fprintf(stderr, "worker batches that started > 8 us after the request: %llu (mean batch index %.1f, %llu of them NOT the first batch after a wake-up); flagged > 25 us: %llu\n", t[12],
                                   t[12] ? double(t[13]) / t[12] : 0.0, t[14], t[15]);
            unsigned long long g[16] = {};
            if (read_tile_dbg(g) == 0 && g[10])
                std::fprintf(stderr, "RAYTRACED batches by rounds 0..7+: %llu %llu %llu %llu %llu %llu %llu %llu; with a listed ray: %llu (listed part %.2f us each); rounds part %.2f us per batch; "
                             "open after round 0: %.2f rays per batch; batches with more than 8 open after round 0: %llu (their rounds part: %.2f us each)\n",
                             g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[8] ? g[9] * 0.01 / g[8] : 0.0,
                             g[10] * 0.01 / (g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] ? g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] : 1),
                             double(g[12]) / (g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] ? g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] : 1), g[13], g[13] ? g[14] * 0.01 / g[13] : 0.0);
            for (int o = 0; o < 8; o += 4)
                if (t[o])
                    std::fprintf(stderr, "tile timing (%s waves): %llu batches, input %.2f us, rays %.2f us, output %.2f us per batch\n", o ? "worker" : "slot", t[o],
                                 t[o + 1] * 0.01 / t[o], t[o + 2] * 0.01 / t[o], t[o + 3] * 0.01 / t[o]);
        }
    }
#endif
    for (auto &pair : ownTile) for (zoic_tile *&t : pair) if (t) { t->mem.release(); delete t; t = nullptr; }
    if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    if (dState) { (void)hipFree(dState); dState = nullptr; }
    mem.release();
}

namespace {

bool params_lens_changed(const zoic_params &n, const OwnedParams &o)  // cameraParams::lensChanged, zoic.cpp:595-606
{
    if (!o.valid) return true;
    const zoic_params &r = o.p;
    const std::string nb = n.bokehPath ? n.bokehPath : "", nl = n.lensDataPath ? n.lensDataPath : "";
    return n.sensorWidth != r.sensorWidth || n.sensorHeight != r.sensorHeight || n.focalLength != r.focalLength ||
           n.fStop != r.fStop || n.focalDistance != r.focalDistance || (n.useImage != 0) != (r.useImage != 0) ||
           (n.useImage && nb != o.bokehPath) || n.lensModel != r.lensModel ||
           (n.lensModel == ZOIC_RAYTRACED && (nl != o.lensDataPath || (n.kolbSamplingLUT != 0) != (r.kolbSamplingLUT != 0)));
}

bool params_bokeh_changed(const zoic_params &n, const OwnedParams &o)  // cameraParams::bokehChanged, zoic.cpp:608-611
{
    const bool oldUse = o.valid && o.p.useImage != 0;
    const std::string nb = n.bokehPath ? n.bokehPath : "";
    return (n.useImage != 0) != oldUse || (n.useImage && nb != o.bokehPath);
}

zoic_status lens_error_status(LensError e)
{
    switch (e) {
    case LensError::None: return ZOIC_OK;
    case LensError::Columns: return fail(ZOIC_ERR_LENS_COLUMNS, "[ZOIC] Failed to read lens data file: need 4 or 5 columns of data");
    case LensError::Parse: return fail(ZOIC_ERR_LENS_PARSE, "[ZOIC] Failed to read lens data file: token is not a number");
    case LensError::MultiAperture: return fail(ZOIC_ERR_MULTI_APERTURE, "[ZOIC] Multiple apertures found. Provide lens description with 1 aperture.");
    case LensError::NoAperture: return fail(ZOIC_ERR_NO_APERTURE, "[ZOIC] No aperture row (radius 0) in the lens description");
    case LensError::TooManySurfaces: return fail(ZOIC_ERR_TOO_MANY_LENSES, "[ZOIC] More lens surfaces than ZOIC_MAX_LENS_SURFACES");
    }
    return ZOIC_ERR_INVALID_ARGUMENT;
}

// LutTraceFn backed by the GPU probe kernel (same strict arithmetic as the host tracer)
void lut_trace_device(const KolbTable &table, float originX, const float *lensU, const float *lensV, size_t n, uint8_t *accepted,
                      uint32_t *tirCount, void *user)
{
    zoic_camera *cam = static_cast<zoic_camera *>(user);
    bool ok = cam->dProbeU.reserve(n) == hipSuccess && cam->dProbeV.reserve(n) == hipSuccess && cam->dProbeOk.reserve(n) == hipSuccess;
    if (ok && !cam->dProbeTir) ok = hipMalloc(reinterpret_cast<void **>(&cam->dProbeTir), sizeof(unsigned int)) == hipSuccess;
    unsigned int tir = 0;
    ok = ok && hipMemcpy(cam->dProbeU.ptr, lensU, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(cam->dProbeV.ptr, lensV, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(cam->dProbeTir, 0, sizeof(unsigned int)) == hipSuccess;
    ok = ok && launch_lut_probes(table, originX, cam->dProbeU.ptr, cam->dProbeV.ptr, n, cam->dProbeOk.ptr, cam->dProbeTir, nullptr) == 0;
    ok = ok && hipMemcpy(accepted, cam->dProbeOk.ptr, n, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(&tir, cam->dProbeTir, sizeof(tir), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {  // surfaced by zoic_camera_update through g_lastError; accept nothing so the failure is visible
        g_lastError = "exit-pupil LUT probe kernel failed";
        std::memset(accepted, 0, n);
        return;
    }
    *tirCount += tir;
}

// LutBuildFn: the whole exit-pupil LUT on the GPU (lut_build.hip)
int lut_build_whole_device(const KolbTable &table, Rng &rng, LutBox boxes[kLutEntries], uint32_t *tirCount, void *)
{
    return build_lut_device(table, rng, boxes, tirCount);
}

// one padded pyramid: level 0 = the CDF, level j+1 = last element of each 16-chunk of level j (tables.hpp)
struct Pyramid {
    int levels = 0;
    int count[kBokehMaxLevels] = {0, 0, 0}, stride[kBokehMaxLevels] = {0, 0, 0};
};

Pyramid pyramid_shape(int n)
{
    Pyramid p;
    int cnt = n;
    for (int j = 0; j < kBokehMaxLevels; ++j) {
        p.count[j] = cnt;
        p.stride[j] = (cnt + 15) / 16 * 16;
        p.levels = j + 1;
        if (cnt <= 16) return p;
        cnt = (cnt + 15) / 16;
    }
    p.levels = 0;  // more than 16^3 entries: not covered, the kernels binary-search the plain CDF
    return p;
}

// append the padded levels of one CDF (n floats) to `dst`; returns per-level offsets
void pyramid_fill(const float *cdf, const Pyramid &shape, float *const levelBase[kBokehMaxLevels], size_t rowIndex)
{
    const float inf = INFINITY;
    for (int j = 0; j < shape.levels; ++j) {
        float *dst = levelBase[j] + rowIndex * static_cast<size_t>(shape.stride[j]);
        const float *src = j == 0 ? cdf : levelBase[j - 1] + rowIndex * static_cast<size_t>(shape.stride[j - 1]);
        for (int i = 0; i < shape.stride[j]; ++i) {
            if (i >= shape.count[j]) dst[i] = inf;
            else if (j == 0) dst[i] = src[i];
            else dst[i] = src[std::min(16 * i + 15, shape.count[j - 1] - 1)];
        }
    }
}

zoic_status upload_bokeh(zoic_camera *cam)
{
    const BokehCdf &im = cam->image;
    cam->bokehDev = BokehTables{};
    if (!im.valid()) return ZOIC_OK;
    const size_t y = static_cast<size_t>(im.y), xy = static_cast<size_t>(im.x) * im.y;
    ZOIC_HIP(cam->dCdfRow.reserve(y));
    ZOIC_HIP(cam->dRowIdx.reserve(y));
    ZOIC_HIP(cam->dCdfColumn.reserve(xy));
    ZOIC_HIP(cam->dColIdx.reserve(xy));
    ZOIC_HIP(hipMemcpy(cam->dCdfRow.ptr, im.cdfRow.data(), y * sizeof(float), hipMemcpyHostToDevice));
    ZOIC_HIP(hipMemcpy(cam->dRowIdx.ptr, im.rowIndices.data(), y * sizeof(int32_t), hipMemcpyHostToDevice));
    ZOIC_HIP(hipMemcpy(cam->dCdfColumn.ptr, im.cdfColumn.data(), xy * sizeof(float), hipMemcpyHostToDevice));
    ZOIC_HIP(hipMemcpy(cam->dColIdx.ptr, im.columnIndices.data(), xy * sizeof(int32_t), hipMemcpyHostToDevice));
    BokehTables &B = cam->bokehDev;
    B.cdfRow = cam->dCdfRow.ptr; B.rowIndices = cam->dRowIdx.ptr; B.cdfColumn = cam->dCdfColumn.ptr; B.columnIndices = cam->dColIdx.ptr;
    // The pyramid search (count of entries <= u) and the cell records equal std::upper_bound only on non-decreasing,
    // NaN-free CDFs.  Negative or NaN luminance (HDR images) can break that: such tables keep the reference's own binary
    // search (levels == 0, no cell records), which visits the same elements as std::upper_bound on any input.
    const auto monotone = [](const float *a, int n) {
        for (int i = 0; i < n; ++i) if (!(a[i] == a[i]) || (i > 0 && a[i] < a[i - 1])) return false;
        return true;
    };
    bool sorted = monotone(im.cdfRow.data(), im.y);
    for (size_t r = 0; sorted && r < y; ++r) sorted = monotone(im.cdfColumn.data() + r * im.x, im.x);
    if (!sorted) return ZOIC_OK;
    // search pyramids (device layout only; same numbers as the reference tables)
    const Pyramid rp = pyramid_shape(im.y), cp = pyramid_shape(im.x);
    if (rp.levels == 0 || cp.levels == 0) return ZOIC_OK;
    const int levels = std::max(rp.levels, cp.levels);  // the kernel walks `levels` levels for both: pad the shorter one
    Pyramid rshape = rp, cshape = cp;
    for (int j = 0; j < levels; ++j) {
        if (j >= rp.levels) { rshape.count[j] = 1; rshape.stride[j] = 16; }
        if (j >= cp.levels) { cshape.count[j] = 1; cshape.stride[j] = 16; }
    }
    rshape.levels = cshape.levels = levels;
    size_t total = 0, rowOff[kBokehMaxLevels], colOff[kBokehMaxLevels];
    for (int j = 0; j < levels; ++j) { rowOff[j] = total; total += rshape.stride[j]; }
    for (int j = 0; j < levels; ++j) { colOff[j] = total; total += static_cast<size_t>(cshape.stride[j]) * y; }
    std::vector<float> host(total);
    float *rbase[kBokehMaxLevels] = {nullptr, nullptr, nullptr}, *cbase[kBokehMaxLevels] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < levels; ++j) { rbase[j] = host.data() + rowOff[j]; cbase[j] = host.data() + colOff[j]; }
    pyramid_fill(im.cdfRow.data(), rshape, rbase, 0);
    for (size_t r = 0; r < y; ++r) pyramid_fill(im.cdfColumn.data() + r * im.x, cshape, cbase, r);
    ZOIC_HIP(cam->dPyramid.reserve(total));
    ZOIC_HIP(hipMemcpy(cam->dPyramid.ptr, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    for (int j = 0; j < levels; ++j) {
        B.rowLevel[j] = cam->dPyramid.ptr + rowOff[j];
        B.colLevel[j] = cam->dPyramid.ptr + colOff[j];
        B.colStride[j] = cshape.stride[j];
        B.rowCount[j] = rshape.count[j];
        B.colCount[j] = cshape.count[j];
    }
    B.levels = levels;
    // cell records (tables.hpp): rows <= 2048 (the row records live in LDS: 32 KB at most), columns <= 4096
    if (im.x <= 4096 && im.y <= 2048) {
        {
            const auto cellCount = [](int n) { int g = 16; while (g < n) g <<= 1; return g; };
            const int gRow = cellCount(im.y), gCol = cellCount(im.x);
            const size_t nCells = static_cast<size_t>(gRow) + y * static_cast<size_t>(gCol);
            // records (4 dwords each), then the bounds (1 dword each); built on the GPU from the tables uploaded above
            // (one lane per cell), ZOIC_CELLS_HOST=1 keeps the host build for A/B -- both produce identical words
            ZOIC_HIP(cam->dBokehCells.reserve(nCells * 5));
            const char *envHost = std::getenv("ZOIC_CELLS_HOST");
            if (!(envHost && envHost[0] == '1')) {
                const int rc = build_bokeh_cells_device(cam->dCdfRow.ptr, cam->dRowIdx.ptr, cam->dCdfColumn.ptr, cam->dColIdx.ptr, im.x, im.y,
                                                        gRow, gCol, cam->dBokehCells.ptr);
                if (rc != 0) { g_lastError = "bokeh cell-record kernel failed"; return ZOIC_ERR_HIP; }
            } else {
                std::vector<uint32_t> cells(nCells * 5);
                uint32_t *bounds = cells.data() + nCells * 4;
                // records of one CDF: cdf[n] non-decreasing, idx[n] pixel indices relative to `idxBase` (each < 65536)
                const auto fill = [](const float *cdf, const int32_t *idx, int32_t idxBase, int n, int g, uint32_t *rec, uint32_t *bnd) {
                    int lo = 0, hi = 0;                            // both only move forward as the cell edge grows
                    for (int c = 0; c < g; ++c, rec += 4, ++bnd) {
                        const float lower = static_cast<float>(c) / static_cast<float>(g), upper = static_cast<float>(c + 1) / static_cast<float>(g);
                        while (lo < n && cdf[lo] <= lower) ++lo;   // lo = #{cdf <= lower}
                        if (hi < lo) hi = lo;
                        while (hi < n && cdf[hi] < upper) ++hi;    // hi = #{cdf <  upper}
                        // the first two DISTINCT values above the lower edge (a run of equal values is one decision), as build_cells_kernel
                        const float inf = INFINITY;
                        const float a = lo < n ? cdf[lo] : inf;
                        const int j = lo < n ? lo + upper_bound_idx(cdf + lo, n - lo, a) : n;
                        const float b = j < n ? cdf[j] : inf;
                        const int k = j < n ? j + upper_bound_idx(cdf + j, n - j, b) : n;
                        const int e[3] = {std::min(lo, n - 1), std::min(j, n - 1), std::min(k, n - 1)};
                        uint32_t id[3];
                        for (int q = 0; q < 3; ++q) id[q] = static_cast<uint32_t>(idx[e[q]] - idxBase) & 0xffffu;
                        std::memcpy(rec + 0, &a, 4);
                        std::memcpy(rec + 1, &b, 4);
                        rec[2] = id[0] | (id[1] << 16);
                        rec[3] = id[2] | ((k < hi) ? 0x80000000u : 0u);
                        *bnd = static_cast<uint32_t>(lo) | (static_cast<uint32_t>(hi) << 16);
                    }
                };
                fill(im.cdfRow.data(), im.rowIndices.data(), 0, im.y, gRow, cells.data(), bounds);
                for (size_t r = 0; r < y; ++r)
                    fill(im.cdfColumn.data() + r * im.x, im.columnIndices.data() + r * im.x, static_cast<int32_t>(r * im.x), im.x, gCol,
                         cells.data() + (static_cast<size_t>(gRow) + r * gCol) * 4, bounds + gRow + r * gCol);
                ZOIC_HIP(hipMemcpy(cam->dBokehCells.ptr, cells.data(), cells.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
            B.rowCells = cam->dBokehCells.ptr;
            B.colCells = cam->dBokehCells.ptr + static_cast<size_t>(gRow) * 4;
            B.rowBounds = cam->dBokehCells.ptr + nCells * 4;
            B.colBounds = B.rowBounds + gRow;
            B.ldsWords = gRow * 4;
            B.rowCellCount = gRow;
            B.colCellCount = gCol;
        }
    }
    return ZOIC_OK;
}

BokehTables bokeh_tables(const zoic_camera *cam) { return cam->bokehDev; }

void exposure_terms(float exposureControl, float &mul, int32_t &on)  // zoic.cpp:1981-1987
{
    const float e2 = exposureControl * exposureControl;
    on = 0; mul = 1.0f;
    if (exposureControl > 0.0f) { on = 1; mul = 1.0f + e2; }
    else if (exposureControl < 0.0f) { on = 1; mul = 1.0f / (1.0f + e2); }
}

class ContextLease {   // a CallContext for the duration of one host-buffer call
    zoic_camera *cam_;
    CallContext *ctx_;
    hipError_t err_ = hipSuccess;
public:
    explicit ContextLease(zoic_camera *cam) : cam_(cam), ctx_(cam->lease_context(err_)) {}
    ~ContextLease() { if (ctx_) cam_->return_context(ctx_); }
    explicit operator bool() const { return ctx_ != nullptr; }
    hipError_t error() const { return err_; }
    CallContext &operator*() const { return *ctx_; }
    ContextLease(const ContextLease &) = delete;
    ContextLease &operator=(const ContextLease &) = delete;
};

zoic_status check_ray_call(const zoic_camera *cam)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device == ZOIC_DEVICE_NONE) return fail(ZOIC_ERR_NO_DEVICE, "tables-only camera: rays need a gfx950 device (no CPU path)");
    if (!cam->updated) return fail(ZOIC_ERR_NOT_UPDATED, "zoic_camera_update has not succeeded yet");
    return ZOIC_OK;
}

// One launch of camera_create_ray over n samples on `stream`, asynchronous.  Safe to call from many host threads at once.
// modeOverride (0 STRICT / 1 decision-safe FAST / 2 unchecked; -1: the camera's) and counted = false (no counter is touched)
// serve node_update's self-check of the FAST modes.
zoic_status launch_rays(zoic_camera *cam, uint64_t n, const float *d_samples, const uint32_t *d_rng, uint64_t rayBase, RayRecord *d_rays,
                        hipStream_t stream, int modeOverride = -1, bool counted = true)
{
    static_assert(sizeof(zoic_ray) == sizeof(RayRecord), "zoic_ray layout");
    const int model = cam->params.p.lensModel;
    if (model != ZOIC_RAYTRACED && model != ZOIC_THINLENS)
        return fail(ZOIC_ERR_INVALID_ARGUMENT, "lensModel NONE produces no rays (zoic.cpp:1966-1968)");
    const int mode = modeOverride >= 0 ? modeOverride : cam->kernel_mode();
    DeviceCounters *const dCounters = counted ? cam->dCounters : nullptr;
    // the Kolb launch's scratch (kernels.hpp): decision-safe FAST's work list, the finish kernel's byte map
    const size_t listEntries = model == ZOIC_RAYTRACED ? kolb_scratch_dwords(cam->kolb, n, mode) : 0;
    const bool needList = listEntries != 0;
    // Slot choice.  A caller that keeps launching on one stream keeps ONE slot (its launches are ordered anyway, and the
    // slot's work-list buffer is allocated once); otherwise the first idle slot; with all 64 busy, the next in turn,
    // behind its previous user's completion event.
    LaunchSlot *slot = nullptr;
    std::unique_lock<std::mutex> lk;
    for (unsigned pass = 0; pass < 2 && !slot; ++pass)
        for (unsigned i = 0; i < kLaunchSlots && !slot; ++i) {
            LaunchSlot &c = cam->slots[i];
            std::unique_lock<std::mutex> t(c.m, std::try_to_lock);
            if (!t.owns_lock()) continue;
            const bool mine = c.recorded && c.lastStream == stream;
            if (pass == 0 ? mine : (!c.recorded || hipEventQuery(c.done) == hipSuccess)) { slot = &c; lk = std::move(t); }
        }
    if (!slot) {
        slot = &cam->slots[cam->nextSlot.fetch_add(1u, std::memory_order_relaxed) % kLaunchSlots];
        lk = std::unique_lock<std::mutex>(slot->m);
    }
    (void)hipGetLastError();   // hipEventQuery's hipErrorNotReady is not an error of this call
// always through the event -- a stream handle can be destroyed and its address reused; on the stream that recorded it the
    // wait is free (measured: no difference on the 8.3 M-ray thin-lens frame or a 1 M-ray Kolb bucket)
    if (slot->recorded) ZOIC_HIP(hipStreamWaitEvent(stream, slot->done, 0));
    if (needList && slot->redo.cap < listEntries) {
        // growing the scratch frees the old one: the slot's previous launch must be over (rare: first use / a larger batch)
        if (slot->recorded) ZOIC_HIP(hipEventSynchronize(slot->done));
        ZOIC_HIP(slot->redo.reserve(listEntries));
    }
    int rc;
    if (model == ZOIC_RAYTRACED) {
        if (slot->cursorsDirty) {
            ZOIC_HIP(hipMemsetAsync(slot->cursor, 0, 2 * kCursorStride * sizeof(unsigned int), stream));
            slot->cursorsDirty = false;
        }
        rc = launch_kolb_rays(cam->kolb, cam->bokehDev, d_samples, d_rng, rayBase, n, d_rays, dCounters, slot->cursor, &slot->parity, mode,
                              needList ? slot->redo.ptr : nullptr, stream);
        if (rc != 0) slot->cursorsDirty = true;
    } else {
        rc = launch_thin_rays(cam->thin, cam->bokehDev, d_samples, d_rng, rayBase, n, d_rays, dCounters, slot->cursor, mode != 0, stream);
        slot->cursorsDirty = true;   // the thin-lens kernels reset the block they use themselves and leave it used
    }
    // whatever the launcher returned, part of the launch (cursor reset, the first kernel) may be queued: the slot's next user
    // must wait behind it
    const hipError_t re = hipEventRecord(slot->done, stream);
    if (re == hipSuccess) { slot->recorded = true; slot->lastStream = stream; }
    if (rc != 0) return fail(ZOIC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
    ZOIC_HIP(re);
    return ZOIC_OK;
}

// samples per piece of the host-buffer pipeline: about eight pieces per call, 256 Ki ... 4 Mi samples each
// (a piece must amortise a launch's fixed cost and still leave something to overlap)
uint64_t host_piece(uint64_t n)
{
    uint64_t p = ((n + 7) / 8 + 65535) / 65536 * 65536;
    if (p < (256ull << 10)) p = 256ull << 10;
    if (p > (4ull << 20)) p = 4ull << 20;
    return p;
}

// one 32-byte record -> the AtCameraOutput fields zoic writes (zoic.cpp:1960-1961, 1974-1977, 1952, 1981-1987)
inline void expand_record(const zoic_ray &r, zoic_camera_output &o)
{
    o.origin = zoic_vec3{r.ox, r.oy, r.oz};
    o.dir = zoic_vec3{r.dx, r.dy, r.dz};
    const float w = r.weight;
    if (w == 0.0f) o.weight[0] = o.weight[1] = o.weight[2] = 0.0f;  // output.weight = 0.0f, zoic.cpp:1825/1952
    else if (w != 1.0f) { o.weight[0] *= w; o.weight[1] *= w; o.weight[2] *= w; }  // exposure factor
    if (r.flags & 1u) { o.dOdy = o.origin; o.dDdy = o.dir; }  // zoic.cpp:1974-1977
}

// node_update's self-check of the FAST modes (RAYTRACED): kFastProbeRays samples spread over the frame go through the STRICT and
// the decision-safe FAST kernels; the camera keeps its FAST modes only if FAST decides these rays as STRICT does (one flip at
// most) and its directions are well within north_star's tolerance (RMSE < 5e-6 over the rays with weight, per slab).  FAST drops the
// reference's renormalisations and takes cos(i) from the hit's geometry (fast_optics.hpp): exact on a lens laid out like a
// lens, not on every table of numbers -- a prescription whose elements graze (a fisheye with an element removed: 6e-5) runs
// STRICT instead.  No counter is touched, no retry stream advanced (per-ray streams of the probe's own ray indices).
// Two slabs of kFastProbeRays samples each (two jitters of the lattice), judged separately: the gate is HALF of north_star's
// tolerance on BOTH (round 3 gated one slab at 1e-5 and the deep fuzz found two machine-made lenses at 1.0-1.1e-5 on another
// slab of the frame: profiles/fuzz_deep_r03.log).
// Returns ZOIC_OK with `keep` set, or the HIP failure that kept the check from running (the caller reports it: a camera that
// silently fell back to STRICT because an allocation failed would claim "outside the FAST domain").
constexpr uint32_t kFastProbeRays = 4096, kFastProbeSlabs = 2;
constexpr double kFastProbeRmse = 5.0e-6;
zoic_status fast_self_check(zoic_camera *cam, bool &keep)
{
    keep = false;
    constexpr uint32_t nAll = kFastProbeRays * kFastProbeSlabs;
    ZOIC_HIP(cam->dFastProbe.reserve(nAll * 4));
    ZOIC_HIP(cam->dFastProbeRays.reserve(2 * nAll));
    if (!cam->fastProbeReady) {
        std::vector<float> h(nAll * 4);
        for (uint32_t k = 0; k < nAll; ++k) {   // per slab: a jittered 64 x 64 lattice over sx in [-1, 1], sy in [-frameMaxSy, frameMaxSy]
            const uint32_t i = k % kFastProbeRays;
            const auto u01 = [](uint32_t v) { return static_cast<float>(pcg_hash(v) >> 8) * (1.0f / 16777216.0f); };
            h[4 * k + 0] = ((static_cast<float>(i & 63u) + u01(4 * k)) / 64.0f) * 2.0f - 1.0f;
            h[4 * k + 1] = (((static_cast<float>(i >> 6) + u01(4 * k + 1)) / 64.0f) * 2.0f - 1.0f) * cam->frameMaxSy;
            h[4 * k + 2] = u01(4 * k + 2);
            h[4 * k + 3] = u01(4 * k + 3);
        }
        ZOIC_HIP(hipMemcpy(cam->dFastProbe.ptr, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        cam->fastProbeReady = true;
    }
    RayRecord *d = cam->dFastProbeRays.ptr;
    if (zoic_status s = launch_rays(cam, nAll, cam->dFastProbe.ptr, nullptr, 0, d, nullptr, 0, false)) return s;
    if (zoic_status s = launch_rays(cam, nAll, cam->dFastProbe.ptr, nullptr, 0, d + nAll, nullptr, 1, false)) return s;
    std::vector<RayRecord> r(2 * nAll);
    ZOIC_HIP(hipMemcpy(r.data(), d, r.size() * sizeof(RayRecord), hipMemcpyDeviceToHost));   // orders behind the null stream
    for (uint32_t slab = 0; slab < kFastProbeSlabs; ++slab) {
        uint32_t flips = 0, live = 0;
        double sum = 0.0;
        for (uint32_t i = slab * kFastProbeRays; i < (slab + 1) * kFastProbeRays; ++i) {
            const RayRecord &a = r[i], &b = r[nAll + i];
            if (a.flags != b.flags) { ++flips; continue; }
            if (a.weight == 0.0f || !std::isfinite(a.dx) || !std::isfinite(a.dy) || !std::isfinite(a.dz)) continue;
            const double ex = static_cast<double>(a.dx) - b.dx, ey = static_cast<double>(a.dy) - b.dy, ez = static_cast<double>(a.dz) - b.dz;
            const double e2 = ex * ex + ey * ey + ez * ez;
            if (!(e2 == e2)) return ZOIC_OK;   // FAST made a NaN where STRICT has a direction
            sum += e2;
            ++live;
        }
        if (flips > 1u || (live != 0u && !(std::sqrt(sum / live) < kFastProbeRmse))) return ZOIC_OK;
    }
    keep = true;
    return ZOIC_OK;
}

}  // namespace

static void detach_tiles(zoic_camera *cam);   // (defined with the tile entry points below)

extern "C" {

int zoic_abi_version(void) { return ZOIC_AMD_ABI_VERSION; }

const char *zoic_status_string(zoic_status s)
{
    switch (s) {
    case ZOIC_OK: return "ZOIC_OK";
    case ZOIC_ERR_INVALID_ARGUMENT: return "ZOIC_ERR_INVALID_ARGUMENT";
    case ZOIC_ERR_LENS_PATH: return "ZOIC_ERR_LENS_PATH";
    case ZOIC_ERR_LENS_COLUMNS: return "ZOIC_ERR_LENS_COLUMNS";
    case ZOIC_ERR_LENS_PARSE: return "ZOIC_ERR_LENS_PARSE";
    case ZOIC_ERR_MULTI_APERTURE: return "ZOIC_ERR_MULTI_APERTURE";
    case ZOIC_ERR_NO_APERTURE: return "ZOIC_ERR_NO_APERTURE";
    case ZOIC_ERR_TOO_MANY_LENSES: return "ZOIC_ERR_TOO_MANY_LENSES";
    case ZOIC_ERR_BOKEH_IMAGE: return "ZOIC_ERR_BOKEH_IMAGE";
    case ZOIC_ERR_NOT_UPDATED: return "ZOIC_ERR_NOT_UPDATED";
    case ZOIC_ERR_HIP: return "ZOIC_ERR_HIP";
    case ZOIC_ERR_NO_DEVICE: return "ZOIC_ERR_NO_DEVICE";
    }
    return "ZOIC_ERR_UNKNOWN";
}

const char *zoic_last_error_string(void) { return g_lastError.c_str(); }

int zoic_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zoic_device_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = static_cast<char>(*c - 'A' + 'a');   // sysfs spells bus ids in lower case
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

void zoic_params_default(zoic_params *p)  // node_parameters, zoic.cpp:1547-1562
{
    if (!p) return;
    p->sensorWidth = 3.6f; p->sensorHeight = 2.4f; p->focalLength = 2.0f; p->fStop = 4.0f; p->focalDistance = 100.0f;
    p->useImage = 0; p->bokehPath = ""; p->lensModel = ZOIC_RAYTRACED; p->lensDataPath = ""; p->kolbSamplingLUT = 1;
    p->useDof = 1; p->opticalVignettingDistance = 0.0f; p->opticalVignettingRadius = 1.0f; p->exposureControl = 0.0f;
}

zoic_status zoic_camera_create(int device, zoic_camera **out)
{
    if (!out) return fail(ZOIC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (device == ZOIC_DEVICE_NONE) {
        // tables-only camera: node_update's host precompute (parse, focus, LUT, CDF) for offline validation.
        // It can never produce a ray: every create_rays entry point returns ZOIC_ERR_NO_DEVICE.
        zoic_camera *cam = new zoic_camera();
        cam->device = ZOIC_DEVICE_NONE;
        cam->lutOnHost = true;
        rng_seed_reference(cam->stream);
        *out = cam;
        return ZOIC_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(ZOIC_ERR_NO_DEVICE, "no HIP device visible: libzoic_amd has no CPU path");
    if (device < 0 || device >= n) return fail(ZOIC_ERR_INVALID_ARGUMENT, "device index out of range");
    DeviceGuard guard(device);
    ZOIC_HIP(guard.error());
    std::unique_ptr<zoic_camera> cam(new zoic_camera());
    cam->device = device;
    rng_seed_reference(cam->stream);
    const char *env = std::getenv("ZOIC_LUT_HOST");   // 1: draws, traces and boxes on the host; 2: GPU traces, host draws + replay (round 1's build)
    cam->lutOnHost = env && env[0] == '1';
    cam->lutHostDraws = env && env[0] == '2';
    cam->tidStates.reset(new std::atomic<TidState *>[kTidStates]);
    for (unsigned i = 0; i < kTidStates; ++i) cam->tidStates[i].store(nullptr, std::memory_order_relaxed);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&cam->dCounters), kCounterSets * sizeof(DeviceCounters));   // kernels.hpp: one set per line
    if (e == hipSuccess) e = hipMemset(cam->dCounters, 0, kCounterSets * sizeof(DeviceCounters));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&cam->dWorkCursor), kLaunchSlots * 2 * kCursorStride * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(cam->dWorkCursor, 0, kLaunchSlots * 2 * kCursorStride * sizeof(unsigned int));
    for (unsigned i = 0; e == hipSuccess && i < kLaunchSlots; ++i) {
        cam->slots[i].cursor = cam->dWorkCursor + i * 2 * kCursorStride;
        e = hipEventCreateWithFlags(&cam->slots[i].done, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();   // the two fills above have RUN (hipMemset on device memory is asynchronous; launches may come on non-blocking streams)
    if (e != hipSuccess) {
        zoic_camera_destroy(cam.release());
        return fail(ZOIC_ERR_HIP, std::string("camera allocation: ") + hipGetErrorString(e));
    }
    *out = cam.release();
    return ZOIC_OK;
}

void zoic_camera_destroy(zoic_camera *cam)
{
    if (!cam) return;
    if (cam->device == ZOIC_DEVICE_NONE) { delete cam; return; }
    {
        DeviceGuard guard(cam->device);
        detach_tiles(cam);              // tiles the caller has not destroyed yet: settled, their arrays released, the handles detached
        cam->mail.release();
        (void)hipDeviceSynchronize();   // launches the caller left in flight still read the tables and cursors freed below
        for (auto &c : cam->contexts) c->release();
        cam->contexts.clear(); cam->freeContexts.clear();
        for (unsigned i = 0; i < kLaunchSlots; ++i) {
            if (cam->slots[i].done) (void)hipEventDestroy(cam->slots[i].done);
            cam->slots[i].redo.release();
        }
        cam->dCdfRow.release(); cam->dCdfColumn.release(); cam->dRowIdx.release(); cam->dColIdx.release(); cam->dPyramid.release(); cam->dBokehCells.release();
        cam->dProbeU.release(); cam->dProbeV.release(); cam->dProbeOk.release(); cam->dFastProbe.release(); cam->dFastProbeRays.release();
        if (cam->dProbeTir) (void)hipFree(cam->dProbeTir);
        if (cam->dCounters) (void)hipFree(cam->dCounters);
        if (cam->dWorkCursor) (void)hipFree(cam->dWorkCursor);
    }
    if (cam->tidStates) for (unsigned i = 0; i < kTidStates; ++i) delete cam->tidStates[i].load(std::memory_order_relaxed);
    delete cam;
}

zoic_status zoic_camera_set_bokeh_image(zoic_camera *cam, int width, int height, int nchannels, const float *pixels)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    cam->pendingPixels.clear();
    cam->pendW = cam->pendH = cam->pendC = 0;
    if (pixels && width > 0 && height > 0 && nchannels > 0) {
        cam->pendingPixels.assign(pixels, pixels + static_cast<size_t>(width) * height * nchannels);
        cam->pendW = width; cam->pendH = height; cam->pendC = nchannels;
    }
    cam->bokehDirty = true;   // new pixels under an unchanged bokehPath still rebuild the CDFs
    return ZOIC_OK;
}

zoic_status zoic_camera_set_lens_text(zoic_camera *cam, const char *text, size_t len)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    cam->haveLensText = text != nullptr;
    cam->lensText.assign(text ? text : "", text ? len : 0);
    cam->lensDirty = true;    // new text under an unchanged lensDataPath still rebuilds the lens
    return ZOIC_OK;
}

zoic_status zoic_camera_set_precision(zoic_camera *cam, zoic_precision mode)
{
    if (!cam || (mode != ZOIC_PRECISION_STRICT && mode != ZOIC_PRECISION_FAST && mode != ZOIC_PRECISION_FAST_UNCHECKED))
        return fail(ZOIC_ERR_INVALID_ARGUMENT, "bad precision");
    if (cam->device != ZOIC_DEVICE_NONE && mode != cam->precision) {   // the resident per-sample kernel is launched for ONE mode
        DeviceGuard guard(cam->device);
        ZOIC_HIP(guard.error());
        ZOIC_HIP(cam->mail.stop());
    }
    cam->precision = mode;
    return ZOIC_OK;
}

zoic_status zoic_camera_set_wait_mode(zoic_camera *cam, zoic_wait_mode mode)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (mode != ZOIC_WAIT_SPIN && mode != ZOIC_WAIT_YIELD && mode != ZOIC_WAIT_SLEEP) return fail(ZOIC_ERR_INVALID_ARGUMENT, "wait mode: ZOIC_WAIT_SPIN, _YIELD or _SLEEP");
    cam->waitMode.store(static_cast<int>(mode), std::memory_order_relaxed);
    return ZOIC_OK;
}

zoic_status zoic_camera_set_frame_aspect(zoic_camera *cam, float max_abs_sy)
{
    if (!cam || !(max_abs_sy > 0.0f) || !std::isfinite(max_abs_sy)) return fail(ZOIC_ERR_INVALID_ARGUMENT, "max_abs_sy must be a positive number");
    if (max_abs_sy != cam->frameMaxSy) {   // other probes: the samples are laid out again and the verdict taken again at the next update
        cam->frameMaxSy = max_abs_sy;
        cam->fastProbeReady = false;
        cam->fastVerdictValid = false;
    }
    return ZOIC_OK;
}

zoic_status zoic_camera_set_seed(zoic_camera *cam, uint32_t seed)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device != ZOIC_DEVICE_NONE && seed != cam->seed) {   // the resident kernel holds the tables -- the seed of a tile's per-ray streams among them -- by value
        DeviceGuard guard(cam->device);
        ZOIC_HIP(guard.error());
        ZOIC_HIP(cam->mail.stop());
    }
    cam->seed = seed;
    cam->kolb.seed = seed;
    cam->thin.seed = seed;
    // the per-thread retry streams of the per-sample adapter restart from the new seed (tid 0 keeps the reference's state)
    if (cam->tidStates) {
        std::lock_guard<std::mutex> lk(cam->tidCreateM);
        for (unsigned i = 1; i < kTidStates; ++i)
            if (TidState *t = cam->tidStates[i].load(std::memory_order_acquire)) {
                std::lock_guard<std::mutex> tl(t->m);
                t->rng = rng_for_ray(seed, (0xA7100000ull | i) << 32);
            }
    }
    return ZOIC_OK;
}

zoic_status zoic_camera_update(zoic_camera *cam, const zoic_params *p)
{
    if (!cam || !p) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    const bool onDevice = cam->device != ZOIC_DEVICE_NONE;
    std::unique_ptr<DeviceGuard> guard;
    if (onDevice) {
        guard.reset(new DeviceGuard(cam->device));
        ZOIC_HIP(guard->error());
        // node_update never runs beside camera_create_ray (Arnold's contract), but launches the caller queued earlier
        // may still be reading the tables rebuilt below; the resident per-sample kernel holds the old tables by value
        ZOIC_HIP(cam->mail.stop());
        ZOIC_HIP(hipDeviceSynchronize());
    }
    const std::string bokehPath = p->bokehPath ? p->bokehPath : "", lensPath = p->lensDataPath ? p->lensDataPath : "";
    // Whatever fails below, the camera is left "not updated" with no remembered parameters, so the next update rebuilds
    // everything instead of trusting a half-applied state.
    const bool hadParams = cam->params.valid;
    OwnedParams previous;
    if (hadParams) previous = cam->params;
    cam->params.valid = false;
    cam->updated = false;
    bool tablesRebuilt = false;   // anything fast_self_check's verdict depends on

    // bokeh image -> CDF tables, zoic.cpp:1587-1593
    if (params_bokeh_changed(*p, previous) || (p->useImage && cam->bokehDirty)) {
        tablesRebuilt = true;
        cam->image.clear();
        cam->bokehDev = BokehTables{};
        if (p->useImage) {
            bool ok = false;
            std::vector<float> filePx;
            const float *px = nullptr; int w = 0, h = 0, c = 0;
            if (!cam->pendingPixels.empty()) { px = cam->pendingPixels.data(); w = cam->pendW; h = cam->pendH; c = cam->pendC; }
            else if (read_pfm(bokehPath, filePx, w, h, c)) px = filePx.data();
            if (px) {
                // bokehProbability on the GPU (bokeh_cdf.hip) when the camera has one; ZOIC_CDF_HOST=1 keeps it on the host
                const char *env = std::getenv("ZOIC_CDF_HOST");
                int rc = -1;
                if (onDevice && !(env && env[0] == '1')) rc = build_bokeh_cdf_device(px, w, h, c, cam->image);
                if (rc > 0) return fail(ZOIC_ERR_HIP, std::string("bokeh CDF kernels: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
                ok = rc == 0 ? cam->image.valid() : cam->image.build(px, w, h, c);
            }
            if (!ok) { cam->image.clear(); return fail(ZOIC_ERR_BOKEH_IMAGE, "[ZOIC] Couldn't open bokeh image!"); }
            if (onDevice) { if (zoic_status s = upload_bokeh(cam)) { cam->image.clear(); return s; } }
        }
        cam->bokehDirty = false;
    }
    const bool imageOn = p->useImage && cam->image.valid();

    switch (p->lensModel) {
    case ZOIC_THINLENS:  // zoic.cpp:1598-1610
        cam->fov = static_cast<float>(2.0f * std::atan(static_cast<double>(p->sensorWidth / (2.0f * p->focalLength))));
        cam->tanFov = tanf(cam->fov / 2.0f);
        cam->apertureRadius = p->focalLength / (2.0f * p->fStop);
        break;
    case ZOIC_RAYTRACED:  // zoic.cpp:1612-1711
        if (params_lens_changed(*p, previous) || cam->lensDirty) {
            tablesRebuilt = true;
            std::string text;
            if (cam->haveLensText) text = cam->lensText;
            else {
                if (lensPath.empty()) return fail(ZOIC_ERR_LENS_PATH, "[ZOIC] Lens Data Path is invalid");
                FILE *f = std::fopen(lensPath.c_str(), "rb");
                if (!f) return fail(ZOIC_ERR_LENS_PATH, "[ZOIC] Lens Data Path is invalid: cannot open " + lensPath);
                char buf[4096]; size_t got;
                while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
                std::fclose(f);
            }
            cam->lensDirty = true;   // until the rebuild below has gone through
            if (zoic_status s = lens_error_status(cam->lens.parse(text.data(), text.size()))) return s;
            g_lastError.clear();
            LensError le = cam->lens.prepare(p->focalLength, p->fStop, p->focalDistance, p->kolbSamplingLUT != 0, cam->stream,
                                             cam->lutOnHost ? lut_trace_host : lut_trace_device, cam,
                                             (cam->lutOnHost || cam->lutHostDraws) ? nullptr : lut_build_whole_device);
            if (zoic_status s = lens_error_status(le)) return s;
            if (!g_lastError.empty()) return ZOIC_ERR_HIP;
            // counters restart with the lens (zoic.cpp:1626-1628); the precompute's TIR bumps stay in (zoic.cpp:1135 ff.)
            DeviceCounters zero{0, 0, cam->lens.precomputeTIR, {}};
            cam->tirInCounters = cam->lens.precomputeTIR;
            if (onDevice) {
                ZOIC_HIP(hipMemset(cam->dCounters, 0, kCounterSets * sizeof(DeviceCounters)));
                ZOIC_HIP(hipMemcpy(cam->dCounters, &zero, sizeof(zero), hipMemcpyHostToDevice));   // set 0 carries the precompute's bumps
            }
            cam->lensDirty = false;
        }
        break;
    default: break;
    }

    // camera->params = parms, zoic.cpp:1719
    cam->params.p = *p;
    cam->params.bokehPath = bokehPath;
    cam->params.lensDataPath = lensPath;
    cam->params.p.bokehPath = cam->params.bokehPath.c_str();
    cam->params.p.lensDataPath = cam->params.lensDataPath.c_str();
    cam->params.valid = true;

    // flatten what the kernels read
    if (p->lensModel == ZOIC_RAYTRACED) {
        cam->lens.fill_table(cam->kolb, p->sensorWidth, imageOn ? cam->image.x : 0, imageOn ? cam->image.y : 0);
        // the FAST modes' domain (include/zoic_amd.h): a lens laid out rear -> front (positive focalLengthRatio: thicknesses keep
        // their signs) with the sensor BEHIND the rear vertex (z = lenses[0].thickness after cleanupLensData) and rays leaving it
        // towards +z -- then every hit is the near-vertex root a short way ahead of the ray
        cam->fastDomain = cam->lens.focalLengthRatio > 0.0f && std::isfinite(cam->lens.focalLengthRatio) && cam->kolb.dirZ > 0.0f &&
                          !cam->lens.rows.empty() && cam->lens.originShift < cam->lens.rows[0].thickness;
        cam->kolb.useLUT = p->kolbSamplingLUT != 0;
        cam->kolb.useImage = imageOn;
        cam->kolb.bokehW = cam->image.x; cam->kolb.bokehH = cam->image.y;
        exposure_terms(p->exposureControl, cam->kolb.exposureMul, cam->kolb.exposureOn);
        cam->kolb.seed = cam->seed;
    } else if (p->lensModel == ZOIC_THINLENS) {
        ThinTable &t = cam->thin;
        t.tanFov = cam->tanFov; t.apertureRadius = cam->apertureRadius; t.focalDistance = p->focalDistance;
        t.ovDistance = p->opticalVignettingDistance; t.ovRadius = p->opticalVignettingRadius;
        t.useDof = p->useDof != 0; t.useImage = imageOn; t.bokehW = cam->image.x; t.bokehH = cam->image.y;
        exposure_terms(p->exposureControl, t.exposureMul, t.exposureOn);
        t.seed = cam->seed;
    }
    cam->updated = true;
    // the FAST modes are kept only for a camera they are good for (fast_self_check above; the geometric test comes first).
    // The verdict depends on the lens tables, the LUT and the bokeh tables only: an update that rebuilt none of them (exposure,
    // vignetting parameters ...) keeps the previous one.
    if (p->lensModel == ZOIC_RAYTRACED && cam->device != ZOIC_DEVICE_NONE && cam->fastDomain) {
        if (tablesRebuilt || !cam->fastVerdictValid) {
            bool keep = false;
            if (zoic_status s = fast_self_check(cam, keep)) {   // the check could not RUN: that is an error of this update, not a verdict
                cam->updated = false; cam->params.valid = false; cam->fastVerdictValid = false;
                return s;
            }
#if defined(ZOIC_EXP_WHATIF) && ZOIC_EXP_WHATIF != 0
            keep = true;   // timing-only what-if builds (kolb_pool_body.hpp): their rays are wrong on purpose, the check would send them to STRICT
#endif
            cam->fastVerdict = keep; cam->fastVerdictValid = true;
        }
        cam->fastDomain = cam->fastVerdict;
    }
    g_lastError.clear();   // ZOIC_OK leaves no stale detail behind (zoic_last_error_string)
    return ZOIC_OK;
}

zoic_status zoic_create_rays_device(zoic_camera *cam, uint64_t n, const float *d_samples, const uint32_t *d_rng_states,
                                    uint64_t ray_index_base, zoic_ray *d_rays, void *stream)
{
    if (zoic_status s = check_ray_call(cam)) return s;
    if (n == 0) return ZOIC_OK;
    if (!d_samples) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_samples is NULL");
    if (reinterpret_cast<uintptr_t>(d_samples) & 15u) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_samples must be 16-byte aligned");
    if (d_rng_states && (reinterpret_cast<uintptr_t>(d_rng_states) & 15u)) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_rng_states must be 16-byte aligned");
    if (!d_rays || (reinterpret_cast<uintptr_t>(d_rays) & 15u)) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_rays must be non-NULL and 16-byte aligned");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    return launch_rays(cam, n, d_samples, d_rng_states, ray_index_base, reinterpret_cast<RayRecord *>(d_rays), static_cast<hipStream_t>(stream));
}

zoic_status zoic_create_rays_host(zoic_camera *cam, uint64_t n, const float *h_samples, const uint32_t *h_rng_states,
                                  uint64_t ray_index_base, zoic_ray *h_rays)
{
    if (zoic_status s = check_ray_call(cam)) return s;
    if (n == 0) return ZOIC_OK;
    if (!h_samples) return fail(ZOIC_ERR_INVALID_ARGUMENT, "h_samples is NULL");
    if (!h_rays) return fail(ZOIC_ERR_INVALID_ARGUMENT, "h_rays is NULL");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    ContextLease lease(cam);
    if (!lease) return fail(ZOIC_ERR_HIP, std::string("call context: ") + hipGetErrorString(lease.error()));
    CallContext &C = *lease;
    // Pieces run through the context's three streams (copy-in, kernels, copy-out), chained by events, on two buffer sets:
    // while piece k is copied out, piece k+1 is traced and piece k+2 is copied in.  With page-locked caller buffers
    // (zoic_host_alloc / zoic_host_register) every copy is truly asynchronous; with pageable ones hipMemcpyAsync stages
    // internally and the pipeline degrades gracefully.
    const uint64_t piece = host_piece(n);
    const size_t cap = static_cast<size_t>(std::min<uint64_t>(n, piece));
    for (int b = 0; b < 2 && (b == 0 || n > piece); ++b) {
        ZOIC_HIP(C.dSamples[b].reserve(cap * 4));
        ZOIC_HIP(C.dRays[b].reserve(cap));
        if (h_rng_states) ZOIC_HIP(C.dRng[b].reserve(cap * 4));
    }
    zoic_status status = ZOIC_OK;
    uint64_t k = 0;
    for (uint64_t off = 0; off < n && status == ZOIC_OK; off += piece, ++k) {
        const int b = static_cast<int>(k & 1u);
        const uint64_t m = std::min<uint64_t>(piece, n - off);
        hipError_t e = hipSuccess;
        // copy-in: dSamples[b] is free once the kernels of piece k-2 are done
        if (k >= 2) e = hipStreamWaitEvent(C.sIn, C.runDone[b], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(C.dSamples[b].ptr, h_samples + off * 4, m * 16, hipMemcpyHostToDevice, C.sIn);
        const uint32_t *dRng = nullptr;
        if (e == hipSuccess && h_rng_states) {
            e = hipMemcpyAsync(C.dRng[b].ptr, h_rng_states + off * 4, m * 16, hipMemcpyHostToDevice, C.sIn);
            dRng = C.dRng[b].ptr;
        }
        if (e == hipSuccess) e = hipEventRecord(C.inDone[b], C.sIn);
        // kernels: need the samples, and dRays[b] back from the copy-out of piece k-2
        if (e == hipSuccess) e = hipStreamWaitEvent(C.sRun, C.inDone[b], 0);
        if (e == hipSuccess && k >= 2) e = hipStreamWaitEvent(C.sRun, C.outDone[b], 0);
        if (e != hipSuccess) { status = fail(ZOIC_ERR_HIP, std::string("H2D: ") + hipGetErrorString(e)); break; }
        status = launch_rays(cam, m, C.dSamples[b].ptr, dRng, ray_index_base + off, C.dRays[b].ptr, C.sRun);
        if (status != ZOIC_OK) break;
        e = hipEventRecord(C.runDone[b], C.sRun);
        // copy-out
        if (e == hipSuccess) e = hipStreamWaitEvent(C.sOut, C.runDone[b], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(h_rays + off, C.dRays[b].ptr, m * sizeof(zoic_ray), hipMemcpyDeviceToHost, C.sOut);
        if (e == hipSuccess) e = hipEventRecord(C.outDone[b], C.sOut);
        if (e != hipSuccess) status = fail(ZOIC_ERR_HIP, std::string("D2H: ") + hipGetErrorString(e));
    }
    {   // own streams only: other threads' calls are not waited for
        const hipError_t e = C.sync_all();
        if (e != hipSuccess && status == ZOIC_OK) status = fail(ZOIC_ERR_HIP, std::string("stream sync: ") + hipGetErrorString(e));
    }
    return status;
}

zoic_status zoic_create_rays_arnold(zoic_camera *cam, uint64_t n, const zoic_camera_input *inputs, zoic_camera_output *outputs,
                                    uint64_t ray_index_base)
{
    if (zoic_status s = check_ray_call(cam)) return s;
    if (n == 0) return ZOIC_OK;
    if (!inputs || !outputs) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    static_assert(sizeof(zoic_camera_input) == 28 && sizeof(zoic_camera_output) == 84, "Arnold POD layout");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    ContextLease lease(cam);
    if (!lease) return fail(ZOIC_ERR_HIP, std::string("call context: ") + hipGetErrorString(lease.error()));
    CallContext &C = *lease;
    // Same three-stream pipeline as zoic_create_rays_host.  The 32-byte records are expanded to AtCameraOutput rows ON THE
    // GPU (kernels.hip expand_outputs_kernel) and leave as whole 84-byte rows straight into the caller's array: the host
    // touches nothing (round 2 expanded on the calling thread: 84 Mrays/s per thread against PCIe's ~0.45 Grays/s).
    const uint64_t piece = host_piece(n);
    const size_t cap = static_cast<size_t>(std::min<uint64_t>(n, piece));
    for (int b = 0; b < 2 && (b == 0 || n > piece); ++b) {
        ZOIC_HIP(C.dInputs7[b].reserve(cap * 7));
        ZOIC_HIP(C.dSamples[b].reserve(cap * 4));
        ZOIC_HIP(C.dRays[b].reserve(cap));
        ZOIC_HIP(C.dOut21[b].reserve(cap * 21));
    }
    zoic_status status = ZOIC_OK;
    uint64_t k = 0;
    for (uint64_t off = 0; off < n && status == ZOIC_OK; off += piece, ++k) {
        const int b = static_cast<int>(k & 1u);
        const uint64_t m = std::min<uint64_t>(piece, n - off);
        hipError_t e = hipSuccess;
        if (k >= 2) e = hipStreamWaitEvent(C.sIn, C.runDone[b], 0);   // dInputs7[b] is free once piece k-2's kernels are done
        if (e == hipSuccess) e = hipMemcpyAsync(C.dInputs7[b].ptr, inputs + off, m * sizeof(zoic_camera_input), hipMemcpyHostToDevice, C.sIn);
        if (e == hipSuccess) e = hipEventRecord(C.inDone[b], C.sIn);
        if (e == hipSuccess) e = hipStreamWaitEvent(C.sRun, C.inDone[b], 0);
        if (e == hipSuccess && k >= 2) e = hipStreamWaitEvent(C.sRun, C.outDone[b], 0);   // dOut21[b] back from piece k-2's copy-out
        if (e == hipSuccess) e = static_cast<hipError_t>(launch_pack_inputs(C.dInputs7[b].ptr, C.dSamples[b].ptr, m, C.sRun));
        if (e != hipSuccess) { status = fail(ZOIC_ERR_HIP, std::string("H2D + pack: ") + hipGetErrorString(e)); break; }
        status = launch_rays(cam, m, C.dSamples[b].ptr, nullptr, ray_index_base + off, C.dRays[b].ptr, C.sRun);
        if (status != ZOIC_OK) break;
        e = static_cast<hipError_t>(launch_expand_outputs(C.dRays[b].ptr, C.dOut21[b].ptr, m, C.sRun));
        if (e == hipSuccess) e = hipEventRecord(C.runDone[b], C.sRun);
        if (e == hipSuccess) e = hipStreamWaitEvent(C.sOut, C.runDone[b], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(outputs + off, C.dOut21[b].ptr, m * sizeof(zoic_camera_output), hipMemcpyDeviceToHost, C.sOut);
        if (e == hipSuccess) e = hipEventRecord(C.outDone[b], C.sOut);
        if (e != hipSuccess) status = fail(ZOIC_ERR_HIP, std::string("expand + D2H: ") + hipGetErrorString(e));
    }
    {   // own streams only: other threads' calls are not waited for
        const hipError_t e = C.sync_all();
        if (e != hipSuccess && status == ZOIC_OK) status = fail(ZOIC_ERR_HIP, std::string("stream sync: ") + hipGetErrorString(e));
    }
    return status;
}

}  // extern "C"

// the resident kernel is running (or has just been started), watching `slot` and -- wantWorkers -- with its tile workers; see mailbox.hip
static zoic_status mailbox_ensure_running(zoic_camera *cam, unsigned slot, bool wantWorkers)
{
    Mailbox &M = cam->mail;
    std::lock_guard<std::mutex> lk(M.launchM);
    ZOIC_HIP(M.init());
    volatile MailHeader *h = M.header();
    const bool moreSlots = slot + 1 > M.slotsInUse.load(std::memory_order_relaxed);
    const bool moreWorkers = wantWorkers && M.workerGroups.load(std::memory_order_relaxed) == 0u;
    if (moreSlots || moreWorkers) {
        // a tid beyond the slots the resident launch watches, or the camera's first tile: it retires (stop flag; tiles in flight are
        // finished first) and starts again watching more / with the worker waves
        if (h->alive != 0u) {
            M.request(0)->stop = 1u;
            std::atomic_thread_fence(std::memory_order_seq_cst);
            ZOIC_HIP(hipStreamSynchronize(M.stream));
            M.request(0)->stop = 0u; h->alive = 0u;
        }
        if (moreSlots) { h->slotsInUse = slot + 1; M.slotsInUse.store(slot + 1, std::memory_order_release); }
        if (moreWorkers) {
            uint32_t groups = kTileWorkerGroups;
            if (const char *env = std::getenv("ZOIC_TILE_WORKER_GROUPS")) { const long v = std::atol(env); if (v >= 1 && v <= static_cast<long>(kTileMaxWorkerWaves / 4u)) groups = static_cast<uint32_t>(v); }
            h->workerGroups = groups;
            M.workerGroups.store(groups, std::memory_order_release);
        }
    }
    if (h->alive != 0u) {
        // alive is cleared by the kernel's last store; a kernel that died without it leaves the stream idle (or in error)
        const hipError_t q = hipStreamQuery(M.stream);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return ZOIC_OK; }
        if (q != hipSuccess) return fail(ZOIC_ERR_HIP, std::string("resident kernel: ") + hipGetErrorString(q));
    }
    ZOIC_HIP(hipStreamSynchronize(M.stream));   // the previous resident kernel has retired (its last store is long done)
    const int model = cam->params.p.lensModel;
    const int mode = cam->kernel_mode();
    M.request(0)->stop = 0u;
    h->alive = 1u;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const int rc = launch_mailbox(cam->kolb, cam->thin, cam->bokehDev, model, mode, M.mem.dev, M.dState, cam->dCounters,
                                  M.workerGroups.load(std::memory_order_relaxed), M.stream);
    if (rc != 0) { h->alive = 0u; return fail(ZOIC_ERR_HIP, std::string("resident kernel launch: ") + hipGetErrorString(static_cast<hipError_t>(rc))); }
    return ZOIC_OK;
}

// Wait until `ready()`; the resident kernel may retire (idle / lifetime) around a call and is started again from here.
// zoic_camera_set_wait_mode: SPIN = a pause loop (a core per waiting thread); YIELD / SLEEP = spin ~2 us (kWaitSpinFirst pauses: the
// per-sample call's reply and a small tile arrive inside that), then sched_yield() / a 20 us sleep between polls -- hosts with more
// render threads than cores.  `tick` counts in pause-equivalents so that the periodic checks keep their rough wall-clock spacing.
constexpr uint64_t kWaitSpinFirst = 256;
template <class Ready>
static zoic_status mailbox_await(zoic_camera *cam, unsigned slot, bool wantWorkers, Ready ready)
{
    Mailbox &M = cam->mail;
    const int mode = cam->waitMode.load(std::memory_order_relaxed);
    std::chrono::steady_clock::time_point t0;
    bool timing = false;
    uint64_t tick = 0, nextAlive = 2048, nextQuery = 0x100000;
    for (uint64_t spins = 1;; ++spins) {
        if (ready()) return ZOIC_OK;
        if (mode == ZOIC_WAIT_SPIN || spins <= kWaitSpinFirst) { __builtin_ia32_pause(); tick += 1; }
        else if (mode == ZOIC_WAIT_YIELD) { sched_yield(); tick += 64; }                                  // ~0.5-1 us per round trip through the scheduler
        else { const timespec ts{0, 20000}; nanosleep(&ts, nullptr); tick += 4096; }                      // 20 us + the timer's slack
        if (tick < nextAlive) continue;
        nextAlive = tick + 2048;
        if (M.header()->alive == 0u) {   // the kernel retired (idle / lifetime) around this call: start it again
            if (zoic_status s = mailbox_ensure_running(cam, slot, wantWorkers)) return s;
        } else if (tick >= nextQuery) {
            // every ~20 ms: a kernel that died (fault) never clears `alive` -- ask the stream; and never wait for ever (20 s)
            nextQuery = tick + 0x100000;
            if (zoic_status s = mailbox_ensure_running(cam, slot, wantWorkers)) return s;
            const auto now = std::chrono::steady_clock::now();
            if (!timing) { t0 = now; timing = true; }
            else if (now - t0 > std::chrono::seconds(20)) {
                // what the host sees of the slot, for whoever has to find out why (a lost batch shows as one flag short)
                char why[320];
                std::snprintf(why, sizeof why, "resident kernel did not answer (slot %u: request %u, tile %u with %u batches of which %u flagged, alive %u, slots in use %u, worker groups %u)",
                              slot, M.seq[slot], M.tileSeq[slot], M.tileBatches[slot], M.tileSeen[slot], M.header()->alive, M.slotsInUse.load(), M.workerGroups.load());
                return fail(ZOIC_ERR_HIP, why);
            }
        }
    }
}

// the tile in flight on `slot` (if any) has been answered; slotM[slot] is held
static zoic_status tile_settle_locked(zoic_camera *cam, unsigned slot)
{
    Mailbox &M = cam->mail;
    const uint32_t seq = M.tileSeq[slot];
    if (seq == 0u) return ZOIC_OK;
    if (zoic_status s = mailbox_await(cam, slot, true, [&] { return M.tile_complete(slot); })) return s;
    std::atomic_thread_fence(std::memory_order_acquire);
    M.tileSeq[slot] = 0u;
    return ZOIC_OK;
}

// zoic_camera_destroy with tiles of the caller still alive (a tile outliving its camera dereferenced freed memory): every tile in
// flight is settled, its page-locked arrays are released with the camera and the handle is DETACHED -- zoic_tile_destroy still frees it,
// every other call on it fails with ZOIC_ERR_INVALID_ARGUMENT, its array getters return NULL.
static void detach_tiles(zoic_camera *cam)
{
    std::vector<zoic_tile *> tiles;
    { std::lock_guard<std::mutex> lk(cam->tilesM); tiles.swap(cam->liveTiles); }
    for (zoic_tile *t : tiles) {
        bool settled = true;
        {
            std::lock_guard<std::mutex> slotLock(cam->mail.slotM[t->slot]);
            if (t->seq != 0u && cam->mail.tileSeq[t->slot] == t->seq) settled = tile_settle_locked(cam, t->slot) == ZOIC_OK;
        }
        if (!settled) (void)cam->mail.stop();   // the kernel did not answer: nothing of it may still write into the arrays released here
        t->mem.release();
        t->inputs = nullptr; t->outputs = nullptr; t->dIn = t->dOut = 0; t->seq = 0u; t->capacity = 0u;
        t->registered = false;
        t->cam = nullptr;
    }
}

extern "C" {

zoic_status zoic_camera_create_ray(zoic_camera *cam, const zoic_camera_input *input, zoic_camera_output *output, uint16_t tid)
{
    // camera_create_ray(node, input, output, tid), zoic.cpp:1752.  The reference ignores `tid` and lets every render thread
    // race on one global xor128 state; here each tid owns a retry stream that carries over from call to call, and tid 0's
    // stream is the camera's own reference state (the one node_update's LUT build draws from), so a single-threaded
    // sequence of update / create_ray calls reproduces the reference process draw for draw.
    // No launch per call: the sample goes to the resident mailbox kernel (mailbox.hip) through mapped pinned memory.
    if (zoic_status s = check_ray_call(cam)) return s;
    if (!input || !output) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    const int model = cam->params.p.lensModel;
    if (model != ZOIC_RAYTRACED && model != ZOIC_THINLENS)
        return fail(ZOIC_ERR_INVALID_ARGUMENT, "lensModel NONE produces no rays (zoic.cpp:1966-1968)");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    TidState *T = cam->tid_state(tid);
    std::lock_guard<std::mutex> tidLock(T->m);
    Rng &rng = tid == 0 ? cam->stream : T->rng;
    Mailbox &M = cam->mail;
    const unsigned slot = tid % kMailSlots;
    std::lock_guard<std::mutex> slotLock(M.slotM[slot]);
    if (M.slotsInUse.load(std::memory_order_acquire) <= slot || M.header()->alive == 0u)
        if (zoic_status s = mailbox_ensure_running(cam, slot, false)) return s;
    if (zoic_status s = tile_settle_locked(cam, slot)) return s;   // a tile submitted on this slot and not waited for yet comes first
    // request: data words first, the sequence number last in every 16-byte chunk (x86 stores stay in program order)
    const uint32_t seq = ++M.seq[slot];
    volatile MailRequest *q = M.request(slot);
    q->rngZ = rng.z; q->rngW = rng.w; q->kind = 0u;
    q->lensy = input->lensy; q->rngX = rng.x; q->rngY = rng.y;
    q->sx = input->sx; q->sy = input->sy; q->lensx = input->lensx;
    std::atomic_thread_fence(std::memory_order_release);
    q->seq2 = seq; q->seq1 = seq; q->seq0 = seq;
    // reply: complete when its three chunks carry this call's number
    volatile MailReply *a = M.reply(slot);
    if (zoic_status s = mailbox_await(cam, slot, false, [&] { return a->seq0 == seq && a->seq1 == seq && a->seq2 == seq; })) return s;
    std::atomic_thread_fence(std::memory_order_acquire);
    zoic_ray r;
    r.ox = a->ox; r.oy = a->oy; r.oz = a->oz; r.dx = a->dx; r.dy = a->dy; r.dz = a->dz; r.weight = a->weight; r.flags = a->flags;
    // every retry drew two numbers (zoic.cpp:1806 / 1881 / 1930)
    const uint32_t tries = (r.flags >> 1) & 31u;
    for (uint32_t i = 0; i < 2u * tries; ++i) (void)xor128(rng);
    expand_record(r, *output);
    return ZOIC_OK;
}

// ---- tiles (include/zoic_amd.h): bucket-sized batches through the resident kernel, no launch ------------------------------------
static zoic_status tile_alloc(zoic_camera *cam, uint32_t capacity, uint16_t tid, zoic_tile **out)
{
    std::unique_ptr<zoic_tile> t(new zoic_tile());
    t->cam = cam; t->tid = tid; t->slot = tid % kMailSlots; t->capacity = capacity;
    const size_t inBytes = (static_cast<size_t>(capacity) * sizeof(zoic_camera_input) + 63u) & ~static_cast<size_t>(63u);
    const hipError_t e = t->mem.reserve(inBytes + static_cast<size_t>(capacity) * sizeof(zoic_camera_output));
    if (e != hipSuccess) return fail(ZOIC_ERR_HIP, std::string("tile buffers: ") + hipGetErrorString(e));
    t->inputs = static_cast<zoic_camera_input *>(t->mem.host);
    t->outputs = reinterpret_cast<zoic_camera_output *>(static_cast<char *>(t->mem.host) + inBytes);
    t->dIn = reinterpret_cast<uint64_t>(t->mem.dev);
    t->dOut = t->dIn + inBytes;
    *out = t.release();
    return ZOIC_OK;
}

// post n rows at (dIn -> dOut) on `slot`; slotM[slot] is held and no tile is in flight on it
static zoic_status tile_post_locked(zoic_camera *cam, unsigned slot, uint32_t n, uint64_t dIn, uint64_t dOut, uint64_t base, uint32_t *seqOut, int rows = ZOIC_TILE_ROWS_ARNOLD)
{
    Mailbox &M = cam->mail;
    if (M.slotsInUse.load(std::memory_order_acquire) <= slot || M.workerGroups.load(std::memory_order_acquire) == 0u || M.header()->alive == 0u)
        if (zoic_status s = mailbox_ensure_running(cam, slot, true)) return s;
    const uint32_t seq = ++M.seq[slot];
    volatile MailTileRequest *q = reinterpret_cast<volatile MailTileRequest *>(M.request(slot));
    q->baseHi = static_cast<uint32_t>(base >> 32); q->pad = static_cast<uint32_t>(rows); q->kind = 1u;
    q->outLo = static_cast<uint32_t>(dOut); q->outHi = static_cast<uint32_t>(dOut >> 32); q->baseLo = static_cast<uint32_t>(base);
    q->inLo = static_cast<uint32_t>(dIn); q->inHi = static_cast<uint32_t>(dIn >> 32); q->n = n;
    std::atomic_thread_fence(std::memory_order_release);   // the caller's input rows and the words above, then the numbers
    q->seq2 = seq; q->seq1 = seq; q->seq0 = seq;
    const uint32_t perBatch = tile_rays_per_batch(cam->params.p.lensModel == ZOIC_THINLENS, n);   // (mailbox.hpp: the kernel's rule)
    M.tileSeq[slot] = seq; M.tileBatches[slot] = (n + perBatch - 1u) / perBatch; M.tileSeen[slot] = 0u;
    *seqOut = seq;
    return ZOIC_OK;
}

static zoic_status check_tile_call(const zoic_camera *cam)
{
    if (zoic_status s = check_ray_call(cam)) return s;
    const int model = cam->params.p.lensModel;
    if (model != ZOIC_RAYTRACED && model != ZOIC_THINLENS)
        return fail(ZOIC_ERR_INVALID_ARGUMENT, "lensModel NONE produces no rays (zoic.cpp:1966-1968)");
    return ZOIC_OK;
}

zoic_status zoic_tile_create(zoic_camera *cam, uint32_t capacity, uint16_t tid, zoic_tile **out)
{
    if (!out) return fail(ZOIC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device == ZOIC_DEVICE_NONE) return fail(ZOIC_ERR_NO_DEVICE, "tables-only camera: rays need a gfx950 device (no CPU path)");
    if (capacity == 0u || capacity > kTileMaxSamples) return fail(ZOIC_ERR_INVALID_ARGUMENT, "tile capacity must be 1 ... ZOIC_TILE_MAX_SAMPLES");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    if (zoic_status s = tile_alloc(cam, capacity, tid, out)) return s;
    std::lock_guard<std::mutex> lk(cam->tilesM);
    cam->liveTiles.push_back(*out);
    (*out)->registered = true;
    return ZOIC_OK;
}

void zoic_tile_destroy(zoic_tile *tile)
{
    if (!tile) return;
    zoic_camera *cam = tile->cam;
    if (!cam) { delete tile; return; }   // detached by zoic_camera_destroy: its arrays went with the camera
    DeviceGuard guard(cam->device);
    // the GPU may still be writing into the arrays freed below; if the resident kernel does not answer (20 s / a HIP error) it is
    // stopped first -- no wave of it outlives the stop
    if (zoic_tile_wait(tile) != ZOIC_OK) (void)cam->mail.stop();
    if (tile->registered) {
        std::lock_guard<std::mutex> lk(cam->tilesM);
        auto it = std::find(cam->liveTiles.begin(), cam->liveTiles.end(), tile);
        if (it != cam->liveTiles.end()) cam->liveTiles.erase(it);
    }
    tile->mem.release();
    delete tile;
}

zoic_camera_input *zoic_tile_inputs(zoic_tile *tile) { return tile ? tile->inputs : nullptr; }
zoic_camera_output *zoic_tile_outputs(zoic_tile *tile) { return tile ? tile->outputs : nullptr; }
uint32_t zoic_tile_capacity(const zoic_tile *tile) { return tile ? tile->capacity : 0u; }

zoic_status zoic_tile_submit(zoic_tile *tile, uint32_t n, uint64_t ray_index_base)
{
    if (!tile) return fail(ZOIC_ERR_INVALID_ARGUMENT, "tile is NULL");
    zoic_camera *cam = tile->cam;
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "the tile's camera has been destroyed");
    if (zoic_status s = check_tile_call(cam)) return s;
    if (n > tile->capacity) return fail(ZOIC_ERR_INVALID_ARGUMENT, "n exceeds the tile's capacity");
    if (n == 0u) return ZOIC_OK;
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    std::lock_guard<std::mutex> slotLock(cam->mail.slotM[tile->slot]);
    // one request per slot at a time: this tile's previous submit, or another tile of the same slot (tids 64 apart), comes first
    // (tile_settle_locked returns at once while nothing is in flight: tileSeq == 0 before the mailbox exists)
    if (zoic_status s = tile_settle_locked(cam, tile->slot)) return s;
    tile->seq = 0u; tile->polls = 0u;
    return tile_post_locked(cam, tile->slot, n, tile->dIn, tile->dOut, ray_index_base, &tile->seq, tile->rows | (tile->ins << 1));
}

zoic_status zoic_tile_wait(zoic_tile *tile)
{
    if (!tile) return fail(ZOIC_ERR_INVALID_ARGUMENT, "tile is NULL");
    zoic_camera *cam = tile->cam;
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "the tile's camera has been destroyed");
    if (tile->seq == 0u) return ZOIC_OK;
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    std::lock_guard<std::mutex> slotLock(cam->mail.slotM[tile->slot]);
    // (a later call on the slot may have settled it already: then tileSeq has moved on and the rows are complete)
    if (cam->mail.tileSeq[tile->slot] == tile->seq) if (zoic_status s = tile_settle_locked(cam, tile->slot)) return s;
    tile->seq = 0u;
    return ZOIC_OK;
}

zoic_status zoic_tile_set_rows(zoic_tile *tile, int rows)
{
    if (!tile) return fail(ZOIC_ERR_INVALID_ARGUMENT, "tile is NULL");
    if (!tile->cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "the tile's camera has been destroyed");
    if (rows != ZOIC_TILE_ROWS_ARNOLD && rows != ZOIC_TILE_ROWS_RAYS) return fail(ZOIC_ERR_INVALID_ARGUMENT, "rows: ZOIC_TILE_ROWS_ARNOLD or ZOIC_TILE_ROWS_RAYS");
    if (tile->seq != 0u) return fail(ZOIC_ERR_INVALID_ARGUMENT, "zoic_tile_set_rows between a submit and its wait");
    tile->rows = rows;
    return ZOIC_OK;
}

const zoic_ray *zoic_tile_rays(const zoic_tile *tile) { return tile ? reinterpret_cast<const zoic_ray *>(tile->outputs) : nullptr; }

zoic_status zoic_tile_set_inputs(zoic_tile *tile, int inputs)
{
    if (!tile) return fail(ZOIC_ERR_INVALID_ARGUMENT, "tile is NULL");
    if (!tile->cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "the tile's camera has been destroyed");
    if (inputs != ZOIC_TILE_INPUTS_ARNOLD && inputs != ZOIC_TILE_INPUTS_SAMPLES) return fail(ZOIC_ERR_INVALID_ARGUMENT, "inputs: ZOIC_TILE_INPUTS_ARNOLD or ZOIC_TILE_INPUTS_SAMPLES");
    if (tile->seq != 0u) return fail(ZOIC_ERR_INVALID_ARGUMENT, "zoic_tile_set_inputs between a submit and its wait");
    tile->ins = inputs;
    return ZOIC_OK;
}

float *zoic_tile_samples(zoic_tile *tile) { return tile ? reinterpret_cast<float *>(tile->inputs) : nullptr; }

int zoic_tile_done(zoic_tile *tile)
{
    if (!tile || !tile->cam || tile->seq == 0u) return 1;
    zoic_camera *cam = tile->cam;
    std::lock_guard<std::mutex> slotLock(cam->mail.slotM[tile->slot]);
    if (cam->mail.tileSeq[tile->slot] != tile->seq) return 1;
    if (cam->mail.tile_complete(tile->slot)) return 1;
    // The resident kernel retires by itself (1 ms idle / 50 ms of life, mailbox.hpp).  If it left between the submit's look at `alive`
    // and the slot wave's look at the request line, nobody is watching the request: a caller that only polls done() would spin for
    // ever.  So an incomplete tile with no kernel alive starts it again here; every 1024th poll also asks the stream (a
    // kernel that died on a fault never clears `alive`) -- an error found that way is reported by the zoic_tile_wait that must follow.
    if (cam->mail.header()->alive == 0u || (++tile->polls & 1023u) == 0u) {
        DeviceGuard guard(cam->device);
        if (guard.error() == hipSuccess) (void)mailbox_ensure_running(cam, tile->slot, true);
    }
    return 0;
}

zoic_status zoic_camera_create_rays_tile(zoic_camera *cam, uint32_t n, const zoic_camera_input *inputs, zoic_camera_output *outputs,
                                         uint64_t ray_index_base, uint16_t tid)
{
    if (zoic_status s = check_tile_call(cam)) return s;
    if (n == 0u) return ZOIC_OK;
    if (!inputs || !outputs) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    Mailbox &M = cam->mail;
    const unsigned slot = tid % kMailSlots;
    // Page-locked, mapped caller arrays (zoic_host_alloc / zoic_host_register) are read and written in place; anything else goes
    // through the slot's own page-locked staging, 16 Ki rows at a time on two buffers, so that the copy-in of piece k+1 and the
    // copy-out of piece k-1 run beside the GPU's work on piece k.
    uint64_t dIn = 0, dOut = 0;
    {
        hipPointerAttribute_t ai, ao;
        const bool pinned = hipPointerGetAttributes(&ai, inputs) == hipSuccess && ai.type == hipMemoryTypeHost && ai.devicePointer &&
                            hipPointerGetAttributes(&ao, outputs) == hipSuccess && ao.type == hipMemoryTypeHost && ao.devicePointer;
        (void)hipGetLastError();   // "not a registered pointer" is an answer, not an error of this call
        if (pinned) { dIn = reinterpret_cast<uint64_t>(ai.devicePointer); dOut = reinterpret_cast<uint64_t>(ao.devicePointer); }
    }
    std::lock_guard<std::mutex> slotLock(M.slotM[slot]);
    if (zoic_status s = tile_settle_locked(cam, slot)) return s;
    uint32_t seq = 0;
    if (dIn != 0 && dOut != 0) {
        for (uint32_t off = 0; off < n; off += kTileMaxSamples) {
            const uint32_t m = std::min<uint32_t>(kTileMaxSamples, n - off);
            if (zoic_status s = tile_post_locked(cam, slot, m, dIn + static_cast<uint64_t>(off) * sizeof(zoic_camera_input),
                                                 dOut + static_cast<uint64_t>(off) * sizeof(zoic_camera_output), ray_index_base + off, &seq)) return s;
            if (zoic_status s = tile_settle_locked(cam, slot)) return s;
        }
        return ZOIC_OK;
    }
    constexpr uint32_t kPiece = 16384;
    const uint32_t piece = std::min<uint32_t>(n, kPiece);
    const unsigned buffers = n > kPiece ? 2u : 1u;
    for (unsigned b = 0; b < buffers; ++b) {
        zoic_tile *&t = M.ownTile[slot][b];
        if (t && t->capacity < piece) { t->mem.release(); delete t; t = nullptr; }
        if (!t) { uint32_t cap = 1024; while (cap < piece) cap <<= 1; if (zoic_status s = tile_alloc(cam, cap, tid, &t)) return s; }
    }
    uint32_t prevOff = 0, prevM = 0;
    unsigned k = 0;
    for (uint32_t off = 0; off < n; off += piece, ++k) {
        const uint32_t m = std::min<uint32_t>(piece, n - off);
        zoic_tile *t = M.ownTile[slot][k & 1u];
        std::memcpy(t->inputs, inputs + off, static_cast<size_t>(m) * sizeof(zoic_camera_input));   // beside the GPU's work on piece k-1
        if (zoic_status s = tile_settle_locked(cam, slot)) return s;                                  // piece k-1 is complete
        if (zoic_status s = tile_post_locked(cam, slot, m, t->dIn, t->dOut, ray_index_base + off, &seq)) return s;
        if (prevM) std::memcpy(outputs + prevOff, M.ownTile[slot][(k - 1u) & 1u]->outputs, static_cast<size_t>(prevM) * sizeof(zoic_camera_output));
        prevOff = off; prevM = m;
    }
    if (zoic_status s = tile_settle_locked(cam, slot)) return s;
    std::memcpy(outputs + prevOff, M.ownTile[slot][(k - 1u) & 1u]->outputs, static_cast<size_t>(prevM) * sizeof(zoic_camera_output));
    return ZOIC_OK;
}

zoic_status zoic_create_rays_device_resident(zoic_camera *cam, uint32_t n, const float *d_samples, zoic_ray *d_rays, uint64_t ray_index_base, uint16_t tid)
{
    // zoic_create_rays_device costs a launch (52-76 us) whatever it carries.  The resident tile workers read a tile's samples
    // and write its records through plain addresses at system scope: device memory serves as well as mapped host memory, so a GPU
    // consumer's batch is a tile request with 16-byte samples in and 32-byte records out -- the layouts of zoic_create_rays_device.
    if (zoic_status s = check_tile_call(cam)) return s;
    if (n == 0u) return ZOIC_OK;
    if (!d_samples || !d_rays) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n > ZOIC_RESIDENT_MAX_SAMPLES) return fail(ZOIC_ERR_INVALID_ARGUMENT, "n exceeds ZOIC_RESIDENT_MAX_SAMPLES: zoic_create_rays_device is the call for large batches");
    if ((reinterpret_cast<uintptr_t>(d_samples) & 15u) || (reinterpret_cast<uintptr_t>(d_rays) & 15u)) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_samples and d_rays must be 16-byte aligned");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    {
        // memory of the camera's device (a host pointer, or another device's memory without peer mapping, would fault inside a kernel
        // that serves every render thread of this camera)
        hipPointerAttribute_t ai, ao;
        const bool ok = hipPointerGetAttributes(&ai, d_samples) == hipSuccess && ai.type == hipMemoryTypeDevice && ai.device == cam->device &&
                        hipPointerGetAttributes(&ao, d_rays) == hipSuccess && ao.type == hipMemoryTypeDevice && ao.device == cam->device;
        (void)hipGetLastError();
        if (!ok) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_samples / d_rays must be device memory of the camera's device");
    }
    Mailbox &M = cam->mail;
    const unsigned slot = tid % kMailSlots;
    std::lock_guard<std::mutex> slotLock(M.slotM[slot]);
    if (zoic_status s = tile_settle_locked(cam, slot)) return s;
    const uint64_t dIn = reinterpret_cast<uint64_t>(d_samples), dOut = reinterpret_cast<uint64_t>(d_rays);
    uint32_t seq = 0;
    for (uint32_t off = 0; off < n; off += kTileMaxSamples) {
        const uint32_t m = std::min<uint32_t>(kTileMaxSamples, n - off);
        if (zoic_status s = tile_post_locked(cam, slot, m, dIn + static_cast<uint64_t>(off) * 16u, dOut + static_cast<uint64_t>(off) * sizeof(zoic_ray), ray_index_base + off, &seq,
                                             ZOIC_TILE_ROWS_RAYS | (ZOIC_TILE_INPUTS_SAMPLES << 1))) return s;
        if (zoic_status s = tile_settle_locked(cam, slot)) return s;
    }
    return ZOIC_OK;
}

int zoic_camera_reverse_ray(const zoic_camera *cam, const zoic_vec3 *Po, float fov, float *Ps, float *relative_time)
{
    // camera_reverse_ray, zoic.cpp:1992-1995: `return false;` -- nothing is written
    (void)cam; (void)Po; (void)fov; (void)Ps; (void)relative_time;
    return 0;
}

zoic_status zoic_host_alloc(size_t bytes, void **out)
{
    if (!out) return fail(ZOIC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (bytes == 0) return ZOIC_OK;
    ZOIC_HIP(hipHostMalloc(out, bytes, hipHostMallocPortable | hipHostMallocMapped));
    return ZOIC_OK;
}

void zoic_host_free(void *p) { if (p) (void)hipHostFree(p); }

zoic_status zoic_host_register(void *p, size_t bytes)
{
    if (!p || bytes == 0) return fail(ZOIC_ERR_INVALID_ARGUMENT, "empty range");
    ZOIC_HIP(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return ZOIC_OK;
}

zoic_status zoic_host_unregister(void *p)
{
    if (!p) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL pointer");
    ZOIC_HIP(hipHostUnregister(p));
    return ZOIC_OK;
}

zoic_status zoic_generate_samples_device(zoic_camera *cam, uint64_t n, uint64_t ray_index_base, uint32_t width, uint32_t height,
                                         uint32_t spp, uint32_t seed, float *d_samples, void *stream)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (!d_samples || width == 0 || height == 0 || spp == 0) return fail(ZOIC_ERR_INVALID_ARGUMENT, "bad sample grid");
    if (cam->device == ZOIC_DEVICE_NONE) return fail(ZOIC_ERR_NO_DEVICE, "tables-only camera");
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    if (int rc = launch_generate_samples(d_samples, ray_index_base, n, width, height, spp, seed, stream))
        return fail(ZOIC_ERR_HIP, std::string("sample kernel: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
    return ZOIC_OK;
}

zoic_status zoic_camera_get_counters(zoic_camera *cam, zoic_counters *out)
{
    if (!cam || !out) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    if (cam->device == ZOIC_DEVICE_NONE) {
        out->succesRays = out->vignettedRays = 0; out->totalInternalReflection = cam->lens.precomputeTIR;
        return ZOIC_OK;
    }
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    ZOIC_HIP(cam->mail.stop());         // the resident per-sample kernel adds its counts when it retires
    ZOIC_HIP(hipDeviceSynchronize());   // every stream of the device: launches of all threads are counted
    std::vector<DeviceCounters> sets(kCounterSets);
    ZOIC_HIP(hipMemcpy(sets.data(), cam->dCounters, kCounterSets * sizeof(DeviceCounters), hipMemcpyDeviceToHost));
    DeviceCounters c{};
    for (const DeviceCounters &s : sets) { c.succes += s.succes; c.vignetted += s.vignetted; c.tir += s.tir; }
    out->succesRays = c.succes; out->vignettedRays = c.vignetted; out->totalInternalReflection = c.tir;
    return ZOIC_OK;
}

zoic_status zoic_camera_reset_counters(zoic_camera *cam)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device == ZOIC_DEVICE_NONE) return ZOIC_OK;
    DeviceGuard guard(cam->device);
    ZOIC_HIP(guard.error());
    ZOIC_HIP(cam->mail.stop());
    ZOIC_HIP(hipDeviceSynchronize());
    ZOIC_HIP(hipMemset(cam->dCounters, 0, kCounterSets * sizeof(DeviceCounters)));
    ZOIC_HIP(hipDeviceSynchronize());   // (a device-memory hipMemset returns before the fill has run, on the null stream: a launch on a non-blocking stream would not wait for it)
    cam->tirInCounters = 0;
    return ZOIC_OK;
}

zoic_status zoic_camera_get_info(const zoic_camera *cam, zoic_lens_info *out)
{
    if (!cam || !out) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(out, 0, sizeof(*out));
    const LensSystem &L = cam->lens;
    out->lensCount = static_cast<int32_t>(L.rows.size());
    out->apertureElement = L.apertureElement;
    out->userApertureRadius = L.userApertureRadius; out->originShift = L.originShift;
    out->apertureDistance = L.apertureDistance; out->focalLengthRatio = L.focalLengthRatio;
    out->tracedFocalLength[0] = L.tracedFocalLength[0]; out->tracedFocalLength[1] = L.tracedFocalLength[1];
    out->fov = cam->fov; out->tan_fov = cam->tanFov; out->apertureRadius = cam->apertureRadius;
    for (size_t i = 0; i < L.rows.size() && i < ZOIC_MAX_LENS_SURFACES; ++i) {
        out->curvature[i] = L.rows[i].radius; out->thickness[i] = L.rows[i].thickness; out->ior[i] = L.rows[i].ior;
        out->aperture[i] = L.rows[i].aperture; out->center[i] = L.rows[i].center;
    }
    out->lutSize = L.hasLUT ? kLutEntries : 0;
    for (int i = 0; i < kLutEntries; ++i) {
        out->lutKey[i] = L.lutKey[i];
        out->lutMaxX[i] = L.lutBox[i].maxX; out->lutMaxY[i] = L.lutBox[i].maxY;
        out->lutMinX[i] = L.lutBox[i].minX; out->lutMinY[i] = L.lutBox[i].minY;
    }
    out->bokehWidth = cam->image.x; out->bokehHeight = cam->image.y;
    out->precomputeTIR = cam->device == ZOIC_DEVICE_NONE ? L.precomputeTIR : cam->tirInCounters;
    out->fastRunsStrict = (cam->params.valid && cam->params.p.lensModel == ZOIC_RAYTRACED && !cam->fastDomain) ? 1 : 0;
    return ZOIC_OK;
}

zoic_status zoic_camera_get_bokeh_tables(const zoic_camera *cam, float *cdfRow, int32_t *rowIndices, float *cdfColumn,
                                         int32_t *columnIndices)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    const BokehCdf &im = cam->image;
    if (!im.valid()) return fail(ZOIC_ERR_BOKEH_IMAGE, "no bokeh image loaded");
    if (cdfRow) std::memcpy(cdfRow, im.cdfRow.data(), im.cdfRow.size() * sizeof(float));
    if (rowIndices) std::memcpy(rowIndices, im.rowIndices.data(), im.rowIndices.size() * sizeof(int32_t));
    if (cdfColumn) std::memcpy(cdfColumn, im.cdfColumn.data(), im.cdfColumn.size() * sizeof(float));
    if (columnIndices) std::memcpy(columnIndices, im.columnIndices.data(), im.columnIndices.size() * sizeof(int32_t));
    return ZOIC_OK;
}

}  // extern "C"
