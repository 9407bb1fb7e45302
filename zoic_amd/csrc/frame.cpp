// frame.cpp -- zoic_frame_*: ONE frame of camera_create_ray (zoic.cpp:1752) over several HIP devices of one process.
//
// The reference is a single process whose render threads share one camera node (node_initialize zoic.cpp:1565-1572,
// camera_create_ray :1752, node_finish :1723-1749); a C++ plug-in therefore cannot use a one-process-per-GPU launcher.
// This file is the in-process form of SURVEY 8(e)'s sharding: a zoic_camera per device, contiguous 256-aligned ray-index
// slabs, per-ray retry streams keyed by the global ray index (=> bit-identical to the one-device frame), and the finished
// slabs moved to the root device chunk by chunk with hipMemcpyPeerAsync while the next chunk is traced.  It is built on
// the public entry points of capi.cpp only (zoic_camera_*, zoic_create_rays_device / _host) plus one packing kernel.
//
// Stream picture of one zoic_frame_render_device call, per PEER device (lane):
//   compute[0]: [trace chunk 0][pack 0]                 [trace chunk 2][pack 2] ...
//   compute[1]:                  [trace chunk 1][pack 1]                 ...          (a chunk's drain + its STRICT list kernel
//   copy      :        wait c0 -> [peer copy 0] wait c1 -> [peer copy 1] ...           run under the next chunk's trace)
// and the root lane: one launch straight into the caller's buffer (RECORDS) or one launch + one pack (PAYLOAD).  The caller's
// root stream waits for every lane's last event; nothing blocks the host.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zoic_amd.h"
#include "host_util.hpp"
#include "kernels.hpp"

using namespace zoic;

namespace {

constexpr uint64_t kTile = 256;                      // rays per workgroup tile: slabs and chunks are aligned to it
constexpr uint64_t kMinChunkPayloadBytes = 64ull << 20;   // chunks of at least 64 MB of payload
constexpr unsigned kChunksPerSlab = 4;
constexpr uint64_t kMaxChunksPerSlab = 64;

#define FRAME_HIP(expr)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail_status(ZOIC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

void slab_of(uint64_t n, int nDevices, int i, uint64_t &lo, uint64_t &hi)
{
    const uint64_t tiles = (n + kTile - 1) / kTile;
    // tiles * i may not overflow for any n < 2^64 / 256 tiles and i < 2^16: tiles < 2^56
    lo = std::min(n, tiles * static_cast<uint64_t>(i) / static_cast<uint64_t>(nDevices) * kTile);
    hi = std::min(n, tiles * static_cast<uint64_t>(i + 1) / static_cast<uint64_t>(nDevices) * kTile);
}

struct Lane {
    int device = 0;
    zoic_camera *cam = nullptr;
    hipStream_t compute[2] = {nullptr, nullptr}, copy = nullptr;
    // the tails of compute[0], compute[1], copy as the LAST render_* call left them.  Every render_* call starts by making all
    // three streams wait for all three (order_behind_previous): the staging buffers (records / payload) and the chunk events are
    // shared by consecutive calls whatever their n, chunk size, sample pointers or root stream -- two calls never overlap on a lane
    hipEvent_t streamDone[3] = {nullptr, nullptr, nullptr};
    bool doneRecorded[3] = {false, false, false};
    hipEvent_t samplesReady = nullptr;                        // zoic_frame_generate_samples' kernel
    std::vector<hipEvent_t> computed;                         // per chunk: trace (+ pack) done
    // ZOIC_FRAME_PAYLOAD_SPARSE: a chunk travels as [tile headers][rows of the rays with weight != 0]; its size is known on the device
    hipStream_t rootExpand = nullptr;                         // on the ROOT device: the expansion of this lane's chunks into the caller's rows
    hipEvent_t expandDone = nullptr;                          // ... its tail for the call in flight (root device)
    bool expandRecorded = false;
    std::vector<hipEvent_t> copied, expanded;                 // per chunk: peer copy landed (this device) / expanded into the output (root device)
    DeviceBuffer<uint32_t> sparse[2];                         // this device: headers + compacted rows of the chunks in flight
    unsigned int *dCount = nullptr;                           // this device: live rays per chunk (kMaxChunksPerSlab dwords)
    unsigned int *hCount = nullptr;                           // page-locked: the same, read by the host before it sizes the copy
    DeviceBuffer<uint32_t> rootStage[2];                      // ROOT device: where the chunks land
    // what the last render_device call did on this lane (zoic_frame_get_lane_info)
    int peerToRoot = 0, rootToPeer = 0;                       // hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess both succeeded
    uint64_t lastRays = 0, lastBytesToRoot = 0; uint32_t lastChunks = 0;
    DeviceBuffer<float> samples;                              // generated samples of this lane's slab
    uint64_t samplesN = 0, samplesBase = 0; bool haveSamples = false;
    DeviceBuffer<zoic_ray> records;                           // this lane's slab, 32 B/ray
    DeviceBuffer<float> payload;                              // PAYLOAD layout: the packed rows on their way to the root
};

}  // namespace

struct zoic_frame {
    std::vector<Lane> lanes;
    uint64_t chunkRays = 0;   // 0: default
    hipEvent_t rootStart = nullptr;   // recorded on the caller's root stream when a render call begins
    // ZOIC_FRAME_PAYLOAD_AUTO: the gather's layout chosen from the camera.  The first AUTO render after an update is dense; the next
    // one reads the counters (zoic_frame_get_counters: one synchronisation, once per update) and from then on the frame ships SPARSE
    // iff at least kAutoSparseZeroWeight of the rays rendered since the update had weight 0.
    bool autoDecided = false, autoSparse = false;
    uint64_t autoBaseSucc = 0, autoBaseVign = 0;   // the counters when the tables were last rebuilt
    uint64_t autoRendered = 0;                     // rays rendered through AUTO since then
    double autoZeroWeight = -1.0;
};
constexpr double kAutoSparseZeroWeight = 0.25;

namespace {

uint64_t chunk_rays_for(const zoic_frame *f, uint64_t slabRays, int bytesPerRay)
{
    uint64_t c = f->chunkRays;
    if (c == 0) {
        c = (slabRays + kChunksPerSlab - 1) / kChunksPerSlab;
        c = std::max<uint64_t>(c, kMinChunkPayloadBytes / static_cast<uint64_t>(bytesPerRay));
    }
    c = std::max<uint64_t>(kTile, c / kTile * kTile);
    // at most kMaxChunksPerSlab chunks (an event, a launch and a peer copy each): a tiny zoic_frame_set_chunk_rays on a large
    // slab is raised to what gives that many
    const uint64_t floorRays = ((slabRays + kMaxChunksPerSlab - 1) / kMaxChunksPerSlab + kTile - 1) / kTile * kTile;
    return std::max(c, floorRays);
}

zoic_status ensure_chunk_events(Lane &L, size_t chunks)
{
    while (L.computed.size() < chunks) {
        hipEvent_t a = nullptr;
        FRAME_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        L.computed.push_back(a);
    }
    return ZOIC_OK;
}

// Every stream of the lane behind everything the previous render_* call queued on ANY of them (device side; the current device
// is the lane's).  A wait on an event of the same stream is free.
zoic_status order_behind_previous(Lane &L)
{
    for (hipStream_t s : {L.compute[0], L.compute[1], L.copy})
        for (int t = 0; t < 3; ++t)
            if (L.doneRecorded[t]) FRAME_HIP(hipStreamWaitEvent(s, L.streamDone[t], 0));
    if (L.expandRecorded)   // a sparse gather's expansion still reads the root's staging of this lane
        for (hipStream_t s : {L.compute[0], L.compute[1], L.copy}) FRAME_HIP(hipStreamWaitEvent(s, L.expandDone, 0));
    return ZOIC_OK;
}

zoic_status for_each_camera(zoic_frame *frame, zoic_status (*fn)(zoic_camera *, const void *), const void *arg)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    for (Lane &L : frame->lanes)
        if (zoic_status s = fn(L.cam, arg)) return s;
    return ZOIC_OK;
}

// the samples a lane renders from: the caller's slab pointer or the generated buffer
zoic_status lane_samples(const zoic_frame *frame, const Lane &L, size_t i, const float *const *d_samples, uint64_t n, uint64_t base, const float *&out)
{
    if (d_samples) {
        out = d_samples[i];
        if (!out) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "d_samples[i] is NULL for a device with a non-empty slab");
        return ZOIC_OK;
    }
    if (!L.haveSamples || L.samplesN != n || L.samplesBase != base)
        return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "d_samples is NULL and zoic_frame_generate_samples has not been called for this (n, ray_index_base)");
    (void)frame;
    out = L.samples.ptr;
    return ZOIC_OK;
}

}  // namespace

extern "C" {

zoic_status zoic_frame_slab(uint64_t n, int n_devices, int i, uint64_t *begin, uint64_t *end)
{
    if (!begin || !end || n_devices <= 0 || i < 0 || i >= n_devices || n_devices > 65536)
        return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "zoic_frame_slab: bad device index / count");
    slab_of(n, n_devices, i, *begin, *end);
    return ZOIC_OK;
}

zoic_status zoic_frame_create(const int *devices, int n_devices, zoic_frame **out)
{
    if (!out) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > 64) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "1 ... 64 devices");
    std::unique_ptr<zoic_frame> f(new zoic_frame());
    f->lanes.resize(static_cast<size_t>(n_devices));
    zoic_status st = ZOIC_OK;
    for (int i = 0; i < n_devices && st == ZOIC_OK; ++i) {
        Lane &L = f->lanes[static_cast<size_t>(i)];
        L.device = devices[i];
        st = zoic_camera_create(L.device, &L.cam);
        if (st != ZOIC_OK) break;
        DeviceGuard guard(L.device);
        hipError_t e = guard.error();
        for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipStreamCreateWithFlags(&L.compute[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.copy, hipStreamNonBlocking);
        for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&L.streamDone[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&L.samplesReady, hipEventDisableTiming);
        if (e == hipSuccess && i == 0) e = hipEventCreateWithFlags(&f->rootStart, hipEventDisableTiming);
        if (e == hipSuccess && i > 0 && L.device != devices[0]) {
            // direct xGMI copies root <-> peer; "already enabled" (another frame of this process) is fine, "not supported"
            // leaves hipMemcpyPeerAsync to stage through the host -- slower, still correct
            // (zoic_frame_get_lane_info reports which of the two it was: a gather that crawls is then explained, not guessed at)
            const auto enabled = [](hipError_t r) { return r == hipSuccess || r == hipErrorPeerAccessAlreadyEnabled; };
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, L.device, devices[0]) == hipSuccess && can) L.peerToRoot = enabled(hipDeviceEnablePeerAccess(devices[0], 0)) ? 1 : 0;
            (void)hipGetLastError();
            DeviceGuard rootGuard(devices[0]);
            if (hipDeviceCanAccessPeer(&can, devices[0], L.device) == hipSuccess && can) L.rootToPeer = enabled(hipDeviceEnablePeerAccess(L.device, 0)) ? 1 : 0;
            (void)hipGetLastError();
        } else { L.peerToRoot = L.rootToPeer = 1; }   // the root itself, or the root's device listed again: no link involved
        if (e != hipSuccess) st = fail_status(ZOIC_ERR_HIP, std::string("frame streams: ") + hipGetErrorString(e));
    }
    if (st != ZOIC_OK) {
        const std::string why = zoic_last_error_string();
        zoic_frame_destroy(f.release());
        return fail_status(st, why);
    }
    *out = f.release();
    return ZOIC_OK;
}

void zoic_frame_destroy(zoic_frame *frame)
{
    if (!frame) return;
    (void)zoic_frame_synchronize(frame);
    for (Lane &L : frame->lanes) {
        if (L.cam) {
            DeviceGuard guard(L.device);
            for (hipStream_t *s : {&L.compute[0], &L.compute[1], &L.copy}) if (*s) { (void)hipStreamDestroy(*s); *s = nullptr; }
            for (hipEvent_t &e : L.streamDone) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            if (L.samplesReady) (void)hipEventDestroy(L.samplesReady);
            for (hipEvent_t e : L.computed) (void)hipEventDestroy(e);
            L.samples.release(); L.records.release(); L.payload.release(); L.sparse[0].release(); L.sparse[1].release();
            for (hipEvent_t e : L.copied) (void)hipEventDestroy(e);
            if (L.dCount) (void)hipFree(L.dCount);
            if (L.hCount) (void)hipHostFree(L.hCount);
            {
                DeviceGuard rootGuard(frame->lanes[0].device);
                if (L.rootExpand) (void)hipStreamDestroy(L.rootExpand);
                if (L.expandDone) (void)hipEventDestroy(L.expandDone);
                for (hipEvent_t e : L.expanded) (void)hipEventDestroy(e);
                L.rootStage[0].release(); L.rootStage[1].release();
            }
            if (&L == &frame->lanes[0] && frame->rootStart) (void)hipEventDestroy(frame->rootStart);
        }
        zoic_camera_destroy(L.cam);
    }
    delete frame;
}

int zoic_frame_device_count(const zoic_frame *frame) { return frame ? static_cast<int>(frame->lanes.size()) : 0; }

zoic_camera *zoic_frame_camera(zoic_frame *frame, int i)
{
    if (!frame || i < 0 || static_cast<size_t>(i) >= frame->lanes.size()) return nullptr;
    return frame->lanes[static_cast<size_t>(i)].cam;
}

zoic_status zoic_frame_set_bokeh_image(zoic_frame *frame, int width, int height, int nchannels, const float *pixels)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    for (Lane &L : frame->lanes)
        if (zoic_status s = zoic_camera_set_bokeh_image(L.cam, width, height, nchannels, pixels)) return s;
    return ZOIC_OK;
}

zoic_status zoic_frame_set_lens_text(zoic_frame *frame, const char *text, size_t len)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    for (Lane &L : frame->lanes)
        if (zoic_status s = zoic_camera_set_lens_text(L.cam, text, len)) return s;
    return ZOIC_OK;
}

zoic_status zoic_frame_set_precision(zoic_frame *frame, zoic_precision mode)
{
    return for_each_camera(frame, [](zoic_camera *c, const void *a) { return zoic_camera_set_precision(c, *static_cast<const zoic_precision *>(a)); }, &mode);
}

zoic_status zoic_frame_set_seed(zoic_frame *frame, uint32_t seed)
{
    return for_each_camera(frame, [](zoic_camera *c, const void *a) { return zoic_camera_set_seed(c, *static_cast<const uint32_t *>(a)); }, &seed);
}

zoic_status zoic_frame_update(zoic_frame *frame, const zoic_params *p)
{
    if (!frame || !p) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    if (zoic_status s = zoic_frame_synchronize(frame)) return s;   // node_update never runs beside camera_create_ray
    // node_update is deterministic (its xor128 stream starts from the reference's seed on every camera and every camera has
    // seen the same sequence of updates), so the n cameras hold identical tables.  One thread per device: the LUT build and
    // the FAST self-check are a millisecond or two of GPU work each, the lens precompute a few hundred microseconds of host.
    std::vector<zoic_status> st(frame->lanes.size(), ZOIC_OK);
    std::vector<std::string> why(frame->lanes.size());
    std::vector<std::thread> th;
    for (size_t i = 1; i < frame->lanes.size(); ++i)
        th.emplace_back([&, i] { st[i] = zoic_camera_update(frame->lanes[i].cam, p); if (st[i] != ZOIC_OK) why[i] = zoic_last_error_string(); });
    st[0] = zoic_camera_update(frame->lanes[0].cam, p);
    if (st[0] != ZOIC_OK) why[0] = zoic_last_error_string();
    for (std::thread &t : th) t.join();
    for (size_t i = 0; i < st.size(); ++i)
        if (st[i] != ZOIC_OK) return fail_status(st[i], why[i]);
    // another camera from here on: ZOIC_FRAME_PAYLOAD_AUTO decides again
    frame->autoDecided = false; frame->autoSparse = false; frame->autoRendered = 0; frame->autoZeroWeight = -1.0;
    zoic_counters c;
    if (zoic_status s = zoic_frame_get_counters(frame, &c)) return s;
    frame->autoBaseSucc = c.succesRays; frame->autoBaseVign = c.vignettedRays;
    return ZOIC_OK;
}

zoic_status zoic_frame_set_chunk_rays(zoic_frame *frame, uint64_t rays)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    frame->chunkRays = rays;
    return ZOIC_OK;
}

zoic_status zoic_frame_generate_samples(zoic_frame *frame, uint64_t n, uint64_t ray_index_base, uint32_t width, uint32_t height,
                                        uint32_t spp, uint32_t seed)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    const int nd = static_cast<int>(frame->lanes.size());
    for (int i = 0; i < nd; ++i) {
        Lane &L = frame->lanes[static_cast<size_t>(i)];
        uint64_t lo, hi;
        slab_of(n, nd, i, lo, hi);
        L.haveSamples = false;
        DeviceGuard guard(L.device);
        FRAME_HIP(guard.error());
        if (hi > lo) {
            // the buffer may still be read by a render queued earlier
            FRAME_HIP(hipStreamSynchronize(L.compute[0]));
            FRAME_HIP(hipStreamSynchronize(L.compute[1]));
            FRAME_HIP(L.samples.reserve((hi - lo) * 4));
            if (zoic_status s = zoic_generate_samples_device(L.cam, hi - lo, ray_index_base + lo, width, height, spp, seed, L.samples.ptr, L.compute[0])) return s;
        }
        FRAME_HIP(hipEventRecord(L.samplesReady, L.compute[0]));
        L.samplesN = n; L.samplesBase = ray_index_base; L.haveSamples = true;
    }
    return ZOIC_OK;
}

}  // extern "C"

namespace {

// A render call that fails half-way has traces and peer copies of earlier lanes / chunks in flight into the caller's buffer and
// the current lane's tails not yet joined to the root stream: the public entry points wait for all of it (the error path may
// block) before they hand the status back, so that a caller who frees or reuses d_out after a failure races with nothing.
zoic_status settle_after_failure(zoic_frame *frame, zoic_status st)
{
    const std::string why = zoic_last_error_string();
    (void)zoic_frame_synchronize(frame);
    for (Lane &L : frame->lanes) { for (bool &r : L.doneRecorded) r = false; L.expandRecorded = false; }   // everything has drained: nothing left to order behind
    return fail_status(st, why);
}

zoic_status render_device_impl(zoic_frame *frame, uint64_t n, const float *const *d_samples, uint64_t ray_index_base, void *d_out,
                               zoic_frame_layout layout, void *root_stream)
{
    const int nd = static_cast<int>(frame->lanes.size());
    const bool payload = layout == ZOIC_FRAME_PAYLOAD;
    const int rowBytes = payload ? 28 : 32;
    Lane &R = frame->lanes[0];
    hipStream_t rootStream = static_cast<hipStream_t>(root_stream);
    char *const out = static_cast<char *>(d_out);
    {
        DeviceGuard guard(R.device);
        FRAME_HIP(guard.error());
        FRAME_HIP(hipEventRecord(frame->rootStart, rootStream));   // whatever the caller queued before (the last reader of d_out) comes first
    }
    for (int i = 0; i < nd; ++i) {
        Lane &L = frame->lanes[static_cast<size_t>(i)];
        uint64_t lo, hi;
        slab_of(n, nd, i, lo, hi);
        if (hi <= lo) continue;
        const float *samples = nullptr;
        if (zoic_status s = lane_samples(frame, L, static_cast<size_t>(i), d_samples, n, ray_index_base, samples)) return s;
        DeviceGuard guard(L.device);
        FRAME_HIP(guard.error());
        const bool root = i == 0;
        const uint64_t slab = hi - lo;
        if (zoic_status s = order_behind_previous(L)) return s;
        L.lastRays = slab; L.lastBytesToRoot = 0; L.lastChunks = 0;
        // the root's slab is ONE launch (nothing to overlap it with: sub-launches cost 8-13 % on one GPU)
        const uint64_t chunk = root ? slab : chunk_rays_for(frame, slab, rowBytes);
        const size_t chunks = static_cast<size_t>((slab + chunk - 1) / chunk);
        if (zoic_status s = ensure_chunk_events(L, chunks)) return s;
        zoic_ray *records = nullptr;
        if (root && !payload) records = reinterpret_cast<zoic_ray *>(out + lo * 32);   // straight into the caller's buffer
        else {
            if (L.records.cap < slab) {   // growing frees the old buffer: earlier calls must be through with it
                for (hipStream_t s : {L.compute[0], L.compute[1], L.copy}) FRAME_HIP(hipStreamSynchronize(s));
                FRAME_HIP(L.records.reserve(slab));
            }
            records = L.records.ptr;
        }
        if (payload && !root && L.payload.cap < slab * 7) {
            for (hipStream_t s : {L.compute[0], L.compute[1], L.copy}) FRAME_HIP(hipStreamSynchronize(s));
            FRAME_HIP(L.payload.reserve(slab * 7));
        }
        bool used[3] = {false, false, false};
        for (size_t k = 0; k < chunks; ++k) {
            const uint64_t a = lo + k * chunk, b = std::min(hi, a + chunk), m = b - a;
            hipStream_t cs = L.compute[k & 1];
            if (!used[k & 1]) {
                used[k & 1] = true;
                FRAME_HIP(hipStreamWaitEvent(cs, frame->rootStart, 0));
                if (!d_samples) FRAME_HIP(hipStreamWaitEvent(cs, L.samplesReady, 0));
            }
            zoic_ray *dst = records + (a - lo);
            if (zoic_status s = zoic_create_rays_device(L.cam, m, samples + (a - lo) * 4, nullptr, ray_index_base + a, dst, cs)) return s;
            if (payload) {
                float *rows = root ? reinterpret_cast<float *>(out + a * 28) : L.payload.ptr + (a - lo) * 7;
                if (int rc = launch_pack_payload(reinterpret_cast<const RayRecord *>(dst), rows, m, cs))
                    return fail_status(ZOIC_ERR_HIP, std::string("pack kernel: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
            }
            if (!root) {
                FRAME_HIP(hipEventRecord(L.computed[k], cs));
                if (!used[2]) { used[2] = true; FRAME_HIP(hipStreamWaitEvent(L.copy, frame->rootStart, 0)); }
                FRAME_HIP(hipStreamWaitEvent(L.copy, L.computed[k], 0));
                const void *src = payload ? static_cast<const void *>(L.payload.ptr + (a - lo) * 7) : static_cast<const void *>(dst);
                char *to = out + a * static_cast<uint64_t>(rowBytes);
                const size_t bytes = static_cast<size_t>(m) * static_cast<size_t>(rowBytes);
                if (L.device == R.device) FRAME_HIP(hipMemcpyAsync(to, src, bytes, hipMemcpyDeviceToDevice, L.copy));
                else FRAME_HIP(hipMemcpyPeerAsync(to, R.device, src, L.device, bytes, L.copy));
                L.lastBytesToRoot += bytes;
            }
        }
        // the caller's root stream continues behind everything this lane queued
        hipStream_t tails[3] = {L.compute[0], L.compute[1], L.copy};
        L.lastChunks = static_cast<uint32_t>(chunks);
        for (int t = 0; t < 3; ++t) {
            if (!used[t]) continue;
            FRAME_HIP(hipEventRecord(L.streamDone[t], tails[t]));
            L.doneRecorded[t] = true;
            DeviceGuard rootGuard(R.device);
            FRAME_HIP(rootGuard.error());
            FRAME_HIP(hipStreamWaitEvent(rootStream, L.streamDone[t], 0));
        }
    }
    return ZOIC_OK;
}

zoic_status render_local_impl(zoic_frame *frame, uint64_t n, const float *const *d_samples, uint64_t ray_index_base, zoic_ray *const *d_rays)
{
    const int nd = static_cast<int>(frame->lanes.size());
    for (int i = 0; i < nd; ++i) {
        Lane &L = frame->lanes[static_cast<size_t>(i)];
        uint64_t lo, hi;
        slab_of(n, nd, i, lo, hi);
        if (hi <= lo) continue;
        const float *samples = nullptr;
        if (zoic_status s = lane_samples(frame, L, static_cast<size_t>(i), d_samples, n, ray_index_base, samples)) return s;
        DeviceGuard guard(L.device);
        FRAME_HIP(guard.error());
        zoic_ray *dst = d_rays ? d_rays[i] : nullptr;
        if (!dst) {
            if (L.records.cap < hi - lo) {
                for (hipStream_t s : {L.compute[0], L.compute[1], L.copy}) FRAME_HIP(hipStreamSynchronize(s));
                FRAME_HIP(L.records.reserve(hi - lo));
            }
            dst = L.records.ptr;
        }
        // behind whatever the previous call left on ANY stream of the lane: an earlier gather may still be copying out of the
        // records, an earlier render_device may still be tracing into them on compute[1]
        if (zoic_status s = order_behind_previous(L)) return s;
        if (!d_samples) FRAME_HIP(hipStreamWaitEvent(L.compute[0], L.samplesReady, 0));
        if (zoic_status s = zoic_create_rays_device(L.cam, hi - lo, samples, nullptr, ray_index_base + lo, dst, L.compute[0])) return s;
        FRAME_HIP(hipEventRecord(L.streamDone[0], L.compute[0]));
        L.doneRecorded[0] = true;
    }
    return ZOIC_OK;
}

// ZOIC_FRAME_PAYLOAD_SPARSE.  SURVEY 8(e)'s gather ships 28 bytes for every ray; a ray with weight 0 -- four fifths of a wide-open
// PETZVAL frame (zoic.cpp:1951-1953), a fifth of a TESSAR's -- carries nothing a consumer reads.  A peer's chunk travels as a
// 256-bit live mask per 256-ray tile + the compacted 28-byte rows of the rays with weight != 0 (pack_sparse_kernel) and is
// expanded on the root (rows of weight-0 rays: seven zeros; their origin / direction -- the reference's partial state -- and their
// try counts stay on the device that traced them).  The size of a chunk is only known on the device: the host reads the chunk's
// live count (4 bytes, page-locked) before it queues the copy, i.e. it waits for the chunk's trace -- with the NEXT chunk of every
// lane already queued, so no device idles.  The loop runs chunk-major over the lanes for that reason; the call returns when the
// last chunk's size is known (the copies and expansions may still be in flight behind root_stream, as with the other layouts).
zoic_status render_sparse_impl(zoic_frame *frame, uint64_t n, const float *const *d_samples, uint64_t ray_index_base, void *d_out, void *root_stream)
{
    const int nd = static_cast<int>(frame->lanes.size());
    Lane &R = frame->lanes[0];
    hipStream_t rootStream = static_cast<hipStream_t>(root_stream);
    char *const out = static_cast<char *>(d_out);
    {
        DeviceGuard guard(R.device);
        FRAME_HIP(guard.error());
        FRAME_HIP(hipEventRecord(frame->rootStart, rootStream));
    }
    struct Plan { uint64_t lo = 0, hi = 0, chunk = 0; size_t chunks = 0; const float *samples = nullptr; };
    std::vector<Plan> plan(static_cast<size_t>(nd));
    size_t rounds = 0;
    for (int i = 0; i < nd; ++i) {
        Lane &L = frame->lanes[static_cast<size_t>(i)];
        Plan &P = plan[static_cast<size_t>(i)];
        slab_of(n, nd, i, P.lo, P.hi);
        L.lastRays = P.hi - P.lo; L.lastBytesToRoot = 0; L.lastChunks = 0;
        if (P.hi <= P.lo) continue;
        if (zoic_status s = lane_samples(frame, L, static_cast<size_t>(i), d_samples, n, ray_index_base, P.samples)) return s;
        const uint64_t slab = P.hi - P.lo;
        P.chunk = i == 0 ? slab : chunk_rays_for(frame, slab, 28);
        P.chunks = static_cast<size_t>((slab + P.chunk - 1) / P.chunk);
        L.lastChunks = static_cast<uint32_t>(P.chunks);
        rounds = std::max(rounds, P.chunks);
        DeviceGuard guard(L.device);
        FRAME_HIP(guard.error());
        if (zoic_status s = order_behind_previous(L)) return s;
        if (zoic_status s = ensure_chunk_events(L, P.chunks)) return s;
        if (L.records.cap < slab) {
            for (hipStream_t s : {L.compute[0], L.compute[1], L.copy}) FRAME_HIP(hipStreamSynchronize(s));
            FRAME_HIP(L.records.reserve(slab));
        }
        for (hipStream_t s : {L.compute[0], L.compute[1]}) {
            FRAME_HIP(hipStreamWaitEvent(s, frame->rootStart, 0));
            if (!d_samples) FRAME_HIP(hipStreamWaitEvent(s, L.samplesReady, 0));
        }
        if (i == 0) continue;
        FRAME_HIP(hipStreamWaitEvent(L.copy, frame->rootStart, 0));
        const size_t words = (sparse_header_bytes(P.chunk) + static_cast<size_t>(P.chunk) * 28u) / 4u;
        while (L.copied.size() < P.chunks) { hipEvent_t e = nullptr; FRAME_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); L.copied.push_back(e); }
        if (!L.dCount) FRAME_HIP(hipMalloc(reinterpret_cast<void **>(&L.dCount), kMaxChunksPerSlab * sizeof(unsigned int)));
        if (!L.hCount) FRAME_HIP(hipHostMalloc(reinterpret_cast<void **>(&L.hCount), kMaxChunksPerSlab * sizeof(unsigned int), hipHostMallocDefault));
        for (int b = 0; b < 2; ++b)
            if (L.sparse[b].cap < words) {
                for (hipStream_t s : {L.compute[0], L.compute[1], L.copy}) FRAME_HIP(hipStreamSynchronize(s));
                FRAME_HIP(L.sparse[b].reserve(words));
            }
        {   // the root's side of this lane
            DeviceGuard rootGuard(R.device);
            FRAME_HIP(rootGuard.error());
            if (!L.rootExpand) FRAME_HIP(hipStreamCreateWithFlags(&L.rootExpand, hipStreamNonBlocking));
            if (!L.expandDone) FRAME_HIP(hipEventCreateWithFlags(&L.expandDone, hipEventDisableTiming));
            while (L.expanded.size() < P.chunks) { hipEvent_t e = nullptr; FRAME_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); L.expanded.push_back(e); }
            for (int b = 0; b < 2; ++b)
                if (L.rootStage[b].cap < words) { FRAME_HIP(hipStreamSynchronize(L.rootExpand)); FRAME_HIP(L.rootStage[b].reserve(words)); }
            FRAME_HIP(hipStreamWaitEvent(L.rootExpand, frame->rootStart, 0));   // d_out's previous readers first
        }
    }
    // chunk k of a peer: its live count is on the host -> the copy, sized; then the expansion on the root
    const auto finish_chunk = [&](int i, size_t k) -> zoic_status {
        Lane &L = frame->lanes[static_cast<size_t>(i)];
        const Plan &P = plan[static_cast<size_t>(i)];
        const uint64_t a = P.lo + k * P.chunk, b = std::min(P.hi, a + P.chunk), m = b - a;
        DeviceGuard guard(L.device);
        FRAME_HIP(guard.error());
        FRAME_HIP(hipEventSynchronize(L.computed[k]));
        const size_t bytes = sparse_header_bytes(m) + static_cast<size_t>(L.hCount[k]) * 28u;
        FRAME_HIP(hipStreamWaitEvent(L.copy, L.computed[k], 0));
        if (k >= 2) FRAME_HIP(hipStreamWaitEvent(L.copy, L.expanded[k - 2], 0));   // the root's staging buffer is free again
        if (L.device == R.device) FRAME_HIP(hipMemcpyAsync(L.rootStage[k & 1].ptr, L.sparse[k & 1].ptr, bytes, hipMemcpyDeviceToDevice, L.copy));
        else FRAME_HIP(hipMemcpyPeerAsync(L.rootStage[k & 1].ptr, R.device, L.sparse[k & 1].ptr, L.device, bytes, L.copy));
        FRAME_HIP(hipEventRecord(L.copied[k], L.copy));
        L.lastBytesToRoot += bytes;
        DeviceGuard rootGuard(R.device);
        FRAME_HIP(rootGuard.error());
        FRAME_HIP(hipStreamWaitEvent(L.rootExpand, L.copied[k], 0));
        if (int rc = launch_expand_sparse(L.rootStage[k & 1].ptr, reinterpret_cast<float *>(out + a * 28u), m, L.rootExpand))
            return fail_status(ZOIC_ERR_HIP, std::string("expand kernel: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
        FRAME_HIP(hipEventRecord(L.expanded[k], L.rootExpand));
        return ZOIC_OK;
    };
    for (size_t k = 0; k < rounds + 1; ++k) {
        for (int i = 0; i < nd && k < rounds; ++i) {
            Lane &L = frame->lanes[static_cast<size_t>(i)];
            const Plan &P = plan[static_cast<size_t>(i)];
            if (k >= P.chunks) continue;
            const uint64_t a = P.lo + k * P.chunk, b = std::min(P.hi, a + P.chunk), m = b - a;
            DeviceGuard guard(L.device);
            FRAME_HIP(guard.error());
            hipStream_t cs = L.compute[k & 1];
            zoic_ray *dst = L.records.ptr + (a - P.lo);
            if (zoic_status s = zoic_create_rays_device(L.cam, m, P.samples + (a - P.lo) * 4, nullptr, ray_index_base + a, dst, cs)) return s;
            if (i == 0) {   // the root's own slab: the same rows, straight into the output
                if (int rc = launch_pack_payload_live(reinterpret_cast<const RayRecord *>(dst), reinterpret_cast<float *>(out + a * 28u), m, cs))
                    return fail_status(ZOIC_ERR_HIP, std::string("pack kernel: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
                continue;
            }
            if (k >= 2) FRAME_HIP(hipStreamWaitEvent(cs, L.copied[k - 2], 0));   // this device's sparse buffer has left
            FRAME_HIP(hipMemsetAsync(L.dCount + k, 0, sizeof(unsigned int), cs));
            if (int rc = launch_pack_sparse(reinterpret_cast<const RayRecord *>(dst), L.sparse[k & 1].ptr, L.dCount + k, m, cs))
                return fail_status(ZOIC_ERR_HIP, std::string("pack kernel: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
            FRAME_HIP(hipMemcpyAsync(L.hCount + k, L.dCount + k, sizeof(unsigned int), hipMemcpyDeviceToHost, cs));
            FRAME_HIP(hipEventRecord(L.computed[k], cs));
        }
        if (k == 0) continue;
        for (int i = 1; i < nd; ++i)
            if (k - 1 < plan[static_cast<size_t>(i)].chunks)
                if (zoic_status s = finish_chunk(i, k - 1)) return s;
    }
    // the caller's root stream continues behind everything
    for (int i = 0; i < nd; ++i) {
        Lane &L = frame->lanes[static_cast<size_t>(i)];
        if (plan[static_cast<size_t>(i)].chunks == 0) continue;
        DeviceGuard guard(L.device);
        FRAME_HIP(guard.error());
        hipStream_t tails[3] = {L.compute[0], L.compute[1], L.copy};
        for (int t = 0; t < 3; ++t) {
            if (i == 0 && t > 0) continue;   // the root lane is one launch on compute[0]
            FRAME_HIP(hipEventRecord(L.streamDone[t], tails[t]));
            L.doneRecorded[t] = true;
            DeviceGuard rootGuard(R.device);
            FRAME_HIP(rootGuard.error());
            FRAME_HIP(hipStreamWaitEvent(rootStream, L.streamDone[t], 0));
        }
        if (i > 0) {
            DeviceGuard rootGuard(R.device);
            FRAME_HIP(rootGuard.error());
            FRAME_HIP(hipEventRecord(L.expandDone, L.rootExpand));
            L.expandRecorded = true;
            FRAME_HIP(hipStreamWaitEvent(rootStream, L.expandDone, 0));
        }
    }
    return ZOIC_OK;
}

}  // namespace

extern "C" {

zoic_status zoic_frame_render_device(zoic_frame *frame, uint64_t n, const float *const *d_samples, uint64_t ray_index_base, void *d_out,
                                     zoic_frame_layout layout, void *root_stream)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    if (layout != ZOIC_FRAME_RECORDS && layout != ZOIC_FRAME_PAYLOAD && layout != ZOIC_FRAME_PAYLOAD_SPARSE && layout != ZOIC_FRAME_PAYLOAD_AUTO)
        return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "bad layout");
    if (n == 0) return ZOIC_OK;
    if (!d_out || (reinterpret_cast<uintptr_t>(d_out) & 15u)) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "d_out must be non-NULL and 16-byte aligned");
    if (layout == ZOIC_FRAME_PAYLOAD_AUTO) {
        if (!frame->autoDecided && frame->autoRendered != 0) {
            // the frame has rendered since its tables were built: what share of those rays had weight 0?  (Any ray call counts -- the
            // cameras' counters do not know who asked.)  One synchronisation, once per update.
            if (zoic_status s = zoic_frame_synchronize(frame)) return s;
            zoic_counters c;
            if (zoic_status s = zoic_frame_get_counters(frame, &c)) return s;
            const uint64_t succ = c.succesRays - std::min<uint64_t>(frame->autoBaseSucc, c.succesRays), vign = c.vignettedRays - std::min<uint64_t>(frame->autoBaseVign, c.vignettedRays);
            if (succ + vign != 0) {
                frame->autoZeroWeight = static_cast<double>(vign) / static_cast<double>(succ + vign);
                frame->autoSparse = frame->autoZeroWeight >= kAutoSparseZeroWeight;
                frame->autoDecided = true;
            }
        }
        frame->autoRendered += n;
        layout = (frame->autoDecided && frame->autoSparse) ? ZOIC_FRAME_PAYLOAD_SPARSE : ZOIC_FRAME_PAYLOAD;
    }
    if (layout == ZOIC_FRAME_PAYLOAD_SPARSE) {
        if (zoic_status s = render_sparse_impl(frame, n, d_samples, ray_index_base, d_out, root_stream)) return settle_after_failure(frame, s);
        return ZOIC_OK;
    }
    if (zoic_status s = render_device_impl(frame, n, d_samples, ray_index_base, d_out, layout, root_stream)) return settle_after_failure(frame, s);
    return ZOIC_OK;
}

int zoic_frame_auto_layout(const zoic_frame *frame, double *zero_weight_fraction)
{
    if (!frame) return -1;
    if (zero_weight_fraction) *zero_weight_fraction = frame->autoZeroWeight;
    if (!frame->autoDecided) return -1;
    return frame->autoSparse ? ZOIC_FRAME_PAYLOAD_SPARSE : ZOIC_FRAME_PAYLOAD;
}

zoic_status zoic_frame_render_local(zoic_frame *frame, uint64_t n, const float *const *d_samples, uint64_t ray_index_base, zoic_ray *const *d_rays)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    if (n == 0) return ZOIC_OK;
    if (zoic_status s = render_local_impl(frame, n, d_samples, ray_index_base, d_rays)) return settle_after_failure(frame, s);
    return ZOIC_OK;
}

zoic_status zoic_frame_get_lane_info(const zoic_frame *frame, int i, zoic_frame_lane_info *out)
{
    if (!frame || !out || i < 0 || static_cast<size_t>(i) >= frame->lanes.size()) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "bad lane");
    const Lane &L = frame->lanes[static_cast<size_t>(i)];
    std::memset(out, 0, sizeof(*out));
    out->device = L.device;
    out->peer_access_to_root = L.peerToRoot; out->peer_access_from_root = L.rootToPeer;
    out->rays = L.lastRays; out->bytes_to_root = L.lastBytesToRoot; out->chunks = L.lastChunks;
    return ZOIC_OK;
}

zoic_status zoic_frame_render_host(zoic_frame *frame, uint64_t n, const float *h_samples, uint64_t ray_index_base, zoic_ray *h_rays)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    if (n == 0) return ZOIC_OK;
    if (!h_samples || !h_rays) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "NULL host buffer");
    const int nd = static_cast<int>(frame->lanes.size());
    std::vector<zoic_status> st(static_cast<size_t>(nd), ZOIC_OK);
    std::vector<std::string> why(static_cast<size_t>(nd));
    const auto run = [&](int i) {
        uint64_t lo, hi;
        slab_of(n, nd, i, lo, hi);
        if (hi <= lo) return;
        const size_t k = static_cast<size_t>(i);
        st[k] = zoic_create_rays_host(frame->lanes[k].cam, hi - lo, h_samples + lo * 4, nullptr, ray_index_base + lo, h_rays + lo);
        if (st[k] != ZOIC_OK) why[k] = zoic_last_error_string();
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nd; ++i) th.emplace_back(run, i);
    run(0);
    for (std::thread &t : th) t.join();
    for (size_t i = 0; i < st.size(); ++i)
        if (st[i] != ZOIC_OK) return fail_status(st[i], why[i]);
    return ZOIC_OK;
}

zoic_status zoic_frame_synchronize(zoic_frame *frame)
{
    if (!frame) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "frame is NULL");
    for (Lane &L : frame->lanes) {
        if (!L.cam) continue;
        DeviceGuard guard(L.device);
        FRAME_HIP(guard.error());
        for (hipStream_t s : {L.compute[0], L.compute[1], L.copy})
            if (s) FRAME_HIP(hipStreamSynchronize(s));
        if (L.rootExpand) {
            DeviceGuard rootGuard(frame->lanes[0].device);
            FRAME_HIP(rootGuard.error());
            FRAME_HIP(hipStreamSynchronize(L.rootExpand));
        }
    }
    return ZOIC_OK;
}

zoic_status zoic_frame_get_counters(zoic_frame *frame, zoic_counters *sum)
{
    if (!frame || !sum) return fail_status(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    zoic_counters total{0, 0, 0};
    for (Lane &L : frame->lanes) {
        zoic_counters c;
        if (zoic_status s = zoic_camera_get_counters(L.cam, &c)) return s;
        total.succesRays += c.succesRays; total.vignettedRays += c.vignettedRays;
        total.totalInternalReflection += c.totalInternalReflection;
        // node_update's own traces bump the TIR counter (zoic.cpp:1135 ff.) on EVERY device's camera; the reference has one
        // node: they are counted once (the root's)
        if (&L != &frame->lanes[0]) {
            zoic_lens_info info;
            if (zoic_status s = zoic_camera_get_info(L.cam, &info)) return s;
            total.totalInternalReflection -= std::min<uint64_t>(info.precomputeTIR, total.totalInternalReflection);
        }
    }
    *sum = total;
    return ZOIC_OK;
}

}  // extern "C"
