// capi.cpp -- the C-ABI of libzoic_amd.so (include/zoic_amd.h): zoic's Arnold node methods restated over plain
// pointers.  Host logic only; every ray is produced by the HIP kernels (kernels.hip / kolb_fast.hip).
//
//   zoic_camera_create   <- node_initialize  zoic.cpp:1565-1572
//   zoic_camera_update   <- node_update      zoic.cpp:1575-1720
//   zoic_create_rays_*   <- camera_create_ray zoic.cpp:1752-1990
//   zoic_camera_destroy  <- node_finish      zoic.cpp:1723-1749
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/zoic_amd.h"
#include "kernels.hpp"
#include "lens_system.hpp"

#pragma STDC FP_CONTRACT OFF

using namespace zoic;

namespace {

thread_local std::string g_lastError;

constexpr unsigned kWorkCursors = 64;   // launches of one camera that may be in flight at once
constexpr unsigned kCursorStride = kCursorParts * kCursorPartStride;  // one launch's set of partition cursors (kernels.hpp)
// a >2^31-sample call splits into several launches, each taking the next slot of the ring (kernels.hip)

zoic_status fail(zoic_status s, const std::string &msg)
{
    g_lastError = msg;
    return s;
}

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

template <class T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

}  // namespace

struct zoic_camera {  // struct cameraData, zoic.cpp:627-643
    int device = 0;
    OwnedParams params;            // camera->params (last applied)
    LensSystem lens;               // camera->lens
    BokehCdf image;                // camera->image
    float fov = 0, tanFov = 0, apertureRadius = 0;
    Rng stream{};                  // the xor128 function-static state (zoic.cpp:648): LUT build draws from it
    zoic_precision precision = ZOIC_PRECISION_STRICT;
    uint32_t seed = 1;
    bool updated = false;
    bool lutOnHost = false;
    // pending inputs
    std::vector<float> pendingPixels; int pendW = 0, pendH = 0, pendC = 0;
    std::string lensText; bool haveLensText = false;
    // flattened tables
    KolbTable kolb{};
    ThinTable thin{};
    // device state
    DeviceBuffer<float> dCdfRow, dCdfColumn, dPyramid;
    DeviceBuffer<uint32_t> dBokehCells;
    DeviceBuffer<int32_t> dRowIdx, dColIdx;
    BokehTables bokehDev{};
    DeviceCounters *dCounters = nullptr;
    DeviceBuffer<float> dSamples, dInputs7, dProbeU, dProbeV;
    DeviceBuffer<RayRecord> dRays;
    DeviceBuffer<uint32_t> dRng;
    DeviceBuffer<uint8_t> dProbeOk;
    unsigned int *dProbeTir = nullptr;
    unsigned int *dWorkCursor = nullptr;   // ring of kWorkCursors chunk cursors: launches in flight on different streams never share one
    unsigned int nextCursor = 0;
};

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
    // cell records (tables.hpp): rows <= 2048 (the row records live in LDS: 32 KB at most), columns <= 4096, both CDFs
    // non-decreasing and NaN-free (anything else keeps the pyramid / reference search)
    if (im.x <= 4096 && im.y <= 2048) {
        const auto monotone = [](const float *a, int n) {
            for (int i = 0; i < n; ++i) if (!(a[i] == a[i]) || (i > 0 && a[i] < a[i - 1])) return false;
            return true;
        };
        bool ok = monotone(im.cdfRow.data(), im.y);
        for (size_t r = 0; ok && r < y; ++r) ok = monotone(im.cdfColumn.data() + r * im.x, im.x);
        if (ok) {
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
                        const float inf = INFINITY;
                        const float a = lo < n ? cdf[lo] : inf, b = lo + 1 < n ? cdf[lo + 1] : inf;
                        uint32_t id[3];
                        for (int k = 0; k < 3; ++k) id[k] = static_cast<uint32_t>(idx[std::min(lo + k, n - 1)] - idxBase) & 0xffffu;
                        std::memcpy(rec + 0, &a, 4);
                        std::memcpy(rec + 1, &b, 4);
                        rec[2] = id[0] | (id[1] << 16);
                        rec[3] = id[2] | ((hi - lo > 2) ? 0x80000000u : 0u);
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

}  // namespace

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
    ZOIC_HIP(hipSetDevice(device));
    zoic_camera *cam = new zoic_camera();
    cam->device = device;
    rng_seed_reference(cam->stream);
    const char *env = std::getenv("ZOIC_LUT_HOST");
    cam->lutOnHost = env && env[0] == '1';
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&cam->dCounters), sizeof(DeviceCounters));
    if (e == hipSuccess) e = hipMemset(cam->dCounters, 0, sizeof(DeviceCounters));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&cam->dWorkCursor), kWorkCursors * kCursorStride * sizeof(unsigned int));
    if (e != hipSuccess) {
        delete cam;
        return fail(ZOIC_ERR_HIP, std::string("counter allocation: ") + hipGetErrorString(e));
    }
    *out = cam;
    return ZOIC_OK;
}

void zoic_camera_destroy(zoic_camera *cam)
{
    if (!cam) return;
    if (cam->device == ZOIC_DEVICE_NONE) { delete cam; return; }
    (void)hipSetDevice(cam->device);
    cam->dCdfRow.release(); cam->dCdfColumn.release(); cam->dRowIdx.release(); cam->dColIdx.release(); cam->dPyramid.release(); cam->dBokehCells.release();
    cam->dSamples.release(); cam->dRays.release(); cam->dInputs7.release(); cam->dRng.release();
    cam->dProbeU.release(); cam->dProbeV.release(); cam->dProbeOk.release();
    if (cam->dProbeTir) (void)hipFree(cam->dProbeTir);
    if (cam->dCounters) (void)hipFree(cam->dCounters);
    if (cam->dWorkCursor) (void)hipFree(cam->dWorkCursor);
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
    return ZOIC_OK;
}

zoic_status zoic_camera_set_lens_text(zoic_camera *cam, const char *text, size_t len)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    cam->haveLensText = text != nullptr;
    cam->lensText.assign(text ? text : "", text ? len : 0);
    return ZOIC_OK;
}

zoic_status zoic_camera_set_precision(zoic_camera *cam, zoic_precision mode)
{
    if (!cam || (mode != ZOIC_PRECISION_STRICT && mode != ZOIC_PRECISION_FAST)) return fail(ZOIC_ERR_INVALID_ARGUMENT, "bad precision");
    cam->precision = mode;
    return ZOIC_OK;
}

zoic_status zoic_camera_set_seed(zoic_camera *cam, uint32_t seed)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    cam->seed = seed;
    cam->kolb.seed = seed;
    cam->thin.seed = seed;
    return ZOIC_OK;
}

zoic_status zoic_camera_update(zoic_camera *cam, const zoic_params *p)
{
    if (!cam || !p) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    const bool onDevice = cam->device != ZOIC_DEVICE_NONE;
    if (onDevice) ZOIC_HIP(hipSetDevice(cam->device));
    zoic_status status = ZOIC_OK;
    const std::string bokehPath = p->bokehPath ? p->bokehPath : "", lensPath = p->lensDataPath ? p->lensDataPath : "";

    // bokeh image -> CDF tables, zoic.cpp:1587-1593
    if (params_bokeh_changed(*p, cam->params)) {
        cam->image.clear();
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
            if (!ok) status = fail(ZOIC_ERR_BOKEH_IMAGE, "[ZOIC] Couldn't open bokeh image!");
            else if (onDevice) { if (zoic_status s = upload_bokeh(cam)) return s; }
        }
    }
    const bool imageOn = p->useImage && cam->image.valid();

    switch (p->lensModel) {
    case ZOIC_THINLENS:  // zoic.cpp:1598-1610
        cam->fov = static_cast<float>(2.0f * std::atan(static_cast<double>(p->sensorWidth / (2.0f * p->focalLength))));
        cam->tanFov = tanf(cam->fov / 2.0f);
        cam->apertureRadius = p->focalLength / (2.0f * p->fStop);
        break;
    case ZOIC_RAYTRACED:  // zoic.cpp:1612-1711
        if (params_lens_changed(*p, cam->params)) {
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
            cam->updated = false;
            if (zoic_status s = lens_error_status(cam->lens.parse(text.data(), text.size()))) return s;
            g_lastError.clear();
            LensError le = cam->lens.prepare(p->focalLength, p->fStop, p->focalDistance, p->kolbSamplingLUT != 0, cam->stream,
                                             cam->lutOnHost ? lut_trace_host : lut_trace_device, cam);
            if (zoic_status s = lens_error_status(le)) return s;
            if (!g_lastError.empty()) return ZOIC_ERR_HIP;
            // counters restart with the lens (zoic.cpp:1626-1628); the precompute's TIR bumps stay in (zoic.cpp:1135 ff.)
            DeviceCounters zero{0, 0, cam->lens.precomputeTIR};
            if (onDevice) ZOIC_HIP(hipMemcpy(cam->dCounters, &zero, sizeof(zero), hipMemcpyHostToDevice));
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
        cam->lens.fill_table(cam->kolb, p->sensorWidth);
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
    cam->updated = (status == ZOIC_OK);
    return status;
}

zoic_status zoic_create_rays_device(zoic_camera *cam, uint64_t n, const float *d_samples, const uint32_t *d_rng_states,
                                    uint64_t ray_index_base, zoic_ray *d_rays, void *stream)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device == ZOIC_DEVICE_NONE) return fail(ZOIC_ERR_NO_DEVICE, "tables-only camera: rays need a gfx950 device (no CPU path)");
    if (!cam->updated) return fail(ZOIC_ERR_NOT_UPDATED, "zoic_camera_update has not succeeded yet");
    if (n == 0) return ZOIC_OK;
    if (!d_samples) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_samples is NULL");
    if (reinterpret_cast<uintptr_t>(d_samples) & 15u) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_samples must be 16-byte aligned");
    if (d_rng_states && (reinterpret_cast<uintptr_t>(d_rng_states) & 15u)) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_rng_states must be 16-byte aligned");
    if (!d_rays || (reinterpret_cast<uintptr_t>(d_rays) & 15u)) return fail(ZOIC_ERR_INVALID_ARGUMENT, "d_rays must be non-NULL and 16-byte aligned");
    static_assert(sizeof(zoic_ray) == sizeof(RayRecord), "zoic_ray layout");
    ZOIC_HIP(hipSetDevice(cam->device));
    RayRecord *planes = reinterpret_cast<RayRecord *>(d_rays);
    int rc = 0;
    switch (cam->params.p.lensModel) {
    case ZOIC_RAYTRACED:
        rc = launch_kolb_rays(cam->kolb, bokeh_tables(cam), d_samples, d_rng_states, ray_index_base, n, planes, cam->dCounters,
                              cam->dWorkCursor + (cam->nextCursor++ % kWorkCursors) * kCursorStride,
                              cam->precision == ZOIC_PRECISION_FAST, stream);
        break;
    case ZOIC_THINLENS:
        rc = launch_thin_rays(cam->thin, bokeh_tables(cam), d_samples, d_rng_states, ray_index_base, n, planes, cam->dCounters,
                              cam->dWorkCursor + (cam->nextCursor++ % kWorkCursors) * kCursorStride,
                              cam->precision == ZOIC_PRECISION_FAST, stream);
        break;
    default:
        return fail(ZOIC_ERR_INVALID_ARGUMENT, "lensModel NONE produces no rays (zoic.cpp:1966-1968)");
    }
    if (rc != 0) return fail(ZOIC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
    return ZOIC_OK;
}

zoic_status zoic_create_rays_host(zoic_camera *cam, uint64_t n, const float *h_samples, const uint32_t *h_rng_states,
                                  uint64_t ray_index_base, zoic_ray *h_rays)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device == ZOIC_DEVICE_NONE) return fail(ZOIC_ERR_NO_DEVICE, "tables-only camera: rays need a gfx950 device (no CPU path)");
    if (!cam->updated) return fail(ZOIC_ERR_NOT_UPDATED, "zoic_camera_update has not succeeded yet");
    if (n == 0) return ZOIC_OK;
    if (!h_samples) return fail(ZOIC_ERR_INVALID_ARGUMENT, "h_samples is NULL");
    ZOIC_HIP(hipSetDevice(cam->device));
    if (!h_rays) return fail(ZOIC_ERR_INVALID_ARGUMENT, "h_rays is NULL");
    ZOIC_HIP(cam->dSamples.reserve(n * 4));
    ZOIC_HIP(cam->dRays.reserve(n));
    ZOIC_HIP(hipMemcpy(cam->dSamples.ptr, h_samples, n * 16, hipMemcpyHostToDevice));
    const uint32_t *dRng = nullptr;
    if (h_rng_states) {
        ZOIC_HIP(cam->dRng.reserve(n * 4));
        ZOIC_HIP(hipMemcpy(cam->dRng.ptr, h_rng_states, n * 16, hipMemcpyHostToDevice));
        dRng = cam->dRng.ptr;
    }
    if (zoic_status s = zoic_create_rays_device(cam, n, cam->dSamples.ptr, dRng, ray_index_base,
                                                reinterpret_cast<zoic_ray *>(cam->dRays.ptr), nullptr)) return s;
    ZOIC_HIP(hipDeviceSynchronize());
    ZOIC_HIP(hipMemcpy(h_rays, cam->dRays.ptr, n * sizeof(zoic_ray), hipMemcpyDeviceToHost));
    return ZOIC_OK;
}

zoic_status zoic_create_rays_arnold(zoic_camera *cam, uint64_t n, const zoic_camera_input *inputs, zoic_camera_output *outputs,
                                    uint64_t ray_index_base)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device == ZOIC_DEVICE_NONE) return fail(ZOIC_ERR_NO_DEVICE, "tables-only camera: rays need a gfx950 device (no CPU path)");
    if (!cam->updated) return fail(ZOIC_ERR_NOT_UPDATED, "zoic_camera_update has not succeeded yet");
    if (n == 0) return ZOIC_OK;
    if (!inputs || !outputs) return fail(ZOIC_ERR_INVALID_ARGUMENT, "NULL argument");
    static_assert(sizeof(zoic_camera_input) == 28 && sizeof(zoic_camera_output) == 84, "Arnold POD layout");
    ZOIC_HIP(hipSetDevice(cam->device));
    ZOIC_HIP(cam->dInputs7.reserve(n * 7));
    ZOIC_HIP(cam->dSamples.reserve(n * 4));
    ZOIC_HIP(cam->dRays.reserve(n));
    ZOIC_HIP(hipMemcpy(cam->dInputs7.ptr, inputs, n * sizeof(zoic_camera_input), hipMemcpyHostToDevice));
    if (int rc = launch_pack_inputs(cam->dInputs7.ptr, cam->dSamples.ptr, n, nullptr))
        return fail(ZOIC_ERR_HIP, std::string("pack kernel: ") + hipGetErrorString(static_cast<hipError_t>(rc)));
    if (zoic_status s = zoic_create_rays_device(cam, n, cam->dSamples.ptr, nullptr, ray_index_base,
                                                reinterpret_cast<zoic_ray *>(cam->dRays.ptr), nullptr)) return s;
    std::vector<zoic_ray> h(n);
    ZOIC_HIP(hipMemcpy(h.data(), cam->dRays.ptr, n * sizeof(zoic_ray), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        zoic_camera_output &o = outputs[i];
        const zoic_ray &r = h[i];
        o.origin = zoic_vec3{r.ox, r.oy, r.oz};
        o.dir = zoic_vec3{r.dx, r.dy, r.dz};
        const float w = r.weight;
        if (w == 0.0f) o.weight[0] = o.weight[1] = o.weight[2] = 0.0f;  // output.weight = 0.0f, zoic.cpp:1825/1952
        else if (w != 1.0f) { o.weight[0] *= w; o.weight[1] *= w; o.weight[2] *= w; }  // exposure factor
        if (r.flags & 1u) { o.dOdy = o.origin; o.dDdy = o.dir; }  // zoic.cpp:1974-1977
    }
    return ZOIC_OK;
}

zoic_status zoic_camera_create_ray(zoic_camera *cam, const zoic_camera_input *input, zoic_camera_output *output, uint16_t tid)
{
    // camera_create_ray(node, input, output, tid): `tid` is unused by the reference as well (zoic.cpp:1752).
    (void)tid;
    return zoic_create_rays_arnold(cam, 1, input, output, 0);
}

zoic_status zoic_generate_samples_device(zoic_camera *cam, uint64_t n, uint64_t ray_index_base, uint32_t width, uint32_t height,
                                         uint32_t spp, uint32_t seed, float *d_samples, void *stream)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (!d_samples || width == 0 || height == 0 || spp == 0) return fail(ZOIC_ERR_INVALID_ARGUMENT, "bad sample grid");
    if (cam->device == ZOIC_DEVICE_NONE) return fail(ZOIC_ERR_NO_DEVICE, "tables-only camera");
    ZOIC_HIP(hipSetDevice(cam->device));
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
    ZOIC_HIP(hipSetDevice(cam->device));
    ZOIC_HIP(hipDeviceSynchronize());
    DeviceCounters c{};
    ZOIC_HIP(hipMemcpy(&c, cam->dCounters, sizeof(c), hipMemcpyDeviceToHost));
    out->succesRays = c.succes; out->vignettedRays = c.vignetted; out->totalInternalReflection = c.tir;
    return ZOIC_OK;
}

zoic_status zoic_camera_reset_counters(zoic_camera *cam)
{
    if (!cam) return fail(ZOIC_ERR_INVALID_ARGUMENT, "cam is NULL");
    if (cam->device == ZOIC_DEVICE_NONE) return ZOIC_OK;
    ZOIC_HIP(hipSetDevice(cam->device));
    ZOIC_HIP(hipMemset(cam->dCounters, 0, sizeof(DeviceCounters)));
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
