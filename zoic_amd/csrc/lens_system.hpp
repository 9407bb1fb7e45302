// lens_system.hpp -- host side of node_update for the RAYTRACED model (zoic.cpp:1612-1711): parse the tabular
// lens prescription, normalise it, rescale it to the requested focal length, focus it, and build the
// exit-pupil LUT.  Produces the KolbTable the HIP kernels take as a kernel argument.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "optics.hpp"
#include "tables.hpp"

namespace zoic {

enum class LensError {
    None = 0,
    Columns,        // zoic.cpp:745-754
    Parse,          // std::stof would throw, zoic.cpp:774 ff.
    MultiAperture,  // zoic.cpp:926-929
    NoAperture,     // apertureElement never written (zoic.cpp:922) -> reading it is UB: rejected
    TooManySurfaces
};

// One row of the prescription after parsing (LensElement, zoic.cpp:522-525)
struct LensRow {
    float radius = 0, thickness = 0, ior = 0, aperture = 0, abbe = 0, center = 0;
};

// decision-safe FAST mode (lens_system.cpp fill_surfaces): interfaces whose rounding-noise estimate eps*|R|/sqrt(housing2)
// exceeds kGuardMinRelBand are guarded with a band of kGuardScale x the estimate
#ifndef ZOIC_GUARD_SCALE
#define ZOIC_GUARD_SCALE 2.0f   // experiments: -DZOIC_GUARD_SCALE=x (0: no ray is ever listed)
#endif
constexpr float kGuardScale = ZOIC_GUARD_SCALE;
// (experiments: ZOIC_FAST_STABLE_STOP in tables.hpp; -DZOIC_GUARD_SCALE_FLAT=x then sizes the band of a near-planar interface)
#ifndef ZOIC_GUARD_SCALE_FLAT
#define ZOIC_GUARD_SCALE_FLAT ZOIC_GUARD_SCALE   // the band of such an interface, in units of eps*|R|/sqrt(housing2)
#endif
constexpr float kGuardScaleFlat = ZOIC_FAST_STABLE_STOP ? ZOIC_GUARD_SCALE_FLAT : ZOIC_GUARD_SCALE;
// share of the sensor square the retry-dead test must classify for the in-kernel completion of such rays to be compiled in
// (fill_table; 0: for every camera with a LUT, >= 1: never -- experiments: -DZOIC_RETRY_DEAD_MIN_SHARE=x)
#ifndef ZOIC_RETRY_DEAD_MIN_SHARE
#define ZOIC_RETRY_DEAD_MIN_SHARE 0.02
#endif
constexpr double kRetryDeadMinShare = ZOIC_RETRY_DEAD_MIN_SHARE;
constexpr float kGuardMinRelBand = 2.0e-5f;
// smallest band of an interface, relative to housing^2: FAST's error at the front elements is what it accumulated on the way
// (measured: flips at well-conditioned interfaces with margins up to 1.5e-6, tools/flip_analysis.py on 33 M rays per config)
#ifndef ZOIC_GUARD_FLOOR
#define ZOIC_GUARD_FLOOR 3.8e-6f   // 64 eps
#endif
constexpr float kGuardFloorRel = ZOIC_GUARD_FLOOR;
#ifndef ZOIC_GUARD_ALL
#define ZOIC_GUARD_ALL 1   // every interface carries its band (0: only those whose estimate exceeds kGuardMinRelBand, round 2)
#endif


struct LutBox { float maxX = 0, maxY = 0, minX = 0, minY = 0; };  // boundingBox2d, zoic.cpp:490-493

// Accept/reject of one batch of exit-pupil probe rays.  The default implementation traces on the host;
// the GPU build swaps in a kernel launch (same strict arithmetic) -- see capi.cpp.
using LutTraceFn = void (*)(const KolbTable &table, float originX, const float *lensU, const float *lensV, size_t n,
                            uint8_t *accepted, uint32_t *tirCount, void *user);

// The whole exit-pupil LUT in one go (lut_build.hip: draws, traces and boxes on the GPU).  Returns 0 when `boxes`, `*tirCount`
// and the advanced `rng` are the reference's; anything else leaves them untouched and the per-entry path below runs.
using LutBuildFn = int (*)(const KolbTable &table, Rng &rng, LutBox boxes[kLutEntries], uint32_t *tirCount, void *user);
int build_lut_device(const KolbTable &table, Rng &rng, LutBox boxes[kLutEntries], uint32_t *tirCount);

class LensSystem {
public:
    // readTabularLensData, zoic.cpp:708-914
    LensError parse(const char *text, size_t len);
    // cleanupLensData .. computeLensCenters (+ exitPupilLUT when useLUT), zoic.cpp:1648-1692.
    // `rng` is the process-wide xorshift128 stream the LUT build draws from (zoic.cpp:1411-1412).
    LensError prepare(float focalLength, float fStop, float focalDistance, bool useLUT, Rng &rng,
                      LutTraceFn trace = nullptr, void *traceUser = nullptr, LutBuildFn whole = nullptr);
    // flatten for the kernels
    // bokehW / bokehH: the bokeh image the lens samples come from (0: the concentric disk mapping) -- bounds the retry-dead test
    void fill_table(KolbTable &t, float sensorWidth, int bokehW = 0, int bokehH = 0) const;

    std::vector<LensRow> rows;       // rear -> front after parse()
    int apertureElement = -1;
    float userApertureRadius = 0, originShift = 0, apertureDistance = 0, focalLengthRatio = 0;
    float tracedFocalLength[2] = {0, 0};
    uint32_t precomputeTIR = 0;      // ld->totalInternalReflection bumps during precompute
    bool hasLUT = false;
    float lutKey[kLutEntries] = {};
    LutBox lutBox[kLutEntries];

private:
    float trace_focal_length();                       // traceThroughLensElementsForFocalLength, zoic.cpp:1161-1228
    float image_distance(float objectDistance);       // calculateImageDistance, zoic.cpp:1054-1095
    void build_lut(Rng &rng, LutTraceFn trace, void *user, LutBuildFn whole);  // exitPupilLUT, zoic.cpp:1391-1452
    void fill_surfaces(KolbTable &t) const;
};

// host accept/reject (reference behaviour of traceThroughLensElementsForApertureSize, zoic.cpp:1309-1350)
void lut_trace_host(const KolbTable &table, float originX, const float *lensU, const float *lensV, size_t n,
                    uint8_t *accepted, uint32_t *tirCount, void *user);

// imageData::bokehProbability, zoic.cpp:222-417.  Ties in the two descending sorts (std::sort is unstable) are
// broken by ascending index -- the rule the oracle documents.
struct BokehCdf {
    int x = 0, y = 0;
    std::vector<float> cdfRow, cdfColumn;
    std::vector<int32_t> rowIndices, columnIndices;
    bool valid() const { return x > 0 && y > 0; }
    // pixels: row-major, nchannels interleaved floats (what AiTextureLoad(path, true, 0, buf) fills, zoic.cpp:101-103)
    bool build(const float *pixels, int width, int height, int nchannels);
    void clear();
};

// bokeh_cdf.hip: the same tables built on the GPU (row-parallel sequential sums + LDS bitonic sorts), bit-identical.
// 0 = ok, -1 = image not covered (dimension > 4096 or < 2: use BokehCdf::build), > 0 = hipError_t
int build_bokeh_cdf_device(const float *pixels, int width, int height, int nchannels, BokehCdf &out);

// bokeh_cdf.hip: the cell records of tables.hpp built from the device-resident reference tables (0 = ok, else hipError_t)
int build_bokeh_cells_device(const float *dCdfRow, const int32_t *dRowIdx, const float *dCdfCol, const int32_t *dColIdx, int x, int y,
                             int gRow, int gCol, uint32_t *dCells);

// minimal .pfm reader for bokehPath (the reference loads through Arnold's texture system, absent here)
bool read_pfm(const std::string &path, std::vector<float> &pixels, int &w, int &h, int &nc);

}  // namespace zoic
