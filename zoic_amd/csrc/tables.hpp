// tables.hpp -- the wave-uniform constant tables the ray kernels read.
//
// One KolbTable (< 2 KB) is passed BY VALUE as a kernel argument: every field is wave-uniform, so the
// compiler fetches it with s_load into SGPRs through the scalar cache -- no VGPRs, no LDS traffic, and the
// per-surface loop index is uniform, so `surf[i]` stays a scalar load.  (LDS is used only where lanes index
// divergently: the bokeh row CDF.)
//
// Reference structures flattened here: LensElement / Lensdata (zoic.cpp:522-541), the
// std::map<float,boundingBox2d> exit-pupil LUT (zoic.cpp:540, 1391-1452), cameraData's thin-lens
// scalars (zoic.cpp:627-630) and the parameters camera_create_ray reads (zoic.cpp:1752-1990).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define ZOIC_HD __host__ __device__ inline
#else
#define ZOIC_HD inline
#endif

namespace zoic {

constexpr int kMaxSurfaces = 32;
constexpr int kLutEntries = 32;
constexpr int kMaxTries = 25;  // const int maxtries = 25, zoic.cpp:1767

// One spherical interface, rear -> front order (index 0 = rear-most, zoic.cpp:913).
struct Surface {
    float center;     // sphere centre z, computeLensCenters zoic.cpp:963-969
    float radius;     // signed radius of curvature (cm, after mm->cm and focal-length rescale)
    float radius2;    // radius*radius in f32, as raySphereIntersection computes it (zoic.cpp:978)
    float sign;       // radius < 0 ? -1 : 1 (zoic.cpp:986, 1000)
    float eta;        // ior2 == 1.0 ? ior1 : ior1/ior2 (zoic.cpp:1013), ior2 = next surface's ior, 1.0 after the last
    float housing2;   // largest f32 <= ((double)aperture*0.5)^2: `h2 > housing2` in f32 == the f64 compare of zoic.cpp:1114;
                      // for the stop: min(that, userApertureRadius^2) -- both clips of zoic.cpp:1114-1115 in one compare
    float invRadius;  // 1/radius (fast mode: unit normal = (c - hit) * invRadius)
    uint32_t tirPossible;  // ior1 > ior2 (zoic.cpp:1019)
};

// The same interface pre-digested for the FAST kernel (fast_optics.hpp): with a unit direction the Snell step needs
// no dot product -- cos(incidence) = thc/|R| -- so everything that multiplies thc or thc^2 is folded on the host.
// EXPERIMENT, off in the product build.  Near-planar interfaces (noise estimate eps*|R|/sqrt(housing2) above kGuardMinRelBand,
// lens_system.hpp: the stop, traced as a sphere of |R| ~ 1e4 cm, zoic.cpp:933): how FAST takes the hit there (fast_optics.hpp fast_hit)
//   0  like everywhere else (the product)
//   1  the root in its conjugate form: accurate -- and MORE disagreements with the reference
//   2  the reference's own operations in the reference's order: fewer disagreements, none smaller: the band cannot shrink
// (profiles/ab_r05/ab_stop.log)
#ifndef ZOIC_FAST_STABLE_STOP
#define ZOIC_FAST_STABLE_STOP 0
#endif
struct FastSurface {
    float center;        // sphere centre z
    float radius2;       // R^2
    float sign;          // sgn(R)  (ZOIC_FAST_STABLE_STOP builds: 2R on a near-planar interface)
    float housing2;      // clip limit on x^2+y^2 (same f32 as Surface::housing2)
    float eta;           // n1/n2
    float qOffset;       // (1 - eta^2) R^2 / eta^2 : q = thc^2 + qOffset, 1 - cs2 = (eta/R)^2 q, TIR <=> q < 0
    float krScale;       // eta / (|R| R)           : u' = eta u + krScale (thc - sqrt(q)) (c - hit)
    // Guard band of the decision-safe FAST mode: a FAST clip decision is trusted only when h^2 lies outside
    // (housingLo, housingHi]; otherwise the ray is handed to the STRICT kernel.  housing2 -/+ the band, rounded outwards, on
    // guarded (ill-conditioned) interfaces -- in practice the stop (zoic.cpp:1111-1117) -- and == housing2 everywhere else:
    // the guarded clip is two compares, clipped-or-unsure = h^2 > housingLo, unsure = that and !(h^2 > housingHi)
    float housingLo, housingHi;
    float pad0, pad1, pad2;
};

struct KolbTable {
    int32_t lensCount;
    int32_t apertureElement;
    float userAperture2;  // userApertureRadius^2 in f32 (zoic.cpp:1115)
    float originShift;    // sensor z (zoic.cpp:1855)
    float dirZ;           // -lenses[0].thickness (zoic.cpp:1924)
    float rearAperture;   // lenses[0].aperture: LUT-off sampling scale (zoic.cpp:1874-1875)
    float halfSensor;     // sensorWidth*0.5, exact in f32 (zoic.cpp:1853-1854)
    int32_t useLUT;       // kolbSamplingLUT
    int32_t useImage;     // image-based bokeh lens sampling
    int32_t bokehW, bokehH;
    int32_t lutSize;
    float exposureMul;    // 1+e^2, 1/(1+e^2) or 1 (zoic.cpp:1981-1987)
    int32_t exposureOn;
    uint32_t seed;
    float bandLutBin;     // decision-safe FAST: |dist*8 - round(dist*8)| below this leaves the LUT bin to STRICT
    // "Retry-dead" shortcut (kolb_refill.hip): the reference translates a RETRY's lens sample by the LUT centroid in BOTH
    // components (zoic.cpp:1933; the first try in x only, :1914), which moves the retry disk off the rear element for every
    // pixel far enough from the axis -- all 26 retries then die at interface 0.  These constants bound, per ray, the region
    // of the lens-sample plane from which a ray can reach the rear element's cap (host: lens_system.cpp fill_table).
    int32_t retryOn;      // 0: geometry outside the shortcut's assumptions -> never taken
    float retryK1;        // centre of that region = o.xy * retryK1
    float retryRho0;      // its radius = retryRho0 + |o.xy| * retrySpread
    float retrySpread;
    float retryMaxD;      // the bounds above hold for hits on the VERTEX-side cap of the rear sphere; the root the reference takes
                          // (zoic.cpp:986: ONE signed root, t < 0 never rejected) can only land on the opposite |xy| <= a cap
                          // when |d.xy| / dirZ > sqrt(R^2 - a^2) / a: rays with |d.xy| above retryMaxD are never classified
    int32_t twoLevel;     // the retry search rejects draws per ray before evaluating them (kolb_pool_body.hpp kTwoLevelDraws): set by the host for cameras
                          // where enough of the sensor's retries are rejectable that way (the TESSAR at 10 cm: 58 % of the draws; the wide-open
                          // PETZVAL: 3 % -- there the bound's instructions cost more than they save)
    float retryLensK;     // bound of |lens sample| x the parabola rotation's 1.0011: 1.0023 for the disk mapping; for a bokeh image
                          // what zoic.cpp:441,466 can return -- the centring swaps width and height, so an image that is not
                          // square samples far outside the unit square (2 x 7 pixels: x in [-3, -2])
    Surface surf[kMaxSurfaces];
    FastSurface fsurf[kMaxSurfaces];
    float lutMaxScale[kLutEntries];  // boundingBox2d::getMaxScale per LUT entry (zoic.cpp:503-517)
    float lutCentroidX[kLutEntries]; // boundingBox2d::getCentroid().x          (zoic.cpp:495-498)
};

struct ThinTable {
    float tanFov;          // zoic.cpp:1607
    float apertureRadius;  // zoic.cpp:1608
    float focalDistance;
    float ovDistance, ovRadius;  // opticalVignettingDistance / Radius
    int32_t useDof, useImage, bokehW, bokehH;
    float exposureMul;
    int32_t exposureOn;
    uint32_t seed;
};

// device pointers of the bokeh CDF tables (bokehProbability, zoic.cpp:222-417).
// Besides the reference's four arrays the device keeps a 16-ary search pyramid over each CDF: level 0 is the CDF
// itself, level j+1 holds the LAST element of every 16-element chunk of level j, every level padded with +inf to a
// multiple of 16 floats so that one chunk is one aligned 64-byte line (4 x global_load_dwordx4, one round trip).
// std::upper_bound over 256 entries then costs 2 dependent loads instead of 8 (device_search.hpp).
constexpr int kBokehMaxLevels = 3;  // 16^3 = 4096 entries per CDF; larger images use the plain binary search
struct BokehTables {
    const float *cdfRow;          // y            (unpadded, reference layout)
    const int32_t *rowIndices;    // y
    const float *cdfColumn;       // x*y          (unpadded, reference layout)
    const int32_t *columnIndices; // x*y
    const float *rowLevel[kBokehMaxLevels];  // padded pyramid of cdfRow
    const float *colLevel[kBokehMaxLevels];  // padded pyramid of cdfColumn, one padded row per image row
    int32_t colStride[kBokehMaxLevels];      // floats per image row at each level (multiple of 16)
    int32_t rowCount[kBokehMaxLevels];       // valid entries per level of the row pyramid
    int32_t colCount[kBokehMaxLevels];       // valid entries per level of one column pyramid row
    int32_t levels;                          // 0: pyramid not built (CDF longer than 4096) -> binary search
    // Cell records (y <= 2048, x <= 4096; built when both CDFs are non-decreasing): the unit interval of the sample is
    // cut into G = 2^k >= n cells; for cell g = floor(u*G) (exact: G is a power of two) the record holds everything
    // std::upper_bound needs when at most two CDF entries fall inside the cell:
    //   .x = a as bits, .y = b as bits: the first two DISTINCT CDF values above g/G (+inf past the end); lo = #{cdf <= g/G},
    //        j = first entry > a, k = first entry > b  (a run of equal values -- the zero-luminance tail -- is one decision)
    //   .z = idx[lo] | idx[j] << 16                                              idx = pixel index (clamped to n-1)
    //   .w = idx[k] | exceptional << 31                                          exceptional: a third distinct value < (g+1)/G
    // so indices[min(upper_bound(u), n-1)] = u < a ? idx[lo] : u < b ? idx[j] : idx[k], out of the same 16 bytes.
    // Exceptional cells finish with std::upper_bound over [lo, hi) of the reference arrays; their bounds
    // (lo | hi << 16, hi = #{cdf < (g+1)/G}) sit in side tables only that path reads.
    // The row record table (rowCells) is copied to LDS once per workgroup: a lens sample is ONE ds_read_b128 plus ONE
    // global_load_dwordx4 (colCells[row*colCellCount + g]) instead of 15 dependent LDS reads + 5 global loads.
    const uint32_t *rowCells;    // rowCellCount records of 4 dwords
    const uint32_t *colCells;    // y * colCellCount records of 4 dwords
    const uint32_t *rowBounds;   // rowCellCount
    const uint32_t *colBounds;   // y * colCellCount
    int32_t ldsWords;            // 4 * rowCellCount; 0: not available
    int32_t rowCellCount;
    int32_t colCellCount;
    int32_t pad0;
};

}  // namespace zoic
