// lens_system.cpp -- host precompute for the Kolb (RAYTRACED) lens model.  See lens_system.hpp.
//
// Everything here runs once per parameter change (node_update); it is arithmetic-exact with respect to the
// reference because the tables it emits steer every accept/reject decision of the hot path.  Strict IEEE:
// compiled with -ffp-contract=off, f64 intermediates where zoic.cpp has them.
#include "lens_system.hpp"
#include "optics.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#pragma STDC FP_CONTRACT OFF

namespace zoic {

// ------------------------------------------------------------------------------------------------ parsing
namespace {

bool is_delim(char c) { return c == '\t' || c == ',' || c == ';' || c == ':' || c == ' '; }  // zoic.cpp:728

// Calls fn(begin, end) for each line the reference's getline loop does not skip (empty / '#', zoic.cpp:724).
template <class F>
void for_each_data_line(const char *text, size_t len, F fn)
{
    size_t pos = 0;
    while (pos < len) {
        size_t eol = pos;
        while (eol < len && text[eol] != '\n') ++eol;
        if (eol > pos && text[pos] != '#') fn(text + pos, text + eol);
        pos = eol + 1;
    }
}

// std::stof semantics: longest valid prefix, failure if none
bool to_float(const char *b, const char *e, float &out)
{
    char buf[96];
    size_t n = std::min(static_cast<size_t>(e - b), sizeof(buf) - 1);
    std::memcpy(buf, b, n);
    buf[n] = 0;
    char *end = nullptr;
    out = std::strtof(buf, &end);
    return end != buf;
}

}  // namespace

LensError LensSystem::parse(const char *text, size_t len)
{
    rows.clear();
    // pass 1 (zoic.cpp:723-741): column count = floor(non-empty tokens / data lines)
    int tokens = 0, lines = 0;
    for_each_data_line(text, len, [&](const char *b, const char *e) {
        bool in_token = false;
        for (const char *p = b; p < e; ++p) {
            if (is_delim(*p)) in_token = false;
            else if (!in_token) { in_token = true; ++tokens; }
        }
        ++lines;
    });
    if (lines == 0) return LensError::Columns;
    const int columns = static_cast<int>(static_cast<float>(tokens) / static_cast<float>(lines));
    if (columns < 4 || columns > 5) return LensError::Columns;

    // pass 2 (zoic.cpp:762-812 / 835-891).  The reference keeps ONE field cursor across lines and advances it on
    // every delimiter, including empty tokens, so a doubled delimiter or a short row shifts the following fields;
    // the running `row` keeps stale values.  Reproduced as is: files that parse differently here would render
    // differently there.
    int cursor = 0;
    LensRow row;
    bool bad = false;
    auto store = [&](float v) {
        const int last = columns - 1;
        if (cursor == 0) row.radius = v;
        else if (cursor == 1) row.thickness = v;
        else if (cursor == 2) row.ior = v;
        else if (columns == 5 && cursor == 3) row.abbe = v;
        else if (cursor == last) { row.aperture = v; cursor = -1; }
    };
    for_each_data_line(text, len, [&](const char *b, const char *e) {
        const char *tok = b;
        for (const char *p = b; p < e; ++p) {
            if (!is_delim(*p)) continue;
            if (p > tok) {
                float v;
                if (!to_float(tok, p, v)) { bad = true; return; }
                store(v);
            }
            tok = p + 1;
            ++cursor;
        }
        if (tok < e) {
            float v;
            if (!to_float(tok, e, v)) { bad = true; return; }
            store(v);
            ++cursor;
        }
        rows.push_back(row);
    });
    if (bad) return LensError::Parse;
    if (rows.size() > static_cast<size_t>(kMaxSurfaces)) return LensError::TooManySurfaces;
    std::reverse(rows.begin(), rows.end());  // rear-most surface first, zoic.cpp:913
    return LensError::None;
}

// ------------------------------------------------------------------------------------------ paraxial helpers
namespace {

// raySphereIntersection with tracingRealRays=false (no miss test), zoic.cpp:973-995
V3 sphere_hit(V3 dir, V3 org, float centerZ, float radius, bool reverse)
{
    V3 u = normalize3(dir);
    V3 L{0.0f - org.x, 0.0f - org.y, centerZ - org.z};
    float tca = dot3(L, u);
    float r2 = radius * radius;
    float d2 = dot3(L, L) - (tca * tca);
    float thc = sqrtf(fabsf(r2 - d2));
    float sign = radius < 0.0f ? -1.0f : 1.0f;
    float t = reverse ? (tca - thc * sign) : (tca + thc * sign);
    return V3{org.x + u.x * t, org.y + u.y * t, org.z + u.z * t};
}

// intersectionNormal, zoic.cpp:999-1004
V3 sphere_normal(V3 hit, float centerZ, float radius)
{
    float sign = radius < 0.0f ? -1.0f : 1.0f;
    V3 n = normalize3(V3{0.0f - hit.x, 0.0f - hit.y, centerZ - hit.z});
    return V3{n.x * sign, n.y * sign, n.z * sign};
}

// calculateTransmissionVector, zoic.cpp:1008-1025; returns false on TIR (only when `real`)
bool refract(V3 &dir, float ior1, float ior2, V3 normal, bool real)
{
    V3 I = normalize3(dir), N = normalize3(normal);
    float eta = (ior2 == 1.0f) ? ior1 : ior1 / ior2;
    float c1 = -dot3(I, N);
    float cs2 = static_cast<float>(static_cast<double>(eta * eta) * (1.0 - static_cast<double>(c1 * c1)));
    if (real && ior1 > ior2 && cs2 > 1.0f) return false;
    float k = static_cast<float>(static_cast<double>(eta * c1) - std::sqrt(std::fabs(1.0 - static_cast<double>(cs2))));
    dir = V3{I.x * eta + N.x * k, I.y * eta + N.y * k, I.z * eta + N.z * k};
    return true;
}

// linePlaneIntersection with the fixed plane y = 0, zoic.cpp:1043-1049; returns the z of the hit
float axis_crossing_z(V3 org, V3 dir)
{
    V3 u = normalize3(dir);
    V3 coord = normalize3(V3{100.0f, 0.0f, 100.0f});
    const V3 n{0.0f, 1.0f, 0.0f};
    float num = dot3(coord, n) - dot3(n, org);
    float inv = 1.0f / dot3(n, u);  // AtVector / float multiplies by the reciprocal
    return org.z + (u.z * num) * inv;
}

// lineLineIntersection(...).x, zoic.cpp:1029-1039
float line_line_x(V3 l1o, V3 l1d, V3 l2o, V3 l2d)
{
    float A1 = l1d.y - l1o.y, B1 = l1o.z - l1d.z, C1 = A1 * l1o.z + B1 * l1o.y;
    float A2 = l2d.y - l2o.y, B2 = l2o.z - l2d.z, C2 = A2 * l2o.z + B2 * l2o.y;
    float delta = A1 * B2 - A2 * B1;
    return (B2 * C1 - B1 * C2) / delta;
}

}  // namespace

float LensSystem::trace_focal_length()
{
    const int n = static_cast<int>(rows.size());
    float focalPoint = 0.0f, principalPlane = 0.0f, summed = 0.0f;
    const float height = static_cast<float>(static_cast<double>(rows[0].aperture) * 0.1);  // zoic.cpp:1163
    V3 org{0.0f, height, 0.0f}, dir{0.0f, 0.0f, 99999.0f}, hit{0, 0, 0};
    for (int i = 0; i < n; ++i) {
        summed = (i == 0) ? rows[0].thickness : summed + rows[i].thickness;
        const float cz = summed - rows[i].radius;
        hit = sphere_hit(dir, org, cz, rows[i].radius, false);
        V3 nrm = sphere_normal(hit, cz, rows[i].radius);
        const float iorNext = (i != n - 1) ? rows[i + 1].ior : 1.0f;
        if (!refract(dir, rows[i].ior, iorNext, nrm, true)) ++precomputeTIR;
        if (i == n - 1) {
            // note: `org` is still the PREVIOUS surface's hit point here (zoic.cpp:1186-1204 run before :1214)
            V3 p1s{0.0f, height, 0.0f}, p1e{0.0f, height, 999999.0f};
            V3 p2e{0.0f, static_cast<float>(static_cast<double>(org.y) + static_cast<double>(dir.y) * 100000.0),
                   static_cast<float>(static_cast<double>(org.z) + static_cast<double>(dir.z) * 100000.0)};
            principalPlane = line_line_x(p1s, p1e, org, p2e);
            focalPoint = axis_crossing_z(org, dir);
        }
        org = hit;
    }
    return focalPoint - principalPlane;
}

float LensSystem::image_distance(float objectDistance)
{
    const int n = static_cast<int>(rows.size());
    V3 org{0.0f, 0.0f, objectDistance};
    V3 dir{0.0f, (rows[n - 1].aperture / 2.0f) * 0.05f, -objectDistance};
    float summed = 0.0f, result = 0.0f;
    for (int k = 0; k < n; ++k) summed += rows[k].thickness;
    for (int i = 0; i < n; ++i) {  // front -> rear
        const int j = n - 1 - i;
        if (i != 0) summed -= rows[n - i].thickness;
        const float cz = summed - rows[j].radius;
        V3 hit = sphere_hit(dir, org, cz, rows[j].radius, true);
        V3 nrm = sphere_normal(hit, cz, -rows[j].radius);
        const float iorFrom = (i == 0) ? 1.0f : rows[n - i].ior;
        if (!refract(dir, iorFrom, rows[j].ior, nrm, false)) ++precomputeTIR;
        if (i == n - 1) result = axis_crossing_z(hit, dir);
        org = hit;
    }
    return result;
}

void LensSystem::fill_surfaces(KolbTable &t) const
{
    const int n = static_cast<int>(rows.size());
    t.lensCount = n;
    t.apertureElement = apertureElement;
    t.userAperture2 = userApertureRadius * userApertureRadius;
    t.originShift = originShift;
    t.dirZ = -rows[0].thickness;
    t.rearAperture = rows[0].aperture;
    // Guard bands of the decision-safe FAST mode.  FAST and STRICT evaluate the same formulas with different roundings
    // (FMA, rsq/sqrt approximations, folded constants), so they can only disagree where the reference's own f32 rounding
    // noise decides.  That noise is large at ONE kind of interface: the stop, which the reference traces as a sphere of
    // |R| ~ 10^4 cm (radius 0 -> 99999 mm, zoic.cpp:933).  There t = tca - thc cancels two numbers of magnitude |R|, the hit
    // point is only good to ~ulp(|R|) ~ 1e-3 cm, and the clip h^2 > r_stop^2 is decided by rounding for rays within
    // eps*|R|/r_stop (1e-4 ... 5e-3, relative) of the edge.  Measured (tools/flip_analysis.py, 8.4 M rays per config): every
    // FAST/STRICT disagreement but ~1e-7 of the rays is decided at the stop, with relative margins up to 0.9 x eps*|R|/r_stop.
    // Every interface carries a band of kGuardScale times its estimate (kGuardFloorRel at least): the well-conditioned ones sit at ~1e-6
    // and cost a few more listed rays for half of the residual flips (16 M rays per config: C2 3 -> 1, C3 7 -> 4, C5 3 -> 1;
    // the two compares are in the kernel anyway).  Round 2 guarded only estimates above kGuardMinRelBand (-DZOIC_GUARD_ALL=0);
    // -DZOIC_GUARD_SCALE=x for experiments.
    const float guardScale = kGuardScale;
    const float eps = 5.9604645e-8f;
    t.bandLutBin = 16.0f * eps * 32.0f;   // dist*8 <= 31: a few ulps of the bin coordinate
    for (int i = 0; i < n; ++i) {
        Surface &s = t.surf[i];
        const LensRow &r = rows[i];
        s.center = r.center;
        s.radius = r.radius;
        s.radius2 = r.radius * r.radius;
        s.sign = r.radius < 0.0f ? -1.0f : 1.0f;
        const float iorNext = (i != n - 1) ? rows[i + 1].ior : 1.0f;
        s.eta = (iorNext == 1.0f) ? r.ior : r.ior / iorNext;
        s.tirPossible = r.ior > iorNext ? 1u : 0u;
        // (double)h2 > half*half  <=>  h2 > largest f32 <= half*half   (h2 is an f32; half*half is exact in f64)
        const double half = static_cast<double>(r.aperture) * 0.5;
        const double lim = half * half;
        float f = static_cast<float>(lim);
        if (static_cast<double>(f) > lim) f = std::nextafterf(f, -INFINITY);
        // at the stop the reference also rejects h2 > userApertureRadius^2 (zoic.cpp:1115); two `>` tests against
        // constants are one `>` test against the smaller constant
        if (i == apertureElement && t.userAperture2 < f) f = t.userAperture2;
        s.housing2 = f;
        s.invRadius = 1.0f / r.radius;
        FastSurface &q = t.fsurf[i];
        q = FastSurface{};
        q.center = s.center; q.radius2 = s.radius2; q.sign = s.sign; q.housing2 = s.housing2;
        q.eta = s.eta;
        const double eta = s.eta, R = static_cast<double>(r.radius);
        q.qOffset = static_cast<float>((1.0 - eta * eta) * R * R / (eta * eta));
        q.krScale = static_cast<float>(eta / (std::fabs(R) * R));
        const float relBand = eps * std::fabs(r.radius) / std::sqrt(s.housing2);   // relative to housing2
        const bool flat = relBand > kGuardMinRelBand;                                // near-planar: in practice the stop
#if ZOIC_FAST_STABLE_STOP
        if (flat) q.sign = 2.0f * r.radius;   // fast_hit takes this interface's root in its conjugate form (2R where the others carry sgn(R))
#endif
        const float scaleHere = flat ? kGuardScaleFlat : guardScale;
#if ZOIC_GUARD_ALL
        const float relAll = scaleHere * relBand > kGuardFloorRel ? scaleHere * relBand : kGuardFloorRel;
        const float band = (guardScale > 0.0f) ? relAll * s.housing2 : 0.0f;
#else
        const float band = flat ? scaleHere * relBand * s.housing2 : 0.0f;
#endif
        q.housingLo = q.housingHi = s.housing2;
        if (band > 0.0f) {   // outward rounding: (housingLo, housingHi] contains every h2 with |h2 - housing2| < band
            q.housingLo = std::nextafterf(s.housing2 - band, -INFINITY);
            q.housingHi = std::nextafterf(s.housing2 + band, INFINITY);
        }
    }
}

LensError LensSystem::prepare(float focalLength, float fStop, float focalDistance, bool useLUT, Rng &rng, LutTraceFn trace,
                              void *traceUser, LutBuildFn whole)
{
    const int n = static_cast<int>(rows.size());
    precomputeTIR = 0;
    hasLUT = false;
    // cleanupLensData, zoic.cpp:917-959
    apertureElement = -1;
    int stops = 0;
    for (int i = 0; i < n; ++i) {
        if (rows[i].radius == 0.0f) {
            apertureElement = i;
            if (++stops > 1) return LensError::MultiAperture;
            rows[i].radius = 99999.0f;  // the stop is traced as a very flat sphere
        }
        if (rows[i].ior == 0.0f) rows[i].ior = 1.0f;
    }
    if (apertureElement < 0) return LensError::NoAperture;
    for (LensRow &r : rows) {  // mm -> cm: f64 multiply by 0.1, narrowed (zoic.cpp:946-950)
        r.radius = static_cast<float>(static_cast<double>(r.radius) * 0.1);
        r.thickness = static_cast<float>(static_cast<double>(r.thickness) * 0.1);
        r.aperture = static_cast<float>(static_cast<double>(r.aperture) * 0.1);
    }
    float total = 0.0f;
    for (const LensRow &r : rows) total += r.thickness;
    rows[0].thickness -= total;  // front vertex at z = 0

    // focal length -> rescale -> focal length again, zoic.cpp:1651-1661
    tracedFocalLength[0] = trace_focal_length();
    focalLengthRatio = focalLength / tracedFocalLength[0];
    for (LensRow &r : rows) {
        r.radius *= focalLengthRatio;
        r.thickness *= focalLengthRatio;
        r.aperture *= focalLengthRatio;
    }
    tracedFocalLength[1] = trace_focal_length();
    userApertureRadius = static_cast<float>(static_cast<double>(tracedFocalLength[1]) / (2.0 * static_cast<double>(fStop)));
    if (userApertureRadius > rows[apertureElement].aperture) userApertureRadius = rows[apertureElement].aperture;  // :1668-1672

    originShift = image_distance(focalDistance);  // zoic.cpp:1675
    apertureDistance = 0.0f;                      // zoic.cpp:1678-1685
    for (int i = 0; i < n; ++i) {
        apertureDistance += rows[i].thickness;
        if (i == apertureElement) break;
    }
    float summed = 0.0f;  // computeLensCenters, zoic.cpp:963-969
    for (int i = 0; i < n; ++i) {
        summed = (i == 0) ? rows[0].thickness : summed + rows[i].thickness;
        rows[i].center = summed - rows[i].radius;
    }
    if (useLUT) build_lut(rng, trace ? trace : lut_trace_host, traceUser, whole);
    return LensError::None;
}

void lut_trace_host(const KolbTable &table, float originX, const float *lensU, const float *lensV, size_t n, uint8_t *accepted,
                    uint32_t *tirCount, void *)
{
    const float ap0 = table.rearAperture;
    uint32_t tir = 0;
    for (size_t b = 0; b < n; ++b) {
        V3 o{originX, 0.0f, table.originShift};
        V3 d{(lensU[b] * ap0) - originX, (lensV[b] * ap0) - 0.0f, table.dirZ};
        accepted[b] = trace_lens_strict(table, o, d, tir) ? 1 : 0;
    }
    *tirCount += tir;
}

// exitPupilLUT(&ld, 32, 100000), zoic.cpp:1391-1452.  Every probe consumes exactly two draws whatever its fate
// (zoic.cpp:1411-1412), so the sample set is a pure function of the stream position: draw all of them, trace
// them as one batch (host, or the GPU kernel), then replay the order-dependent bounding-box update.
void LensSystem::build_lut(Rng &rng, LutTraceFn trace, void *user, LutBuildFn whole)
{
    constexpr int kFilmSamples = kLutEntries;
    constexpr int kBoundsSamples = 100000;
    KolbTable t{};
    fill_surfaces(t);
    const float spacing = 4.0f / static_cast<float>(kFilmSamples);
    if (whole) {   // draws, traces and boxes on the GPU (lut_build.hip); anything but 0: the sequential path below
        uint32_t tir = 0;
        if (whole(t, rng, lutBox, &tir, user) == 0) {
            for (int i = 0; i < kFilmSamples; ++i) lutKey[i] = static_cast<float>(spacing * static_cast<float>(i));
            precomputeTIR += tir;
            hasLUT = true;
            return;
        }
    }
    const float ap0 = rows[0].aperture;
    std::vector<float> U(kBoundsSamples), V(kBoundsSamples);
    std::vector<uint8_t> ok(kBoundsSamples);
    for (int i = 0; i < kFilmSamples; ++i) {
        const float ox = static_cast<float>(spacing * static_cast<float>(i));
        for (int b = 0; b < kBoundsSamples; ++b) {
            U[b] = (rng_unit(xor128(rng)) * 2.0f) - 1.0f;
            V[b] = (rng_unit(xor128(rng)) * 2.0f) - 1.0f;
        }
        trace(t, ox, U.data(), V.data(), kBoundsSamples, ok.data(), &precomputeTIR, user);
        LutBox box;
        for (int b = 0; b < kBoundsSamples; ++b) {
            if (!ok[b]) continue;
            const float px = U[b] * ap0, py = V[b] * ap0;
            if ((box.minX + box.minY) == 0.0f) box = LutBox{px, py, px, py};  // zoic.cpp:1423-1428 (order dependent)
            if (px > box.maxX) box.maxX = px;
            if (py > box.maxY) box.maxY = py;
            if (px < box.minX) box.minX = px;
            if (py < box.minY) box.minY = py;
        }
        lutKey[i] = ox;
        lutBox[i] = box;
    }
    hasLUT = true;
}

// |lens sample| can be at most this (x 1.0011 for the parabola rotation that follows).  Disk mapping: |r| <= 1 times a parabola
// cos/sin pair of norm <= 1.0011.  Bokeh image (zoic.cpp:441,466,479-480): x = (col - (H-1)/2) / W * 2, y = -(row - (W-1)/2) / H * 2
// with col in [0, W-1], row in [0, H-1] and INTEGER halves -- width and height are swapped in the centring, so only a square
// image stays inside the unit square.
static float lens_sample_bound(int bokehW, int bokehH)
{
    if (bokehW <= 0 || bokehH <= 0) return 1.0011f * 1.0011f + 1.0e-4f;
    const int c0 = (bokehH - 1) / 2, r0 = (bokehW - 1) / 2;
    const double mx = 2.0 * std::max(std::abs(0 - c0), std::abs(bokehW - 1 - c0)) / bokehW;
    const double my = 2.0 * std::max(std::abs(0 - r0), std::abs(bokehH - 1 - r0)) / bokehH;
    return static_cast<float>(std::sqrt(mx * mx + my * my) * 1.0011 * 1.0001 + 1.0e-4);
}

void LensSystem::fill_table(KolbTable &t, float sensorWidth, int bokehW, int bokehH) const
{
    fill_surfaces(t);
    t.halfSensor = sensorWidth * 0.5f;  // exact; sx*(sensorWidth*0.5) in f64 rounds once, like the f32 product
    t.lutSize = hasLUT ? kLutEntries : 0;
    for (int i = 0; i < kLutEntries; ++i) {
        const LutBox &b = lutBox[i];
        const float cx = (b.minX + b.maxX) * 0.5f, cy = (b.minY + b.maxY) * 0.5f;  // getCentroid, zoic.cpp:495-498
        const float x1 = b.maxX - cx, y2 = b.maxY - cy;                            // getMaxScale, zoic.cpp:503-517
        const float sx = sqrtf(x1 * x1), sy = sqrtf(y2 * y2);
        t.lutMaxScale[i] = (sx >= sy) ? sx : sy;
        t.lutCentroidX[i] = cx;
    }
    // Retry-dead shortcut.  A try starts at o = (ox, oy, originShift) with d = (lens - o.xy, dirZ); it survives interface 0
    // only if it meets the rear sphere at a point H with |H.xy| <= a (housing radius) -- a point of the cap between the
    // vertex plane z_v and the rim plane z_rim.  With H = o + lambda * d, lambda = (H.z - o.z) / dirZ lies in
    // [lambda_lo, lambda_hi], and lens = o.xy * (1 - 1/lambda) + H.xy / lambda: every lens point that can pass lies within
    // a/lambda_lo + |o.xy| * (1/lambda_lo - 1/lambda_hi)/2 of o.xy * (1 - mean(1/lambda)).  The kernel compares that disk
    // with the disk the retries sample (centre: the doubly translated LUT centroid, radius: maxScale) with a 1 % margin.
    // H must be THAT cap: raySphereIntersection takes one signed root (zoic.cpp:986) and never rejects t < 0, and the sphere
    // has a second cap with |xy| <= a on its far side.  At a point of that cap the outward normal n has |n.xy| <= a/|R| and
    // |n.z| >= sqrt(R^2 - a^2)/|R|; the root the reference takes is the ray's entry (R < 0) or exit (R > 0) point, which needs
    // n.d < 0 resp. > 0 against the sign of n.z -- possible only for |d.xy| / dirZ > sqrt(R^2 - a^2) / a.  retryMaxD is that
    // bound on |d.xy| (1 % margin); the per-ray test leaves rays that could exceed it to their 26 draws.
    t.retryOn = 0; t.twoLevel = 0; t.retryK1 = t.retryRho0 = t.retrySpread = t.retryMaxD = 0.0f;
    t.retryLensK = lens_sample_bound(bokehW, bokehH);
    if (hasLUT && !rows.empty() && kRetryDeadMinShare < 1.0) {
        const double R = rows[0].radius, a = std::sqrt(static_cast<double>(t.surf[0].housing2)), dirZ = t.dirZ, oz = originShift;
        if (a < std::fabs(R) && dirZ > 0.0) {
            const double sag = std::fabs(R) - std::sqrt(R * R - a * a);
            const double zv = static_cast<double>(rows[0].center) + R, zrim = zv - (R < 0.0 ? -1.0 : 1.0) * sag;
            const double l1 = (zv - oz) / dirZ, l2 = (zrim - oz) / dirZ;
            const double lo = std::min(l1, l2), hi = std::max(l1, l2);
            if (lo > 1.0e-3 && std::isfinite(hi)) {
                t.retryOn = 1;
                t.retryK1 = static_cast<float>(1.0 - 0.5 * (1.0 / lo + 1.0 / hi));
                t.retryRho0 = static_cast<float>(a / lo);
                t.retrySpread = static_cast<float>(0.5 * (1.0 / lo - 1.0 / hi));
                t.retryMaxD = static_cast<float>(0.99 * dirZ * std::sqrt(R * R - a * a) / a);
            }
        }
    }
    // The shortcut has a price -- a byte map to clear, a finish kernel to launch, staging in the pass loop -- that a camera
    // with a handful of such pixels should not pay (the double Gauss at 50 mm: +1.5 % of the launch for 0.1 % of the rays).
    // Estimate the share of the sensor square the per-ray test (kolb_refill_body.hpp setup_ray) classifies and keep the
    // shortcut for cameras where that is a real part of the frame; without it the same rays simply draw their 26 samples.
    if (t.retryOn) {
        const int grid = 64;
        int hits = 0, live = 0, rejecting = 0;   // rejecting: positions whose retries the per-draw bound rejects at least a quarter of the time (rmin >= 0.5)
        for (int iy = 0; iy < grid; ++iy)
            for (int ix = 0; ix < grid; ++ix) {
                const float ox = (static_cast<float>(ix) + 0.5f) / grid * 2.0f - 1.0f, oy = (static_cast<float>(iy) + 0.5f) / grid * 2.0f - 1.0f;
                const float o0x = ox * t.halfSensor, o0y = oy * t.halfSensor, dist = std::sqrt(o0x * o0x + o0y * o0y);
                float maxScale = 0.0f, translation = 0.0f;
                if (!lut_lookup(t, dist, maxScale, translation) || (maxScale == 0.0f && translation == 0.0f)) continue;   // dead pixels have their own shortcut
                const float theta = std::atan2(o0y, o0x), sn = std::sin(theta), cs = std::cos(theta);
                const float ccx = translation * (cs - sn) - o0x * t.retryK1, ccy = translation * (sn + cs) - o0y * t.retryK1;
                const float reach = (t.retryRho0 + dist * t.retrySpread + std::fabs(maxScale) * t.retryLensK) * 1.01f + 1.0e-4f;
                const float dxyMax = std::fabs(maxScale) * t.retryLensK + std::fabs(translation) * 1.4158f + dist;
                if (ccx * ccx + ccy * ccy > reach * reach && dxyMax <= t.retryMaxD) ++hits;
                else {
                    ++live;
                    // the two-level retry search's bound for this position (kolb_pool_body.hpp setup_ray): draws with max(|2u-1|, |2v-1|) < rmin cannot
                    // reach the rear element; a position counts when that is a quarter of its draws or more
                    const float pass = (t.retryRho0 + dist * t.retrySpread) * 1.01f + 1.0e-4f;
                    const float rm = (std::sqrt(ccx * ccx + ccy * ccy) * 0.999f - pass) / (std::fabs(maxScale) * t.retryLensK * 1.001f + 1.0e-30f);
                    if (dxyMax <= t.retryMaxD && rm >= 0.5f) ++rejecting;
                }
            }
        if (hits < kRetryDeadMinShare * grid * grid) t.retryOn = 0;
        // Two-level retry search: only where it pays.  [MI355X] the TESSAR at 10 cm (C2: 58 % of the retries' draws rejectable) +3.9 %, the wide-open
        // PETZVAL (C5: 2.8 %) -2.5 % when forced on.  Disk sampler only (an image's lens samples are not bounded by the draw).
        t.twoLevel = (t.retryOn && bokehW <= 0 && live > 0 && rejecting >= 0.15 * live) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ bokeh CDF
void BokehCdf::clear()
{
    x = y = 0;
    cdfRow.clear(); cdfColumn.clear(); rowIndices.clear(); columnIndices.clear();
}

bool BokehCdf::build(const float *px, int width, int height, int nchannels)
{
    clear();
    // isValid(), zoic.cpp:135-137
    if (!px || width <= 0 || height <= 0 || nchannels < 3) return false;
    x = width; y = height;
    const int npixels = x * y;
    std::vector<float> lum(npixels), pdf(npixels), rowMass(y), cond(npixels);
    float total = 0.0f;
    for (int i = 0; i < npixels; ++i) {  // zoic.cpp:243-249, sequential f32 sum
        const float *p = px + static_cast<size_t>(i) * nchannels;
        lum[i] = p[0] * 0.3f + p[1] * 0.59f + p[2] * 0.11f;
        total += lum[i];
    }
    const float invTotal = 1.0f / total;
    for (int i = 0; i < npixels; ++i) pdf[i] = lum[i] * invTotal;
    for (int r = 0; r < y; ++r) {  // zoic.cpp:283-293
        float s = 0.0f;
        for (int c = 0; c < x; ++c) s += pdf[r * x + c];
        rowMass[r] = s;
    }
    auto sort_desc = [](int32_t *first, int32_t *last, const float *key) {
        std::stable_sort(first, last, [key](int32_t a, int32_t b) { return key[a] > key[b]; });
    };
    rowIndices.resize(y);
    std::iota(rowIndices.begin(), rowIndices.end(), 0);
    sort_desc(rowIndices.data(), rowIndices.data() + y, rowMass.data());  // zoic.cpp:317
    cdfRow.resize(y);
    float run = 0.0f;
    for (int r = 0; r < y; ++r) { run = run + rowMass[rowIndices[r]]; cdfRow[r] = run; }  // zoic.cpp:333-337
    for (int r = 0; r < y; ++r)
        for (int c = 0; c < x; ++c) {  // zoic.cpp:352-364
            const int i = r * x + c;
            cond[i] = (pdf[i] != 0 && rowMass[r] != 0) ? pdf[i] / rowMass[r] : 0.0f;
        }
    columnIndices.resize(npixels);
    std::iota(columnIndices.begin(), columnIndices.end(), 0);
    for (int r = 0; r < y; ++r) sort_desc(columnIndices.data() + r * x, columnIndices.data() + (r + 1) * x, cond.data());  // :380-382
    cdfColumn.resize(npixels);
    for (int r = 0; r < y; ++r) {  // zoic.cpp:398-407
        run = 0.0f;
        for (int c = 0; c < x; ++c) { const int i = r * x + c; run = run + cond[columnIndices[i]]; cdfColumn[i] = run; }
    }
    return true;
}

bool read_pfm(const std::string &path, std::vector<float> &pixels, int &w, int &h, int &nc)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    float scale = 0.0f;
    bool ok = std::fscanf(f, "%2s %d %d %f", magic, &w, &h, &scale) == 4 && w > 0 && h > 0;
    ok = ok && (std::strcmp(magic, "PF") == 0 || std::strcmp(magic, "Pf") == 0);
    if (ok) {
        std::fgetc(f);  // single whitespace after the scale
        nc = (magic[1] == 'F') ? 3 : 1;
        std::vector<float> raw(static_cast<size_t>(w) * h * nc);
        ok = std::fread(raw.data(), sizeof(float), raw.size(), f) == raw.size();
        if (ok) {
            if (scale > 0.0f) {  // big endian payload
                for (float &v : raw) {
                    unsigned char *b = reinterpret_cast<unsigned char *>(&v);
                    std::swap(b[0], b[3]); std::swap(b[1], b[2]);
                }
            }
            // PFM stores rows bottom-to-top; present top-to-bottom, 3 channels
            pixels.assign(static_cast<size_t>(w) * h * 3, 0.0f);
            for (int r = 0; r < h; ++r)
                for (int c = 0; c < w; ++c)
                    for (int k = 0; k < 3; ++k)
                        pixels[(static_cast<size_t>(r) * w + c) * 3 + k] = raw[(static_cast<size_t>(h - 1 - r) * w + c) * nc + (nc == 3 ? k : 0)];
            nc = 3;
        }
    }
    std::fclose(f);
    return ok;
}

}  // namespace zoic
