/*
 * oracle/zoic_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of zpelgrims/zoic's per-sample lens hot path
 * (camera_create_ray, src/zoic.cpp:1752-1990) and of the node_update
 * precompute (zoic.cpp:1575-1720) that builds the tables it reads.
 *
 * WHY A RESTATEMENT: the reference is one C++ translation unit that includes
 * <ai.h> (Arnold SDK 5.0.2.0, reference Makefile:4).  The SDK is closed source
 * and absent from this image, and a reference build against hand-written
 * stand-in headers is not allowed, so the reference is UNBUILDABLE here and no
 * oracle/_ref exists.
 *
 * PINNING STATUS: pinned against the one known-answer artefact the reference
 * holds, src/draw.zoic (a committed _DRAW dump; values re-entered under
 * tests/golden/draw_zoic_*.json):
 *   - header (lens centres/curvatures, IORs, aperture element/distance,
 *     user aperture radius, max aperture, image distance): the whole
 *     precompute chain for F_2.0_DOUBLE_GAUSS @ f=5.0 f/2.8 focus 23 cm,
 *     reproduced to all 10 printed decimals;
 *   - RAYS{}: 109 complete traced rays (per-surface hit points and exit
 *     direction of traceThroughLensElements) replayed by tests/test_oracle_kat.py.
 * The functions the dump does not exercise (concentricDiskSample, fastSin/Cos,
 * the exit-pupil LUT lookup, bokehSample, the retry loop, thin-lens) have no
 * reference-held vectors: for those this oracle is "parity unpinned" -- it is a
 * line-by-line restatement only.
 *
 * THIRD-PARTY ARITHMETIC (Arnold SDK 5.0.2.0 inlines, not under /root/reference):
 *   AiV3Dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z            (float, left to right)
 *   AiV3Length(a)     = sqrtf(AiV3Dot(a,a))
 *   AiV3Normalize(a)  = a * (1/len), len==0 -> a*0
 *   AtVector op float = componentwise float; AtVector / f = a * (1/f)
 *   AI_PI = 3.14159265358979323846f, AI_PIOVER2 = 1.57079632679489661923f, AI_P2_ZERO=(0,0)
 * Call sites: zoic.cpp:974,977,979,1002,1009-1010,1015,1046-1048,1777,1800,1810,1816.
 *
 * TWO ASSUMPTIONS NO REFERENCE-HELD VECTOR CAN PIN (stated here so that nobody mistakes them for facts):
 *  (i)  ARGUMENT EVALUATION ORDER.  The reference draws a retry's two random numbers inside ONE argument list --
 *       `concentricDiskSample(xor128() / 4294967296.0, xor128() / 4294967296.0, &lens)` (zoic.cpp:1806, 1881, 1930;
 *       likewise bokehSample at 1808, 1883, 1932).  C++ leaves the order of the two xor128() calls unsequenced.  This
 *       restatement takes LEFT TO RIGHT (first draw -> first parameter), which is what clang does and clang++ is the
 *       compiler the reference's Makefile names (reference Makefile:6).  g++ evaluates right to left and would swap u and v
 *       of every retry: a g++ build of the reference is a different (equally valid) sample sequence, not a bug here.
 *  (ii) THE ARNOLD INLINES listed above.  The SDK headers are not in the tree; the formulas are the published Arnold 5
 *       ones and the draw.zoic replay agrees with them to float rounding, but their exact operation order (e.g. whether
 *       AiV3Normalize multiplies by a reciprocal or divides) is pinned only to that level.
 * tests/test_oracle_properties.py holds property pins for the functions without reference vectors, so that an edit of this
 * file cannot drift silently.
 *
 * FP DISCIPLINE: strict IEEE binary32 with the reference's scattered binary64
 * intermediates (unsuffixed literals) mirrored one by one.
 * Build with -ffp-contract=off, no fast-math.  `atan2(float,float)` at
 * zoic.cpp:1899 is taken as the C library's double atan2 narrowed to float
 * (what unqualified lookup finds through <cmath> on glibc/libstdc++).
 */
#include "zoic_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define AI_PI      3.14159265358979323846f
#define AI_PIOVER2 1.57079632679489661923f

/* ------------------------------------------------------------------ vectors */
static inline zo_v3 V3(float x, float y, float z) { zo_v3 r = { x, y, z }; return r; }
static inline zo_v3 v3_sub(zo_v3 a, zo_v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline zo_v3 v3_add(zo_v3 a, zo_v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline zo_v3 v3_mulf(zo_v3 a, float f) { return V3(a.x * f, a.y * f, a.z * f); }
static inline zo_v3 v3_divf(zo_v3 a, float f) { float c = 1.0f / f; return V3(a.x * c, a.y * c, a.z * c); }
static inline float ai_v3_dot(zo_v3 a, zo_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline zo_v3 ai_v3_normalize(zo_v3 a)
{
    float tmp = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    if (tmp != 0.0f) tmp = 1.0f / tmp;
    return V3(a.x * tmp, a.y * tmp, a.z * tmp);
}

/* ------------------------------------------------------------------ structs */
typedef struct zo_image {          /* class imageData, zoic.cpp:115-486 */
    int x, y, nchannels;
    float *pixelData, *cdfRow, *cdfColumn;
    int *rowIndices, *columnIndices;
} zo_image;

typedef struct zo_lensdata {       /* struct Lensdata, zoic.cpp:528-541 */
    zo_lens_element lenses[ZO_MAX_LENSES];
    int lensCount;
    float userApertureRadius;
    int apertureElement;
    int apertureElementSet;        /* fence for the uninitialised read, see ZO_ERR_NO_APERTURE */
    int vignettedRays, succesRays, totalInternalReflection;
    long long surfaceVisits;       /* NOT in the reference: interfaces entered by traceThroughLensElements since the last lens
                                      rebuild -- the work measure behind SURVEY 8(d)'s FLOP/ray (bench.py's flop_frac) */
    float apertureDistance, focalLengthRatio, filmDiagonal, originShift, focalDistance;
    float tracedFocal[2];
    int lutSize;                   /* std::map<float,boundingBox2d> as sorted arrays */
    float lutKeys[ZO_LUT_SIZE];
    zo_bbox2 lutBoxes[ZO_LUT_SIZE];
} zo_lensdata;

typedef struct zo_params_owned {   /* struct cameraParams, zoic.cpp:544-612 */
    zo_params p;
    char *bokehPath, *lensDataPath;
} zo_params_owned;

struct zo_camera {                 /* struct cameraData, zoic.cpp:627-643 */
    float fov, tan_fov, apertureRadius;
    zo_image image;
    zo_params_owned params;
    zo_lensdata lens;
    zo_rng rng;                    /* the function-static state of xor128 */
    /* pending inputs for the next update */
    int pend_w, pend_h, pend_nc; float *pend_pixels;
    char *lens_text; size_t lens_text_len;
};

/* ---------------------------------------------------------------------- RNG */
void zo_rng_seed(zo_rng *r) { r->x = 123456789u; r->y = 362436069u; r->z = 521288629u; r->w = 88675123u; }

/* zoic.cpp:647-652 */
uint32_t zo_xor128(zo_rng *r)
{
    uint32_t t = r->x ^ (r->x << 11);
    r->x = r->y; r->y = r->z; r->z = r->w;
    return r->w = (r->w ^ (r->w >> 19) ^ t ^ (t >> 8));
}

/* zoic.cpp:655-657 */
static inline float linearInterpolate(float perc, float a, float b) { return a + perc * (b - a); }

/* zoic.cpp:661-668 */
float zo_fast_sin(float x)
{
    x = (float)(fmod((double)(x + AI_PI), (double)(AI_PI * 2)) - (double)AI_PI);
    const float B = 4.0f / AI_PI;
    const float C = -4.0f / (AI_PI * AI_PI);
    float y = B * x + C * x * fabsf(x);
    const float P = 0.225f;
    return P * (y * fabsf(y) - y) + y;
}

/* zoic.cpp:671-681 */
float zo_fast_cos(float x)
{
    x = (float)((double)x + (double)AI_PI * 0.5);                      /* :673, f64 add */
    x = (float)(fmod((double)(x + AI_PI), (double)(AI_PI * 2)) - (double)AI_PI);
    const float B = 4.0f / AI_PI;
    const float C = -4.0f / (AI_PI * AI_PI);
    float y = B * x + C * x * fabsf(x);
    const float P = 0.225f;
    return P * (y * fabsf(y) - y) + y;
}

/* zoic.cpp:686-704 */
void zo_concentric_disk_sample(float ox, float oy, zo_v2 *lens)
{
    float phi, r;
    float a = (float)(2.0 * (double)ox - 1.0);                         /* :690 */
    float b = (float)(2.0 * (double)oy - 1.0);                         /* :691 */
    if ((a * a) > (b * b)) {
        r = a;
        phi = (0.78539816339f) * (b / a);
    } else {
        r = b;
        phi = (AI_PIOVER2) - (0.78539816339f) * (a / b);
    }
    lens->x = r * zo_fast_cos(phi);
    lens->y = r * zo_fast_sin(phi);
}

/* ------------------------------------------------------------- bokeh image */
static void image_invalidate(zo_image *im)     /* zoic.cpp:139-166 */
{
    free(im->pixelData); free(im->cdfRow); free(im->cdfColumn);
    free(im->rowIndices); free(im->columnIndices);
    memset(im, 0, sizeof(*im));
}

static int image_valid(const zo_image *im)     /* zoic.cpp:135-137 */
{
    return (im->x * im->y * im->nchannels > 0 && im->nchannels >= 3);
}

/* std::sort(first,last, arrayCompare(values)) -- zoic.cpp:106-112, 317, 381.
 * std::sort is unstable: the order of equal keys is implementation defined.  The oracle
 * defines ties as "ascending original index" (a stable descending sort); fixtures are tie-free
 * where the permutation matters and the product must reproduce this rule. */
static const float *g_sort_values;
static int cmp_desc(const void *pa, const void *pb)
{
    int a = *(const int *)pa, b = *(const int *)pb;
    float va = g_sort_values[a], vb = g_sort_values[b];
    if (va > vb) return -1;
    if (va < vb) return 1;
    return (a > b) - (a < b);
}
static pthread_mutex_t g_sort_mutex = PTHREAD_MUTEX_INITIALIZER;
static void sort_desc(int *idx, int n, const float *values)
{
    pthread_mutex_lock(&g_sort_mutex);
    g_sort_values = values;
    qsort(idx, (size_t)n, sizeof(int), cmp_desc);
    pthread_mutex_unlock(&g_sort_mutex);
}

/* imageData::bokehProbability, zoic.cpp:222-417 */
static void image_bokeh_probability(zo_image *im)
{
    if (!image_valid(im)) return;
    const int x = im->x, y = im->y, nchannels = im->nchannels;
    int npixels = x * y;
    float *pixelValues = malloc(sizeof(float) * npixels);
    float *normalizedPixelValues = malloc(sizeof(float) * npixels);
    int o1 = (nchannels >= 2 ? 1 : 0);
    int o2 = (nchannels >= 3 ? 2 : o1);
    float totalValue = 0.0f;
    for (int i = 0, j = 0; i < npixels; ++i, j += nchannels) {          /* :243-249 */
        pixelValues[i] = im->pixelData[j] * 0.3f + im->pixelData[j + o1] * 0.59f + im->pixelData[j + o2] * 0.11f;
        totalValue += pixelValues[i];
    }
    float invTotalValue = 1.0f / totalValue;                            /* :259 */
    for (int i = 0; i < npixels; ++i)
        normalizedPixelValues[i] = pixelValues[i] * invTotalValue;      /* :263 */

    float *summedRowValues = malloc(sizeof(float) * y);
    for (int i = 0, k = 0; i < y; ++i) {                                /* :283-293 */
        summedRowValues[i] = 0.0f;
        for (int j = 0; j < x; ++j, ++k) summedRowValues[i] += normalizedPixelValues[k];
    }
    im->rowIndices = malloc(sizeof(int) * y);
    for (int i = 0; i < y; ++i) im->rowIndices[i] = i;
    sort_desc(im->rowIndices, y, summedRowValues);                      /* :317 */

    im->cdfRow = malloc(sizeof(float) * y);
    float prevVal = 0.0f;
    for (int i = 0; i < y; ++i) {                                       /* :333-337 */
        im->cdfRow[i] = prevVal + summedRowValues[im->rowIndices[i]];
        prevVal = im->cdfRow[i];
    }
    float *normalizedValuesPerRow = malloc(sizeof(float) * npixels);
    for (int r = 0, i = 0; r < y; ++r)                                  /* :352-364 */
        for (int c = 0; c < x; ++c, ++i) {
            if ((normalizedPixelValues[i] != 0) && (summedRowValues[r] != 0))
                normalizedValuesPerRow[i] = normalizedPixelValues[i] / summedRowValues[r];
            else
                normalizedValuesPerRow[i] = 0;
        }
    im->columnIndices = malloc(sizeof(int) * npixels);
    for (int i = 0; i < npixels; i++) im->columnIndices[i] = i;
    for (int i = 0; i < npixels; i += x)                                /* :380-382 */
        sort_desc(im->columnIndices + i, x, normalizedValuesPerRow);

    im->cdfColumn = malloc(sizeof(float) * npixels);
    for (int r = 0, i = 0; r < y; ++r) {                                /* :398-407 */
        prevVal = 0.0f;
        for (int c = 0; c < x; ++c, ++i) {
            im->cdfColumn[i] = prevVal + normalizedValuesPerRow[im->columnIndices[i]];
            prevVal = im->cdfColumn[i];
        }
    }
    free(pixelValues); free(normalizedPixelValues); free(summedRowValues); free(normalizedValuesPerRow);
}

/* std::upper_bound: first element > v */
static int upper_bound_f(const float *a, int n, float v)
{
    int lo = 0, len = n;
    while (len > 0) {
        int half = len >> 1;
        if (!(v < a[lo + half])) { lo += half + 1; len -= half + 1; }
        else len = half;
    }
    return lo;
}

/* imageData::bokehSample, zoic.cpp:420-485 */
static void image_bokeh_sample(const zo_image *im, float randomNumberRow, float randomNumberColumn, float *dx, float *dy)
{
    if (!image_valid(im)) { *dx = 0.0f; *dy = 0.0f; return; }          /* :421-426 */
    const int x = im->x, y = im->y;
    int ub = upper_bound_f(im->cdfRow, y, randomNumberRow);            /* :432 */
    int r = (ub >= y) ? y - 1 : ub;                                     /* :435 */
    int actualPixelRow = im->rowIndices[r];
    int recalulatedPixelRow = actualPixelRow - ((x - 1) / 2);          /* :441 (x, not y) */
    int startPixel = actualPixelRow * x;
    int ubc = upper_bound_f(im->cdfColumn + startPixel, x, randomNumberColumn); /* :458 */
    int c = (ubc >= x) ? startPixel + x - 1 : startPixel + ubc;        /* :461 */
    int actualPixelColumn = im->columnIndices[c];
    int relativePixelColumn = actualPixelColumn - startPixel;
    int recalulatedPixelColumn = relativePixelColumn - ((y - 1) / 2);  /* :466 (y, not x) */
    float flippedRow = (float)recalulatedPixelColumn;                   /* :479 */
    float flippedColumn = recalulatedPixelRow * -1.0f;                  /* :480 */
    *dx = (float)((double)(flippedRow / (float)x) * 2.0);               /* :483 */
    *dy = (float)((double)(flippedColumn / (float)y) * 2.0);            /* :484 */
}

/* imageData::read, zoic.cpp:168-219, with the Arnold texture calls replaced by the pending buffer */
static int image_read(zo_camera *cam)
{
    zo_image *im = &cam->image;
    image_invalidate(im);
    if (!cam->pend_pixels || cam->pend_w <= 0 || cam->pend_h <= 0 || cam->pend_nc <= 0) return 0;
    im->x = cam->pend_w; im->y = cam->pend_h; im->nchannels = cam->pend_nc;
    size_t n = (size_t)im->x * im->y * im->nchannels;
    im->pixelData = malloc(sizeof(float) * n);
    memcpy(im->pixelData, cam->pend_pixels, sizeof(float) * n);
    image_bokeh_probability(im);
    return 1;
}

/* --------------------------------------------------------------- lens file */
static size_t find_first_of(const char *line, size_t L, size_t prev)   /* delimiters zoic.cpp:728 */
{
    for (size_t i = prev; i < L; ++i) {
        char c = line[i];
        if (c == '\t' || c == ',' || c == ';' || c == ':' || c == ' ') return i;
    }
    return (size_t)-1;
}

static int stof_token(const char *s, size_t n, float *out)             /* std::stof */
{
    char buf[128];
    if (n >= sizeof(buf)) n = sizeof(buf) - 1;
    memcpy(buf, s, n); buf[n] = 0;
    char *end = NULL;
    float v = strtof(buf, &end);
    if (end == buf) return 0;
    *out = v;
    return 1;
}

static void assign_field(zo_lens_element *lens, int totalColumns, int *counter, float v)
{
    /* zoic.cpp:773-785 (4 columns) and 846-861 (5 columns) */
    int c = *counter;
    if (totalColumns == 4) {
        if (c == 0) lens->curvature = v; else if (c == 1) lens->thickness = v;
        else if (c == 2) lens->ior = v; else if (c == 3) { lens->aperture = v; *counter = -1; }
    } else {
        if (c == 0) lens->curvature = v; else if (c == 1) lens->thickness = v;
        else if (c == 2) lens->ior = v; else if (c == 3) lens->abbe = v;
        else if (c == 4) { lens->aperture = v; *counter = -1; }
    }
}

/* readTabularLensData, zoic.cpp:708-914 (on a text buffer; std::getline line splitting) */
static int read_tabular_lens_data(const char *text, size_t len, zo_lensdata *ld)
{
    int columns = 0, lines = 0;
    for (size_t ls = 0; ls < len; ) {                                   /* pass 1, :723-737 */
        size_t le = ls; while (le < len && text[le] != '\n') ++le;
        const char *line = text + ls; size_t L = le - ls;
        ls = le + 1;
        if (L == 0 || line[0] == '#') continue;
        size_t prev = 0, pos;
        while ((pos = find_first_of(line, L, prev)) != (size_t)-1) {
            if (pos > prev) ++columns;
            prev = pos + 1;
        }
        if (prev < L) ++columns;
        ++lines;
    }
    if (lines == 0) return ZO_ERR_LENS_COLUMNS;
    int totalColumns = (int)((float)columns / (float)lines);            /* :741 */
    if (totalColumns < 4 || totalColumns > 5) return ZO_ERR_LENS_COLUMNS; /* :745-754 */

    int lensDataCounter = 0, n = 0;
    zo_lens_element lens; memset(&lens, 0, sizeof(lens));
    for (size_t ls = 0; ls < len; ) {                                   /* pass 2, :762-812 / 835-891 */
        size_t le = ls; while (le < len && text[le] != '\n') ++le;
        const char *line = text + ls; size_t L = le - ls;
        ls = le + 1;
        if (L == 0 || line[0] == '#') continue;
        size_t prev = 0, pos; float v;
        while ((pos = find_first_of(line, L, prev)) != (size_t)-1) {
            if (pos > prev) {
                if (!stof_token(line + prev, pos - prev, &v)) return ZO_ERR_LENS_PARSE;
                assign_field(&lens, totalColumns, &lensDataCounter, v);
            }
            prev = pos + 1;
            ++lensDataCounter;
        }
        if (prev < L) {
            if (!stof_token(line + prev, L - prev, &v)) return ZO_ERR_LENS_PARSE;
            assign_field(&lens, totalColumns, &lensDataCounter, v);
            ++lensDataCounter;
        }
        if (n >= ZO_MAX_LENSES) return ZO_ERR_TOO_MANY_LENSES;
        ld->lenses[n++] = lens;                                         /* :810 / :889 */
    }
    ld->lensCount = n;
    for (int i = 0; i < n / 2; ++i) {                                   /* std::reverse :913 */
        zo_lens_element t = ld->lenses[i]; ld->lenses[i] = ld->lenses[n - 1 - i]; ld->lenses[n - 1 - i] = t;
    }
    return ZO_OK;
}

/* cleanupLensData, zoic.cpp:917-959 */
static int cleanup_lens_data(zo_lensdata *ld)
{
    int apertureCount = 0;
    for (int i = 0; i < ld->lensCount; i++) {
        if (ld->lenses[i].curvature == 0.0) {
            ld->apertureElement = i; ld->apertureElementSet = 1;
            ++apertureCount;
            if (apertureCount > 1) return ZO_ERR_MULTI_APERTURE;       /* :926-929 */
            ld->lenses[i].curvature = 99999.0;                          /* :933 */
        }
        if (ld->lenses[i].ior == 0.0) ld->lenses[i].ior = 1.0;         /* :937-940 */
    }
    for (int i = 0; i < ld->lensCount; i++) {                           /* :946-950, f64 multiply */
        ld->lenses[i].curvature = (float)((double)ld->lenses[i].curvature * 0.1);
        ld->lenses[i].thickness = (float)((double)ld->lenses[i].thickness * 0.1);
        ld->lenses[i].aperture  = (float)((double)ld->lenses[i].aperture * 0.1);
    }
    float summedThickness = 0.0;
    for (int i = 0; i < ld->lensCount; i++) summedThickness += ld->lenses[i].thickness;
    ld->lenses[0].thickness -= summedThickness;                         /* :958 */
    return ZO_OK;
}

/* computeLensCenters, zoic.cpp:963-969 */
static void compute_lens_centers(zo_lensdata *ld)
{
    float summedThickness = 0.0f;
    for (int i = 0; i < ld->lensCount; i++) {
        if (i == 0) summedThickness = ld->lenses[0].thickness; else summedThickness += ld->lenses[i].thickness;
        ld->lenses[i].center = summedThickness - ld->lenses[i].curvature;
    }
}

/* --------------------------------------------------------------- optics */
/* Diagnostic hook (tests / tools only; never alters a result): while g_margin_probe points at a record, the three
 * accept/reject decisions of traceThroughLensElements note how close each call was -- the smallest |relative margin|
 * seen, the interface it occurred at and its kind (0 housing/stop clip :1114-1115, 1 sphere miss :980, 2 total internal
 * reflection :1019).  Used to calibrate the guard bands of the product's decision-safe FAST mode. */
static __thread zo_margin_probe *g_margin_probe = NULL;
static __thread int g_probe_iface = 0;
static inline void probe_note(float value, float limit, int kind)
{
    zo_margin_probe *p = g_margin_probe;
    if (!p) return;
    float m = fabsf(value - limit) / (fabsf(limit) > 0.0f ? fabsf(limit) : 1.0f);
    if (m < p->min_rel_margin) { p->min_rel_margin = m; p->iface = g_probe_iface; p->kind = kind; }
}

/* raySphereIntersection, zoic.cpp:973-995 */
static inline int raySphereIntersection(zo_v3 *hit_point, zo_v3 ray_direction, zo_v3 ray_origin, zo_v3 sphere_center,
                                        float sphere_radius, int reverse, int tracingRealRays)
{
    ray_direction = ai_v3_normalize(ray_direction);
    zo_v3 L = v3_sub(sphere_center, ray_origin);
    float tca = ai_v3_dot(L, ray_direction);
    float radius2 = sphere_radius * sphere_radius;
    float d2 = ai_v3_dot(L, L) - (tca * tca);
    if (tracingRealRays) probe_note(d2, radius2, 1);
    if (tracingRealRays && (d2 > radius2)) return 0;
    float thc = sqrtf(fabsf(radius2 - d2));
    float sign = (sphere_radius < 0.0f ? -1.0f : 1.0f);
    if (reverse) *hit_point = v3_add(ray_origin, v3_mulf(ray_direction, (tca - thc * sign)));
    else         *hit_point = v3_add(ray_origin, v3_mulf(ray_direction, (tca + thc * sign)));
    return 1;
}

/* intersectionNormal, zoic.cpp:999-1004 */
static inline void intersectionNormal(zo_v3 hit_point, zo_v3 sphere_center, float sphere_radius, zo_v3 *hit_point_normal)
{
    float sign = (sphere_radius < 0.0f ? -1.0f : 1.0f);
    *hit_point_normal = v3_mulf(ai_v3_normalize(v3_sub(sphere_center, hit_point)), sign);
}

/* calculateTransmissionVector, zoic.cpp:1008-1025.  ior2 arrives as float (the literal 1.0 at the
 * call sites converts exactly). */
static inline int calculateTransmissionVector(zo_v3 *ray_direction, float ior1, float ior2, zo_v3 incidentVector,
                                              zo_v3 normalVector, int tracingRealRays)
{
    incidentVector = ai_v3_normalize(incidentVector);
    normalVector = ai_v3_normalize(normalVector);
    float eta;
    if (ior2 == 1.0) eta = ior1; else eta = ior1 / ior2;                /* :1013 */
    float c1 = -ai_v3_dot(incidentVector, normalVector);
    float cs2 = (float)((double)(eta * eta) * (1.0 - (double)(c1 * c1))); /* :1016 */
    if (tracingRealRays && (ior1 > ior2)) probe_note(cs2, 1.0f, 2);
    if ((tracingRealRays) && (ior1 > ior2) && ((double)cs2 > 1.0)) return 0; /* :1019 */
    float k = (float)((double)(eta * c1) - sqrt(fabs(1.0 - (double)cs2))); /* :1023, narrowed by operator*(float) */
    *ray_direction = v3_add(v3_mulf(incidentVector, eta), v3_mulf(normalVector, k));
    return 1;
}

/* lineLineIntersection, zoic.cpp:1029-1039 */
static zo_v2 lineLineIntersection(zo_v3 l1o, zo_v3 l1d, zo_v3 l2o, zo_v3 l2d)
{
    float A1 = l1d.y - l1o.y;
    float B1 = l1o.z - l1d.z;
    float C1 = A1 * l1o.z + B1 * l1o.y;
    float A2 = l2d.y - l2o.y;
    float B2 = l2o.z - l2d.z;
    float C2 = A2 * l2o.z + B2 * l2o.y;
    float delta = A1 * B2 - A2 * B1;
    zo_v2 rv = { (B2 * C1 - B1 * C2) / delta, (A1 * C2 - A2 * C1) / delta };
    return rv;
}

/* linePlaneIntersection, zoic.cpp:1043-1049 */
static zo_v3 linePlaneIntersection(zo_v3 rayOrigin, zo_v3 rayDirection)
{
    zo_v3 coord = V3(100.0, 0.0, 100.0);
    zo_v3 planeNormal = V3(0.0, 1.0, 0.0);
    rayDirection = ai_v3_normalize(rayDirection);
    coord = ai_v3_normalize(coord);
    float num = ai_v3_dot(coord, planeNormal) - ai_v3_dot(planeNormal, rayOrigin);
    return v3_add(rayOrigin, v3_divf(v3_mulf(rayDirection, num), ai_v3_dot(planeNormal, rayDirection)));
}

/* calculateImageDistance, zoic.cpp:1054-1095 */
static float calculateImageDistance(float objectDistance, zo_lensdata *ld)
{
    const int n = ld->lensCount;
    zo_v3 ray_origin = V3(0.0f, 0.0f, objectDistance);
    zo_v3 ray_direction = V3(0.0f, (ld->lenses[n - 1].aperture / 2.0f) * 0.05f, -objectDistance);
    float summedThickness = 0.0, imageDistance = 0.0;
    zo_v3 hit_point_normal = V3(0, 0, 0), hit_point = V3(0, 0, 0);
    for (int k = 0; k < n; k++) summedThickness += ld->lenses[k].thickness;
    for (int i = 0; i < n; i++) {
        if (i != 0) summedThickness -= ld->lenses[n - i].thickness;
        zo_v3 sphere_center = V3(0.0f, 0.0f, summedThickness - ld->lenses[n - 1 - i].curvature);
        raySphereIntersection(&hit_point, ray_direction, ray_origin, sphere_center, ld->lenses[n - 1 - i].curvature, 1, 0);
        intersectionNormal(hit_point, sphere_center, -ld->lenses[n - 1 - i].curvature, &hit_point_normal);
        if (i == 0) {
            if (!calculateTransmissionVector(&ray_direction, 1.0, ld->lenses[n - i - 1].ior, ray_direction, hit_point_normal, 0))
                ld->totalInternalReflection++;
        } else {
            if (!calculateTransmissionVector(&ray_direction, ld->lenses[n - i].ior, ld->lenses[n - i - 1].ior, ray_direction, hit_point_normal, 0))
                ld->totalInternalReflection++;
        }
        if (i == n - 1) imageDistance = linePlaneIntersection(hit_point, ray_direction).z;
        ray_origin = hit_point;
    }
    return imageDistance;
}

/* traceThroughLensElements, zoic.cpp:1099-1158 (== traceThroughLensElementsForApertureSize 1309-1350
 * arithmetically; the latter takes its rays by value).  rec/nrec: the _DRAW dump hook. */
static inline int traceThroughLensElements(zo_v3 *ray_origin, zo_v3 *ray_direction, zo_lensdata *ld, zo_v3 *rec, int *nrec)
{
    zo_v3 hit_point, hit_point_normal, sphere_center;
    const int n = ld->lensCount;
    for (int i = 0; i < n; i++) {
        g_probe_iface = i;
        ld->surfaceVisits++;
        sphere_center.x = 0.0f; sphere_center.y = 0.0f; sphere_center.z = ld->lenses[i].center;
        if (!raySphereIntersection(&hit_point, *ray_direction, *ray_origin, sphere_center, ld->lenses[i].curvature, 0, 1))
            return 0;
        float hitPoint2 = hit_point.x * hit_point.x + hit_point.y * hit_point.y;
        /* :1114-1115 -- the housing clip is evaluated in f64, the user-aperture clip in f32 */
        double half = (double)ld->lenses[i].aperture * 0.5;
        if (g_margin_probe) {
            float lim = (float)(half * half);
            if (i == ld->apertureElement && ld->userApertureRadius * ld->userApertureRadius < lim) lim = ld->userApertureRadius * ld->userApertureRadius;
            probe_note(hitPoint2, lim, 0);
        }
        if (((double)hitPoint2 > half * half)
            || ((i == ld->apertureElement) && (hitPoint2 > (ld->userApertureRadius * ld->userApertureRadius))))
            return 0;
        intersectionNormal(hit_point, sphere_center, ld->lenses[i].curvature, &hit_point_normal);
        if (rec) rec[(*nrec)++] = hit_point;
        *ray_origin = hit_point;
        if (i != n - 1) {
            if (!calculateTransmissionVector(ray_direction, ld->lenses[i].ior, ld->lenses[i + 1].ior, *ray_direction, hit_point_normal, 1)) {
                ld->totalInternalReflection++;
                return 0;
            }
        } else {
            if (!calculateTransmissionVector(ray_direction, ld->lenses[i].ior, 1.0, *ray_direction, hit_point_normal, 1)) {
                ld->totalInternalReflection++;
                return 0;
            }
        }
    }
    return 1;
}

int zo_trace_record(zo_camera *cam, zo_v3 *origin, zo_v3 *dir, zo_v3 *hits, int *nhits)
{
    *nhits = 0;
    return traceThroughLensElements(origin, dir, &cam->lens, hits, nhits);
}

/* traceThroughLensElementsForFocalLength, zoic.cpp:1161-1228 */
static float traceThroughLensElementsForFocalLength(zo_lensdata *ld)
{
    float tracedFocalLength = 0.0, focalPointDistance = 0.0, principlePlaneDistance = 0.0, summedThickness = 0.0;
    float rayOriginHeight = (float)((double)ld->lenses[0].aperture * 0.1);   /* :1163 */
    zo_v3 hit_point = V3(0, 0, 0), hit_point_normal;
    zo_v3 ray_origin = V3(0.0, rayOriginHeight, 0.0);
    zo_v3 ray_direction = V3(0.0, 0.0, 99999.0);
    const int n = ld->lensCount;
    for (int i = 0; i < n; i++) {
        if (i == 0) summedThickness = ld->lenses[0].thickness; else summedThickness += ld->lenses[i].thickness;
        zo_v3 sphere_center = V3(0.0, 0.0, summedThickness - ld->lenses[i].curvature);
        raySphereIntersection(&hit_point, ray_direction, ray_origin, sphere_center, ld->lenses[i].curvature, 0, 0);
        intersectionNormal(hit_point, sphere_center, ld->lenses[i].curvature, &hit_point_normal);
        if (i != n - 1) {
            if (!calculateTransmissionVector(&ray_direction, ld->lenses[i].ior, ld->lenses[i + 1].ior, ray_direction, hit_point_normal, 1))
                ld->totalInternalReflection++;
        } else {
            if (!calculateTransmissionVector(&ray_direction, ld->lenses[i].ior, 1.0, ray_direction, hit_point_normal, 1))
                ld->totalInternalReflection++;
            zo_v3 pp_line1start = V3(0.0, rayOriginHeight, 0.0);
            zo_v3 pp_line1end = V3(0.0, rayOriginHeight, 999999.0);
            zo_v3 pp_line2end = V3(0.0,
                                   (float)((double)ray_origin.y + ((double)ray_direction.y * 100000.0)),  /* :1192 */
                                   (float)((double)ray_origin.z + ((double)ray_direction.z * 100000.0))); /* :1193 */
            principlePlaneDistance = lineLineIntersection(pp_line1start, pp_line1end, ray_origin, pp_line2end).x;
            focalPointDistance = linePlaneIntersection(ray_origin, ray_direction).z;
        }
        ray_origin = hit_point;
    }
    tracedFocalLength = focalPointDistance - principlePlaneDistance;
    return tracedFocalLength;
}

/* adjustFocalLength, zoic.cpp:1231-1237 */
static void adjustFocalLength(zo_lensdata *ld)
{
    for (int i = 0; i < ld->lensCount; i++) {
        ld->lenses[i].curvature *= ld->focalLengthRatio;
        ld->lenses[i].thickness *= ld->focalLengthRatio;
        ld->lenses[i].aperture *= ld->focalLengthRatio;
    }
}

/* empericalOpticalVignetting, zoic.cpp:1297-1305 */
static int empericalOpticalVignetting(zo_v3 origin, zo_v3 direction, float apertureRadius, float opticalVignettingRadius,
                                      float opticalVignettingDistance)
{
    zo_v3 p = v3_sub(v3_mulf(direction, opticalVignettingDistance), origin);
    float pointHypotenuse = sqrtf((p.x * p.x) + (p.y * p.y));
    float virtualApertureTrueRadius = apertureRadius * opticalVignettingRadius;
    return fabsf(pointHypotenuse) < virtualApertureTrueRadius;
}

/* exitPupilLUT, zoic.cpp:1391-1452 */
static void exitPupilLUT(zo_lensdata *ld, zo_rng *rng, int filmSamplesX, int boundsSamples)
{
    float filmWidth = 4.0;
    float filmSpacingX = filmWidth / (float)filmSamplesX;
    ld->lutSize = 0;
    for (int i = 0; i < filmSamplesX; i++) {
        zo_v3 sampleOrigin = V3((float)(filmSpacingX * (float)i), 0.0, ld->originShift);
        zo_bbox2 ab; ab.min.x = ab.min.y = ab.max.x = ab.max.y = 0.0f;
        zo_v3 boundsDirection;
        float lensU = 0.0, lensV = 0.0;
        const float ap0 = ld->lenses[0].aperture;
        for (int b = 0; b < boundsSamples; b++) {
            lensU = (((float)zo_xor128(rng) / 4294967296.0f) * 2.0f) - 1.0f;     /* :1411 */
            lensV = (((float)zo_xor128(rng) / 4294967296.0f) * 2.0f) - 1.0f;     /* :1412 */
            boundsDirection.x = (lensU * ap0) - sampleOrigin.x;
            boundsDirection.y = (lensV * ap0) - sampleOrigin.y;
            boundsDirection.z = -ld->lenses[0].thickness;
            zo_v3 o = sampleOrigin, d = boundsDirection;
            if (traceThroughLensElements(&o, &d, ld, NULL, NULL)) {
                if ((ab.min.x + ab.min.y) == 0.0) {                                /* :1423 */
                    ab.min.x = lensU * ap0; ab.min.y = lensV * ap0;
                    ab.max.x = lensU * ap0; ab.max.y = lensV * ap0;
                }
                if ((lensU * ap0) > ab.max.x) ab.max.x = lensU * ap0;
                if ((lensV * ap0) > ab.max.y) ab.max.y = lensV * ap0;
                if ((lensU * ap0) < ab.min.x) ab.min.x = lensU * ap0;
                if ((lensV * ap0) < ab.min.y) ab.min.y = lensV * ap0;
            }
        }
        ld->lutKeys[ld->lutSize] = sampleOrigin.x;                      /* map.insert :1450 (keys ascending, unique) */
        ld->lutBoxes[ld->lutSize] = ab;
        ld->lutSize++;
    }
}

/* boundingBox2d::getCentroid / getMaxScale, zoic.cpp:495-517 */
static inline zo_v2 bbox_centroid(const zo_bbox2 *b)
{
    zo_v2 rv = { (b->min.x + b->max.x) * 0.5f, (b->min.y + b->max.y) * 0.5f };
    return rv;
}
static inline float bbox_max_scale(const zo_bbox2 *b)
{
    zo_v2 c = bbox_centroid(b);
    float x1 = b->max.x - c.x;
    float y2 = b->max.y - c.y;
    float scaleX = sqrtf(x1 * x1);
    float scaleY = sqrtf(y2 * y2);
    return (scaleX >= scaleY) ? scaleX : scaleY;
}

/* ----------------------------------------------------------- node methods */
zo_camera *zo_camera_new(void)
{
    zo_camera *c = calloc(1, sizeof(zo_camera));
    c->params.p.lensModel = ZO_NONE;                                    /* cameraParams() :560-572 */
    zo_rng_seed(&c->rng);
    return c;
}

void zo_camera_free(zo_camera *c)
{
    if (!c) return;
    image_invalidate(&c->image);
    free(c->params.bokehPath); free(c->params.lensDataPath);
    free(c->pend_pixels); free(c->lens_text);
    free(c);
}

void zo_camera_reset_rng(zo_camera *c) { zo_rng_seed(&c->rng); }
zo_rng *zo_camera_rng(zo_camera *c) { return &c->rng; }

void zo_camera_set_bokeh_pixels(zo_camera *c, int w, int h, int nc, const float *px)
{
    free(c->pend_pixels); c->pend_pixels = NULL;
    c->pend_w = w; c->pend_h = h; c->pend_nc = nc;
    if (px && w > 0 && h > 0 && nc > 0) {
        size_t n = (size_t)w * h * nc;
        c->pend_pixels = malloc(sizeof(float) * n);
        memcpy(c->pend_pixels, px, sizeof(float) * n);
    }
}

void zo_camera_set_lens_text(zo_camera *c, const char *text, size_t len)
{
    free(c->lens_text); c->lens_text = NULL; c->lens_text_len = 0;
    if (text) {
        c->lens_text = malloc(len + 1);
        memcpy(c->lens_text, text, len); c->lens_text[len] = 0;
        c->lens_text_len = len;
    }
}

static int streq(const char *a, const char *b) { return strcmp(a ? a : "", b ? b : "") == 0; }

/* cameraParams::lensChanged / bokehChanged, zoic.cpp:595-611 */
static int lens_changed(const zo_params *n, const zo_params_owned *o)
{
    const zo_params *r = &o->p;
    return (n->sensorWidth != r->sensorWidth || n->sensorHeight != r->sensorHeight || n->focalLength != r->focalLength
            || n->fStop != r->fStop || n->focalDistance != r->focalDistance || n->useImage != r->useImage
            || (n->useImage && !streq(n->bokehPath, o->bokehPath)) || n->lensModel != r->lensModel
            || (n->lensModel == ZO_RAYTRACED && (!streq(n->lensDataPath, o->lensDataPath) || n->kolbSamplingLUT != r->kolbSamplingLUT)));
}
static int bokeh_changed(const zo_params *n, const zo_params_owned *o)
{
    return (n->useImage != o->p.useImage || (n->useImage && !streq(n->bokehPath, o->bokehPath)));
}

static char *dupstr(const char *s) { size_t n = strlen(s ? s : ""); char *d = malloc(n + 1); memcpy(d, s ? s : "", n); d[n] = 0; return d; }

/* node_update, zoic.cpp:1575-1720 */
int zo_camera_update(zo_camera *camera, const zo_params *parms)
{
    int status = ZO_OK;
    if (bokeh_changed(parms, &camera->params)) {                        /* :1587-1593 */
        image_invalidate(&camera->image);
        if (parms->useImage && !image_read(camera)) status = ZO_ERR_BOKEH;
    }
    switch (parms->lensModel) {
    case ZO_THINLENS:                                                   /* :1598-1610 */
        camera->fov = (float)(2.0f * atan((double)(parms->sensorWidth / (2.0f * parms->focalLength))));
        camera->tan_fov = tanf(camera->fov / 2.0f);
        camera->apertureRadius = (parms->focalLength) / (2.0f * parms->fStop);
        break;
    case ZO_RAYTRACED:
        if (lens_changed(parms, &camera->params)) {                     /* :1615 */
            zo_lensdata *ld = &camera->lens;
            ld->lensCount = 0;                                          /* lenses.clear() */
            ld->vignettedRays = 0; ld->succesRays = 0; ld->totalInternalReflection = 0; ld->surfaceVisits = 0;
            ld->originShift = 0.0; ld->lutSize = 0;
            ld->apertureElementSet = 0;
            ld->filmDiagonal = sqrtf((parms->sensorWidth * parms->sensorWidth) + (parms->sensorHeight * parms->sensorHeight));
            ld->focalDistance = parms->focalDistance;
            const int have_text = camera->lens_text != NULL;
            if (!have_text && (!parms->lensDataPath || !parms->lensDataPath[0])) {
                status = ZO_ERR_LENS_PATH;                              /* :1639-1642 */
            } else {
                char *text = NULL; size_t len = 0; int rc;
                if (have_text) { text = camera->lens_text; len = camera->lens_text_len; }
                else {
                    FILE *f = fopen(parms->lensDataPath, "rb");
                    if (!f) return ZO_ERR_LENS_PATH;
                    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
                    text = malloc((size_t)sz + 1); len = fread(text, 1, (size_t)sz, f); text[len] = 0; fclose(f);
                }
                rc = read_tabular_lens_data(text, len, ld);             /* :1645 */
                if (!have_text) free(text);
                if (rc != ZO_OK) return rc;
                rc = cleanup_lens_data(ld);                             /* :1648 */
                if (rc != ZO_OK) return rc;
                if (!ld->apertureElementSet) return ZO_ERR_NO_APERTURE;
                float kolbFocalLength = traceThroughLensElementsForFocalLength(ld);       /* :1651 */
                ld->tracedFocal[0] = kolbFocalLength;
                ld->focalLengthRatio = parms->focalLength / kolbFocalLength;                /* :1654 */
                adjustFocalLength(ld);                                                      /* :1658 */
                kolbFocalLength = traceThroughLensElementsForFocalLength(ld);              /* :1661 */
                ld->tracedFocal[1] = kolbFocalLength;
                ld->userApertureRadius = (float)((double)kolbFocalLength / (2.0 * (double)parms->fStop)); /* :1664 */
                if (ld->userApertureRadius > ld->lenses[ld->apertureElement].aperture)     /* :1668-1672 */
                    ld->userApertureRadius = ld->lenses[ld->apertureElement].aperture;
                ld->originShift = calculateImageDistance(parms->focalDistance, ld);        /* :1675 */
                ld->apertureDistance = 0.0;                                                 /* :1678-1685 */
                for (int i = 0; i < ld->lensCount; i++) {
                    ld->apertureDistance += ld->lenses[i].thickness;
                    if (i == ld->apertureElement) break;
                }
                compute_lens_centers(ld);                                                   /* :1688 */
                if (parms->kolbSamplingLUT) exitPupilLUT(ld, &camera->rng, 32, 100000);     /* :1691-1692 */
            }
        }
        break;
    default: break;
    }
    /* camera->params = parms  :1719 */
    free(camera->params.bokehPath); free(camera->params.lensDataPath);
    camera->params.p = *parms;
    camera->params.bokehPath = dupstr(parms->bokehPath);
    camera->params.lensDataPath = dupstr(parms->lensDataPath);
    camera->params.p.bokehPath = camera->params.bokehPath;
    camera->params.p.lensDataPath = camera->params.lensDataPath;
    return status;
}

/* std::map::lower_bound over the sorted key array: first key >= v, lutSize if none */
static int lut_lower_bound(const zo_lensdata *ld, float v)
{
    int i = 0;
    while (i < ld->lutSize && ld->lutKeys[i] < v) ++i;
    return i;
}

static inline void sample_lens(zo_camera *camera, float u, float v, zo_v2 *lens)
{
    if (!camera->params.p.useImage) zo_concentric_disk_sample(u, v, lens);
    else image_bokeh_sample(&camera->image, u, v, &lens->x, &lens->y);
}

/* camera_create_ray, zoic.cpp:1752-1990 */
void zo_create_ray(zo_camera *camera, const zo_input *input, zo_output *output, zo_rng *rng, int *tries_out)
{
    const zo_params *params = &camera->params.p;
    zo_lensdata *ld = &camera->lens;
    if (!rng) rng = &camera->rng;
    int tries = 0, lut_miss = 0;
    const int maxtries = 25;

    switch (params->lensModel) {
    case ZO_THINLENS: {                                                 /* :1771-1846 */
        zo_v3 p = V3(input->sx * camera->tan_fov, input->sy * camera->tan_fov, 1.0);
        output->dir = ai_v3_normalize(v3_sub(p, output->origin));       /* :1777 reads caller's origin */
        zo_v3 originOriginal = output->origin;
        if (params->useDof) {
            zo_v2 lens = { 0.0, 0.0 };
            sample_lens(camera, input->lensx, input->lensy, &lens);     /* :1787 */
            lens.x *= camera->apertureRadius; lens.y *= camera->apertureRadius;
            output->origin.x = lens.x; output->origin.y = lens.y; output->origin.z = 0.0;
            float intersection = fabsf(params->focalDistance / output->dir.z);          /* :1798 */
            zo_v3 focusPoint = v3_mulf(output->dir, intersection);
            output->dir = ai_v3_normalize(v3_sub(focusPoint, output->origin));
            if (params->opticalVignettingDistance > 0.0f) {
                while (!empericalOpticalVignetting(output->origin, output->dir, camera->apertureRadius,
                                                   params->opticalVignettingRadius, params->opticalVignettingDistance)
                       && tries <= maxtries) {
                    float u = (float)zo_xor128(rng) / 4294967296.0f;    /* :1806 */
                    float v = (float)zo_xor128(rng) / 4294967296.0f;
                    sample_lens(camera, u, v, &lens);
                    lens.x *= camera->apertureRadius; lens.y *= camera->apertureRadius;
                    output->dir = ai_v3_normalize(v3_sub(p, originOriginal));
                    output->origin.x = lens.x; output->origin.y = lens.y; output->origin.z = 0.0;
                    float intersection2 = fabsf(params->focalDistance / output->dir.z);
                    zo_v3 focusPoint2 = v3_mulf(output->dir, intersection2);
                    output->dir = ai_v3_normalize(v3_sub(focusPoint2, output->origin));
                    ++tries;
                }
            }
            if (tries > maxtries) {                                     /* :1824-1830 */
                output->weight[0] = output->weight[1] = output->weight[2] = 0.0f;
                ++ld->vignettedRays;
            } else ++ld->succesRays;
        }
        output->dir.z *= -1.0f;                                         /* :1845 */
    } break;

    case ZO_RAYTRACED: {                                                /* :1850-1964 */
        output->origin.x = (float)((double)input->sx * ((double)params->sensorWidth * 0.5));   /* :1853 */
        output->origin.y = (float)((double)input->sy * ((double)params->sensorWidth * 0.5));   /* :1854 */
        output->origin.z = ld->originShift;
        zo_v3 kolb_origin_original = output->origin;
        zo_v2 lens = { 0.0, 0.0 };
        sample_lens(camera, input->lensx, input->lensy, &lens);         /* :1870 */

        if (!params->kolbSamplingLUT) {                                 /* :1873-1888 */
            output->dir.x = (lens.x * ld->lenses[0].aperture) - output->origin.x;
            output->dir.y = (lens.y * ld->lenses[0].aperture) - output->origin.y;
            output->dir.z = -ld->lenses[0].thickness;
            while (!traceThroughLensElements(&output->origin, &output->dir, ld, NULL, NULL) && tries <= maxtries) {
                output->origin = kolb_origin_original;
                float u = (float)((double)zo_xor128(rng) / 4294967296.0);   /* :1881 f64 divide, f32 argument */
                float v = (float)((double)zo_xor128(rng) / 4294967296.0);
                sample_lens(camera, u, v, &lens);
                output->dir.x = (lens.x * ld->lenses[0].aperture) - output->origin.x;
                output->dir.y = (lens.y * ld->lenses[0].aperture) - output->origin.y;
                output->dir.z = -ld->lenses[0].thickness;
                ++tries;
            }
        } else {                                                        /* :1889-1948 */
            float samplingErrorCorrection = 1.05;
            float distanceFromOrigin = fabsf(sqrtf(output->origin.x * output->origin.x + output->origin.y * output->origin.y));
            int low = lut_lower_bound(ld, distanceFromOrigin);          /* :1895 */
            float theta = (float)atan2((double)output->origin.y, (double)output->origin.x);   /* :1899 */
            float sin = zo_fast_sin(theta);
            float cos = zo_fast_cos(theta);
            float maxScale, translation;
            if (ld->lutSize <= 0 || !(distanceFromOrigin <= ld->lutKeys[ld->lutSize - 1])) {
                /* lower_bound()==end() is dereferenced at :1896 (UB, d > 3.875 cm; NaN lands here too).
                 * Fenced: such a sample is outside every tabulated image circle; use the all-zero entry
                 * semantics and report it (flag bit 6 of the batch driver). */
                maxScale = 0.0f; translation = 0.0f;
                lut_miss = 1;
            } else if (low == 0) {
                /* --begin() at :1905 is UB (d == 0).  Fenced with the reference's own d==0 branch of
                 * testAperturesLUT (zoic.cpp:1512-1518): entry 0, no interpolation. */
                maxScale = bbox_max_scale(&ld->lutBoxes[0]) * samplingErrorCorrection;
                translation = bbox_centroid(&ld->lutBoxes[0]).x;
            } else {
                float lowerBound = ld->lutKeys[low];
                float prev = ld->lutKeys[low - 1];                      /* --low :1905-1906 */
                float percentage = (distanceFromOrigin - lowerBound) / (prev - lowerBound);
                maxScale = linearInterpolate(percentage, bbox_max_scale(&ld->lutBoxes[low]), bbox_max_scale(&ld->lutBoxes[low - 1]))
                           * samplingErrorCorrection;                   /* :1910 */
                translation = linearInterpolate(percentage, bbox_centroid(&ld->lutBoxes[low]).x, bbox_centroid(&ld->lutBoxes[low - 1]).x);
            }
            lens.x *= maxScale; lens.y *= maxScale;                     /* :1913 */
            lens.x += translation;                                      /* :1914 (x only) */
            float lensx_rotated = lens.x * cos - lens.y * sin;
            float lensy_rotated = lens.x * sin + lens.y * cos;
            lens.x = lensx_rotated; lens.y = lensy_rotated;
            output->dir.x = lens.x - output->origin.x;
            output->dir.y = lens.y - output->origin.y;
            output->dir.z = -ld->lenses[0].thickness;
            while (!traceThroughLensElements(&output->origin, &output->dir, ld, NULL, NULL) && tries <= maxtries) {
                output->origin = kolb_origin_original;
                float u = (float)((double)zo_xor128(rng) / 4294967296.0);   /* :1930 */
                float v = (float)((double)zo_xor128(rng) / 4294967296.0);
#ifdef ZO_VARIANT_SWAP_UV   /* tests/test_oracle_assumptions.py only: g++'s right-to-left evaluation of the two xor128() arguments */
                { float t_ = u; u = v; v = t_; }
#endif
                sample_lens(camera, u, v, &lens);
                lens.x *= maxScale; lens.y *= maxScale;
#ifdef ZO_VARIANT_RETRY_X_ONLY   /* tests/test_oracle_assumptions.py only: what :1933 would be if it read like :1914 */
                lens.x += translation;
#else
                lens.x += translation; lens.y += translation;           /* :1933 (both components) */
#endif
                lensx_rotated = lens.x * cos - lens.y * sin;
                lensy_rotated = lens.x * sin + lens.y * cos;
                lens.x = lensx_rotated; lens.y = lensy_rotated;
                output->dir.x = lens.x - output->origin.x;
                output->dir.y = lens.y - output->origin.y;
                output->dir.z = -ld->lenses[0].thickness;
                ++tries;
            }
        }
        if (tries > maxtries) {                                         /* :1951-1957 */
            output->weight[0] = output->weight[1] = output->weight[2] = 0.0f;
            ++ld->vignettedRays;
        } else ++ld->succesRays;
        output->dir = v3_mulf(output->dir, -1.0f);                      /* :1960 */
        output->origin = v3_mulf(output->origin, -1.0f);                /* :1961 */
    } break;
    default: break;
    }

    if (tries > 0) {                                                    /* :1974-1977 */
        output->dOdy = output->origin;
        output->dDdy = output->dir;
    }
    float e2 = (params->exposureControl * params->exposureControl);     /* :1981-1987 */
    if (params->exposureControl > 0.0f) {
        for (int k = 0; k < 3; ++k) output->weight[k] *= 1.0f + e2;
    } else if (params->exposureControl < 0.0f) {
        for (int k = 0; k < 3; ++k) output->weight[k] *= 1.0f / (1.0f + e2);
    }
    if (tries_out) *tries_out = tries | (lut_miss << 8);
}

/* diagnostic: the closest accept/reject call of every decision ray i's evaluation took (see g_margin_probe) */
void zo_create_rays_probe(zo_camera *cam, size_t n, const float *in4, const uint32_t *rng_states, zo_margin_probe *out)
{
    for (size_t i = 0; i < n; ++i) {
        zo_input in = { in4[4 * i], in4[4 * i + 1], 0, 0, in4[4 * i + 2], in4[4 * i + 3], 0 };
        zo_output o; memset(&o, 0, sizeof(o));
        o.weight[0] = o.weight[1] = o.weight[2] = 1.0f;
        zo_rng r = { rng_states[4 * i], rng_states[4 * i + 1], rng_states[4 * i + 2], rng_states[4 * i + 3] };
        out[i].min_rel_margin = 3.0e38f; out[i].iface = -1; out[i].kind = -1;
        g_margin_probe = &out[i];
        int tries = 0;
        zo_create_ray(cam, &in, &o, &r, &tries);
        g_margin_probe = NULL;
        out[i].tries = tries & 0xff;
    }
}

/* ------------------------------------------------------------ batch driver */
static void one_ray(zo_camera *cam, size_t n, size_t i, const float *in4, float *planes, uint8_t *flags,
                    zo_rng *rng, int *tries_out)
{
    zo_input in; memset(&in, 0, sizeof(in));
    in.sx = in4[4 * i + 0]; in.sy = in4[4 * i + 1]; in.lensx = in4[4 * i + 2]; in.lensy = in4[4 * i + 3];
    zo_output out; memset(&out, 0, sizeof(out));
    out.weight[0] = out.weight[1] = out.weight[2] = 1.0f;               /* Arnold hands in origin=0, weight=1 */
    int tries = 0;
    zo_create_ray(cam, &in, &out, rng, &tries);
    planes[0 * n + i] = out.origin.x; planes[1 * n + i] = out.origin.y; planes[2 * n + i] = out.origin.z;
    planes[3 * n + i] = out.dir.x;    planes[4 * n + i] = out.dir.y;    planes[5 * n + i] = out.dir.z;
    planes[6 * n + i] = out.weight[0];
    int lut_miss = tries >> 8;
    tries &= 0xff;
    if (flags) flags[i] = (uint8_t)((tries > 0 ? 1 : 0) | (tries << 1) | (lut_miss << 6));
    if (tries_out) *tries_out = tries;
}

void zo_create_rays(zo_camera *cam, size_t n, const float *in4, float *planes, uint8_t *flags,
                    const uint32_t *rng_states, uint32_t *first_retry_states)
{
    for (size_t i = 0; i < n; ++i) {
        int tries = 0;
        if (rng_states) {
            zo_rng r = { rng_states[4 * i], rng_states[4 * i + 1], rng_states[4 * i + 2], rng_states[4 * i + 3] };
            one_ray(cam, n, i, in4, planes, flags, &r, &tries);
        } else {
            zo_rng before = cam->rng;       /* the first retry of this ray (if any) draws from this state */
            one_ray(cam, n, i, in4, planes, flags, NULL, &tries);
            if (first_retry_states) {
                uint32_t *s = first_retry_states + 4 * i;
                if (tries > 0) { s[0] = before.x; s[1] = before.y; s[2] = before.z; s[3] = before.w; }
                else s[0] = s[1] = s[2] = s[3] = 0;
            }
        }
    }
}

typedef struct mt_job { zo_camera *cam; size_t n; size_t *next; const float *in4; float *planes; uint8_t *flags; const uint32_t *rng; int succ, vign, tir; long long visits; } mt_job;
#define ZO_MT_CHUNK 4096   /* rays a thread takes at a time: image regions differ in cost, equal slabs would leave cores idle */

static void *mt_worker(void *arg)
{
    mt_job *j = arg;
    /* private copy of the camera head so the racy counters (zoic.cpp:1135,1142,1953-1956) stay per thread;
     * tables are shared read-only */
    zo_camera local = *j->cam;
    local.lens.succesRays = local.lens.vignettedRays = local.lens.totalInternalReflection = 0;
    local.lens.surfaceVisits = 0;
    for (;;) {
        const size_t lo = __atomic_fetch_add(j->next, (size_t)ZO_MT_CHUNK, __ATOMIC_RELAXED);
        if (lo >= j->n) break;
        const size_t hi = lo + ZO_MT_CHUNK < j->n ? lo + ZO_MT_CHUNK : j->n;
        for (size_t i = lo; i < hi; ++i) {
            zo_rng r = { j->rng[4 * i], j->rng[4 * i + 1], j->rng[4 * i + 2], j->rng[4 * i + 3] };
            one_ray(&local, j->n, i, j->in4, j->planes, j->flags, &r, NULL);
        }
    }
    j->succ = local.lens.succesRays; j->vign = local.lens.vignettedRays; j->tir = local.lens.totalInternalReflection;
    j->visits = local.lens.surfaceVisits;
    return NULL;
}

void zo_create_rays_mt(zo_camera *cam, size_t n, const float *in4, float *planes, uint8_t *flags,
                       const uint32_t *rng_states, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    mt_job *jobs = malloc(sizeof(mt_job) * nthreads);
    size_t next = 0;
    for (int t = 0; t < nthreads; ++t) {
        mt_job j = { cam, n, &next, in4, planes, flags, rng_states, 0, 0, 0, 0 };
        jobs[t] = j;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        cam->lens.succesRays += jobs[t].succ; cam->lens.vignettedRays += jobs[t].vign;
        cam->lens.totalInternalReflection += jobs[t].tir;
        cam->lens.surfaceVisits += jobs[t].visits;
    }
    free(th); free(jobs);
}

/* ---------------------------------------------------------------- getters */
int   zo_lens_count(const zo_camera *c) { return c->lens.lensCount; }
int   zo_aperture_element(const zo_camera *c) { return c->lens.apertureElement; }
const zo_lens_element *zo_lenses(const zo_camera *c) { return c->lens.lenses; }
float zo_user_aperture_radius(const zo_camera *c) { return c->lens.userApertureRadius; }
float zo_origin_shift(const zo_camera *c) { return c->lens.originShift; }
float zo_aperture_distance(const zo_camera *c) { return c->lens.apertureDistance; }
float zo_focal_length_ratio(const zo_camera *c) { return c->lens.focalLengthRatio; }
float zo_traced_focal_length(const zo_camera *c, int which) { return c->lens.tracedFocal[which ? 1 : 0]; }
int   zo_lut_size(const zo_camera *c) { return c->lens.lutSize; }
const float *zo_lut_keys(const zo_camera *c) { return c->lens.lutKeys; }
const zo_bbox2 *zo_lut_boxes(const zo_camera *c) { return c->lens.lutBoxes; }
float zo_fov(const zo_camera *c) { return c->fov; }
float zo_tan_fov(const zo_camera *c) { return c->tan_fov; }
float zo_aperture_radius(const zo_camera *c) { return c->apertureRadius; }
void  zo_counters(const zo_camera *c, int *s, int *v, int *t)
{
    if (s) *s = c->lens.succesRays;
    if (v) *v = c->lens.vignettedRays;
    if (t) *t = c->lens.totalInternalReflection;
}
long long zo_surface_visits(const zo_camera *c) { return c->lens.surfaceVisits; }
int   zo_bokeh_dims(const zo_camera *c, int *x, int *y) { if (x) *x = c->image.x; if (y) *y = c->image.y; return image_valid(&c->image); }
const float *zo_bokeh_cdf_row(const zo_camera *c) { return c->image.cdfRow; }
const float *zo_bokeh_cdf_column(const zo_camera *c) { return c->image.cdfColumn; }
const int   *zo_bokeh_row_indices(const zo_camera *c) { return c->image.rowIndices; }
const int   *zo_bokeh_column_indices(const zo_camera *c) { return c->image.columnIndices; }
void  zo_bokeh_sample(const zo_camera *c, float u1, float u2, float *dx, float *dy) { image_bokeh_sample(&c->image, u1, u2, dx, dy); }
