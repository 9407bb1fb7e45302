/*
 * oracle/zoic_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the per-sample lens hot path of zpelgrims/zoic
 * (src/zoic.cpp) and of the cold precompute that builds the tables the hot
 * path reads.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed
 * CPU baseline.  The product (zoic_amd/csrc) never links or calls it.
 *
 * Every function cites the reference file:line it follows ("zoic.cpp:N").
 */
#ifndef ZOIC_ORACLE_H
#define ZOIC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } zo_v3;
typedef struct { float x, y; } zo_v2;

/* zoic.cpp:84-88 */
enum { ZO_THINLENS = 0, ZO_RAYTRACED = 1, ZO_NONE = 2 };

/* the 14 node parameters, zoic.cpp:1547-1562 (defaults) / 578-593 (read) */
typedef struct zo_params {
    float sensorWidth;
    float sensorHeight;
    float focalLength;
    float fStop;
    float focalDistance;
    int   useImage;
    int   lensModel;
    int   kolbSamplingLUT;
    int   useDof;
    float opticalVignettingDistance;
    float opticalVignettingRadius;
    float exposureControl;
    const char *bokehPath;     /* identity only: pixels come from zo_camera_set_bokeh_pixels */
    const char *lensDataPath;  /* read with fopen unless lens text was supplied */
} zo_params;

/* zoic.cpp:522-525 */
typedef struct zo_lens_element {
    float curvature, thickness, ior, aperture, abbe, center;
} zo_lens_element;

/* zoic.cpp:490-493 */
typedef struct zo_bbox2 { zo_v2 max, min; } zo_bbox2;

/* zoic.cpp:647-652 state */
typedef struct zo_rng { uint32_t x, y, z, w; } zo_rng;

/* AtCameraInput / AtCameraOutput field order (Arnold 5 SDK) */
typedef struct zo_input  { float sx, sy, dsx, dsy, lensx, lensy, relative_time; } zo_input;
typedef struct zo_output { zo_v3 origin, dir, dOdx, dOdy, dDdx, dDdy; float weight[3]; } zo_output;

#define ZO_MAX_LENSES 64
#define ZO_LUT_SIZE   32

typedef struct zo_camera zo_camera;

/* status codes of zo_camera_update */
enum {
    ZO_OK = 0,
    ZO_ERR_LENS_PATH = 1,      /* zoic.cpp:1639-1642 */
    ZO_ERR_LENS_COLUMNS = 2,   /* zoic.cpp:745-754 */
    ZO_ERR_MULTI_APERTURE = 3, /* zoic.cpp:926-929 */
    ZO_ERR_NO_APERTURE = 4,    /* zoic.cpp:922 is the only write of apertureElement: UB fenced off */
    ZO_ERR_BOKEH = 5,          /* zoic.cpp:1589-1592 */
    ZO_ERR_LENS_PARSE = 6,     /* std::stof would throw (zoic.cpp:774 ff.) */
    ZO_ERR_TOO_MANY_LENSES = 7
};

zo_camera *zo_camera_new(void);                    /* node_initialize zoic.cpp:1565-1572 */
void       zo_camera_free(zo_camera *);            /* node_finish     zoic.cpp:1747 */
void       zo_rng_seed(zo_rng *);                  /* zoic.cpp:648 seed */
uint32_t   zo_xor128(zo_rng *);                    /* zoic.cpp:647-652 */
void       zo_camera_reset_rng(zo_camera *);       /* fresh-process state of the function-static xor128 */
zo_rng    *zo_camera_rng(zo_camera *);

/* stand-in for AiTextureGetResolution/GetNumChannels/AiTextureLoad (zoic.cpp:176-186):
 * the pixels the next update will "load" when useImage is set */
void zo_camera_set_bokeh_pixels(zo_camera *, int w, int h, int nchannels, const float *pixels);
/* optional: lens prescription text instead of fopen(lensDataPath) */
void zo_camera_set_lens_text(zo_camera *, const char *text, size_t len);

int zo_camera_update(zo_camera *, const zo_params *);   /* node_update zoic.cpp:1575-1720 */

/* camera_create_ray zoic.cpp:1752-1990.  rng==NULL -> the camera-global stream
 * (reference semantics, single thread).  *tries_out receives `tries` | (outside-LUT fence << 8). */
void zo_create_ray(zo_camera *, const zo_input *in, zo_output *out, zo_rng *rng, int *tries_out);

/* batch driver used by tests / cpu baseline.
 *   in4        : n x (sx, sy, lensx, lensy)
 *   planes     : 7 planes of n floats: ox oy oz dx dy dz weight
 *   flags      : n bytes: bit0 = retried (tries>0), bits1-5 = tries (0..26), bit6 = outside the LUT (fenced UB)
 *   rng_states : NULL -> global sequential stream; else n x 4 u32 per-ray states (read only)
 *   first_retry_states : optional out, n x 4 u32: stream state when the ray's first retry drew
 *                        (state before the draw); zeros for rays that never retried */
void zo_create_rays(zo_camera *, size_t n, const float *in4, float *planes, uint8_t *flags,
                    const uint32_t *rng_states, uint32_t *first_retry_states);

/* multi-threaded cpu baseline: per-ray states are mandatory (the reference's shared RNG is a race) */
void zo_create_rays_mt(zo_camera *, size_t n, const float *in4, float *planes, uint8_t *flags,
                       const uint32_t *rng_states, int nthreads);

/* table access for parity tests */
int   zo_lens_count(const zo_camera *);
int   zo_aperture_element(const zo_camera *);
const zo_lens_element *zo_lenses(const zo_camera *);
float zo_user_aperture_radius(const zo_camera *);
float zo_origin_shift(const zo_camera *);
float zo_aperture_distance(const zo_camera *);
float zo_focal_length_ratio(const zo_camera *);
float zo_traced_focal_length(const zo_camera *, int which); /* 0: before, 1: after adjust */
int   zo_lut_size(const zo_camera *);
/* not a reference quantity: interfaces entered by traceThroughLensElements (hot path + precompute) since the last lens rebuild;
 * the work measure of SURVEY 8(d)'s FLOP model (~106 FLOP per interface visit + ~130 per try) */
long long zo_surface_visits(const zo_camera *);
const float    *zo_lut_keys(const zo_camera *);
const zo_bbox2 *zo_lut_boxes(const zo_camera *);
float zo_fov(const zo_camera *);
float zo_tan_fov(const zo_camera *);
float zo_aperture_radius(const zo_camera *);
void  zo_counters(const zo_camera *, int *succes, int *vignetted, int *tir);
int   zo_bokeh_dims(const zo_camera *, int *x, int *y);
const float *zo_bokeh_cdf_row(const zo_camera *);
const float *zo_bokeh_cdf_column(const zo_camera *);
const int   *zo_bokeh_row_indices(const zo_camera *);
const int   *zo_bokeh_column_indices(const zo_camera *);

/* exposed pieces for known-answer tests */
void  zo_concentric_disk_sample(float ox, float oy, zo_v2 *lens);      /* zoic.cpp:686-704 */
float zo_fast_sin(float x);                                           /* zoic.cpp:661-668 */
float zo_fast_cos(float x);                                           /* zoic.cpp:671-681 */
void  zo_bokeh_sample(const zo_camera *, float u1, float u2, float *dx, float *dy); /* zoic.cpp:420-485 */
/* traceThroughLensElements with the _DRAW hit-point dump (zoic.cpp:1099-1158, 1121-1128,1146-1153):
 * hits receives up to lensCount (z,y,x) triples of the accepted hit points; returns 1 on success and
 * writes the number of recorded hits to *nhits. */
int   zo_trace_record(zo_camera *, zo_v3 *origin, zo_v3 *dir, zo_v3 *hits, int *nhits);

/* diagnostic (tools/tests only): how close the closest accept/reject decision of a ray's evaluation was.
 * kind: 0 housing/stop clip (zoic.cpp:1114-1115), 1 sphere miss (:980), 2 total internal reflection (:1019) */
typedef struct zo_margin_probe { float min_rel_margin; int iface, kind, tries; } zo_margin_probe;
void zo_create_rays_probe(zo_camera *cam, size_t n, const float *in4, const uint32_t *rng_states, zo_margin_probe *out);

#ifdef __cplusplus
}
#endif
#endif
