/*
 * include/zoic_amd.h -- C-ABI of libzoic_amd.so, the MI355X (gfx950) camera-ray generator for
 * zoic's per-sample lens hot path.
 *
 * Drop-in boundary.  zoic is an Arnold camera node: Arnold calls the method table
 * (AI_CAMERA_NODE_EXPORT_METHODS(zoicMethods), zoic.cpp:61; returned by NodeLoader,
 * zoic.cpp:1999-2007) -- node_parameters / node_initialize / node_update / node_finish /
 * camera_create_ray.  Each entry point below replaces one of those methods (file:line cited
 * per function) with plain pointers and sizes; no C++ or torch types cross this boundary.
 * INTEGRATION.md shows the binding a zoic maintainer would add inside zoic.cpp.
 *
 * All compute entry points run hand-written HIP kernels; there is no CPU fallback.  Every
 * function returns a zoic_status; ZOIC_OK == 0.
 *
 * Threading contract -- the reference's own (Arnold calls node_initialize / node_update / node_finish from one
 * thread with no sample in flight, and camera_create_ray concurrently from every render thread, zoic.cpp:1752):
 *   - zoic_camera_create / _update / _destroy / _set_* / _reset_counters: one thread at a time per camera, and no
 *     ray call of that camera running on another thread.  (_update and _destroy wait for launches still queued.)
 *   - zoic_create_rays_device / _host / _arnold / _device_resident, zoic_camera_create_ray, zoic_camera_create_rays_tile, zoic_tile_submit / _wait /
 *     _done (one tile per thread), zoic_camera_reverse_ray, zoic_camera_get_counters, zoic_camera_set_wait_mode: any number of host threads on one camera at once.  Each call works on private
 *     scratch and private HIP streams and waits only for its own work; results do not depend on the interleaving
 *     (batched calls key every ray's retry stream by its global ray index, the per-sample call by its tid).
 *   - No entry point changes the calling thread's current HIP device.
 */
#ifndef ZOIC_AMD_H
#define ZOIC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (a plug-in shim checks zoic_abi_version() == ZOIC_AMD_ABI_VERSION at load and refuses anything else):
 *   1  round 1.
 *   2  round 2: zoic_create_rays_arnold, zoic_host_*, per-tid retry streams.
 *   3  round 3/4: zoic_lens_info gained the trailing fastRunsStrict (zoic_camera_get_info writes sizeof(zoic_lens_info) bytes:
 *      a v2 caller's struct is 8 bytes short, see the end of this entry); zoic_create_rays_arnold WRITES EVERY OUTPUT ROW WHOLE (v2 updated fields in
 *      place); zoic_camera_create_ray runs through a resident kernel, so a caller's hipDeviceSynchronize / hipFree can wait up
 *      to 50 ms for it to retire (zoic_camera_get_counters / _update / _destroy stop it first); the zoic_frame_* entry points
 *      (one frame over several devices of this process); zoic_lens_info gained precomputeTIR behind fastRunsStrict (8 bytes in all).
 *   4  round 5: zoic_tile_* / zoic_camera_create_rays_tile (bucket-sized batches through the resident kernel, no launch);
 *      zoic_frame_get_lane_info; ZOIC_FRAME_PAYLOAD_SPARSE; zoic_camera_set_frame_aspect; zoic_tile_set_rows / zoic_tile_rays, zoic_tile_set_inputs / zoic_tile_samples.
 *      Nothing of ABI 3 changed shape.
 *   5  round 6: zoic_camera_set_wait_mode (how a render thread waits for the resident kernel: spin / yield / sleep); a zoic_tile that
 *      outlives its camera is DETACHED by zoic_camera_destroy (every call but zoic_tile_destroy fails, the array getters return NULL)
 *      instead of dangling; zoic_tile_done restarts a resident kernel that retired under the poll; ZOIC_FRAME_PAYLOAD_AUTO +
 *      zoic_frame_auto_layout (the gather's layout chosen from the camera's dead-ray fraction); zoic_create_rays_device_resident (device
 *      buffers through the resident kernel, no launch).  Nothing of ABI 4 changed shape. */
#define ZOIC_AMD_ABI_VERSION 5

typedef enum zoic_status {
    ZOIC_OK = 0,
    ZOIC_ERR_INVALID_ARGUMENT = 1,
    ZOIC_ERR_LENS_PATH = 2,        /* "[ZOIC] Lens Data Path is invalid"            zoic.cpp:1639-1642 */
    ZOIC_ERR_LENS_COLUMNS = 3,     /* "<4 / >5 columns of data"                     zoic.cpp:745-754   */
    ZOIC_ERR_LENS_PARSE = 4,       /* std::stof would throw                         zoic.cpp:774 ff.   */
    ZOIC_ERR_MULTI_APERTURE = 5,   /* "Multiple apertures found"                    zoic.cpp:926-929   */
    ZOIC_ERR_NO_APERTURE = 6,      /* no zero-radius row: apertureElement would be read uninitialised (zoic.cpp:922,1115,1668) */
    ZOIC_ERR_TOO_MANY_LENSES = 7,  /* > ZOIC_MAX_LENS_SURFACES rows */
    ZOIC_ERR_BOKEH_IMAGE = 8,      /* "[ZOIC] Couldn't open bokeh image!"           zoic.cpp:1589-1592 */
    ZOIC_ERR_NOT_UPDATED = 9,      /* create_rays before a successful update */
    ZOIC_ERR_HIP = 10,             /* HIP runtime error; zoic_last_error_string() has the text */
    ZOIC_ERR_NO_DEVICE = 11        /* no gfx950 device visible: this library has no CPU path */
} zoic_status;

/* enum LensModel, zoic.cpp:84-88 */
typedef enum zoic_lens_model { ZOIC_THINLENS = 0, ZOIC_RAYTRACED = 1, ZOIC_LENS_NONE = 2 } zoic_lens_model;

/* arithmetic mode of the kernels */
typedef enum zoic_precision {
    ZOIC_PRECISION_STRICT = 0, /* the reference's operation order, no FMA contraction, its f64 intermediates: bit-exact vs the CPU oracle */
    ZOIC_PRECISION_FAST = 1,   /* same algorithm, f32 only, FMA/rsq, redundant normalisations removed: direction RMSE < 1e-5.
                                  Decision-safe: housing / stop clips that lie inside the interface's guard band (wide at the stop,
                                  which the reference traces as a sphere of |R| ~ 1e4 cm; a few ulps elsewhere), and the
                                  exit-pupil LUT's edge, are re-taken in STRICT arithmetic (a second kernel over the few rays
                                  concerned).  Sphere-miss and TIR decisions are not guarded: residual flips of try count /
                                  weight: none in 33.5 M rays of each benchmark configuration (tests hold 5e-5); origin / direction
                                  differ in low-order bits */
    ZOIC_PRECISION_FAST_UNCHECKED = 2 /* FAST without the decision check (A/B; decisions flip where the reference's own f32
                                         rounding noise decides, ~1e-5 ... 1e-3 of the rays depending on the lens) */
    /* Domain of the FAST modes: they drop the reference's per-interface renormalisations (the refracted direction stays unit
       by construction as long as every hit lies ON its sphere to rounding), which holds while hits are the near-vertex roots
       of a lens laid out rear -> front with the sensor behind it.  A prescription whose traced focal length is negative is
       rescaled by a NEGATIVE focalLengthRatio (zoic.cpp:1651-1661: thicknesses and radii change sign), and one whose focus
       computation puts the sensor in FRONT of the rear vertex (originShift >= lenses[0].thickness) sends its rays away from
       the lens: the reference's one signed root (zoic.cpp:986, t < 0 never rejected) then lands far behind the ray, and the
       hit-point rounding no longer is small against the radii.  Such a camera (zoic_lens_info::fastRunsStrict) runs STRICT in
       every mode.  So does a camera that fails zoic_camera_update's self-check: TWO slabs of 4096 probe samples each (two jitters
       of a 64 x 64 lattice over sx in [-1, 1], sy in [-a, a], a = zoic_camera_set_frame_aspect's value, 1.0 by default) go through
       the STRICT and the decision-safe FAST kernels, and the FAST modes are kept only if, on EACH slab, at most one ray is decided
       differently and the direction RMSE of the others is < 5e-6 -- half of the 1e-5 tolerance (0.3 ms per update that rebuilds
       tables; no counter, no retry stream is touched; a HIP failure inside the check is an error of the update). */
} zoic_precision;

#define ZOIC_MAX_LENS_SURFACES 32
#define ZOIC_DEVICE_NONE (-1) /* zoic_camera_create: tables-only camera (host precompute of node_update; cannot make rays) */
#define ZOIC_LUT_ENTRIES 32 /* exitPupilLUT(&ld, 32, 100000), zoic.cpp:1692 */

/* The 14 node parameters: names, types and defaults of node_parameters (zoic.cpp:1547-1562),
 * read by cameraParams::fromNode (zoic.cpp:578-593). */
typedef struct zoic_params {
    float sensorWidth;               /* 3.6  cm */
    float sensorHeight;              /* 2.4  cm */
    float focalLength;               /* 2.0  cm */
    float fStop;                     /* 4.0     */
    float focalDistance;             /* 100  cm */
    int32_t useImage;                /* false   */
    const char *bokehPath;           /* ""  : identity of the bokeh image (change detection, zoic.cpp:608-611);
                                              a ".pfm" path is loaded from disk unless pixels were supplied by
                                              zoic_camera_set_bokeh_image */
    int32_t lensModel;               /* RAYTRACED */
    const char *lensDataPath;        /* ""  : tabular .dat lens prescription */
    int32_t kolbSamplingLUT;         /* true    */
    int32_t useDof;                  /* true    */
    float opticalVignettingDistance; /* 0.0     */
    float opticalVignettingRadius;   /* 1.0     */
    float exposureControl;           /* 0.0     */
} zoic_params;

/* AtCameraInput / AtCameraOutput (Arnold 5 ai_cameras.h) as PODs; zoic reads
 * input.{sx,sy,lensx,lensy} and writes output.{origin,dir,weight,dOdy,dDdy} (zoic.cpp:1752-1990). */
typedef struct zoic_camera_input { float sx, sy, dsx, dsy, lensx, lensy, relative_time; } zoic_camera_input;
typedef struct zoic_vec3 { float x, y, z; } zoic_vec3;
typedef struct zoic_camera_output {
    zoic_vec3 origin, dir, dOdx, dOdy, dDdx, dDdy;
    float weight[3];
} zoic_camera_output;

/* Batch output: one 32-byte record per ray (the fields of AtCameraOutput that zoic writes, zoic.cpp:1752-1990).
 * flags: bit0 = retried (tries > 0  => the caller sets dOdy=origin, dDdy=dir, zoic.cpp:1974-1977),
 *        bits1-5 = tries (0..26; 26 => weight 0, zoic.cpp:1951-1953), bit6 = outside the exit-pupil LUT (fenced UB). */
typedef struct zoic_ray {
    float ox, oy, oz;   /* output.origin */
    float dx, dy, dz;   /* output.dir    */
    float weight;       /* output.weight (r == g == b; the caller's initial weight is taken as 1) */
    uint32_t flags;
} zoic_ray;

/* struct cameraData (zoic.cpp:627-643) + its device tables */
typedef struct zoic_camera zoic_camera;
/* succesRays / vignettedRays / totalInternalReflection of struct Lensdata (zoic.cpp:533-534), printed by node_finish */
typedef struct zoic_counters { uint64_t succesRays, vignettedRays, totalInternalReflection; } zoic_counters;

/* ---- environment ---------------------------------------------------------------------------
 * The library reads three environment variables, all of them about WHERE node_update builds its tables (the tables are
 * identical either way; tests/test_parity_gpu.py compares both builds).  None selects a CPU ray path: there is none.
 *   ZOIC_LUT_HOST=1    exit-pupil LUT (draws, probe traces, bounding boxes) built by the host loop instead of lut_build.hip;
 *                      =2: probes traced on the GPU, draws and the order-dependent box replay on the host (round 1's build)
 *   ZOIC_CDF_HOST=1    bokeh CDFs (bokehProbability) built by the host loop instead of bokeh_cdf.hip
 *   ZOIC_CELLS_HOST=1  bokeh cell records built by the host loop instead of build_cells_kernel
 * Kernel tuning constants are compile-time (-D, tools/build_variant.sh): ZOIC_MIN_SEARCHING, ZOIC_CHUNK_RAYS, ZOIC_GRID_BLOCKS,
 * ZOIC_GUARD_SCALE, ZOIC_RETRY_DEAD_MIN_SHARE, ZOIC_POOL_SLIM, ZOIC_TRACE_PREFETCH. */

/* ---- library ------------------------------------------------------------------------------- */
int         zoic_abi_version(void);
const char *zoic_status_string(zoic_status);
const char *zoic_last_error_string(void);          /* thread-local detail of the last failure */
int         zoic_device_count(void);               /* gfx950 devices visible to HIP */
/* NUMA node of the host the device hangs off (sysfs, via its PCI bus id), -1 if unknown: render threads that call
 * zoic_camera_create_ray save ~1 us per call when they run on that node */
int         zoic_device_numa_node(int device);

/* ---- node lifetime ------------------------------------------------------------------------- */
/* node_parameters defaults, zoic.cpp:1547-1562 */
void zoic_params_default(zoic_params *p);
/* node_initialize, zoic.cpp:1565-1572 (`new cameraData()`); binds the camera to one HIP device.
 * device == ZOIC_DEVICE_NONE gives a tables-only camera: update() runs the host precompute (lens parse, focus,
 * exit-pupil LUT, bokeh CDF) and the get_info/get_bokeh_tables getters work, but every create_rays call fails
 * with ZOIC_ERR_NO_DEVICE -- there is no CPU ray path. */
zoic_status zoic_camera_create(int device, zoic_camera **out);
/* node_finish, zoic.cpp:1723-1749 (`delete camera`) */
void zoic_camera_destroy(zoic_camera *cam);
/* node_update, zoic.cpp:1575-1720: bokeh CDF build, lens parse + precompute + exit-pupil LUT, upload */
zoic_status zoic_camera_update(zoic_camera *cam, const zoic_params *p);
/* replaces AiTextureGetResolution / AiTextureGetNumChannels / AiTextureLoad (zoic.cpp:101-103,176-186):
 * the pixels (row-major, nchannels interleaved floats) the next update uses when useImage is set */
zoic_status zoic_camera_set_bokeh_image(zoic_camera *cam, int width, int height, int nchannels, const float *pixels);
/* lens prescription text in memory instead of fopen(lensDataPath) (readTabularLensData, zoic.cpp:708-914) */
zoic_status zoic_camera_set_lens_text(zoic_camera *cam, const char *text, size_t len);
zoic_status zoic_camera_set_precision(zoic_camera *cam, zoic_precision mode);
/* The largest |sy| the renderer will send (Arnold's convention: sx in [-1, 1], sy in [-1/aspect, 1/aspect]): the extent
 * of the frame zoic_camera_update's self-check of the FAST modes spreads its probe samples over.  Default 1.0 (a square frame; a
 * 3:2 frame is inside it); a portrait frame passes its own 1/aspect > 1.  Takes effect at the next update that rebuilds tables
 * (or at once for the verdict of the next update when the value changed). */
zoic_status zoic_camera_set_frame_aspect(zoic_camera *cam, float max_abs_sy);
/* How a render thread waits for the camera's RESIDENT kernel (zoic_camera_create_ray, zoic_tile_wait, zoic_camera_create_rays_tile).
 *   ZOIC_WAIT_SPIN   (default) a `pause` loop on the reply / the tile's flags: lowest latency, one core per waiting thread.  Right when
 *                    the render threads have a core each.
 *   ZOIC_WAIT_YIELD  spins for ~2 us, then sched_yield() between polls: for hosts with more render threads than CORES.  (Under a CPU quota
 *                    with idle cores -- a container -- sched_yield returns at once and burns quota like the spin: use ZOIC_WAIT_SLEEP there.)
 *   ZOIC_WAIT_SLEEP  spins for ~2 us, then sleeps 20 us between polls: a waiting thread costs next to nothing; +10-20 us per call.
 *                    [MI355X box, 16-CPU quota, 4096-sample tiles] 64 threads: 573 Mrays/s, p99 0.58 ms (spinning: 240 Mrays/s, p99 10 ms;
 *                    16 threads spinning: 490); 128 threads: 575 Mrays/s (profiles/ab_r06/tile_threads_wait_modes.txt).
 * Any thread, any time; takes effect at the next wait. */
typedef enum zoic_wait_mode { ZOIC_WAIT_SPIN = 0, ZOIC_WAIT_YIELD = 1, ZOIC_WAIT_SLEEP = 2 } zoic_wait_mode;
zoic_status zoic_camera_set_wait_mode(zoic_camera *cam, zoic_wait_mode mode);
/* seed of the per-ray retry streams (see zoic_create_rays_device) */
zoic_status zoic_camera_set_seed(zoic_camera *cam, uint32_t seed);

/* ---- the hot path: camera_create_ray, zoic.cpp:1752-1990 ----------------------------------- */
/* n samples already resident in device memory.
 *   d_samples    : n x (sx, sy, lensx, lensy) f32, 16-byte aligned (AtCameraInput fields zoic reads)
 *   d_rng_states : NULL, or n x 4 u32 xorshift128 states (zoic.cpp:647-652), one private retry stream per ray.
 *                  NULL => ray i uses the stream seeded from (seed, ray_index_base + i), so results do not
 *                  depend on how the image is split over launches or GPUs.
 *   d_rays       : n zoic_ray records in device memory, 16-byte aligned
 *   stream       : hipStream_t (NULL = default stream).  Asynchronous.
 * The reference draws retries from ONE process-global stream shared (racily) by all render threads
 * (zoic.cpp:648); a per-ray stream is the only order-independent restatement. */
zoic_status zoic_create_rays_device(zoic_camera *cam, uint64_t n, const float *d_samples, const uint32_t *d_rng_states,
                                    uint64_t ray_index_base, zoic_ray *d_rays, void *stream);
/* same, host buffers: H2D, kernels, D2H in pieces on three private streams (copy-in, kernels, copy-out); returns when h_rays is complete */
zoic_status zoic_create_rays_host(zoic_camera *cam, uint64_t n, const float *h_samples, const uint32_t *h_rng_states,
                                  uint64_t ray_index_base, zoic_ray *h_rays);
/* Arnold-layout batch: n AtCameraInput -> n AtCameraOutput (host arrays; page-locked ones -- zoic_host_alloc /
 * zoic_host_register -- move at PCIe rate).  Every output row is WRITTEN WHOLE (the expansion runs on the GPU): origin, dir,
 * weight[3] = the exposure factor or 0 (the caller's initial weight is taken as 1), dOdy = origin and dDdy = dir for rays
 * that retried (zoic.cpp:1974-1977), and 0 in what camera_create_ray leaves alone (dOdx, dDdx; dOdy, dDdy of first-try
 * rays) -- the values of a zero-initialised AtCameraOutput.  The thin-lens branch reads output.origin (zoic.cpp:1777): it is
 * taken as 0, as Arnold hands it in.  Ray i draws its retries from the stream keyed by ray_index_base + i. */
zoic_status zoic_create_rays_arnold(zoic_camera *cam, uint64_t n, const zoic_camera_input *inputs,
                                    zoic_camera_output *outputs, uint64_t ray_index_base);
/* camera_create_ray(node, input, output, tid), zoic.cpp:1752: the per-sample signature.  No launch per call: the sample goes
 * to a resident kernel through mapped pinned memory (csrc/mailbox.hip; ~7 us per call; the kernel retires by itself after 1 ms
 * without a call and is started again by the next one).  Re-entrant: every tid owns a retry stream that carries over from call
 * to call, so two samples that retry never see the same draws; tid 0's stream is the reference's process-global xor128 state
 * (seeded 123456789..., advanced by node_update's LUT build and by every retry), so ONE render thread reproduces the
 * reference's sequential output exactly (STRICT precision).
 * Output fields: as the reference, the call UPDATES the caller's AtCameraOutput in place -- origin, dir written; weight set to 0
 * (zoic.cpp:1825/1952) or multiplied by the exposure factor (zoic.cpp:1981-1987); dOdy/dDdy written for retried rays only;
 * dOdx/dDdx never touched -- whereas zoic_create_rays_arnold writes whole rows from a zero-initialised output with weight 1.
 * For an output that Arnold hands in (zeroed, weight 1) the two agree.  THINLENS is evaluated in the reference's arithmetic in
 * every precision mode here; under ZOIC_PRECISION_FAST with opticalVignettingDistance > 0 the batch entry points use the fast
 * arithmetic, so the same sample can differ in low-order bits between the two (decisions never differ). */
zoic_status zoic_camera_create_ray(zoic_camera *cam, const zoic_camera_input *input, zoic_camera_output *output,
                                   uint16_t tid);
/* ---- tiles: what a render thread buffers of camera_create_ray (zoic.cpp:1752), answered without a launch --------------------
 * A renderer works in buckets (Arnold: 64 x 64 pixels x AA^2 samples = 36 K ... 150 K samples per bucket and thread).  Served by
 * kernel launches such a batch costs 50-80 us of launch, stream and copy overhead whatever it carries; a zoic_tile goes to the
 * camera's RESIDENT kernel instead (csrc/mailbox.hip): the request is one 64-byte line in mapped page-locked memory, the kernel's
 * worker waves take 64 samples each at full lane width -- the same device functions, hence the same bits, as the batch kernels --
 * read the AtCameraInput rows and write the AtCameraOutput rows in place across PCIe, and the render thread spins on one word.
 *   zoic_tile_create  page-locked input / output arrays of `capacity` rows (<= ZOIC_TILE_MAX_SAMPLES) for render thread `tid`
 *                     (tile requests of tids 64 apart share a mailbox slot and are served one after the other);
 *   zoic_tile_inputs / _outputs   the arrays: the caller fills inputs[0 .. n) (the fields zoic reads: sx, sy, lensx, lensy);
 *   zoic_tile_submit  posts rows [0, n) and returns at once; ray i draws its retries from the stream keyed by
 *                     ray_index_base + i, exactly as zoic_create_rays_arnold does: outputs[i] equals that call's row bit for bit
 *                     (whole rows: origin, dir, weight[3], dOdy / dDdy for retried rays, zeros elsewhere);
 *   zoic_tile_wait    returns when outputs[0 .. n) are complete; zoic_tile_done polls (1 = complete, nothing pending).
 * One submit per tile at a time (a second submit waits for the first).  A tile belongs to one render thread; different tiles may
 * be used from different threads at once.  zoic_camera_update: wait for the camera's tiles first.  zoic_camera_destroy settles the
 * tiles still alive and DETACHES them: their arrays go with the camera (zoic_tile_inputs / _outputs / _rays / _samples return NULL,
 * zoic_tile_capacity 0), every call on them but zoic_tile_destroy fails with ZOIC_ERR_INVALID_ARGUMENT.
 * zoic_tile_done may be polled without a zoic_tile_wait in between (it restarts a resident kernel that retired under the poll); a
 * HIP error it runs into is reported by the next zoic_tile_wait / _submit.
 * zoic_camera_create_rays_tile is the one-call form for arrays the caller owns: page-locked mapped arrays (zoic_host_alloc /
 * zoic_host_register) are used in place, anything else is staged through the slot's own page-locked buffers (two 16 Ki-row
 * pieces in flight).  Any n.  Counters: tile rays count like every other ray.
 *   zoic_tile_set_rows(tile, ZOIC_TILE_ROWS_RAYS)  the tile's answer is n zoic_ray RECORDS (32 bytes: origin, dir, weight, flags --
 *                     what zoic_create_rays_device writes, bit for bit) at zoic_tile_rays() instead of n AtCameraOutput rows (84 bytes,
 *                     51 of them zeros or copies): with many render threads a tile costs what its rows cost on PCIe, and a
 *                     renderer that reads origin / dir / weight (dOdy = origin, dDdy = dir when flags bit 0 is set, zoic.cpp:1974-1977)
 *                     needs nothing else.  ZOIC_TILE_ROWS_ARNOLD (the default) switches back.  Between a wait and the next submit only.
 *   zoic_tile_set_inputs(tile, ZOIC_TILE_INPUTS_SAMPLES)  the tile is filled with 16-byte (sx, sy, lensx, lensy) samples at zoic_tile_samples()
 *                     -- the four fields zoic reads (zoic.cpp:1853-1854, 1870), what zoic_create_rays_device takes -- instead of 28-byte
 *                     AtCameraInput rows.  ZOIC_TILE_INPUTS_ARNOLD (the default) switches back.  Between a wait and the next submit only. */
#define ZOIC_TILE_MAX_SAMPLES 65536u
#define ZOIC_TILE_ROWS_ARNOLD 0
#define ZOIC_TILE_ROWS_RAYS 1
#define ZOIC_TILE_INPUTS_ARNOLD 0
#define ZOIC_TILE_INPUTS_SAMPLES 1
typedef struct zoic_tile zoic_tile;
zoic_status zoic_tile_create(zoic_camera *cam, uint32_t capacity, uint16_t tid, zoic_tile **out);
void        zoic_tile_destroy(zoic_tile *tile);
zoic_camera_input  *zoic_tile_inputs(zoic_tile *tile);
zoic_camera_output *zoic_tile_outputs(zoic_tile *tile);
uint32_t    zoic_tile_capacity(const zoic_tile *tile);
zoic_status zoic_tile_submit(zoic_tile *tile, uint32_t n, uint64_t ray_index_base);
zoic_status zoic_tile_wait(zoic_tile *tile);
int         zoic_tile_done(zoic_tile *tile);
zoic_status zoic_tile_set_rows(zoic_tile *tile, int rows);
const zoic_ray *zoic_tile_rays(const zoic_tile *tile);
zoic_status zoic_tile_set_inputs(zoic_tile *tile, int inputs);
float *zoic_tile_samples(zoic_tile *tile);   /* capacity x 4 floats: the same memory as zoic_tile_inputs */
zoic_status zoic_camera_create_rays_tile(zoic_camera *cam, uint32_t n, const zoic_camera_input *inputs, zoic_camera_output *outputs,
                                         uint64_t ray_index_base, uint16_t tid);
/* The same resident kernel for a GPU consumer's mid-size batch: n <= ZOIC_RESIDENT_MAX_SAMPLES (sx, sy, lensx, lensy) samples in DEVICE memory
 * -> n zoic_ray records in DEVICE memory, what zoic_create_rays_device(cam, n, d_samples, NULL, ray_index_base, d_rays, stream) writes, bit
 * for bit, without a kernel launch: a launch-based call costs 52-76 us whatever it carries; this one 30 us for 4096
 * samples (launch + synchronise: 90) and 68 us for 65 536 (115).  NOT stream-ordered: d_samples must be complete when the call is made (synchronise the stream that produced them), the
 * call returns when d_rays is complete and visible to any kernel launched afterwards.  Pieces of 65536 samples are served one after the
 * other on the mailbox slot of `tid` (several threads, several slots: 1.9 Grays/s from four): beyond a few hundred thousand samples per call
 * zoic_create_rays_device's one launch is the faster call. */
#define ZOIC_RESIDENT_MAX_SAMPLES 1048576u
zoic_status zoic_create_rays_device_resident(zoic_camera *cam, uint32_t n, const float *d_samples, zoic_ray *d_rays, uint64_t ray_index_base,
                                             uint16_t tid);
/* camera_reverse_ray, zoic.cpp:1992-1995: the reference returns false and writes nothing; so does this (returns 0). */
int zoic_camera_reverse_ray(const zoic_camera *cam, const zoic_vec3 *Po, float fov, float *Ps /* [2] */,
                            float *relative_time);

/* Page-locked host memory for the buffers of zoic_create_rays_host: with pinned samples/rays the call runs as a
 * three-stream pipeline (copy-in of piece k+2, trace of piece k+1 and copy-out of piece k at once) at PCIe rate.  zoic_host_register pins
 * memory the caller already owns (keep it registered across calls: registration costs more than one transfer). */
zoic_status zoic_host_alloc(size_t bytes, void **out);
void        zoic_host_free(void *p);
zoic_status zoic_host_register(void *p, size_t bytes);
zoic_status zoic_host_unregister(void *p);

/* synthetic sample generator used by bench/tests: pixel-jittered screen samples, uniform lens samples */
zoic_status zoic_generate_samples_device(zoic_camera *cam, uint64_t n, uint64_t ray_index_base, uint32_t width,
                                         uint32_t height, uint32_t spp, uint32_t seed, float *d_samples, void *stream);

/* ---- one frame over several devices of ONE process ---------------------------------------------
 * The reference is one process whose render threads all call camera_create_ray on one node (zoic.cpp:1752; lifetime
 * :1565-1572, :1723-1749).  A zoic_frame is that node spread over n HIP devices of the calling process: one zoic_camera per
 * device (identical tables: node_update is deterministic), the samples of a call split into contiguous ray-index slabs
 * (zoic_frame_slab: 256-ray tiles, sizes differ by at most one tile), every device renders its slab with
 * zoic_create_rays_device's kernels, and the finished rays are moved to the root device (devices[0]).  Retry streams are
 * keyed by the GLOBAL ray index (ray_index_base + i), so the frame equals the one-device result bit for bit however many
 * devices it is split over.  No collective library is involved: a slab is cut into chunks, chunk k travels to the root with
 * hipMemcpyPeerAsync on the sending device's copy stream (xGMI is point to point: every peer pushes over its own link)
 * while chunk k+1 is traced; chunks alternate between two compute streams so that one's drain runs under the next one's
 * trace.  The root's own slab is ONE launch straight into the output; with one device the call IS
 * zoic_create_rays_device.  The same device may be listed more than once (a 1-GPU box exercises every code path with
 * devices = {0, 0}).
 * Threading: zoic_frame_create / _update / _set_* / _destroy as the camera's node_* methods (one thread, nothing in flight);
 * zoic_frame_render_* from one thread at a time per frame (a frame owns its streams and staging buffers; render threads that
 * want concurrency create one frame each, or call zoic_create_rays_* on zoic_frame_camera(frame, i) directly). */
typedef struct zoic_frame zoic_frame;
typedef enum zoic_frame_layout {
    ZOIC_FRAME_RECORDS = 0, /* n zoic_ray records (32 B/ray, flag word included) */
    ZOIC_FRAME_PAYLOAD = 1, /* n rows of 7 f32: ox oy oz dx dy dz weight (28 B/ray, SURVEY 8e's gather; the flag word stays on
                               the device that traced the ray) */
    ZOIC_FRAME_PAYLOAD_SPARSE = 2, /* the same n rows on the root, but only the rays with weight != 0 are TRANSPORTED: a peer's chunk
                               travels as a 256-bit live mask per 256-ray tile + the compacted rows of its live rays and is expanded
                               on the root.  Rows of weight-0 rays arrive as seven zeros: their origin / direction (the reference's
                               partial state of the last try, zoic.cpp:1951-1961) and their try counts are NOT transported -- they stay
                               in the tracing device's records.  Rows of live rays are bit-identical to ZOIC_FRAME_PAYLOAD's.  A wide-open
                               PETZVAL frame (79 % weight 0) moves 4.5x fewer bytes into the root.  The size of a chunk is only known on
                               the device: zoic_frame_render_device reads one 4-byte count per chunk back before it queues the copy, so
                               with this layout the call returns when the last chunk has been TRACED (copies and expansions may still
                               be in flight behind root_stream); zoic_frame_get_lane_info::bytes_to_root says what moved */
    ZOIC_FRAME_PAYLOAD_AUTO = 3 /* ZOIC_FRAME_PAYLOAD or ZOIC_FRAME_PAYLOAD_SPARSE, chosen from the camera: the first AUTO render after a
                               zoic_frame_update gathers dense; the next one reads the frame's ray counters (one synchronisation, once per
                               update) and from then on the gather is SPARSE iff at least 25 % of the rays rendered since the update had
                               weight 0 -- a wide-open PETZVAL (79 %) yes, the double Gauss with its bokeh image (0.07 %) never: there the
                               sparse layout's count read-back and expansion pass cost more than its headers save.  Rows of live rays are
                               the same bits either way; rows of weight-0 rays follow the layout chosen (dense: the reference's partial
                               state; sparse: zeros).  zoic_frame_auto_layout says what was chosen. */
} zoic_frame_layout;
/* ZOIC_FRAME_PAYLOAD_AUTO's decision for the tables the frame holds: ZOIC_FRAME_PAYLOAD, ZOIC_FRAME_PAYLOAD_SPARSE, or -1 while
 * undecided (no AUTO render since the last update, or only one).  zero_weight_fraction (may be NULL): what it measured, -1 undecided. */
int zoic_frame_auto_layout(const zoic_frame *frame, double *zero_weight_fraction);

/* [begin, end) of device i's slab of an n-sample call over n_devices devices.  Pure arithmetic, callable without a device. */
zoic_status zoic_frame_slab(uint64_t n, int n_devices, int i, uint64_t *begin, uint64_t *end);
/* node_initialize for every device (zoic_camera_create each); devices[0] is the root.  Enables peer access root <-> peers. */
zoic_status zoic_frame_create(const int *devices, int n_devices, zoic_frame **out);
void        zoic_frame_destroy(zoic_frame *frame);          /* node_finish for every device; waits for queued work */
int         zoic_frame_device_count(const zoic_frame *frame);
zoic_camera *zoic_frame_camera(zoic_frame *frame, int i);   /* device i's camera (getters, counters, per-sample calls) */
/* the camera's setters and node_update, applied to every device's camera (first failure is returned) */
zoic_status zoic_frame_set_bokeh_image(zoic_frame *frame, int width, int height, int nchannels, const float *pixels);
zoic_status zoic_frame_set_lens_text(zoic_frame *frame, const char *text, size_t len);
zoic_status zoic_frame_set_precision(zoic_frame *frame, zoic_precision mode);
zoic_status zoic_frame_set_seed(zoic_frame *frame, uint32_t seed);
zoic_status zoic_frame_update(zoic_frame *frame, const zoic_params *p);
/* rays per chunk of a peer's slab (rounded down to 256-ray tiles); 0 = default: a quarter of a slab, at least 64 MB of payload.
 * A slab is never cut into more than 64 chunks: a smaller value is raised to what gives 64. */
zoic_status zoic_frame_set_chunk_rays(zoic_frame *frame, uint64_t rays);
/* camera_create_ray over n samples resident on the devices.
 *   d_samples : n_devices pointers; d_samples[i] = device i's slab, (end - begin) x (sx, sy, lensx, lensy) f32 in device i's
 *               memory, complete before the call.  NULL: the slabs zoic_frame_generate_samples produced for this (n, base).
 *   d_out     : root-device memory for n records / payload rows (16-byte aligned), in global ray order
 *   root_stream : a stream of the root device (NULL = its default stream).  The call is asynchronous; everything it queued is
 *               ordered before whatever is queued on root_stream afterwards, and behind what was queued on it before. */
zoic_status zoic_frame_render_device(zoic_frame *frame, uint64_t n, const float *const *d_samples, uint64_t ray_index_base,
                                     void *d_out, zoic_frame_layout layout, void *root_stream);
/* same without moving anything to the root: every device leaves its slab's records in its own memory (d_rays[i], or the
 * frame's own buffers when d_rays is NULL) -- the "compute only" leg of a scaling measurement */
zoic_status zoic_frame_render_local(zoic_frame *frame, uint64_t n, const float *const *d_samples, uint64_t ray_index_base,
                                    zoic_ray *const *d_rays);
/* host buffers: device i moves its slab in and out over ITS OWN PCIe link (zoic_create_rays_host per device, all at once);
 * nothing crosses xGMI.  Returns when h_rays is complete. */
zoic_status zoic_frame_render_host(zoic_frame *frame, uint64_t n, const float *h_samples, uint64_t ray_index_base, zoic_ray *h_rays);
/* synthetic samples of zoic_generate_samples_device, every device its own slab, into frame-owned buffers */
zoic_status zoic_frame_generate_samples(zoic_frame *frame, uint64_t n, uint64_t ray_index_base, uint32_t width, uint32_t height,
                                        uint32_t spp, uint32_t seed);
zoic_status zoic_frame_synchronize(zoic_frame *frame);      /* every stream of the frame, on every device */
/* What device i's lane did in the last zoic_frame_render_device call, and how it reaches the root: peer_access_* = 1 when
 * hipDeviceCanAccessPeer said yes AND hipDeviceEnablePeerAccess succeeded in that direction (0: hipMemcpyPeerAsync stages through
 * the host -- correct, slow; the root itself and a device listed twice report 1: no link involved). */
typedef struct zoic_frame_lane_info {
    int32_t device, peer_access_to_root, peer_access_from_root;
    uint32_t chunks;              /* sub-launches of the slab */
    uint64_t rays;                /* the slab */
    uint64_t bytes_to_root;       /* bytes the lane's copies moved into the root's memory (0 for the root's own slab) */
} zoic_frame_lane_info;
zoic_status zoic_frame_get_lane_info(const zoic_frame *frame, int i, zoic_frame_lane_info *out);
zoic_status zoic_frame_get_counters(zoic_frame *frame, zoic_counters *sum);   /* summed over the devices */

/* ---- statistics (node_finish prints them, zoic.cpp:1729-1732) ------------------------------ */
zoic_status zoic_camera_get_counters(zoic_camera *cam, zoic_counters *out); /* synchronises the device */
zoic_status zoic_camera_reset_counters(zoic_camera *cam);

/* ---- introspection of the precomputed tables (parity tests) -------------------------------- */
typedef struct zoic_lens_info {
    int32_t lensCount, apertureElement;
    float userApertureRadius, originShift, apertureDistance, focalLengthRatio;
    float tracedFocalLength[2];
    float fov, tan_fov, apertureRadius;            /* thin lens, zoic.cpp:1606-1608 */
    float curvature[ZOIC_MAX_LENS_SURFACES], thickness[ZOIC_MAX_LENS_SURFACES], ior[ZOIC_MAX_LENS_SURFACES],
          aperture[ZOIC_MAX_LENS_SURFACES], center[ZOIC_MAX_LENS_SURFACES];
    int32_t lutSize;
    float lutKey[ZOIC_LUT_ENTRIES];
    float lutMaxX[ZOIC_LUT_ENTRIES], lutMaxY[ZOIC_LUT_ENTRIES], lutMinX[ZOIC_LUT_ENTRIES], lutMinY[ZOIC_LUT_ENTRIES];
    int32_t bokehWidth, bokehHeight;
    int32_t fastRunsStrict; /* 1: this camera is outside the FAST modes' domain (see zoic_precision) and runs STRICT whatever the mode */
    uint32_t precomputeTIR; /* totalInternalReflection bumps of node_update's own traces (zoic.cpp:1135 ff.) that the camera's
                               counters currently include (0 after zoic_camera_reset_counters) */
} zoic_lens_info;
zoic_status zoic_camera_get_info(const zoic_camera *cam, zoic_lens_info *out);
/* copies of the bokeh CDF tables (bokehProbability, zoic.cpp:222-417); arrays sized by bokehWidth/Height */
zoic_status zoic_camera_get_bokeh_tables(const zoic_camera *cam, float *cdfRow, int32_t *rowIndices, float *cdfColumn,
                                         int32_t *columnIndices);

#ifdef __cplusplus
}
#endif
#endif /* ZOIC_AMD_H */
