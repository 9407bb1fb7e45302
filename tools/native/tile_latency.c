/* tile_latency.c -- what a renderer's bucket costs through the resident tile server (zoic_tile_*, csrc/mailbox.hip), as the
 * render threads see it: every thread owns one tile, fills its page-locked input rows, submits, waits, reads a checksum of the rows
 * that came back.
 *   tile_latency <lens.dat> [threads=16] [samples_per_tile=65536] [tiles_per_thread=200] [precision 0|1|2] [lensModel 0|1] [mode]
 * mode: 0 = zoic_tile_submit + zoic_tile_wait on the tile's own arrays (zero copy), 1 = zoic_camera_create_rays_tile on malloc'd
 * arrays (staged through the slot's page-locked buffers), 2 = zoic_create_rays_arnold on malloc'd arrays (launch-based: the call the
 * tile server replaces), 3 = zoic_create_rays_arnold on page-locked arrays, 4 = zoic_create_rays_device_resident on DEVICE buffers (16-byte
 * samples in, 32-byte records out: no PCIe rows, no launch), 5 = zoic_create_rays_device + a stream synchronise on the same device buffers
 * (the launch-based call mode 4 replaces).  [rows] [ins] as zoic_tile_set_rows / _set_inputs (mode 0); [wait] = zoic_camera_set_wait_mode.
 * Prints one JSON line: p50 / p90 / p99 / mean microseconds per tile call, aggregate Mrays/s over the wall clock, and the PCIe
 * floor of a call (112 B per sample at 55 GB/s: 28 B in, 84 B out). */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zoic_amd.h"

/* modes 4 / 5 need device memory, which the C-ABI does not hand out (a GPU consumer owns its buffers): the three HIP entry points
 * they need, declared by hand so that this file stays plain C for gcc (linked with -lamdhip64, which libzoic_amd.so loads anyway) */
extern int hipMalloc(void **p, size_t bytes);
extern int hipFree(void *p);
extern int hipMemcpy(void *dst, const void *src, size_t bytes, int kind);   /* 1 = host to device, 2 = device to host */
extern int hipStreamSynchronize(void *stream);

static zoic_camera *cam;
static int tiles = 200, mode = 0, rows = 0, ins = 0;   /* ins: 1 = 16-byte samples instead of AtCameraInput rows (mode 0 only) */
/* rows: 1 = zoic_ray records instead of AtCameraOutput rows (mode 0 only) */
static uint32_t per_tile = 65536;
static double *lat;   /* threads x tiles */
static pthread_barrier_t go;
static double t_start[256], t_end[256];
static double sums[256];

static double now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static int cmp(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

static int pin_to_gpu_node(void)
{
    const int node = zoic_device_numa_node(0);
    if (node < 0) return -1;
    char path[128], list[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    if (!fgets(list, sizeof list, f)) { fclose(f); return -1; }
    fclose(f);
    cpu_set_t set;
    CPU_ZERO(&set);
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(NULL, ",\n")) {
        int a, b;
        if (sscanf(tok, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b; ++c) CPU_SET(c, &set); }
        else if (sscanf(tok, "%d", &a) == 1) CPU_SET(a, &set);
    }
    return sched_setaffinity(0, sizeof set, &set) == 0 ? node : -1;
}

/* a bucket of a 3840 x 2160 frame: consecutive pixels of a 64-pixel-wide block, every sample jittered */
static void fill(zoic_camera_input *in0, size_t at, uint32_t n, uint32_t *state, int tid, int k)
{
    zoic_camera_input *in = in0 + at;
    float *sm = (float *)in0 + 4 * at;   /* ins: the same memory as 16-byte samples */
    uint32_t s = *state;
    const float bx = (float)((tid * 7 + k * 13) % 60) / 60.0f * 1.9f - 0.95f, by = (float)((tid * 5 + k * 11) % 33) / 33.0f * 1.0f - 0.5f;
    for (uint32_t i = 0; i < n; ++i) {
        float sx, sy, lx, ly;
        s = s * 1664525u + 1013904223u; sx = bx + (float)(s >> 8) / 16777216.0f * (64.0f / 1920.0f);
        s = s * 1664525u + 1013904223u; sy = by + (float)(s >> 8) / 16777216.0f * (64.0f / 1920.0f);
        s = s * 1664525u + 1013904223u; lx = (float)(s >> 8) / 16777216.0f;
        s = s * 1664525u + 1013904223u; ly = (float)(s >> 8) / 16777216.0f;
        if (ins) { sm[4 * i] = sx; sm[4 * i + 1] = sy; sm[4 * i + 2] = lx; sm[4 * i + 3] = ly; }
        else { in[i].sx = sx; in[i].sy = sy; in[i].lensx = lx; in[i].lensy = ly; in[i].dsx = in[i].dsy = in[i].relative_time = 0.0f; }
    }
    *state = s;
}

static void *worker(void *arg)
{
    const int tid = (int)(intptr_t)arg;
    uint32_t s = 12345u + 977u * (uint32_t)tid;
    zoic_tile *tile = NULL;
    zoic_camera_input *in = NULL;
    zoic_camera_output *out = NULL;
    float *d_in = NULL; zoic_ray *d_out = NULL;
    if (mode >= 4) {
        in = malloc(sizeof(*in) * per_tile); out = malloc(sizeof(*out) * per_tile);
        if (hipMalloc((void **)&d_in, 16u * (size_t)per_tile) != 0 || hipMalloc((void **)&d_out, sizeof(zoic_ray) * (size_t)per_tile) != 0) { fprintf(stderr, "hipMalloc failed\n"); exit(2); }
    } else if (mode == 0) {
        if (zoic_tile_create(cam, per_tile, (uint16_t)tid, &tile) != ZOIC_OK) { fprintf(stderr, "tile: %s\n", zoic_last_error_string()); exit(2); }
        if (ins && zoic_tile_set_inputs(tile, ZOIC_TILE_INPUTS_SAMPLES) != ZOIC_OK) { fprintf(stderr, "tile inputs: %s\n", zoic_last_error_string()); exit(2); }
        if (rows && zoic_tile_set_rows(tile, ZOIC_TILE_ROWS_RAYS) != ZOIC_OK) { fprintf(stderr, "tile rows: %s\n", zoic_last_error_string()); exit(2); }
        in = zoic_tile_inputs(tile); out = zoic_tile_outputs(tile);
    } else if (mode == 3) {
        if (zoic_host_alloc(sizeof(*in) * per_tile, (void **)&in) != ZOIC_OK || zoic_host_alloc(sizeof(*out) * per_tile, (void **)&out) != ZOIC_OK) exit(2);
    } else {
        in = malloc(sizeof(*in) * per_tile); out = malloc(sizeof(*out) * per_tile);
    }
    memset(out, 0, sizeof(*out) * per_tile);
    fill(in, 0, per_tile, &s, tid, 0);
    if (mode >= 4 && hipMemcpy(d_in, in, 16u * (size_t)per_tile, 1) != 0) { fprintf(stderr, "hipMemcpy failed\n"); exit(2); }
    double sum = 0.0;
    const int warm = tiles > 20 ? 10 : 2;
    for (int k = -warm; k < tiles; ++k) {
        if (k == 0) { pthread_barrier_wait(&go); t_start[tid] = now_us(); }
        /* a fresh quarter of the inputs per tile (a renderer writes all of them; the generator here is slower than the GPU) */
        if (mode < 4) fill(in, (size_t)((k + warm) & 3) * (per_tile / 4), per_tile / 4, &s, tid, k);   /* (device buffers: a GPU producer made them) */
        const uint64_t base = ((uint64_t)tid << 40) + (uint64_t)(k + warm) * per_tile;
        const double t0 = now_us();
        zoic_status st;
        if (mode == 0) { st = zoic_tile_submit(tile, per_tile, base); if (st == ZOIC_OK) st = zoic_tile_wait(tile); }
        else if (mode == 1) st = zoic_camera_create_rays_tile(cam, per_tile, in, out, base, (uint16_t)tid);
        else if (mode == 4) st = zoic_create_rays_device_resident(cam, per_tile, d_in, d_out, base, (uint16_t)tid);
        else if (mode == 5) { st = zoic_create_rays_device(cam, per_tile, d_in, NULL, base, d_out, NULL); if (st == ZOIC_OK && hipStreamSynchronize(NULL) != 0) st = ZOIC_ERR_HIP; }
        else st = zoic_create_rays_arnold(cam, per_tile, in, out, base);
        if (st != ZOIC_OK) { fprintf(stderr, "tile call: %s\n", zoic_last_error_string()); exit(2); }
        if (k >= 0) lat[(size_t)tid * tiles + k] = now_us() - t0;
        if (mode >= 4) { if (k == tiles - 1) { zoic_ray r; hipMemcpy(&r, d_out + per_tile - 1, sizeof r, 2); sum += r.dz + r.weight; } }
        else if (rows) { const zoic_ray *r = (const zoic_ray *)out; sum += r[(size_t)(k + warm) % per_tile].dz + r[per_tile - 1].weight; }
        else sum += out[(size_t)(k + warm) % per_tile].dir.z + out[per_tile - 1].weight[0];
    }
    t_end[tid] = now_us();
    sums[tid] = sum;
    if (mode >= 4) { hipFree(d_in); hipFree(d_out); free(in); free(out); }
    else if (tile) zoic_tile_destroy(tile);
    else if (mode == 3) { zoic_host_free(in); zoic_host_free(out); }
    else { free(in); free(out); }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: tile_latency lens.dat [threads] [samples_per_tile] [tiles_per_thread] [precision] [lensModel] [mode] [rows] [ins] [wait]\n"); return 1; }
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    if (argc > 3) per_tile = (uint32_t)atoi(argv[3]);
    if (argc > 4) tiles = atoi(argv[4]);
    const int precision = argc > 5 ? atoi(argv[5]) : ZOIC_PRECISION_FAST, model = argc > 6 ? atoi(argv[6]) : ZOIC_RAYTRACED;
    if (argc > 7) mode = atoi(argv[7]);
    if (argc > 8) rows = atoi(argv[8]);
    if (argc > 9) ins = atoi(argv[9]);
    const int waitMode = argc > 10 ? atoi(argv[10]) : 0;
    if (mode >= 4) ins = 1;   /* fill() writes 16-byte samples */
    if (threads < 1 || threads > 256 || per_tile < 4 || (mode <= 0 && per_tile > ZOIC_TILE_MAX_SAMPLES) || (mode == 4 && per_tile > ZOIC_RESIDENT_MAX_SAMPLES) || tiles < 1) { fprintf(stderr, "bad arguments\n"); return 1; }
    zoic_params p;
    zoic_params_default(&p);
    p.lensDataPath = argv[1]; p.lensModel = model; p.focalLength = 5.0f; p.fStop = 2.0f;
    if (zoic_camera_create(0, &cam) != ZOIC_OK || zoic_camera_update(cam, &p) != ZOIC_OK ||
        zoic_camera_set_precision(cam, (zoic_precision)precision) != ZOIC_OK) { fprintf(stderr, "camera: %s\n", zoic_last_error_string()); return 2; }
    if (zoic_camera_set_wait_mode(cam, (zoic_wait_mode)waitMode) != ZOIC_OK) { fprintf(stderr, "wait mode: %s\n", zoic_last_error_string()); return 2; }
    const int node = pin_to_gpu_node();   /* inherited by the worker threads */
    lat = malloc(sizeof(double) * (size_t)threads * tiles);
    pthread_barrier_init(&go, NULL, (unsigned)threads);
    pthread_t th[256];
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    double first = t_start[0], last = t_end[0], check = 0.0;
    for (int t = 0; t < threads; ++t) { if (t_start[t] < first) first = t_start[t]; if (t_end[t] > last) last = t_end[t]; check += sums[t]; }
    const size_t n = (size_t)threads * tiles;
    double mean = 0;
    for (size_t i = 0; i < n; ++i) mean += lat[i];
    qsort(lat, n, sizeof(double), cmp);
    zoic_counters c;
    zoic_camera_get_counters(cam, &c);
    const double floor_us = mode >= 4 ? 0.0 : 112.0 * per_tile / 55e9 * 1e6;
    printf("{\"mode\": %d, \"wait_mode\": %d, \"threads\": %d, \"samples_per_tile\": %u, \"tiles_per_thread\": %d, \"precision\": %d, \"lensModel\": %d, \"p50_us\": %.2f, "
           "\"p90_us\": %.2f, \"p99_us\": %.2f, \"mean_us\": %.2f, \"mrays_s\": %.1f, \"pcie_floor_us\": %.2f, \"rays_counted\": %llu, \"checksum\": %.6g, \"numa_node\": %d}\n",
           mode, waitMode, threads, per_tile, tiles, precision, model, lat[n / 2], lat[n * 9 / 10], lat[n * 99 / 100], mean / n,
           (double)n * per_tile / (last - first), floor_us, (unsigned long long)(c.succesRays + c.vignettedRays), check, node);
    zoic_camera_destroy(cam);
    return 0;
}
