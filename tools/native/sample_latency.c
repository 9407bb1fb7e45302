/* sample_latency.c -- latency of zoic_camera_create_ray (the per-sample signature of zoic.cpp:1752) as a render thread sees it.
 *   sample_latency <lens.dat> [threads=1] [calls=200000] [precision 0|1|2] [lensModel 0|1]
 * prints one JSON line: median / p90 / p99 / mean microseconds per call (per thread) and the aggregate call rate. */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zoic_amd.h"

static zoic_camera *cam;
static int calls = 200000;
static double *lat;   /* threads x calls */

static double now_us(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}

static int cmp(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

/* run on the CPUs of the NUMA node the GPU hangs off (worth ~1 us per call); returns the node or -1 */
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

static void *worker(void *arg)
{
    const int tid = (int)(intptr_t)arg;
    uint32_t s = 12345u + 977u * (uint32_t)tid;
    zoic_camera_input in;
    zoic_camera_output out;
    memset(&in, 0, sizeof in);
    for (int i = -2000; i < calls; ++i) {   /* 2000 warm-up calls */
        s = s * 1664525u + 1013904223u; in.sx = (float)(s >> 8) / 16777216.0f * 1.6f - 0.8f;
        s = s * 1664525u + 1013904223u; in.sy = (float)(s >> 8) / 16777216.0f * 0.9f - 0.45f;
        s = s * 1664525u + 1013904223u; in.lensx = (float)(s >> 8) / 16777216.0f;
        s = s * 1664525u + 1013904223u; in.lensy = (float)(s >> 8) / 16777216.0f;
        memset(&out, 0, sizeof out);
        out.weight[0] = out.weight[1] = out.weight[2] = 1.0f;
        const double t0 = now_us();
        if (zoic_camera_create_ray(cam, &in, &out, (uint16_t)tid) != ZOIC_OK) { fprintf(stderr, "create_ray: %s\n", zoic_last_error_string()); exit(2); }
        if (i >= 0) lat[(size_t)tid * calls + i] = now_us() - t0;
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: sample_latency lens.dat [threads] [calls] [precision] [lensModel]\n"); return 1; }
    const int threads = argc > 2 ? atoi(argv[2]) : 1;
    if (argc > 3) calls = atoi(argv[3]);
    const int precision = argc > 4 ? atoi(argv[4]) : ZOIC_PRECISION_FAST, model = argc > 5 ? atoi(argv[5]) : ZOIC_RAYTRACED;
    zoic_params p;
    zoic_params_default(&p);
    p.lensDataPath = argv[1]; p.lensModel = model; p.focalLength = 5.0f; p.fStop = 2.8f;
    if (zoic_camera_create(0, &cam) != ZOIC_OK || zoic_camera_update(cam, &p) != ZOIC_OK ||
        zoic_camera_set_precision(cam, (zoic_precision)precision) != ZOIC_OK) { fprintf(stderr, "camera: %s\n", zoic_last_error_string()); return 2; }
    const int node = pin_to_gpu_node();   /* inherited by the worker threads */
    lat = malloc(sizeof(double) * (size_t)threads * calls);
    pthread_t th[256];
    const double t0 = now_us();
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    const double wall = now_us() - t0;
    const size_t n = (size_t)threads * calls;
    double mean = 0;
    for (size_t i = 0; i < n; ++i) mean += lat[i];
    qsort(lat, n, sizeof(double), cmp);
    zoic_counters c;
    zoic_camera_get_counters(cam, &c);
    printf("{\"threads\": %d, \"calls_per_thread\": %d, \"precision\": %d, \"lensModel\": %d, \"median_us\": %.2f, \"p90_us\": %.2f, \"p99_us\": %.2f, "
           "\"mean_us\": %.2f, \"calls_per_s\": %.0f, \"rays_counted\": %llu, \"numa_node\": %d}\n", threads, calls, precision, model, lat[n / 2], lat[n * 9 / 10],
           lat[n * 99 / 100], mean / n, (double)(n + 2000.0 * threads) / wall * 1e6, (unsigned long long)(c.succesRays + c.vignettedRays), node);
    zoic_camera_destroy(cam);
    return 0;
}
