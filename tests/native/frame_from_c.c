/* tests/native/frame_from_c.c -- the zoic_frame_* entry points called from plain C (gcc, no HIP headers, no C++): what a
 * plug-in written in the reference's own language links against.  Renders one TESSAR frame of 1 M samples over
 * devices = {0, 0}, twice (host buffers), and compares the two results and a one-device camera's byte for byte.
 *   gcc -O2 -Iinclude tests/native/frame_from_c.c -o frame_from_c -Lzoic_amd -lzoic_amd -Wl,-rpath,$PWD/zoic_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zoic_amd.h"

#define CHECK(call) do { zoic_status s_ = (call); if (s_ != ZOIC_OK) { fprintf(stderr, "%s -> %s: %s\n", #call, zoic_status_string(s_), zoic_last_error_string()); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: frame_from_c lens.dat\n"); return 2; }
    if (zoic_abi_version() != ZOIC_AMD_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    const uint64_t n = 1000003;
    uint64_t a, b, c, d;
    CHECK(zoic_frame_slab(n, 2, 0, &a, &b));
    CHECK(zoic_frame_slab(n, 2, 1, &c, &d));
    if (a != 0 || b != c || d != n || b % 256) { fprintf(stderr, "bad partition\n"); return 1; }
    float *samples = (float *)malloc(n * 16);
    zoic_ray *r1 = (zoic_ray *)malloc(n * sizeof(zoic_ray)), *r2 = (zoic_ray *)malloc(n * sizeof(zoic_ray)), *r3 = (zoic_ray *)malloc(n * sizeof(zoic_ray));
    uint32_t x = 12345u;
    for (uint64_t i = 0; i < n * 4; ++i) {   /* sx, sy in (-1, 1) x (-0.56, 0.56), lens samples in [0, 1) */
        x = x * 1664525u + 1013904223u;
        const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
        samples[i] = (i & 3) == 0 ? 2.0f * u - 1.0f : (i & 3) == 1 ? (2.0f * u - 1.0f) * 0.5625f : u;
    }
    zoic_params p;
    zoic_params_default(&p);
    p.lensDataPath = argv[1]; p.focalLength = 10.0f; p.fStop = 2.8f;
    const int devices[2] = {0, 0};
    zoic_frame *frame = NULL;
    CHECK(zoic_frame_create(devices, 2, &frame));
    CHECK(zoic_frame_set_precision(frame, ZOIC_PRECISION_FAST));
    CHECK(zoic_frame_update(frame, &p));
    CHECK(zoic_frame_render_host(frame, n, samples, 7000, r1));
    CHECK(zoic_frame_render_host(frame, n, samples, 7000, r2));
    zoic_camera *cam = NULL;
    CHECK(zoic_camera_create(0, &cam));
    CHECK(zoic_camera_set_precision(cam, ZOIC_PRECISION_FAST));
    CHECK(zoic_camera_update(cam, &p));
    CHECK(zoic_create_rays_host(cam, n, samples, NULL, 7000, r3));
    zoic_counters fc, cc;
    CHECK(zoic_frame_get_counters(frame, &fc));
    CHECK(zoic_camera_get_counters(cam, &cc));
    const int same = memcmp(r1, r2, n * sizeof(zoic_ray)) == 0 && memcmp(r1, r3, n * sizeof(zoic_ray)) == 0;
    const int counted = fc.succesRays == 2 * cc.succesRays && fc.vignettedRays == 2 * cc.vignettedRays && fc.succesRays + fc.vignettedRays == 2 * n;
    printf("{\"rays\": %llu, \"identical\": %d, \"counters_ok\": %d, \"succes\": %llu}\n", (unsigned long long)n, same, counted, (unsigned long long)cc.succesRays);
    zoic_camera_destroy(cam);
    zoic_frame_destroy(frame);
    free(samples); free(r1); free(r2); free(r3);
    return same && counted ? 0 : 1;
}
