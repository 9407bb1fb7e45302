// boundary_stress.cpp -- native driver for the sanitizer builds of libzoic_amd.so (tests/test_sanitizers.py).
//
// Exercises the host C++ of the product (capi.cpp, lens_system.cpp) through the C-ABI only, the way a renderer would:
//   part 1 (always, no GPU needed): several threads each own a tables-only camera (ZOIC_DEVICE_NONE) and run node_update's
//           whole host precompute -- parse, focus, exit-pupil LUT, bokeh CDF -- plus the error paths;
//   part 3 (when a HIP device is visible): zoic_frame_* over devices {0, 0, 0} -- update threads, host-render threads, chunked
//          device render -- against one camera's result
//   part 2 (when a HIP device is visible): 16 render threads hammer ONE camera with camera_create_ray / tile / Arnold-layout /
//           host-buffer calls, the contract of zoic.cpp:1752; the results are checked against a serial replay.
// Exit code 0 = clean; the sanitizer runtime turns any report into a non-zero exit.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "zoic_amd.h"

// a 7-surface triplet in the tabular format (radius thickness ior abbe aperture-diameter, mm; radius 0 = stop)
static const char *kTriplet =
    "# radius thickness ior abbe aperture\n"
    "42.200\t2.070\t1.649\t53.3\t20.0\n"
    "-283.530\t3.000\t0.0\t0.0\t20.0\n"
    "-84.930\t9.170\t1.673\t32.2\t15.6\n"
    "33.230\t7.67\t0.0\t0.0\t13.5\n"
    "0.0\t7.00\t0.0\t0.0\t13.6\n"
    "84.930\t6.00\t1.694\t53.3\t14.0\n"
    "-84.930\t65.475\t0.0\t0.0\t14.0\n";

static std::atomic<int> g_failures{0};
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, zoic_last_error_string()); ++g_failures; } \
    } while (0)

static std::vector<float> make_image(int w, int h, unsigned seed)
{
    std::vector<float> px(static_cast<size_t>(w) * h * 3);
    unsigned s = seed;
    for (float &v : px) { s = s * 1664525u + 1013904223u; v = static_cast<float>(s >> 8) * (1.0f / 16777216.0f); }
    return px;
}

static void tables_only_worker(int t)
{
    zoic_camera *cam = nullptr;
    CHECK(zoic_camera_create(ZOIC_DEVICE_NONE, &cam) == ZOIC_OK);
    zoic_params p;
    zoic_params_default(&p);
    p.lensDataPath = "mem:triplet";
    p.focalLength = 5.0f + 0.25f * static_cast<float>(t);
    p.fStop = 2.8f;
    // error paths first: bad text, two stops, missing image
    CHECK(zoic_camera_set_lens_text(cam, "1 2 3\n4 5 6\n", 12) == ZOIC_OK);   // 3 columns
    CHECK(zoic_camera_update(cam, &p) == ZOIC_ERR_LENS_COLUMNS);
    const std::string twoStops = std::string(kTriplet) + "0 1 1 50 10\n";
    CHECK(zoic_camera_set_lens_text(cam, twoStops.c_str(), twoStops.size()) == ZOIC_OK);
    CHECK(zoic_camera_update(cam, &p) == ZOIC_ERR_MULTI_APERTURE);
    CHECK(zoic_camera_set_lens_text(cam, kTriplet, std::strlen(kTriplet)) == ZOIC_OK);
    p.useImage = 1; p.bokehPath = "mem:none";
    CHECK(zoic_camera_update(cam, &p) == ZOIC_ERR_BOKEH_IMAGE);
    CHECK(zoic_camera_update(cam, &p) == ZOIC_ERR_BOKEH_IMAGE);   // a failure is not forgotten
    const std::vector<float> img = make_image(48 + t, 32 + t, 7u + static_cast<unsigned>(t));
    CHECK(zoic_camera_set_bokeh_image(cam, 48 + t, 32 + t, 3, img.data()) == ZOIC_OK);
    CHECK(zoic_camera_update(cam, &p) == ZOIC_OK);                // parse + focus + 3.2 M-probe LUT on the host + CDFs
    zoic_lens_info info;
    CHECK(zoic_camera_get_info(cam, &info) == ZOIC_OK);
    CHECK(info.lensCount == 7 && info.apertureElement == 2 && info.lutSize == ZOIC_LUT_ENTRIES);
    CHECK(info.bokehWidth == 48 + t && info.bokehHeight == 32 + t);
    std::vector<float> cdfRow(info.bokehHeight), cdfCol(static_cast<size_t>(info.bokehWidth) * info.bokehHeight);
    std::vector<int32_t> ri(cdfRow.size()), ci(cdfCol.size());
    CHECK(zoic_camera_get_bokeh_tables(cam, cdfRow.data(), ri.data(), cdfCol.data(), ci.data()) == ZOIC_OK);
    CHECK(std::fabs(cdfRow.back() - 1.0f) < 1e-3f);
    // a tables-only camera can never make a ray
    zoic_camera_input in{0.1f, 0.1f, 0, 0, 0.5f, 0.25f, 0};
    zoic_camera_output out{};
    CHECK(zoic_camera_create_ray(cam, &in, &out, static_cast<uint16_t>(t)) == ZOIC_ERR_NO_DEVICE);
    float ps[2] = {0, 0}, rt = 0;
    zoic_vec3 po{1, 2, 3};
    CHECK(zoic_camera_reverse_ray(cam, &po, 0.5f, ps, &rt) == 0);
    zoic_camera_destroy(cam);
}

struct ThreadLog { std::vector<float> values; };

static void render_worker(zoic_camera *cam, int t, int calls, ThreadLog *log)
{
    unsigned s = 12345u + 977u * static_cast<unsigned>(t);
    auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return static_cast<float>(s >> 8) * (1.0f / 16777216.0f); };
    // a tile of this render thread's own (zoic_tile_*: page-locked rows the resident kernel reads and writes in place)
    zoic_tile *tile = nullptr;
    CHECK(zoic_tile_create(cam, 2048, static_cast<uint16_t>(t), &tile) == ZOIC_OK);
    for (int i = 0; i < calls; ++i) {
        const float kind = rnd();
        if (kind < 0.06f && tile) {
            // a bucket through the tile server: sometimes the tile's own arrays (submit, a per-sample call on the same slot in between,
            // wait), sometimes the one-call form on pageable arrays (staged)
            const uint32_t m = 1u + static_cast<uint32_t>(rnd() * 1023.0f);
            const uint64_t base = 9000000ull * static_cast<unsigned>(t) + static_cast<unsigned>(i);
            if (rnd() < 0.5f) {
                zoic_camera_input *in = zoic_tile_inputs(tile);
                for (uint32_t k = 0; k < m; ++k) in[k] = zoic_camera_input{2.0f * rnd() - 1.0f, (2.0f * rnd() - 1.0f) * 0.5625f, 0, 0, rnd(), rnd(), 0};
                CHECK(zoic_tile_submit(tile, m, base) == ZOIC_OK);
                (void)zoic_tile_done(tile);
                CHECK(zoic_tile_wait(tile) == ZOIC_OK);
                const zoic_camera_output *out = zoic_tile_outputs(tile);
                for (uint32_t k = 0; k < m; k += 61) log->values.insert(log->values.end(), {out[k].dir.x, out[k].dir.z, out[k].weight[1], out[k].dOdy.y});
            } else {
                std::vector<zoic_camera_input> in(m);
                std::vector<zoic_camera_output> out(m);
                for (uint32_t k = 0; k < m; ++k) in[k] = zoic_camera_input{2.0f * rnd() - 1.0f, (2.0f * rnd() - 1.0f) * 0.5625f, 0, 0, rnd(), rnd(), 0};
                CHECK(zoic_camera_create_rays_tile(cam, m, in.data(), out.data(), base, static_cast<uint16_t>(t)) == ZOIC_OK);
                for (uint32_t k = 0; k < m; k += 61) log->values.insert(log->values.end(), {out[k].dir.x, out[k].dir.z, out[k].weight[1], out[k].dOdy.y});
            }
        } else if (kind < 0.7f) {
            zoic_camera_input in{2.0f * rnd() - 1.0f, (2.0f * rnd() - 1.0f) * 0.5625f, 0, 0, rnd(), rnd(), 0};
            zoic_camera_output out{};
            out.weight[0] = out.weight[1] = out.weight[2] = 1.0f;
            CHECK(zoic_camera_create_ray(cam, &in, &out, static_cast<uint16_t>(t)) == ZOIC_OK);
            log->values.insert(log->values.end(), {out.origin.x, out.dir.x, out.dir.y, out.dir.z, out.weight[0], out.dDdy.x});
        } else if (kind < 0.9f) {
            const size_t m = 1 + static_cast<size_t>(rnd() * 2000.0f);
            std::vector<zoic_camera_input> in(m);
            std::vector<zoic_camera_output> out(m);
            for (size_t k = 0; k < m; ++k) {
                in[k] = zoic_camera_input{2.0f * rnd() - 1.0f, (2.0f * rnd() - 1.0f) * 0.5625f, 0, 0, rnd(), rnd(), 0};
                std::memset(&out[k], 0, sizeof(out[k]));
                out[k].weight[0] = out[k].weight[1] = out[k].weight[2] = 1.0f;
            }
            CHECK(zoic_create_rays_arnold(cam, m, in.data(), out.data(), 1000000ull * static_cast<unsigned>(t) + static_cast<unsigned>(i)) == ZOIC_OK);
            for (size_t k = 0; k < m; k += 97) log->values.insert(log->values.end(), {out[k].dir.x, out[k].dir.z, out[k].weight[1]});
        } else {
            const size_t m = 1 + static_cast<size_t>(rnd() * 30000.0f);
            std::vector<float> smp(m * 4);
            for (size_t k = 0; k < m; ++k) { smp[4 * k] = 2.0f * rnd() - 1.0f; smp[4 * k + 1] = (2.0f * rnd() - 1.0f) * 0.5625f; smp[4 * k + 2] = rnd(); smp[4 * k + 3] = rnd(); }
            std::vector<zoic_ray> rays(m);
            CHECK(zoic_create_rays_host(cam, m, smp.data(), nullptr, 5000000ull * static_cast<unsigned>(t) + static_cast<unsigned>(i), rays.data()) == ZOIC_OK);
            for (size_t k = 0; k < m; k += 211) log->values.insert(log->values.end(), {rays[k].dx, rays[k].oy, rays[k].weight, static_cast<float>(rays[k].flags)});
        }
    }
    zoic_tile_destroy(tile);
}

static zoic_camera *render_camera()
{
    zoic_camera *cam = nullptr;
    if (zoic_camera_create(0, &cam) != ZOIC_OK) return nullptr;
    zoic_params p;
    zoic_params_default(&p);
    p.lensDataPath = "mem:triplet"; p.focalLength = 5.0f; p.fStop = 2.5f;
    CHECK(zoic_camera_set_lens_text(cam, kTriplet, std::strlen(kTriplet)) == ZOIC_OK);
    CHECK(zoic_camera_update(cam, &p) == ZOIC_OK);
    return cam;
}

int main(int argc, char **argv)
{
    const int hostThreads = argc > 1 ? std::atoi(argv[1]) : 4;
    const int renderThreads = argc > 2 ? std::atoi(argv[2]) : 16;
    const int calls = argc > 3 ? std::atoi(argv[3]) : 300;
    {
        std::vector<std::thread> th;
        for (int t = 0; t < hostThreads; ++t) th.emplace_back(tables_only_worker, t);
        for (auto &x : th) x.join();
    }
    std::printf("part 1: %d tables-only cameras updated concurrently, failures %d\n", hostThreads, g_failures.load());
    if (zoic_device_count() > 0 && renderThreads > 0) {
        std::vector<ThreadLog> serial(renderThreads), parallel(renderThreads);
        zoic_camera *cam = render_camera();
        CHECK(cam != nullptr);
        if (cam) {
            for (int t = 0; t < renderThreads; ++t) render_worker(cam, t, calls, &serial[t]);
            zoic_counters c1{}; CHECK(zoic_camera_get_counters(cam, &c1) == ZOIC_OK);
            zoic_camera_destroy(cam);
            cam = render_camera();
            std::vector<std::thread> th;
            for (int t = 0; t < renderThreads; ++t) th.emplace_back(render_worker, cam, t, calls, &parallel[t]);
            for (auto &x : th) x.join();
            zoic_counters c2{}; CHECK(zoic_camera_get_counters(cam, &c2) == ZOIC_OK);
            CHECK(c1.succesRays == c2.succesRays && c1.vignettedRays == c2.vignettedRays && c1.totalInternalReflection == c2.totalInternalReflection);
            for (int t = 0; t < renderThreads; ++t) {
                const bool same = serial[t].values.size() == parallel[t].values.size() &&
                                  std::memcmp(serial[t].values.data(), parallel[t].values.data(), serial[t].values.size() * sizeof(float)) == 0;
                if (!same) { std::fprintf(stderr, "thread %d: parallel run differs from its serial replay\n", t); ++g_failures; }
            }
            zoic_camera_destroy(cam);
            std::printf("part 2: %d render threads x %d calls on one camera, failures %d\n", renderThreads, calls, g_failures.load());
        }
    } else {
        std::printf("part 2 skipped: no HIP device\n");
    }
    // part 3: one camera node over three "devices" of this process (zoic_frame_*, csrc/frame.cpp): the frame's own update
    // threads, its per-device host-render threads and the event-chained device render, compared with one camera's result
    if (zoic_device_count() > 0 && renderThreads > 0) {
        const int devices[3] = {0, 0, 0};
        zoic_frame *frame = nullptr;
        CHECK(zoic_frame_create(devices, 3, &frame) == ZOIC_OK);
        zoic_camera *cam = render_camera();
        CHECK(cam != nullptr);
        if (frame && cam) {
            zoic_params p;
            zoic_params_default(&p);
            p.lensDataPath = "mem:triplet"; p.focalLength = 5.0f; p.fStop = 2.5f;
            CHECK(zoic_frame_set_lens_text(frame, kTriplet, std::strlen(kTriplet)) == ZOIC_OK);
            CHECK(zoic_frame_update(frame, &p) == ZOIC_OK);
            CHECK(zoic_frame_set_chunk_rays(frame, 4096) == ZOIC_OK);
            const size_t m = 100003;
            unsigned s = 99u;
            auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return static_cast<float>(s >> 8) * (1.0f / 16777216.0f); };
            std::vector<float> smp(m * 4);
            for (size_t k = 0; k < m; ++k) { smp[4 * k] = 2.0f * rnd() - 1.0f; smp[4 * k + 1] = (2.0f * rnd() - 1.0f) * 0.5625f; smp[4 * k + 2] = rnd(); smp[4 * k + 3] = rnd(); }
            std::vector<zoic_ray> one(m), many(m);
            CHECK(zoic_create_rays_host(cam, m, smp.data(), nullptr, 424242, one.data()) == ZOIC_OK);
            for (int rep = 0; rep < 3; ++rep) {
                std::memset(many.data(), 0, m * sizeof(zoic_ray));
                CHECK(zoic_frame_render_host(frame, m, smp.data(), 424242, many.data()) == ZOIC_OK);
                CHECK(std::memcmp(one.data(), many.data(), m * sizeof(zoic_ray)) == 0);
            }
            CHECK(zoic_frame_generate_samples(frame, m, 0, 640, 360, 4, 1) == ZOIC_OK);
            CHECK(zoic_frame_render_local(frame, m, nullptr, 0, nullptr) == ZOIC_OK);
            CHECK(zoic_frame_synchronize(frame) == ZOIC_OK);
            zoic_counters fc{};
            CHECK(zoic_frame_get_counters(frame, &fc) == ZOIC_OK);
            CHECK(fc.succesRays + fc.vignettedRays == 4 * m);
        }
        zoic_camera_destroy(cam);
        zoic_frame_destroy(frame);
        std::printf("part 3: one frame over 3 lanes of device 0, failures %d\n", g_failures.load());
    }
    // Leave without running the ROCm runtime's exit-time destructors: under ROCm's AddressSanitizer runtime they trip an
    // internal CHECK of its device allocator (sanitizer_allocator_device.h, "dev_runtime_unloaded_") inside
    // libhsa-runtime64's __cxa_finalize handlers -- after main, outside the product.  Every report about the product's
    // own code has been made by now; the sanitizer exit codes still propagate from reports raised earlier.
    std::fflush(stdout); std::fflush(stderr);
    _Exit(g_failures.load() ? 1 : 0);
}
