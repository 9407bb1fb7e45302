// tests/native/tile_buffer_test.cpp -- arnold/zoic_tile_buffer.hpp (accumulate -> flush -> serve) on a GPU box, from plain C++ against the
// C-ABI: a bucket's rows equal zoic_create_rays_arnold's bit for bit, serve() updates a caller's AtCameraOutput the way
// zoic_camera_create_ray does, several render threads with a buffer each.  Prints "tile_buffer_test OK" and exits 0.
//   tile_buffer_test <lens.dat> [precision 0|1] [ray records 0|1] [16-byte samples 0|1]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../arnold/zoic_tile_buffer.hpp"

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

static int check(bool ok, const char *what) { if (!ok) { std::fprintf(stderr, "FAILED: %s (%s)\n", what, zoic_last_error_string()); std::exit(1); } return 0; }

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: tile_buffer_test lens.dat [precision]\n"); return 2; }
    zoic_camera *cam = nullptr;
    zoic_params p;
    zoic_params_default(&p);
    p.lensDataPath = argv[1]; p.focalLength = 10.0f; p.fStop = 2.8f; p.exposureControl = 0.5f;
    check(zoic_camera_create(0, &cam) == ZOIC_OK && zoic_camera_update(cam, &p) == ZOIC_OK, "camera");
    check(zoic_camera_set_precision(cam, argc > 2 ? static_cast<zoic_precision>(std::atoi(argv[2])) : ZOIC_PRECISION_STRICT) == ZOIC_OK, "precision");
    const bool rayRecords = argc > 3 && std::atoi(argv[3]) != 0;   // the buffers are answered with zoic_ray records (zoic_tile_set_rows)
    const bool samples16 = argc > 4 && std::atoi(argv[4]) != 0;    // the buffers are filled with (sx, sy, lensx, lensy) samples (zoic_tile_set_inputs)
    const uint32_t n = 64 * 64 * 4;
    const int threads = 6;
    std::vector<int> bad(threads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([&, t] {
        ZoicTileBuffer tile(cam, n, static_cast<uint16_t>(t), rayRecords, samples16);
        if (tile.sample_inputs() != samples16 || tile.ray_records() != rayRecords) { bad[t] = 7; return; }
        std::vector<zoic_camera_input> in(n);
        std::vector<zoic_camera_output> ref(n);
        uint32_t s = 99u + 31u * t;
        for (int bucket = 0; bucket < 5; ++bucket) {
            tile.clear();
            for (uint32_t i = 0; i < n; ++i) {
                const float sx = lcg(s) / 16777216.0f * 2.0f - 1.0f, sy = lcg(s) / 16777216.0f * 1.2f - 0.6f;
                const float lx = lcg(s) / 16777216.0f, ly = lcg(s) / 16777216.0f;
                const uint32_t at = tile.push(sx, sy, lx, ly);
                std::memset(&in[at], 0, sizeof in[at]);
                in[at].sx = sx; in[at].sy = sy; in[at].lensx = lx; in[at].lensy = ly;
            }
            // one sample too many: refused, nothing stored (the bucket's last sample and its first output row stay what they were)
            if (!tile.full() || tile.push(9.0f, 9.0f, 0.9f, 0.9f) != ZoicTileBuffer::kFull || tile.size() != n) { bad[t] = 6; return; }
            const uint64_t base = (static_cast<uint64_t>(t) << 32) + static_cast<uint64_t>(bucket) * n;
            if (tile.flush(base) != ZOIC_OK || tile.wait() != ZOIC_OK || !tile.done()) { bad[t] = 1; return; }
            if (zoic_create_rays_arnold(cam, n, in.data(), ref.data(), base) != ZOIC_OK) { bad[t] = 2; return; }
            uint32_t retried = 0, dead = 0;
            for (uint32_t i = 0; i < n; ++i) {
                if (!rayRecords && std::memcmp(&tile.row(i), &ref[i], sizeof(zoic_camera_output)) != 0) { bad[t] = 3; return; }
                if (rayRecords && (std::memcmp(&tile.ray(i).ox, &ref[i].origin, 12) != 0 || std::memcmp(&tile.ray(i).dx, &ref[i].dir, 12) != 0)) { bad[t] = 3; return; }
                // serve(): what camera_create_ray does to the caller's output (weight 0.25 handed in, derivatives pre-set)
                zoic_camera_output o;
                std::memset(&o, 0, sizeof o);
                o.weight[0] = o.weight[1] = o.weight[2] = 0.25f;
                o.dOdx.x = 5.0f; o.dOdy.y = 6.0f; o.dDdy.z = 7.0f;
                tile.serve(i, o);
                const zoic_camera_output &r = ref[i];
                const bool wasRetried = std::memcmp(&r.dOdy, &r.origin, 12) == 0 && std::memcmp(&r.dDdy, &r.dir, 12) == 0;
                retried += wasRetried; dead += r.weight[0] == 0.0f;
                bool ok = std::memcmp(&o.origin, &r.origin, 12) == 0 && std::memcmp(&o.dir, &r.dir, 12) == 0 && o.dOdx.x == 5.0f;
                ok = ok && o.weight[0] == (r.weight[0] == 0.0f ? 0.0f : 0.25f * r.weight[0]);
                if (wasRetried) ok = ok && std::memcmp(&o.dOdy, &r.origin, 12) == 0 && std::memcmp(&o.dDdy, &r.dir, 12) == 0;
                else ok = ok && o.dOdy.y == 6.0f && o.dDdy.z == 7.0f;
                if (!ok) { bad[t] = 4; return; }
            }
            if (retried == 0 || dead == 0) { bad[t] = 5; return; }   // the TESSAR at 10 cm has both in every bucket of this size
        }
    });
    for (std::thread &x : th) x.join();
    for (int t = 0; t < threads; ++t) if (bad[t]) { std::fprintf(stderr, "thread %d failed at check %d\n", t, bad[t]); return 1; }
    zoic_counters c;
    check(zoic_camera_get_counters(cam, &c) == ZOIC_OK, "counters");
    check(c.succesRays + c.vignettedRays == 2ull * threads * 5 * n, "every ray counted once per call (tile + arnold)");
    zoic_camera_destroy(cam);
    std::printf("tile_buffer_test OK\n");
    return 0;
}
