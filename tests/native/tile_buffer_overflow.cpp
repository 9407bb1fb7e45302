// tests/native/tile_buffer_overflow.cpp -- arnold/zoic_tile_buffer.hpp against a MOCK of the zoic_tile_* entry points (plain malloc'd
// arrays, no GPU, no libzoic_amd), built with -fsanitize=address by tests/test_arnold_shim.py: push() at capacity must store nothing
// and return kFull in every input layout.  VERDICT r5 / ADVICE r5: round 5's push() wrote in_[n_] with no check -- this program is a
// heap-buffer-overflow report under ASan with that header, and "tile_buffer_overflow OK" with this one.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../arnold/zoic_tile_buffer.hpp"

struct zoic_tile { uint32_t capacity; char *in; char *out; int rows, ins; uint32_t submitted; };
static zoic_tile *g_tile = nullptr;   // the tile the buffer under test made
extern "C" {
const char *zoic_last_error_string(void) { return "mock"; }
zoic_status zoic_tile_create(zoic_camera *, uint32_t capacity, uint16_t, zoic_tile **out)
{
    zoic_tile *t = new zoic_tile();
    t->capacity = capacity;
    // EXACTLY capacity rows each, separate allocations: one row too many is a heap-buffer-overflow under ASan
    t->in = static_cast<char *>(std::malloc(static_cast<size_t>(capacity) * sizeof(zoic_camera_input)));
    t->out = static_cast<char *>(std::malloc(static_cast<size_t>(capacity) * sizeof(zoic_camera_output)));
    t->rows = 0; t->ins = 0; t->submitted = 0;
    *out = t; g_tile = t;
    return ZOIC_OK;
}
void zoic_tile_destroy(zoic_tile *t) { if (t) { std::free(t->in); std::free(t->out); delete t; } }
zoic_camera_input *zoic_tile_inputs(zoic_tile *t) { return reinterpret_cast<zoic_camera_input *>(t->in); }
zoic_camera_output *zoic_tile_outputs(zoic_tile *t) { return reinterpret_cast<zoic_camera_output *>(t->out); }
uint32_t zoic_tile_capacity(const zoic_tile *t) { return t->capacity; }
zoic_status zoic_tile_submit(zoic_tile *t, uint32_t n, uint64_t) { t->submitted = n; std::memset(t->out, 0, static_cast<size_t>(n) * (t->rows ? sizeof(zoic_ray) : sizeof(zoic_camera_output))); return n <= t->capacity ? ZOIC_OK : ZOIC_ERR_INVALID_ARGUMENT; }
zoic_status zoic_tile_wait(zoic_tile *) { return ZOIC_OK; }
int zoic_tile_done(zoic_tile *) { return 1; }
zoic_status zoic_tile_set_rows(zoic_tile *t, int rows) { t->rows = rows; return ZOIC_OK; }
const zoic_ray *zoic_tile_rays(const zoic_tile *t) { return reinterpret_cast<const zoic_ray *>(t->out); }
zoic_status zoic_tile_set_inputs(zoic_tile *t, int ins) { t->ins = ins; return ZOIC_OK; }
float *zoic_tile_samples(zoic_tile *t) { return reinterpret_cast<float *>(t->in); }
}

static void expect(bool ok, const char *what) { if (!ok) { std::fprintf(stderr, "FAILED: %s\n", what); std::exit(1); } }

int main()
{
    for (int layout = 0; layout < 4; ++layout) {
        const bool records = (layout & 1) != 0, samples16 = (layout & 2) != 0;
        const uint32_t cap = 37;
        ZoicTileBuffer b(nullptr, cap, 3, records, samples16);
        expect(b.capacity() == cap && b.size() == 0 && !b.full(), "fresh buffer");
        for (uint32_t i = 0; i < cap; ++i) expect(b.push(0.1f * i, -0.2f, 0.3f, 0.4f) == i, "push returns the sample's index");
        expect(b.full(), "full at capacity");
        // one sample too many, in both forms: nothing stored, kFull
        expect(b.push(9.0f, 9.0f, 9.0f, 9.0f) == ZoicTileBuffer::kFull, "push at capacity returns kFull");
        zoic_camera_input in;
        std::memset(&in, 0, sizeof in);
        in.sx = 7.0f;
        expect(b.push(in) == ZoicTileBuffer::kFull && b.size() == cap, "push(row) at capacity returns kFull");
        // the last row is still the one pushed last
        if (samples16) expect(zoic_tile_samples(g_tile)[4 * (cap - 1)] == 0.1f * (cap - 1), "last sample intact");
        else expect(zoic_tile_inputs(g_tile)[cap - 1].sx == 0.1f * (cap - 1), "last row intact");
        expect(b.flush(0) == ZOIC_OK && b.wait() == ZOIC_OK && b.done() && b.flushed() == cap, "flush");
        zoic_camera_output o;
        std::memset(&o, 0, sizeof o);
        b.serve(cap - 1, o);
        bool threw = false;
        try { b.serve(cap, o); } catch (const std::out_of_range &) { threw = true; }
        expect(threw, "serve beyond the flushed samples throws");
        b.clear();
        expect(b.push(in) == 0u, "push after clear");
    }
    std::printf("tile_buffer_overflow OK\n");
    return 0;
}
