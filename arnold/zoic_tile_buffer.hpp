// arnold/zoic_tile_buffer.hpp -- the host-side tile accumulator of north_star's surface ("the C++ host buffers a tile of
// (sx, sy, lensx, lensy) samples and calls through a thin C-ABI"): accumulate -> flush -> serve, header-only C++11 over
// include/zoic_amd.h.  No Arnold SDK needed (tests/native/tile_buffer_test.cpp drives it on a GPU box).
//
// The reference answers camera_create_ray(node, input, output, tid) one sample at a time (zoic.cpp:1752).  A host that knows the
// samples of its bucket before it needs their rays -- a renderer that owns its sampler, a bucket pre-pass, a light-field or
// lens-baking tool -- gives every render thread ONE ZoicTileBuffer:
//
//     ZoicTileBuffer tile(cam, 64 * 64 * 16, tid);          // page-locked arrays the GPU reads / writes in place
//     for (sample : bucket)                                  // accumulate: 16 bytes written per sample, no call
//         if (tile.push(sx, sy, lensx, lensy) == ZoicTileBuffer::kFull) { ...flush / wait / serve, then push again... }
//     tile.flush(first_ray_index_of_bucket);                 // ONE 64-byte request to the camera's resident kernel: no launch
//     ... (the thread may build the next bucket's acceleration data here) ...
//     tile.wait();
//     for (i : bucket) tile.serve(i, output);                // exactly what camera_create_ray would have written into `output`
//
// push() NEVER writes past the arrays: at capacity it stores nothing and returns kFull (round 5's version had no check: one sample too
// many scribbled over the first output row, or past the allocation).  Two options trade the drop-in row layouts for PCIe bytes:
// rayRecords (the answer is 32-byte zoic_ray records instead of 84-byte AtCameraOutput rows) and samples16 (the bucket is filled with
// 16-byte (sx, sy, lensx, lensy) samples instead of 28-byte AtCameraInput rows); serve() is the same whatever the layouts.
//
// [MI355X, round 5's driver run, BENCH_r05.json host_path.tile] a flushed bucket of 4096 samples is answered in 28 us (thin lens 18); 16
// threads x 65536-sample buckets run at 577 Mrays/s with the Arnold rows (112 B a sample across PCIe) and at 1.0 Grays/s with samples16 +
// rayRecords (48 B); the per-sample callback costs 6.9 us per SAMPLE.  Rays are those of
// zoic_create_rays_arnold bit for bit: sample i of the bucket draws its retries from the stream keyed by first_ray_index + i, so a
// frame does not depend on which thread rendered which bucket (the reference's single global stream makes it depend on thread
// timing, zoic.cpp:648).
#ifndef ZOIC_TILE_BUFFER_HPP
#define ZOIC_TILE_BUFFER_HPP

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>

#include "zoic_amd.h"

class ZoicTileBuffer {
public:
    static const uint32_t kFull = 0xffffffffu;   // push() at capacity: nothing was stored -- flush, wait, serve, clear, push again

    // rayRecords: the bucket is answered with 32-byte zoic_ray records instead of 84-byte AtCameraOutput rows (zoic_tile_set_rows): what a
    // bucket costs with many render threads is what crosses PCIe.  serve() is the same either way; row() needs the rows.
    // samples16: the bucket is filled with 16-byte (sx, sy, lensx, lensy) samples -- the four fields zoic reads (zoic.cpp:1853-1854, 1870) --
    // instead of 28-byte AtCameraInput rows (zoic_tile_set_inputs); push() is the same either way.
    ZoicTileBuffer(zoic_camera *cam, uint32_t capacity, uint16_t tid, bool rayRecords = false, bool samples16 = false)
        : tile_(nullptr), in_(nullptr), samples_(nullptr), out_(nullptr), rays_(nullptr), capacity_(0), n_(0), flushed_(0)
    {
        if (zoic_tile_create(cam, capacity, tid, &tile_) != ZOIC_OK) throw std::runtime_error(std::string("zoic_tile_create: ") + zoic_last_error_string());
        if (rayRecords) {
            if (zoic_tile_set_rows(tile_, ZOIC_TILE_ROWS_RAYS) != ZOIC_OK) { zoic_tile_destroy(tile_); throw std::runtime_error(std::string("zoic_tile_set_rows: ") + zoic_last_error_string()); }
            rays_ = zoic_tile_rays(tile_);
        }
        if (samples16) {
            if (zoic_tile_set_inputs(tile_, ZOIC_TILE_INPUTS_SAMPLES) != ZOIC_OK) { zoic_tile_destroy(tile_); throw std::runtime_error(std::string("zoic_tile_set_inputs: ") + zoic_last_error_string()); }
            samples_ = zoic_tile_samples(tile_);
        }
        in_ = zoic_tile_inputs(tile_);
        out_ = zoic_tile_outputs(tile_);
        capacity_ = zoic_tile_capacity(tile_);
    }
    ~ZoicTileBuffer() { zoic_tile_destroy(tile_); }   // waits for a flush still in flight
    ZoicTileBuffer(const ZoicTileBuffer &) = delete;
    ZoicTileBuffer &operator=(const ZoicTileBuffer &) = delete;

    uint32_t capacity() const { return capacity_; }
    uint32_t size() const { return n_; }
    bool full() const { return n_ == capacity_; }
    void clear() { n_ = 0; }

    // accumulate: the four AtCameraInput fields zoic reads (zoic.cpp:1853-1854, 1870).  Returns the sample's index in the bucket, or
    // kFull -- and stores NOTHING -- when the bucket holds `capacity()` samples already.  (Not while a flush is in flight: the GPU is
    // reading the rows.)
    uint32_t push(float sx, float sy, float lensx, float lensy)
    {
        if (n_ >= capacity_) return kFull;
        if (samples_) { float *q = samples_ + 4u * static_cast<size_t>(n_); q[0] = sx; q[1] = sy; q[2] = lensx; q[3] = lensy; }
        else { zoic_camera_input &r = in_[n_]; r.sx = sx; r.sy = sy; r.dsx = 0.0f; r.dsy = 0.0f; r.lensx = lensx; r.lensy = lensy; r.relative_time = 0.0f; }
        return n_++;
    }
    uint32_t push(const zoic_camera_input &in)
    {
        if (samples_) return push(in.sx, in.sy, in.lensx, in.lensy);
        if (n_ >= capacity_) return kFull;
        in_[n_] = in;
        return n_++;
    }

    // flush: posts the accumulated samples; returns at once.  ray_index_base: the global index of the bucket's first sample.
    zoic_status flush(uint64_t ray_index_base)
    {
        flushed_ = n_;
        return zoic_tile_submit(tile_, n_, ray_index_base);
    }
    bool sample_inputs() const { return samples_ != nullptr; }
    bool ray_records() const { return rays_ != nullptr; }
    zoic_status wait() { return zoic_tile_wait(tile_); }
    bool done() { return zoic_tile_done(tile_) != 0; }

    // serve: the finished row as the library wrote it (a whole AtCameraOutput from a zero-initialised one with weight 1) ...
    // (i < flushed(): rows beyond what was flushed were never written)
    const zoic_camera_output &row(uint32_t i) const
    {
        if (rays_) throw std::logic_error("ZoicTileBuffer::row: this buffer holds zoic_ray records");
        if (i >= flushed_) throw std::out_of_range("ZoicTileBuffer::row: beyond the flushed samples");
        return out_[i];
    }
    const zoic_ray &ray(uint32_t i) const
    {
        if (!rays_) throw std::logic_error("ZoicTileBuffer::ray: this buffer holds AtCameraOutput rows");
        if (i >= flushed_) throw std::out_of_range("ZoicTileBuffer::ray: beyond the flushed samples");
        return rays_[i];
    }
    // ... or applied to the caller's AtCameraOutput exactly as camera_create_ray updates it in place (zoic.cpp:1960-1961 origin /
    // dir; 1825 / 1952 weight = 0; 1981-1987 weight *= exposure; 1974-1977 dOdy / dDdy for retried rays only; dOdx / dDdx and the
    // derivatives of first-try rays are left alone)
    void serve(uint32_t i, zoic_camera_output &output) const
    {
        if (i >= flushed_) throw std::out_of_range("ZoicTileBuffer::serve: beyond the flushed samples");
        if (rays_) {   // the record says it all: flags bit 0 = retried (zoic_amd.h)
            const zoic_ray &q = rays_[i];
            output.origin.x = q.ox; output.origin.y = q.oy; output.origin.z = q.oz;
            output.dir.x = q.dx; output.dir.y = q.dy; output.dir.z = q.dz;
            if (q.weight == 0.0f) output.weight[0] = output.weight[1] = output.weight[2] = 0.0f;
            else if (q.weight != 1.0f) { output.weight[0] *= q.weight; output.weight[1] *= q.weight; output.weight[2] *= q.weight; }
            if (q.flags & 1u) { output.dOdy = output.origin; output.dDdy = output.dir; }
            return;
        }
        const zoic_camera_output &r = out_[i];
        output.origin = r.origin;
        output.dir = r.dir;
        const float w = r.weight[0];
        if (w == 0.0f) output.weight[0] = output.weight[1] = output.weight[2] = 0.0f;
        else if (w != 1.0f) { output.weight[0] *= w; output.weight[1] *= w; output.weight[2] *= w; }
        // a retried ray's row carries dOdy = origin, dDdy = dir; a first-try ray's row carries zeros there
        static const zoic_vec3 zero = {0.0f, 0.0f, 0.0f};
        const bool retried = std::memcmp(&r.dOdy, &r.origin, sizeof(zoic_vec3)) == 0 && std::memcmp(&r.dDdy, &r.dir, sizeof(zoic_vec3)) == 0 &&
                             std::memcmp(&r.dDdy, &zero, sizeof(zoic_vec3)) != 0;   // (a direction is never all-zero bits; bitwise: NaN rays compare too)
        if (retried) { output.dOdy = r.origin; output.dDdy = r.dir; }
    }
    uint32_t flushed() const { return flushed_; }

private:
    zoic_tile *tile_;
    zoic_camera_input *in_;
    float *samples_;                  // samples16: the same memory as in_, 4 floats per sample
    zoic_camera_output *out_;
    const zoic_ray *rays_;
    uint32_t capacity_, n_, flushed_;
};

#endif  // ZOIC_TILE_BUFFER_HPP
