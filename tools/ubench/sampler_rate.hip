// sampler_rate.hip -- bokeh lens sampling in isolation on a synthetic 256x256 table set, persistent waves, 6 workgroups of
// 256 per CU like the Kolb kernel: the v7 descent sampler stage by stage against the access pattern of the cell-record
// sampler that replaced it (device_search.hpp).  Numbers: profiles/ubench_r01.txt.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include tools/ubench/sampler_rate.hip -o tools/ubench/sampler_rate
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../zoic_amd/csrc/device_search.hpp"

using namespace zoic;

// the round-1 v7 sampler (binary descent over LDS copies of the row CDF and the column chunk maxima + one packed
// 128-byte global line), kept here only as the "before" of the comparison
struct OldTables { const float *ldsImage, *colPacked; int rowStride0, rowLog2, colChunks; };
__device__ __forceinline__ int chunk_count_le16(const float *chunk, float v)
{
    const float4 *p = reinterpret_cast<const float4 *>(chunk);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    int n = 0;
    n += !(v < a.x); n += !(v < a.y); n += !(v < a.z); n += !(v < a.w);
    n += !(v < b.x); n += !(v < b.y); n += !(v < b.z); n += !(v < b.w);
    n += !(v < c.x); n += !(v < c.y); n += !(v < c.z); n += !(v < c.w);
    n += !(v < d.x); n += !(v < d.y); n += !(v < d.z); n += !(v < d.w);
    return n;
}
__device__ __forceinline__ int lds_count_le_pow2(const float *a, int log2n, float v)
{
    int pos = 0;
    for (int step = (1 << log2n) >> 1; step > 0; step >>= 1) pos += !(v < a[pos + step - 1]) ? step : 0;
    pos += !(v < a[pos]) ? 1 : 0;
    return pos;
}
__device__ __forceinline__ V2 old_sample(const OldTables &B, const float *lds, int x, int y, float uRow, float uCol)
{
    int r = lds_count_le_pow2(lds + 16, B.rowLog2, uRow);
    if (r >= y) r = y - 1;
    const int row = reinterpret_cast<const int32_t *>(lds + 16 + B.rowStride0)[r];
    int c = lds_count_le_pow2(lds + 16 + 2 * B.rowStride0 + row * 16, 4, uCol);
    if (c >= B.colChunks) c = B.colChunks - 1;
    const float *line = B.colPacked + (static_cast<size_t>(row) * B.colChunks + c) * 32;
    int e = c * 16 + chunk_count_le16(line, uCol);
    if (e >= x) e = x - 1;
    const int col = reinterpret_cast<const int32_t *>(line + 16)[e & 15];
    return V2{static_cast<float>(col - 127) * (2.0f / 256.0f), static_cast<float>(row - 127) * (-2.0f / 256.0f)};
}
extern __shared__ __align__(16) float dynLds[];

template <int MODE>
__global__ __launch_bounds__(256) void k(OldTables B, int ldsWords, int iters, float *out, const float4 *cells)
{
    for (int i = threadIdx.x; i < ldsWords; i += 256) dynLds[i] = B.ldsImage[i];
    __syncthreads();
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = static_cast<float>(s >> 8) * (1.0f / 16777216.0f);
        s = s * 1664525u + 1013904223u;
        const float v = static_cast<float>(s >> 8) * (1.0f / 16777216.0f);
        if (MODE == 0) { const V2 l = old_sample(B, dynLds, 256, 256, u, v); acc += l.x + l.y; }
        if (MODE == 1) { acc += static_cast<float>(lds_count_le_pow2(dynLds + 16, B.rowLog2, u)); }              // row descent only
        if (MODE == 2) { const int r = lds_count_le_pow2(dynLds + 16, B.rowLog2, u);                               // + rowIdx + colTop
                         const int row = reinterpret_cast<const int32_t *>(dynLds + 16 + B.rowStride0)[r > 255 ? 255 : r];
                         acc += static_cast<float>(lds_count_le_pow2(dynLds + 16 + 2 * B.rowStride0 + row * 16, 4, v)); }
        if (MODE == 4) {   // cell-record prototype: one ds_read_b128 (row cell) + one global dwordx4 (column cell), ~12 VALU each
            const float4 rc = reinterpret_cast<const float4 *>(dynLds)[static_cast<int>(u * 256.0f)];
            const uint32_t pk = __builtin_bit_cast(uint32_t, rc.z);
            const int kr = (!(u < rc.x)) + (!(u < rc.y));
            const int row = (pk >> (8 * kr)) & 0xff;
            const float4 cc = cells[row * 256 + static_cast<int>(v * 256.0f)];
            const uint32_t pk2 = __builtin_bit_cast(uint32_t, cc.z);
            const int kc = (!(v < cc.x)) + (!(v < cc.y));
            acc += static_cast<float>(row) + static_cast<float>((pk2 >> (8 * kc)) & 0xff) + ((pk | pk2) >> 24 ? 1.f : 0.f);
        }
        if (MODE == 3) { acc += u + v; }                                                                            // loop + LCG only
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
void run(const char *name, const OldTables &B, int ldsWords, float *d, const float4 *cells)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400, blocks = 2048;
    const size_t lds = 26 * 1024;   // 6 workgroups per CU
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), lds, 0, B, ldsWords, 4, d, cells);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), lds, 0, B, ldsWords, iters, d, cells);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = double(blocks) * 256 * iters;
    printf("%-34s %.3f ms  %.1f Gsamples/s   (170 M samples = %.2f ms)\n", name, ms, n / (ms * 1e-3) / 1e9, 170e6 / (n / (ms * 1e-3)) * 1e3);
}

int main()
{
    const int x = 256, y = 256, rs0 = 256, chunks = 16;
    const size_t ldsWords = 16 + rs0 * 2 + y * 16, packed = size_t(y) * chunks * 32;
    std::vector<float> img(ldsWords + packed);
    srand(1);
    // row CDF: increasing, ~uniform increments with jitter
    double run_ = 0; std::vector<double> inc(y);
    for (int i = 0; i < y; ++i) { inc[i] = 0.5 + rand() / double(RAND_MAX); run_ += inc[i]; }
    double a = 0;
    for (int i = 0; i < y; ++i) { a += inc[i] / run_; img[16 + i] = float(a); }
    int32_t *ri = reinterpret_cast<int32_t *>(img.data() + 16 + rs0);
    for (int i = 0; i < y; ++i) ri[i] = (i * 97) % y;
    float *pk = img.data() + ldsWords;
    for (int r = 0; r < y; ++r) {
        double tot = 0; std::vector<double> w(x);
        for (int c = 0; c < x; ++c) { w[c] = 0.5 + rand() / double(RAND_MAX); tot += w[c]; }
        double b = 0;
        for (int c = 0; c < x; ++c) {
            b += w[c] / tot;
            float *line = pk + (size_t(r) * chunks + c / 16) * 32;
            line[c & 15] = float(b);
            reinterpret_cast<int32_t *>(line + 16)[c & 15] = (c * 31) % x;
            if ((c & 15) == 15) img[16 + 2 * rs0 + r * 16 + c / 16] = float(b);
        }
    }
    float *dImg; (void)hipMalloc(&dImg, img.size() * 4); (void)hipMemcpy(dImg, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    float *d; (void)hipMalloc(&d, 4);
    OldTables B{dImg, dImg + ldsWords, rs0, 8, chunks};
    std::vector<uint32_t> cw(size_t(256) * 256 * 4);
    for (size_t i = 0; i < cw.size(); ++i) cw[i] = (i & 3) < 2 ? 0x3f000000u : (uint32_t(rand()) & 0x00ffffffu);
    float4 *cells; (void)hipMalloc(&cells, cw.size() * 4); (void)hipMemcpy(cells, cw.data(), cw.size() * 4, hipMemcpyHostToDevice);
    run<3>("loop + LCG only", B, int(ldsWords), d, cells);
    run<4>("cell records (1 LDS b128 + 1 global x4)", B, int(ldsWords), d, cells);
    run<1>("row descent (9 LDS steps)", B, int(ldsWords), d, cells);
    run<2>("row + rowIdx + colTop (15 LDS)", B, int(ldsWords), d, cells);
    run<0>("full sampler", B, int(ldsWords), d, cells);
    return 0;
}
