// bokeh_cdf.hip -- imageData::bokehProbability (zoic.cpp:222-417) on the GPU: SURVEY "next" row f2.
//
// The tables must be identical to the reference's, and the reference builds them with SEQUENTIAL f32 sums (total
// luminance, row masses, both running CDFs) and two descending sorts.  What can run in parallel without changing a
// single rounding:
//   * the luminance total is one sequential chain over every pixel -> stays on the host (O(n), ~1 ns/pixel);
//   * row masses: one lane per row walks its row in order (rows are independent chains);
//   * the row sort and the y per-row column sorts: bitonic sort of 64-bit keys in LDS,
//       key = (~orderable(value) << 32) | index   (ascending key == descending value, ascending index on ties --
//       the deterministic tie rule the oracle documents; std::sort itself leaves ties unspecified);
//   * the running CDFs: one lane per row adds its sorted row in order.
// At 2048^2 the host build takes 135 ms (std::stable_sort dominated); this path takes a few ms.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "lens_system.hpp"

#pragma STDC FP_CONTRACT OFF

namespace zoic {

namespace {

constexpr int kSortThreads = 256;
constexpr int kMaxSortN = 4096;  // 32 KiB of 64-bit keys in LDS

__device__ __forceinline__ uint32_t orderable(float v)
{
    if (v == 0.0f) v = 0.0f;  // -0 and +0 compare equal in the reference's comparator
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t o)
{
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __builtin_bit_cast(float, b);
}

// ascending bitonic sort of N (power of two) keys held in LDS
__device__ void bitonic_sort_lds(unsigned long long *keys, int N)
{
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = keys[i], b = keys[p];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) { keys[i] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// pdf = lum / total, row masses (sequential per row)            zoic.cpp:259-293
__global__ void cdf_normalise_rows(const float *__restrict__ lum, float invTotal, int x, int y, float *__restrict__ pdf,
                                   float *__restrict__ rowMass)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= y) return;
    float s = 0.0f;
    const size_t base = static_cast<size_t>(r) * x;
    for (int c = 0; c < x; ++c) {
        const float q = lum[base + c] * invTotal;
        pdf[base + c] = q;
        s += q;
    }
    rowMass[r] = s;
}

// rows sorted by descending mass + running row CDF               zoic.cpp:313-337
__global__ void cdf_sort_rows(const float *__restrict__ rowMass, int y, int N, int32_t *__restrict__ rowIndices,
                              float *__restrict__ cdfRow)
{
    extern __shared__ unsigned long long keys[];
    for (int i = threadIdx.x; i < N; i += blockDim.x)
        keys[i] = i < y ? (static_cast<unsigned long long>(~orderable(rowMass[i])) << 32) | static_cast<uint32_t>(i) : ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, N);
    for (int i = threadIdx.x; i < y; i += blockDim.x) rowIndices[i] = static_cast<int32_t>(keys[i] & 0xffffffffu);
    if (threadIdx.x == 0) {
        float run = 0.0f;
        for (int i = 0; i < y; ++i) {
            run = run + rowMass[static_cast<uint32_t>(keys[i] & 0xffffffffu)];
            cdfRow[i] = run;
        }
    }
}

// one workgroup per image row: conditional pdf, descending sort, running column CDF     zoic.cpp:352-407
__global__ void cdf_sort_columns(const float *__restrict__ pdf, const float *__restrict__ rowMass, int x, int N,
                                 int32_t *__restrict__ columnIndices, float *__restrict__ cdfColumn)
{
    extern __shared__ unsigned long long keys[];
    const int r = blockIdx.x;
    const size_t base = static_cast<size_t>(r) * x;
    const float mass = rowMass[r];
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        unsigned long long key = ~0ull;
        if (c < x) {
            const float q = pdf[base + c];
            const float w = (q != 0 && mass != 0) ? q / mass : 0.0f;
            key = (static_cast<unsigned long long>(~orderable(w)) << 32) | static_cast<uint32_t>(c);
        }
        keys[c] = key;
    }
    __syncthreads();
    bitonic_sort_lds(keys, N);
    for (int c = threadIdx.x; c < x; c += blockDim.x)
        columnIndices[base + c] = static_cast<int32_t>(base) + static_cast<int32_t>(keys[c] & 0xffffffffu);
    if (threadIdx.x == 0) {
        float run = 0.0f;
        for (int c = 0; c < x; ++c) {
            run = run + from_orderable(~static_cast<uint32_t>(keys[c] >> 32));
            cdfColumn[base + c] = run;
        }
    }
}

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// Cell records of tables.hpp for every CDF of the image: block 0 = the row CDF (y entries, gRow cells), block 1 + r = the
// column CDF of image row r (x entries, gCol cells).  One lane per cell: two binary searches and three index reads.
__global__ void build_cells_kernel(const float *__restrict__ cdfRow, const int32_t *__restrict__ rowIdx, const float *__restrict__ cdfCol,
                                   const int32_t *__restrict__ colIdx, int x, int y, int gRow, int gCol, uint32_t *__restrict__ rowCells,
                                   uint32_t *__restrict__ colCells, uint32_t *__restrict__ rowBounds, uint32_t *__restrict__ colBounds)
{
    const bool isRow = blockIdx.x == 0;
    const int r = static_cast<int>(blockIdx.x) - 1;
    const float *cdf = isRow ? cdfRow : cdfCol + static_cast<size_t>(r) * x;
    const int32_t *idx = isRow ? rowIdx : colIdx + static_cast<size_t>(r) * x;
    const int32_t idxBase = isRow ? 0 : r * x;
    const int n = isRow ? y : x, g = isRow ? gRow : gCol;
    uint32_t *rec = isRow ? rowCells : colCells + static_cast<size_t>(r) * gCol * 4;
    uint32_t *bnd = isRow ? rowBounds : colBounds + static_cast<size_t>(r) * gCol;
    for (int c = threadIdx.x; c < g; c += blockDim.x) {
        const float lower = static_cast<float>(c) / static_cast<float>(g), upper = static_cast<float>(c + 1) / static_cast<float>(g);
        const int lo = upper_bound_idx(cdf, n, lower);                 // #{cdf <= lower}
        int hi = lo, len = n - lo;                                     // lower_bound(upper) over [lo, n): #{cdf < upper}
        while (len > 0) {
            const int half = len >> 1;
            if (cdf[hi + half] < upper) { hi += half + 1; len -= half + 1; } else len = half;
        }
        // thresholds = the first two DISTINCT CDF values above the cell's lower edge: a run of equal values (the zero-luminance
        // tail of every CDF) is ONE decision for std::upper_bound -- u < value: before the run, else: after all of it
        const float inf = __builtin_inff();
        const float a = lo < n ? cdf[lo] : inf;
        const int j = lo < n ? lo + upper_bound_idx(cdf + lo, n - lo, a) : n;      // first entry > a
        const float b = j < n ? cdf[j] : inf;
        const int k = j < n ? j + upper_bound_idx(cdf + j, n - j, b) : n;          // first entry > b
        const int e0 = lo < n ? lo : n - 1, e1 = j < n ? j : n - 1, e2 = k < n ? k : n - 1;
        uint4 out;
        out.x = __builtin_bit_cast(uint32_t, a); out.y = __builtin_bit_cast(uint32_t, b);
        out.z = (static_cast<uint32_t>(idx[e0] - idxBase) & 0xffffu) | ((static_cast<uint32_t>(idx[e1] - idxBase) & 0xffffu) << 16);
        out.w = (static_cast<uint32_t>(idx[e2] - idxBase) & 0xffffu) | ((k < hi) ? 0x80000000u : 0u);   // a third distinct value inside the cell
        reinterpret_cast<uint4 *>(rec)[c] = out;
        bnd[c] = static_cast<uint32_t>(lo) | (static_cast<uint32_t>(hi) << 16);
    }
}

}  // namespace

// Returns 0 on success (tables in `out`), >0 = hipError_t, -1 = image outside what this path covers (caller uses the host build).
int build_bokeh_cdf_device(const float *pixels, int width, int height, int nchannels, BokehCdf &out)
{
    out.clear();
    if (!pixels || width <= 0 || height <= 0 || nchannels < 3) return -1;
    if (width > kMaxSortN || height > kMaxSortN || width < 2 || height < 2) return -1;
    const int x = width, y = height;
    const size_t n = static_cast<size_t>(x) * y;
    // luminance + its sequential f32 total: host (zoic.cpp:243-249, 259)
    std::vector<float> lum(n);
    float total = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float *p = pixels + i * nchannels;
        lum[i] = p[0] * 0.3f + p[1] * 0.59f + p[2] * 0.11f;
        total += lum[i];
    }
    const float invTotal = 1.0f / total;

    float *dLum = nullptr, *dPdf = nullptr, *dRowMass = nullptr, *dCdfRow = nullptr, *dCdfCol = nullptr;
    int32_t *dRowIdx = nullptr, *dColIdx = nullptr;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc(reinterpret_cast<void **>(&dLum), n * 4); alloc(reinterpret_cast<void **>(&dPdf), n * 4);
    alloc(reinterpret_cast<void **>(&dRowMass), y * 4); alloc(reinterpret_cast<void **>(&dCdfRow), y * 4);
    alloc(reinterpret_cast<void **>(&dCdfCol), n * 4); alloc(reinterpret_cast<void **>(&dRowIdx), y * 4);
    alloc(reinterpret_cast<void **>(&dColIdx), n * 4);
    if (e == hipSuccess) e = hipMemcpy(dLum, lum.data(), n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cdf_normalise_rows, dim3((y + 63) / 64), dim3(64), 0, nullptr, dLum, invTotal, x, y, dPdf, dRowMass);
        const int ny = next_pow2(y), nx = next_pow2(x);
        hipLaunchKernelGGL(cdf_sort_rows, dim3(1), dim3(kSortThreads), ny * sizeof(unsigned long long), nullptr, dRowMass, y, ny,
                           dRowIdx, dCdfRow);
        hipLaunchKernelGGL(cdf_sort_columns, dim3(y), dim3(kSortThreads), nx * sizeof(unsigned long long), nullptr, dPdf, dRowMass, x,
                           nx, dColIdx, dCdfCol);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        out.x = x; out.y = y;
        out.cdfRow.resize(y); out.rowIndices.resize(y); out.cdfColumn.resize(n); out.columnIndices.resize(n);
        e = hipMemcpy(out.cdfRow.data(), dCdfRow, y * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(out.rowIndices.data(), dRowIdx, y * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(out.cdfColumn.data(), dCdfCol, n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(out.columnIndices.data(), dColIdx, n * 4, hipMemcpyDeviceToHost);
    }
    for (void *p : {static_cast<void *>(dLum), static_cast<void *>(dPdf), static_cast<void *>(dRowMass), static_cast<void *>(dCdfRow),
                    static_cast<void *>(dCdfCol), static_cast<void *>(dRowIdx), static_cast<void *>(dColIdx)})
        if (p) (void)hipFree(p);
    if (e != hipSuccess) { out.clear(); return static_cast<int>(e); }
    return 0;
}

// Cell records (tables.hpp) from the device-resident reference tables; dCells = records of the row CDF, records of the y
// column CDFs, then the bounds in the same order ((gRow + y*gCol) * 5 dwords).  Identical to the host build in capi.cpp.
int build_bokeh_cells_device(const float *dCdfRow, const int32_t *dRowIdx, const float *dCdfCol, const int32_t *dColIdx, int x, int y,
                             int gRow, int gCol, uint32_t *dCells)
{
    const size_t nCells = static_cast<size_t>(gRow) + static_cast<size_t>(y) * gCol;
    uint32_t *rowCells = dCells, *colCells = dCells + static_cast<size_t>(gRow) * 4;
    uint32_t *rowBounds = dCells + nCells * 4, *colBounds = rowBounds + gRow;
    hipLaunchKernelGGL(build_cells_kernel, dim3(static_cast<unsigned>(y) + 1u), dim3(256), 0, nullptr, dCdfRow, dRowIdx, dCdfCol, dColIdx, x, y,
                       gRow, gCol, rowCells, colCells, rowBounds, colBounds);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return static_cast<int>(e);
}

}  // namespace zoic
