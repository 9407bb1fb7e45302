// device_search.hpp -- std::upper_bound on the bokeh CDFs through the 16-ary pyramid of tables.hpp.
//
// imageData::bokehSample (zoic.cpp:420-485) does two std::upper_bound binary searches per lens sample: 8 + 8
// dependent loads for a 256x256 image, each a different cache line per lane.  On the device one pyramid level is
// one aligned 64-byte line fetched with four global_load_dwordx4 (a single round trip); the position inside the line
// is the COUNT of entries <= u (the array is non-decreasing, so that count is the upper_bound index).  Padding is
// +inf, the result is clamped to n, which makes the index identical to std::upper_bound for every u including
// NaN (count saturates -> n) -- so strict mode stays bit-exact.
#pragma once
#include <hip/hip_runtime.h>

#include "optics.hpp"
#include "tables.hpp"

namespace zoic {

__device__ __forceinline__ int chunk_count_le(const float *chunk, float v)
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

// upper_bound over one CDF given its pyramid levels; `row` selects the padded row at each level (0 for the row CDF)
__device__ __forceinline__ int pyramid_upper_bound(const float *const level[kBokehMaxLevels], const int32_t stride[kBokehMaxLevels],
                                                   const int32_t count[kBokehMaxLevels], int levels, int row, float v)
{
    int pos = 0;
#pragma unroll
    for (int j = kBokehMaxLevels - 1; j >= 0; --j) {
        if (j >= levels) continue;
        const float *chunk = level[j] + static_cast<size_t>(row) * stride[j] + static_cast<size_t>(pos) * 16;
        pos = pos * 16 + chunk_count_le(chunk, v);
        if (pos >= count[j]) return count[0];
    }
    return pos;
}

__device__ __forceinline__ V2 bokeh_sample_device(const BokehTables &B, int x, int y, float uRow, float uCol)
{
    if (B.levels == 0)  // CDF longer than the pyramid covers: the reference's binary search
        return bokeh_sample(B.cdfRow, B.rowIndices, B.cdfColumn, B.columnIndices, x, y, uRow, uCol);
    const int32_t zeroStride[kBokehMaxLevels] = {0, 0, 0};
    int r = pyramid_upper_bound(B.rowLevel, zeroStride, B.rowCount, B.levels, 0, uRow);
    if (r >= y) r = y - 1;
    const int row = B.rowIndices[r];
    int c = pyramid_upper_bound(B.colLevel, B.colStride, B.colCount, B.levels, row, uCol);
    if (c >= x) c = x - 1;
    const int start = row * x;
    const int col = B.columnIndices[start + c] - start;
    const float flippedRow = static_cast<float>(col - ((y - 1) / 2));             // zoic.cpp:466,479 (x/y swapped)
    const float flippedColumn = static_cast<float>(row - ((x - 1) / 2)) * -1.0f;  // zoic.cpp:441,480
    return V2{(flippedRow / static_cast<float>(x)) * 2.0f, (flippedColumn / static_cast<float>(y)) * 2.0f};
}

// Number of entries <= v in a non-decreasing LDS array of 2^log2n floats (+inf padded): the classic branch-free
// descent, one ds_read_b32 + compare + select per level with a wave-uniform step.  == std::upper_bound index.
// The kernels are VALU-issue bound, so 3 VALU per level beats the 32 compares+adds of counting a 16-chunk.
__device__ __forceinline__ int lds_count_le_pow2(const float *a, int log2n, float v)
{
    int pos = 0;
    for (int step = 1 << log2n; step > 0; step >>= 1) {
        const int t = pos + step;
        if (t <= (1 << log2n) && !(v < a[t - 1])) pos = t;
    }
    return pos;
}

// Lens sample with the row tables and the column pyramid tops resident in LDS (`lds` = the workgroup's copy of
// BokehTables::ldsImage).  Identical indices to bokeh_sample_device / std::upper_bound.
template <bool EXACT_DIVIDE>
__device__ __forceinline__ V2 bokeh_sample_lds(const BokehTables &B, const float *lds, int x, int y, float uRow, float uCol)
{
    const float *rowL0 = lds + 16;
    const int32_t *rowIdx = reinterpret_cast<const int32_t *>(lds + 16 + B.rowStride0);
    const float *colTop = lds + 16 + 2 * B.rowStride0;
    int r = lds_count_le_pow2(rowL0, B.rowLog2, uRow);          // padding is +inf: never counted for finite u
    if (r >= y) r = y - 1;                                      // also catches NaN (every compare true -> 2^log2n)
    const int row = rowIdx[r];
    int c = lds_count_le_pow2(colTop + row * 16, 4, uCol);     // which 16-chunk of this row's column CDF
    int col;
    if (c >= B.colCount[1]) {                                   // every entry of this row's CDF <= u: clamp to the last
        const int32_t *iline = reinterpret_cast<const int32_t *>(B.colPacked + (static_cast<size_t>(row) * B.colChunks + (x - 1) / 16) * 32 + 16);
        col = iline[(x - 1) & 15];
    } else {
        const float *line = B.colPacked + (static_cast<size_t>(row) * B.colChunks + c) * 32;
        int k = chunk_count_le(line, uCol);                     // one 64-byte global read; < 16 because the chunk max > u
        int e = c * 16 + k;
        if (e >= x) e = x - 1;
        col = reinterpret_cast<const int32_t *>(line + 16)[e & 15];    // same 128-byte line: L1 hit
    }
    const float flippedRow = static_cast<float>(col - ((y - 1) / 2));
    const float flippedColumn = static_cast<float>(row - ((x - 1) / 2)) * -1.0f;
    if constexpr (EXACT_DIVIDE)
        return V2{(flippedRow / static_cast<float>(x)) * 2.0f, (flippedColumn / static_cast<float>(y)) * 2.0f};
    else  // fast mode: wave-uniform reciprocals (exact when x, y are powers of two)
        return V2{flippedRow * (2.0f * __builtin_amdgcn_rcpf(static_cast<float>(x))),
                  flippedColumn * (2.0f * __builtin_amdgcn_rcpf(static_cast<float>(y)))};
}

}  // namespace zoic
