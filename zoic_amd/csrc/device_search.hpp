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

// One CDF resolved through its cell record (tables.hpp): the pixel index, identical to
// indices[min(upper_bound(cdf, u), n-1)] unless the cell is exceptional (then the caller finishes with upper_bound).
__device__ __forceinline__ int cell_of(float u, int cellCount, bool &inRange)
{
    inRange = (u >= 0.0f) & (u < 1.0f);
    return inRange ? static_cast<int>(u * static_cast<float>(cellCount)) : 0;   // exact: cellCount is a power of two
}

__device__ __forceinline__ int cell_resolve(const uint4 rec, float u, bool inRange, bool &exceptional)
{
    const float a = __builtin_bit_cast(float, rec.x), b = __builtin_bit_cast(float, rec.y);
    const bool ge0 = !(u < a), ge1 = !(u < b);          // sorted: ge1 implies ge0
    exceptional = ((rec.w >> 31) != 0u) | !inRange;
    const uint32_t pair = ge1 ? rec.w : rec.z;           // k = 2 -> idx[lo+2] (low half of .w)
    return static_cast<int>((ge0 & !ge1) ? (pair >> 16) : (pair & 0xffffu));   // k = 1 -> high half of .z
}

// Lens sample through the cell records, in two halves so that a kernel can request the column record of a sample one pass
// before it needs it (kolb_pool_body.hpp): bokeh_cells_issue resolves the row from the workgroup's LDS copy of
// BokehTables::rowCells and issues the ONE dependent global load (the column cell record of that row); bokeh_cells_finish
// turns the record into the lens point.  Identical indices to bokeh_sample / std::upper_bound (zoic.cpp:420-485).
struct CellProbe { uint4 rec; int row; uint32_t cellIndex; };

__device__ __forceinline__ CellProbe bokeh_cells_issue(const BokehTables &B, const float *ldsRowCells, int y, float uRow, float uCol)
{
    bool inR, inC, excR;
    const int gr = cell_of(uRow, B.rowCellCount, inR);
    const int gc = cell_of(uCol, B.colCellCount, inC);
    int row = cell_resolve(reinterpret_cast<const uint4 *>(ldsRowCells)[gr], uRow, inR, excR);
    if (__ballot(excR) != 0ull) {          // rare, wave-uniform: dense cell or a sample outside [0,1)
        if (excR) {
            const uint32_t bnd = B.rowBounds[gr];
            const int lo = inR ? static_cast<int>(bnd & 0xffffu) : 0, hi = inR ? static_cast<int>(bnd >> 16) : y;
            int r = lo + upper_bound_idx(B.cdfRow + lo, hi - lo, uRow);
            if (r >= y) r = y - 1;
            row = B.rowIndices[r];
        }
    }
    CellProbe p;
    p.row = row;
    p.cellIndex = static_cast<uint32_t>(row) * static_cast<uint32_t>(B.colCellCount) + static_cast<uint32_t>(gc);
    p.rec = reinterpret_cast<const uint4 *>(B.colCells)[p.cellIndex];
    return p;
}

template <bool EXACT_DIVIDE>
__device__ __forceinline__ V2 bokeh_cells_finish(const BokehTables &B, int x, int y, float uCol, const CellProbe &p)
{
    const bool inC = (uCol >= 0.0f) & (uCol < 1.0f);
    bool excC;
    const int row = p.row;
    int col = cell_resolve(p.rec, uCol, inC, excC);
    if (__ballot(excC) != 0ull) {
        if (excC) {
            const uint32_t bnd = B.colBounds[p.cellIndex];
            const int lo = inC ? static_cast<int>(bnd & 0xffffu) : 0, hi = inC ? static_cast<int>(bnd >> 16) : x;
            const int start = row * x;
            int c = lo + upper_bound_idx(B.cdfColumn + start + lo, hi - lo, uCol);
            if (c >= x) c = x - 1;
            col = B.columnIndices[start + c] - start;
        }
    }
    const float flippedRow = static_cast<float>(col - ((y - 1) / 2));             // zoic.cpp:466,479 (x/y swapped)
    const float flippedColumn = static_cast<float>(row - ((x - 1) / 2)) * -1.0f;  // zoic.cpp:441,480
    if constexpr (EXACT_DIVIDE)
        return V2{(flippedRow / static_cast<float>(x)) * 2.0f, (flippedColumn / static_cast<float>(y)) * 2.0f};
    else  // fast mode: wave-uniform reciprocals (exact when x, y are powers of two)
        return V2{flippedRow * (2.0f * __builtin_amdgcn_rcpf(static_cast<float>(x))),
                  flippedColumn * (2.0f * __builtin_amdgcn_rcpf(static_cast<float>(y)))};
}

template <bool EXACT_DIVIDE>
__device__ __forceinline__ V2 bokeh_sample_cells(const BokehTables &B, const float *ldsRowCells, int x, int y, float uRow, float uCol)
{
    return bokeh_cells_finish<EXACT_DIVIDE>(B, x, y, uCol, bokeh_cells_issue(B, ldsRowCells, y, uRow, uCol));
}

}  // namespace zoic
