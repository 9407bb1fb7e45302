// kolb_listed.hip -- the listed kernel of a decision-safe FAST launch (kolb_listed_body.hpp): one instantiation per unrolled
// interface count.  A translation unit of its own: it compiles beside the main kernels and carries both arithmetics.
#include "kolb_listed_body.hpp"

namespace zoic {

// Launched right behind the GUARD kernel on its stream; `grid` = the GUARD kernel's (workgroups beyond the list's length retire
// at once: the length is only known on the device).  d_redoCursor: the partition cursors of this kernel inside the launch's
// cursor block (kernels.hpp kRedoCursorOffset).
int launch_kolb_listed(const KolbTable &table, const BokehTables &bokeh, const float4 *d_samples, const uint4 *d_rng, uint64_t rayBase, uint32_t m,
                       RayRecord *out, DeviceCounters *d_counters, unsigned int *d_redoCursor, uint32_t *d_redoList, unsigned int *d_redoCount,
                       unsigned grid, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    // EXPERIMENT (profiles/ab_r06/ab_listed_grid.log): ZOIC_LISTED_GRID=n workgroups instead of the GUARD kernel's grid.  Both paths of the
    // kernel stride over the list with whatever grid they get, so this is a pure performance knob.
    static const long gridOverride = [] { const char *e = std::getenv("ZOIC_LISTED_GRID"); return e ? std::atol(e) : 0L; }();
    if (gridOverride > 0) grid = static_cast<unsigned>(gridOverride);
    const uint32_t ldsWords = kolb_image_cells(table, bokeh) ? static_cast<uint32_t>(bokeh.ldsWords) : 0u;
    const size_t lds = static_cast<size_t>(ldsWords + kLutLdsWords + kWavesPerBlock * kListedWaveWords) * sizeof(float);
#define ZOIC_LAUNCH_LISTED(NS_)                                                                                                          \
    hipLaunchKernelGGL((kolb_listed_kernel<NS_>), dim3(grid), dim3(kRefillBlock), lds, st, table, bokeh, d_samples, d_rng, rayBase, m, out, \
                       d_counters, d_redoCursor, ldsWords, 0u, 0u, kMinSearching, d_redoList, d_redoCount, static_cast<unsigned int *>(nullptr))
    switch (table.lensCount) {
    case 7: ZOIC_LAUNCH_LISTED(7); break;
    case 8: ZOIC_LAUNCH_LISTED(8); break;
    case 9: ZOIC_LAUNCH_LISTED(9); break;
    case 10: ZOIC_LAUNCH_LISTED(10); break;
    case 11: ZOIC_LAUNCH_LISTED(11); break;
    case 12: ZOIC_LAUNCH_LISTED(12); break;
    default: ZOIC_LAUNCH_LISTED(0); break;
    }
#undef ZOIC_LAUNCH_LISTED
    return static_cast<int>(hipGetLastError());
}

}  // namespace zoic
