// kolb_pool.hip -- the Kolb launch (kernels.hpp) and the batch + pool kernels (kolb_pool_body.hpp) for cameras WITHOUT
// retry-dead rays (KolbTable::retryOn == 0: no LUT, or every retry can reach the rear element).
#include "kolb_pool_body.hpp"

namespace zoic {

int launch_kolb_pool_dead(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                          uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor,
                          int mode, uint32_t *d_scratch, void *stream);   // kolb_pool_dead.hip

int launch_kolb_rays(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                     uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor,
                     int mode, uint32_t *d_scratch, void *stream)
{
    if (n == 0) return 0;
    if (table.retryOn) return launch_kolb_pool_dead(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_workCursor, mode, d_scratch, stream);
    if (kolb_image_cells(table, bokeh))
        return launch_kolb_pool_impl<false, true>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_workCursor, mode, d_scratch, stream);
    return launch_kolb_pool_impl<false, false>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_workCursor, mode, d_scratch, stream);
}

}  // namespace zoic
