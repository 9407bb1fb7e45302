// kolb_pool_dead.hip -- the batch + pool Kolb kernels for cameras WITH retry-dead rays (KolbTable::retryOn: off-axis pixels
// whose 26 retries all miss the rear element, tables.hpp): they complete those rays inside the kernel, 64 at a time.  A
// translation unit of its own: the two sets of kernels compile side by side, and cameras without such rays carry none of it.
#include "kolb_pool_body.hpp"

namespace zoic {

int launch_kolb_pool_dead(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                          uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_cursorPair, unsigned *parity,
                          int mode, uint32_t *d_scratch, void *stream)
{
    if (kolb_image_cells(table, bokeh))
        return launch_kolb_pool_impl<true, true>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_cursorPair, parity, mode, d_scratch, stream);
    return launch_kolb_pool_impl<true, false>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_cursorPair, parity, mode, d_scratch, stream);
}

#ifdef ZOIC_PASS_STATS
int read_pass_stats_dead(unsigned long long *acc8, int reset) { return read_pass_stats(acc8, reset); }
int read_region_cycles_dead(unsigned long long *acc16, int reset) { return read_region_cycles(acc16, reset); }
#endif

}  // namespace zoic
