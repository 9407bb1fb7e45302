// kolb_refill_dead.hip -- the Kolb kernels for cameras WITH retry-dead rays (KolbTable::retryOn: off-axis pixels whose 26
// retries all miss the rear element, tables.hpp) + the finish kernel that completes them.  A translation unit of its own:
// the two sets of 28 kernels compile side by side, and the cameras without such rays run kernels that carry none of this.
#include "kolb_refill_body.hpp"

namespace zoic {

int launch_kolb_refill_dead(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                            uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor,
                            int mode, uint32_t *d_scratch, void *stream)
{
    return launch_kolb_refill_impl<true>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_workCursor, mode, d_scratch, stream);
}

#ifdef ZOIC_REGION_TIMERS
int read_region_debug_dead(unsigned long long *acc8, bool passStats, int reset) { return read_region_debug(acc8, passStats, reset); }
int read_wave_log_dead(unsigned long long *out) { return read_wave_log(out); }
#endif

}  // namespace zoic
