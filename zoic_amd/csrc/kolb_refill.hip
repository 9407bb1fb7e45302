// kolb_refill.hip -- the Kolb launch (kernels.hpp) and the kernels for cameras WITHOUT retry-dead rays (KolbTable::retryOn
// == 0: no LUT, or every retry can reach the rear element).  The kernels themselves: kolb_refill_body.hpp.
#include "kolb_refill_body.hpp"

namespace zoic {

int launch_kolb_refill_dead(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                            uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor,
                            int mode, uint32_t *d_scratch, void *stream);   // kolb_refill_dead.hip
#ifdef ZOIC_REGION_TIMERS
int read_region_debug_dead(unsigned long long *acc8, bool passStats, int reset);
int read_wave_log_dead(unsigned long long *out);
#endif

int launch_kolb_refill(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                       uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_workCursor,
                       int mode, uint32_t *d_scratch, void *stream)
{
    if (table.retryOn) return launch_kolb_refill_dead(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_workCursor, mode, d_scratch, stream);
    return launch_kolb_refill_impl<false>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_workCursor, mode, d_scratch, stream);
}

}  // namespace zoic

#ifdef ZOIC_REGION_TIMERS
extern "C" int zoic_debug_pass_stats(unsigned long long *out8, int reset)
{
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    const int e = zoic::read_region_debug(out8, true, reset);
    return e ? e : zoic::read_region_debug_dead(out8, true, reset);
}
// per-wave timeline of the last launch: 8192 x {start, exhausted, end, passes}; dead != 0: the kernels of kolb_refill_dead.hip
extern "C" int zoic_debug_wave_log(unsigned long long *out, int dead) { return dead ? zoic::read_wave_log_dead(out) : zoic::read_wave_log(out); }
extern "C" int zoic_debug_region_cycles(unsigned long long *out8, int reset)
{
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    const int e = zoic::read_region_debug(out8, false, reset);
    return e ? e : zoic::read_region_debug_dead(out8, false, reset);
}
#endif
