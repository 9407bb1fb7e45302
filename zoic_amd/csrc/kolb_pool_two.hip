// kolb_pool_two.hip -- the batch + pool Kolb kernels with the TWO-LEVEL retry search (kolb_pool_body.hpp kTwoLevelDraws), for cameras with
// retry-dead rays whose remaining retries are mostly rejectable per draw (KolbTable::twoLevel, lens_system.cpp fill_table: the TESSAR at
// 10 cm yes, the wide-open PETZVAL no).  Disk sampler only.  A translation unit of its own, like kolb_pool_dead.hip.
#include "kolb_pool_body.hpp"

namespace zoic {

int launch_kolb_pool_two(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                         uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_cursorPair, unsigned *parity,
                         int mode, uint32_t *d_scratch, void *stream)
{
#if ZOIC_TWO_LEVEL_SEARCH > 0
    return launch_kolb_pool_impl<true, false, true>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_cursorPair, parity, mode, d_scratch, stream);
#else   // compiled out (-DZOIC_TWO_LEVEL_SEARCH=0): kolb_pool.hip never comes here
    (void)table; (void)bokeh; (void)d_samples; (void)d_rng; (void)rayBase; (void)n; (void)out; (void)d_counters; (void)d_cursorPair; (void)parity; (void)mode; (void)d_scratch; (void)stream;
    return static_cast<int>(hipErrorNotSupported);
#endif
}

#ifdef ZOIC_PASS_STATS
int read_pass_stats_two(unsigned long long *acc8, int reset) { return read_pass_stats(acc8, reset); }
int read_region_cycles_two(unsigned long long *acc16, int reset) { return read_region_cycles(acc16, reset); }
#endif

}  // namespace zoic
