// kolb_pool.hip -- the Kolb launch (kernels.hpp) and the batch + pool kernels (kolb_pool_body.hpp) for cameras WITHOUT
// retry-dead rays (KolbTable::retryOn == 0: no LUT, or every retry can reach the rear element).
#include "kolb_pool_body.hpp"

namespace zoic {

int launch_kolb_pool_dead(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                          uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_cursorPair, unsigned *parity,
                          int mode, uint32_t *d_scratch, void *stream);   // kolb_pool_dead.hip
int launch_kolb_pool_two(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                         uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_cursorPair, unsigned *parity,
                         int mode, uint32_t *d_scratch, void *stream);    // kolb_pool_two.hip

int launch_kolb_rays(const KolbTable &table, const BokehTables &bokeh, const float *d_samples, const uint32_t *d_rng,
                     uint64_t rayBase, uint64_t n, RayRecord *out, DeviceCounters *d_counters, unsigned int *d_cursorPair, unsigned *parity,
                     int mode, uint32_t *d_scratch, void *stream)
{
    if (n == 0) return 0;
    if (table.retryOn && table.twoLevel && kTwoLevelDraws > 0 && !table.useImage)
        return launch_kolb_pool_two(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_cursorPair, parity, mode, d_scratch, stream);
    if (table.retryOn) return launch_kolb_pool_dead(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_cursorPair, parity, mode, d_scratch, stream);
    if (kolb_image_cells(table, bokeh))
        return launch_kolb_pool_impl<false, true>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_cursorPair, parity, mode, d_scratch, stream);
    return launch_kolb_pool_impl<false, false>(table, bokeh, d_samples, d_rng, rayBase, n, out, d_counters, d_cursorPair, parity, mode, d_scratch, stream);
}

#ifdef ZOIC_PASS_STATS
int read_pass_stats_dead(unsigned long long *acc8, int reset);
int read_region_cycles_dead(unsigned long long *acc16, int reset);
int read_pass_stats_two(unsigned long long *acc8, int reset);
int read_region_cycles_two(unsigned long long *acc16, int reset);
#endif

}  // namespace zoic

#ifdef ZOIC_PASS_STATS
extern "C" int zoic_debug_pass_stats(unsigned long long *out8, int reset)
{
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    const int e = zoic::read_pass_stats(out8, reset);
    const int e2 = e ? e : zoic::read_pass_stats_dead(out8, reset);
    return e2 ? e2 : zoic::read_pass_stats_two(out8, reset);
}
extern "C" int zoic_debug_region_cycles(unsigned long long *out16, int reset)
{
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    const int e = zoic::read_region_cycles(out16, reset);
    const int e2 = e ? e : zoic::read_region_cycles_dead(out16, reset);
    return e2 ? e2 : zoic::read_region_cycles_two(out16, reset);
}
#endif
