#!/bin/bash
# headline spread: launch series with the clock / power sysfs files sampled beside it, then the default headline three times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/clock_series.log
: > $O
ls /sys/class/drm/ | wc -l >> $O 2>&1
rocm-smi --showclocks --showpower --showtemp --showperflevel >> $O 2>&1
timeout 300 python tools/r6_clock_series.py C3 >> $O 2>&1
for i in 1 2 3; do timeout 120 python bench.py --only-headline 2>/dev/null | tail -1 | cut -c1-260 >> $O; done
SERIES_STEPS=300 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C4 >> $O 2>&1
SERIES_STEPS=300 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C3 strict >> $O 2>&1
SERIES_STEPS=60 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C5 >> $O 2>&1
SERIES_STEPS=2000 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C2 >> $O 2>&1
SERIES_STEPS=4000 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C1 >> $O 2>&1
rocm-smi --showclocks --showpower --showtemp >> $O 2>&1
tail -5 $O
