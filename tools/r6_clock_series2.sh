#!/bin/bash
# C3's two states on one box: the series first on the fresh box, again after a minute of heavy frames, with the fabric / memory DPM levels read too
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/clock_series2.log; : > $O
cat /sys/class/drm/card*/device/pp_dpm_fclk 2>/dev/null | head -8 >> $O
for round in 1 2 3; do
  echo "== round $round" >> $O
  SERIES_STEPS=300 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C3 >> $O 2>&1
  SERIES_STEPS=200 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C4 2>&1 | grep -v "^ *-\?[0-9]\+ \+[0-9 ]*$" >> $O
  SERIES_STEPS=100 SERIES_IDLE=0.5 timeout 300 python tools/r6_clock_series.py C5 2>&1 | grep -v "^ *-\?[0-9]\+ \+[0-9 ]*$" >> $O
done
grep "Grays/s:\|== round" $O
