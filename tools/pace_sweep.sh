#!/bin/bash
# store-pacing sweep of the fused range-angle kernel (on the GPU box): tools/pace_sweep.sh CONFIG[:FRAMES] word [word ...]
CFG=${1%%:*}; FR=${1#*:}; [ "$FR" = "$1" ] && FR=0; shift
for W in "$@"; do
  JRC_RA_PACE=$W python bench.py --config $CFG $( [ $FR != 0 ] && echo --frames $FR ) --no-cpu-baseline --no-secondary --steps 40 --windows 3 --oracle-frames 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$CFG F=$FR pace=$W fused %.4f ms frac %.3f step %.4f ok %s' % (j['kernels_ms']['range_angle_fused'], j['roofline']['frac'], j['windows']['ms_per_step_median'], j['check']['ok']))"
done
