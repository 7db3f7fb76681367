#!/bin/bash
# range-Doppler pruned-FFT kernel experiments (GPU box): per-kernel times of tools/rd_probe.py under JRC_RD_EXP variants (work-skipping: needs a
# library whose ctx.hip was built with -DJRC_TIMING_EXPERIMENTS, see tools/detect_exp.sh)
# usage: tools/rd_exp.sh [cfg frames]      RD_EXPS="0 1 2 3"
CFG=${1:-D}; F=${2:-16}
REPO=$(pwd); export TMPDIR=/tmp
for E in ${RD_EXPS:-0 1 2 3}; do
  OUT=$REPO/gpurun_out/rdexp_${CFG}_$E; rm -rf $OUT; mkdir -p $OUT
  export JRC_RD_EXP=$E
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $REPO/tools/rd_probe.py $CFG $F > $OUT/stats.log 2>&1 )
  S=$(find $OUT/stats -name '*.db' | head -1)
  python3 profiles/summarize_rocpd.py $OUT/summary.csv "rd exp $E" $S > /dev/null
  find $OUT -name '*.db' -delete
  echo "== JRC_RD_EXP=$E"; grep -E "range_doppler|stockham|rd_product" $OUT/summary.csv | sed -E 's/\(HIP_vector[^"]*"/"/' | cut -c1-120; tail -1 $OUT/stats.log | cut -c1-200
done
