#!/bin/bash
# Round 6's GPU session in ONE bounded call (the pool was closed to this repository when the round began; this is what runs the moment it opens):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'tools/round6_gpu_plan.sh r06_a'
# Order = value of the evidence: (1) the whole -m gpu suite, stamped; (2) the default bench line; (3) the kernels written without a GPU, vetted and
# timed against the defaults; (4) rocprof of the headline; (5) LAST, because it may hang the device: the 8-rank same-device study.
TAG=${1:-r06_a}
mkdir -p gpurun_out
export TMPDIR=/tmp
# (1) + (2)
tools/final_check.sh $TAG 1000 > gpurun_out/${TAG}_final_check.txt 2>&1
echo "final_check rc $?" | tee -a gpurun_out/${TAG}_plan.txt
# (3) unvetted kernels: their tests on the device (own pytest run: a failure here does not touch the suite's record), then A/B timings
JRC_TEST_UNVETTED=1 timeout 600 python3 -m pytest tests/test_gpu_unvetted.py -m gpu -v -p no:cacheprovider > gpurun_out/${TAG}_unvetted.log 2>&1
echo "unvetted rc $? $(tail -1 gpurun_out/${TAG}_unvetted.log)" | tee -a gpurun_out/${TAG}_plan.txt
for F in 64 256; do
  for M in 0 1; do
    echo "device-resident flowgraph config B, $F packets per pass, JRC_DRF_FUSED_MOD=$M: $(JRC_DRF_FUSED_MOD=$M timeout 300 python3 tools/drf_probe.py $F 2>&1 | tail -1)" | tee -a gpurun_out/${TAG}_plan.txt
  done
done
for F in 64 512; do
  for SW in "JRC_DRF_FUSED_MOD=0 JRC_TSIM_ONCHIP=0" "JRC_DRF_FUSED_MOD=1 JRC_TSIM_ONCHIP=0" "JRC_DRF_FUSED_MOD=0 JRC_TSIM_ONCHIP=1" "JRC_DRF_FUSED_MOD=1 JRC_TSIM_ONCHIP=1"; do
    echo "device-resident flowgraph at the .grc's geometry, $F packets per pass, $SW: $(env $SW timeout 300 python3 tools/drf_probe.py $F grc 2>&1 | tail -1)" | tee -a gpurun_out/${TAG}_plan.txt
  done
done
# (4)
timeout 900 tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.txt 2>&1
echo "profile_round rc $?" | tee -a gpurun_out/${TAG}_plan.txt
# (5)
timeout 1500 tools/hang_bisect.sh "capped8:8:8:- two:2:3:- four:4:3:- uncapped8:8:6:GPU_MAX_HW_QUEUES=4" > gpurun_out/${TAG}_hang_bisect.txt 2>&1
echo "hang_bisect rc $?" | tee -a gpurun_out/${TAG}_plan.txt
cat gpurun_out/${TAG}_plan.txt
tail -5 gpurun_out/${TAG}_hang_bisect.txt
