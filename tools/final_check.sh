#!/bin/bash
# Evidence at the final tree: run on the GPU box via gpurun as the LAST thing after any kernel change.
#   1. the WHOLE `-m gpu` suite, verbose (the log shows the parity files first), stamped with build.source_hash() -> gpurun_out/<TAG>_gpu_suite.json
#   2. the default bench command (what the driver runs): stdout line (compact) + the verbose record
#   3. the flowgraph parity record the suite wrote
# Copy gpurun_out/<TAG>_* into profiles/ afterwards (tests/test_evidence_stamps.py checks the stamp against the tree).
# usage: tools/final_check.sh TAG [pytest timeout seconds]
TAG=${1:-r06}
LIMIT=${2:-900}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout $LIMIT python3 -m pytest tests -m gpu -v -x -p no:cacheprovider --durations=12 > gpurun_out/${TAG}_gpu_suite.log 2>&1
RC=$?
python3 tools/stamp_suite.py gpurun_out/${TAG}_gpu_suite.log $RC gpurun_out/${TAG}_gpu_suite.json
cp gpurun_out/flowgraph_parity.json gpurun_out/${TAG}_flowgraph_parity.json 2>/dev/null
tail -n 25 gpurun_out/${TAG}_gpu_suite.log | cut -c1-300
timeout 600 python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc $? line bytes $(wc -c < gpurun_out/${TAG}_bench_line.json)"
cp gpurun_out/bench_verbose.json gpurun_out/${TAG}_bench_verbose.json 2>/dev/null
cat gpurun_out/${TAG}_bench_line.json
exit $RC
