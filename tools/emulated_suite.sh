#!/bin/bash
# The GPU tier on the emulated kernels (tests/hipcpu), whole: the default pass, the heavy shapes, and a pass under AddressSanitizer +
# UndefinedBehaviorSanitizer (which the GPU pool refuses on the device).  Runs on any x86-64 machine with clang++; no GPU.
# Writes gpurun_out/<TAG>_emulated_*.log and the stamped record gpurun_out/<TAG>_emulated_suite.json (copy it to profiles/).
# usage: tools/emulated_suite.sh TAG [workers] [heavy|noheavy] [asan|noasan]
TAG=${1:-r06}
NW=${2:-6}
HEAVY=${3:-heavy}
ASAN=${4:-asan}
mkdir -p gpurun_out
export JRC_EMULATE=1 OMP_NUM_THREADS=1
python3 -m pytest tests -m gpu -v -n $NW --timeout 1800 -p no:cacheprovider > gpurun_out/${TAG}_emulated_default.log 2>&1
RC1=$?
RC2=0; RC3=0; RC4=0; RC5=0; RC6=0
# the same pass with the emulated waves and lanes run backwards, and in a shuffled order: results must not depend on it
HIPCPU_SCHEDULE=reverse python3 -m pytest tests -m gpu -v -n $NW --timeout 1800 -p no:cacheprovider > gpurun_out/${TAG}_emulated_reverse.log 2>&1
RC4=$?
HIPCPU_SCHEDULE=shuffle:2026 python3 -m pytest tests -m gpu -v -n $NW --timeout 1800 -p no:cacheprovider > gpurun_out/${TAG}_emulated_shuffle.log 2>&1
RC5=$?
if [ "$HEAVY" = heavy ]; then
  # the shapes the default pass leaves out (bench batches, 10^6-sample streams).  "Device" memory is host memory here: the config-D batch property tests
  # hold ~37 GB each and run ALONE afterwards (beside another worker they ran this 62 GB machine out of memory, twice)
  HEAVY_K="benchmarked or long_bursts or 1048576 or million_samples or B-1100 or B-300 or B-700 or D-256 or B-512 or 600] or baseline_batch or test_wide_kernel_batches or config_d_eight or detect_slices_beyond or launch_switches"
  JRC_EMULATE_HEAVY=1 python3 -m pytest tests -m gpu -v -n 3 --timeout 5400 -p no:cacheprovider -k "($HEAVY_K) and not baseline_batch[D]" \
    > gpurun_out/${TAG}_emulated_heavy.log 2>&1
  RC2=$?
  JRC_EMULATE_HEAVY=1 python3 -m pytest tests -m gpu -v -n 0 --timeout 5400 -p no:cacheprovider -k "baseline_batch[D]" \
    > gpurun_out/${TAG}_emulated_heavy_config_d_batches.log 2>&1
  RC6=$?
fi
if [ "$ASAN" = asan ]; then
  RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
  JRC_EMULATE_SANITIZE=address,undefined LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=77 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python3 -m pytest tests/test_gpu_blocks.py tests/test_gpu_chain.py tests/test_gpu_chain_modes.py tests/test_gpu_comm.py tests/test_gpu_tsim.py tests/test_gpu_sync.py \
      tests/test_gpu_codec.py tests/test_gpu_edges.py tests/test_gpu_flowgraph.py tests/test_gpu_flowgraph_parity.py tests/test_host_blocks.py tests/test_golden_fixtures.py tests/test_gpu_unvetted.py \
      -m gpu -v -n $NW --timeout 3600 -p no:cacheprovider > gpurun_out/${TAG}_emulated_asan_ubsan.log 2>&1
  RC3=$?
fi
python3 tools/stamp_emulated.py $TAG $RC1 $RC2 $RC3 $RC4 $RC5 $RC6
