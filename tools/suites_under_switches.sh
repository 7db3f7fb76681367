#!/bin/bash
# The kernels kept behind a switch (INTEGRATION.md "measurement switches") against the same oracle comparisons as the defaults (GPU box):
# each line = one switch setting + the test files whose kernels it re-routes.  tests/test_gpu_switches.py holds the radar chain byte for byte
# to its default outputs under the launch-geometry switches; this script covers the alternative kernel forms of the other rows.
set -u
run() { echo "== $1"; env $1 python3 -m pytest ${@:2} -q -x 2>&1 | tail -1; }
run "JRC_DEC_SINGLE=1" tests/test_gpu_codec.py
run "JRC_SYNC_NAIVE=1" tests/test_gpu_sync.py
run "JRC_FD_SERIAL=1" tests/test_gpu_sync.py
run "JRC_EQ_THREADS=-1" tests/test_gpu_comm.py
run "JRC_EQ_THREADS=128 JRC_EQ_WPE=4" tests/test_gpu_comm.py
run "JRC_EQ_WPE=2" tests/test_gpu_comm.py
run "JRC_DEMOD_SPR=1" tests/test_gpu_chain.py -k time_domain
run "JRC_DEMOD_SPR=2" tests/test_gpu_chain.py -k time_domain
run "JRC_DEMOD_SPR=4" tests/test_gpu_chain.py -k time_domain
run "JRC_NO_WIDE=1" tests/test_gpu_chain.py tests/test_gpu_chain_modes.py
run "JRC_RD_TWO_STEP=1" tests/test_gpu_chain.py -k range_doppler
run "JRC_RD_CHUNK_MB=1" tests/test_gpu_chain.py -k range_doppler
