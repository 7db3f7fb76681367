#!/bin/bash
# The kernels kept behind a switch (INTEGRATION.md "measurement switches") against the same oracle comparisons as the defaults (GPU box):
# each line = one switch setting + the test files whose kernels it re-routes.  tests/test_gpu_switches.py holds the radar chain byte for byte
# to its default outputs under the launch-geometry switches; this script covers the alternative kernel forms of the other rows.
set -u
run() { echo "== $1"; env $1 python3 -m pytest ${@:2} -q -x 2>&1 | tail -1; }
run "JRC_DEC_SINGLE=1" tests/test_gpu_codec.py
run "JRC_SYNC_NAIVE=1" tests/test_gpu_sync.py
run "JRC_FD_SERIAL=1" tests/test_gpu_sync.py
run "JRC_SYNC_STREAMS=1" tests/test_gpu_sync.py
run "JRC_SYNC_TILE=1" tests/test_gpu_sync.py
run "JRC_EQ_THREADS=-1" tests/test_gpu_comm.py
run "JRC_EQ_THREADS=128 JRC_EQ_WPE=4" tests/test_gpu_comm.py
run "JRC_EQ_WPE=2" tests/test_gpu_comm.py
run "JRC_DEMOD_SPR=1" tests/test_gpu_chain.py -k time_domain
run "JRC_DEMOD_SPR=2" tests/test_gpu_chain.py -k time_domain
run "JRC_DEMOD_SPR=4" tests/test_gpu_chain.py -k time_domain
run "JRC_NO_WIDE=1" tests/test_gpu_chain.py tests/test_gpu_chain_modes.py
run "JRC_RD_TWO_STEP=1" tests/test_gpu_chain.py -k range_doppler
run "JRC_RD_CHUNK_MB=1" tests/test_gpu_chain.py -k range_doppler
run "JRC_EQ_SIG_FULL=1" tests/test_gpu_comm.py
run "JRC_EQ_SIG_FULL=1" tests/test_gpu_flowgraph_parity.py -k comm
run "JRC_DETECT_SLICES=3" tests/test_gpu_chain_modes.py tests/test_gpu_fuzz.py
run "JRC_DETECT_SLICES=4 JRC_CHANEST_U2=1" tests/test_gpu_chain_modes.py
run "JRC_CHANEST_U2=1" tests/test_gpu_chain.py tests/test_gpu_blocks.py
run "JRC_RADAR_CHAIN_MAX_AGE_US=0" tests/test_host_blocks.py
run "JRC_RADAR_CHAIN_TX_RESIDENT=0" tests/test_host_blocks.py -k radar_chain_block_runs
run "JRC_RADAR_CHAIN_TX_RESIDENT=0" tests/test_host_blocks.py -k several_devices
