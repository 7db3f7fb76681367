#!/bin/bash
# detect-only kernel timing experiments (GPU box): JRC_DETECT_EXP variants of tools/bench_extra.py's detect legs.  Bits 1 / 2 / 32 skip work and
# exist only in a library whose ctx.hip was built with -DJRC_TIMING_EXPERIMENTS: tools/ra_variants.py build timing:ctx.hip:-DJRC_TIMING_EXPERIMENTS,
# then JRC_LIB_PATH=gr-mimo-ofdm-jrc_amd/lib/variants/timing/libjrc_hip.so
for E in ${DETECT_EXPS:-0 16 8}; do
  echo "== JRC_DETECT_EXP=$E"
  JRC_DETECT_EXP=$E JRC_BENCH_EXTRA_ONLY=detect python3 tools/bench_extra.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'detect-only' in d['what']:
        print('  %-28s step %.4f ms  %.3f M frames/s  kernels %s equal=%s' % (d['what'].split('config ')[1][:1] + (' noise' if 'NOISE' in d['what'] else ' target'), d['ms_per_step'], d['frames_per_s'] / 1e6, {k: round(v, 4) for k, v in d['kernels_ms'].items()}, d['results_equal_map_mode']))
"
done
