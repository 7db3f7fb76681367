#!/bin/bash
# detect-only pipeline (chain.hip chain_run): slices per batch x A1 form, config B / D, target / noise-only frames (GPU box)
for U2 in "" 1; do
for S in ${DETECT_SLICES:-1 2 4 8}; do
  echo "== JRC_DETECT_SLICES=$S JRC_CHANEST_U2=${U2:-0} ${EXTRA_ENV}"
  env JRC_DETECT_SLICES=$S ${U2:+JRC_CHANEST_U2=1} ${EXTRA_ENV} JRC_BENCH_EXTRA_ONLY=${DETECT_LEGS:-detectB,detectB_noise,detectD} python3 tools/bench_extra.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'detect-only' in d['what']:
        print('  %-10s step %.4f ms  %.3f M frames/s  (kernels in series %.4f ms: %s) equal=%s' % (d['what'].split('config ')[1][:1] + (' noise' if 'NOISE' in d['what'] else ' target'), d['ms_per_step'], d['frames_per_s'] / 1e6, d['ms_per_step_kernels_in_series'], {k: round(v, 4) for k, v in d['kernels_ms'].items()}, d['results_equal_map_mode']))
"
done
done
