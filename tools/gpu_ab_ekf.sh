#!/bin/bash
# A/B of library builds x kernel variants on the headline bench: tools/gpu_ab_ekf.sh "lib:variant" ...   (variant 0 = default)
# prints: value(20-step window) ms/step | long-run value | parity max_abs_diff | updates/pass | frac
for spec in "$@"; do
  lib=${spec%%:*}; var=${spec##*:}
  echo -n "$(basename $lib) variant=$var: "
  SLAM_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-once-per-step --waves-per-filter $var 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print(round(d['value']/1e6,2), d['ms_per_step'], '| long', round(c['steady_state_long_run']['value']/1e6,2), '| full', round(c['full_run_from_init']['value']/1e6,2), '| parity', c['parity_check']['max_abs_diff'], '| upd/pass', r['updates_per_pass'], 'passes/step', r['passes_per_instance_step'], 'frac', r['frac'], r['kernel'], 'flag', c['instances_flagged'])"
done
