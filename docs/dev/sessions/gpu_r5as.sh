# round 5, session as: the EKF headline with three wavefronts per filter (155 registers per lane, no scratch; the default four have 128 and 64 B of scratch) - variant codes 1364 / 1354 against the default
# (library built with SLAM_EXTRA_VARIANTS="103,3,6,4,0,1;103,3,5,4,0,1")
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5as
for i in 1 2; do
for v in 0 1364 1354; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --waves-per-filter $v --no-secondary --no-cpu-baseline --no-once-per-step 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant $v:', round(d['value']/1e6,2), 'M frac', r['frac'], 'steady', round((d['config'].get('steady_state_value') or 0)/1e6,1), 'parity', d['config'].get('parity_check',{}).get('max_abs_diff_vs_oracle'), r.get('kernel'))"
done
done | tee gpurun_out/r5as/w3.txt
