# round 5, session af: the EKF headline against the length of the timed launch (same first timestep, one launch of K timesteps): what the 20-step command pays per launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5af
for K in 20 40 80 160 320 20; do
python3 bench.py --gpus 1 --steps $K --warmup 5 --no-long-runs --no-parity-check --no-once-per-step --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('K=$K', round(d['value']/1e6,2), 'M  ms/step', d['ms_per_step'], 'launch ms', r.get('kernel_ms'), 'frac', r['frac'], 'passes/step', d['config'].get('passes_per_step'), 'k', d['config'].get('mean_detections_per_step'))"
done | tee gpurun_out/r5af/window_length.txt
