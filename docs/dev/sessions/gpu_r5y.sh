# round 5, session y: where the L = 50 UKF step goes (step-kernel phases at n up to 104; the kernel is 44 % of the GPU time there)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5y
python3 tools/gpu_ukf_phases.py 50 12 > gpurun_out/r5y/step_phases_L50.txt 2>&1; cat gpurun_out/r5y/step_phases_L50.txt | head -40
python3 bench.py --filter ukf --landmarks 50 --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('L50', d['value'], d['ms_per_step'])"
