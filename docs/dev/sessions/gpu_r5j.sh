# round 5, session j: over-long message routing + quirk switches, adversarial soak, the profiles of the round (EKF driver command: kernel trace + PMC; pose graph: kernel trace + PMC)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "over_long or quirk or streamed" 2>&1 | tail -5 > gpurun_out/r5j/tests.log
timeout 300 python tools/gpu_soak_adversarial.py 150 801 both > gpurun_out/r5j/soak_adversarial.txt 2>&1
tail -n 3 gpurun_out/r5j/tests.log; tail -n 3 gpurun_out/r5j/soak_adversarial.txt | cut -c 1-500
bash tools/profile.sh r05a 20 5 > gpurun_out/r5j/profile.log 2>&1; tail -25 gpurun_out/r5j/profile.log
mkdir -p gpurun_out/r05a; cp gpurun_out/prof_r05a/summary.json gpurun_out/prof_r05a/summary.txt gpurun_out/r05a/ 2>/dev/null
cp $(find gpurun_out/prof_r05a/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r05a/kernel_stats.csv
python tools/gpu_phases.py f64 > gpurun_out/r05a/phases.txt 2>&1
rm -rf gpurun_out/prof_r05a
bash tools/profile_pgs.sh r05_pgs > gpurun_out/r5j/profile_pgs.log 2>&1
python3 tools/summarize_pgs_profile.py gpurun_out/prof_r05_pgs gpurun_out/r05_pgs > gpurun_out/r5j/summarize_pgs.log 2>&1
rm -rf gpurun_out/prof_r05_pgs
cat gpurun_out/r05_pgs/summary.txt | head -40
