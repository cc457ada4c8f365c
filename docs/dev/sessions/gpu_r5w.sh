# round 5, session w: GPU suite + the soaks whose paths the long-message launch pairs touch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5w
timeout 2400 python3 -m pytest tests -q -m gpu -rs > gpurun_out/r5w/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5w/pytest.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r5w/pytest.log | tail -4
timeout 300 python3 tools/gpu_soak_ekf.py 150 2201 both > gpurun_out/r5w/soak_ekf.txt 2>&1; tail -n 1 gpurun_out/r5w/soak_ekf.txt | cut -c 1-300
timeout 300 python3 tools/gpu_soak_api.py 120 2202 > gpurun_out/r5w/soak_api.txt 2>&1; tail -n 1 gpurun_out/r5w/soak_api.txt | cut -c 1-300
timeout 300 python3 tools/gpu_soak_adversarial.py 150 2203 ukf > gpurun_out/r5w/soak_adv_ukf.txt 2>&1; tail -n 1 gpurun_out/r5w/soak_adv_ukf.txt | cut -c 1-300
