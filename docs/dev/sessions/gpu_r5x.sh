# round 5, session x: fp32 storage on the long-message path (ekf_big_step_kernel<float>), oracle without its per-message switch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_parity_ukf_gpu.py -x -q -m gpu -k "long or any_length" > gpurun_out/r5x/new_tests.txt 2>&1
tail -5 gpurun_out/r5x/new_tests.txt
timeout 400 python3 tools/gpu_soak_adversarial.py 300 61 ekf > gpurun_out/r5x/soak_adv.txt 2>&1; tail -3 gpurun_out/r5x/soak_adv.txt
