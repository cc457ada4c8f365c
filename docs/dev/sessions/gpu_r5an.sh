# round 5, session an: fp32 storage beyond the LDS classes (the streamed kernel with float storage for a whole handle)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5an
timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_capi.py -x -q -m gpu -k "fp32_storage_beyond or long or bad_arguments or state_of" > gpurun_out/r5an/tests.txt 2>&1; tail -3 gpurun_out/r5an/tests.txt
timeout 300 python3 tools/gpu_soak_adversarial.py 200 8101 ekf > gpurun_out/r5an/soak_adv.txt 2>&1; tail -2 gpurun_out/r5an/soak_adv.txt | cut -c1-300
