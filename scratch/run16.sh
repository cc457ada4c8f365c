mkdir -p gpurun_out/r06b
python tools/gpu_soak_pgs.py 300 21 > gpurun_out/r06b/soak_pgs.txt 2>&1; tail -2 gpurun_out/r06b/soak_pgs.txt
python tools/gpu_soak_pgs.py 240 22 big > gpurun_out/r06b/soak_pgs_big.txt 2>&1; tail -2 gpurun_out/r06b/soak_pgs_big.txt
python tools/gpu_soak_pgs.py 240 23 wide > gpurun_out/r06b/soak_pgs_wide.txt 2>&1; tail -2 gpurun_out/r06b/soak_pgs_wide.txt
python tools/gpu_soak_pgs_api.py 200 24 > gpurun_out/r06b/soak_pgs_api.txt 2>&1; tail -2 gpurun_out/r06b/soak_pgs_api.txt
python tools/gpu_soak_ekf.py 200 25 > gpurun_out/r06b/soak_ekf.txt 2>&1; tail -2 gpurun_out/r06b/soak_ekf.txt
python tools/gpu_soak_api.py 200 26 > gpurun_out/r06b/soak_api.txt 2>&1; tail -2 gpurun_out/r06b/soak_api.txt
python tools/gpu_soak_adversarial.py 200 27 > gpurun_out/r06b/soak_adversarial.txt 2>&1; tail -2 gpurun_out/r06b/soak_adversarial.txt
