python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -3
mkdir -p gpurun_out/r06_ukf
python tools/gpu_ukf_phases.py 20 12 > gpurun_out/r06_ukf/step_phases_L20.txt 2>&1
python tools/gpu_ukf_phases.py 50 6 > gpurun_out/r06_ukf/step_phases_L50.txt 2>&1
bash tools/pmc_ukf.sh > gpurun_out/r06_ukf/pmc_summary.txt 2>&1
cat gpurun_out/r06_ukf/step_phases_L20.txt | head -30
tail -45 gpurun_out/r06_ukf/pmc_summary.txt
