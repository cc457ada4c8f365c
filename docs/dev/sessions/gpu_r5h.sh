set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py tests/test_soak_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5h/tests.log
timeout 300 python tools/gpu_soak_pgs.py 150 701 > gpurun_out/r5h/soak_pgs.txt 2>&1
# EKF fp32-storage retune (VERDICT r04 item 2 / Weak 6): ring depth, strip height, pass threshold
for lib in base pm4 pm5; do
  echo "== lib_$lib" >> gpurun_out/r5h/f32_variants.txt
  SLAM_HIP_LIB=$PWD/tools/lib_$lib.so timeout 600 python tools/gpu_variants.py f32 1442 1464 1462 1452 1454 >> gpurun_out/r5h/f32_variants.txt 2>&1
done
# what the association of the pre-step costs the control wavefront: the same kernel with the association run 2x / 4x
for rep in 1 2; do
  bash tools/gpu_ab_ekf.sh tools/lib_base.so:0 tools/lib_assoc2.so:0 tools/lib_assoc4.so:0 >> gpurun_out/r5h/assoc_ab.txt 2>&1
done
tail -n 3 gpurun_out/r5h/tests.log; tail -n 2 gpurun_out/r5h/soak_pgs.txt | cut -c 1-400; cat gpurun_out/r5h/f32_variants.txt gpurun_out/r5h/assoc_ab.txt
