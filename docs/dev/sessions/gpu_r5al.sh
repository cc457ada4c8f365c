# round 5, session al: PMC counters of the UKF kernels on the final tree (the measured side of profiles/r05_ukf/lds_bank_model.txt)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5al
bash tools/pmc_ukf.sh > gpurun_out/r5al/pmc_summary.txt 2>&1
grep -A40 "^ukf_sqrt_kernel" gpurun_out/r5al/pmc_summary.txt | head -45
rm -rf gpurun_out/prof_ukf_pmc
