cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_gx3 -- python bench.py --workload gx3 --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --no-dropin-timing > gpurun_out/pmc_fetch_gx3.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_l2_gx3 -- python bench.py --workload gx3 --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --no-dropin-timing > gpurun_out/pmc_l2_gx3.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_l2_gx1 -- python bench.py --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --no-dropin-timing > gpurun_out/pmc_l2_gx1.log 2>&1
echo done
