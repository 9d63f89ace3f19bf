cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/t12full.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/t12full.log | head -5 > gpurun_out/t12.log
python bench.py > gpurun_out/bench_final_gx1.json 2> gpurun_out/bench_final_gx1.err
python bench.py --workload tenth --steps 3 --warmup 1 --cpu-seconds 4 > gpurun_out/bench_final_tenth.json 2> gpurun_out/bench_final_tenth.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final_gx1 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-dropin-timing > gpurun_out/prof_final_gx1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final_tenth -- python bench.py --workload tenth --steps 1 --warmup 0 --no-cpu-baseline --no-dropin-timing > gpurun_out/prof_final_tenth.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_final_tenth -- python bench.py --workload tenth --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --no-dropin-timing --calibrate > gpurun_out/pmcf_final_tenth.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_final_tenth -- python bench.py --workload tenth --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --no-dropin-timing --calibrate > gpurun_out/pmcw_final_tenth.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_final_gx1 -- python bench.py --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --no-dropin-timing --calibrate > gpurun_out/pmcf_final_gx1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_final_gx1 -- python bench.py --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --no-dropin-timing --calibrate > gpurun_out/pmcw_final_gx1.log 2>&1
echo done
