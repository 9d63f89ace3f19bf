#!/bin/bash
# full GPU validation + the numbers and profiles that go into profiles/ and DESIGN.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
mkdir -p gpurun_out
B="--no-cpu-baseline --no-dropin-timing"
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1 || { grep -v "^ " gpurun_out/final_tests.log | tail -40; exit 1; }
grep -E "passed|failed" gpurun_out/final_tests.log | tail -2
timeout -k 10 500 python bench.py > gpurun_out/final_bench_gx1.json 2> gpurun_out/final_bench_gx1.err
timeout -k 10 500 python bench.py --workload tenth --steps 3 --warmup 1 --cpu-seconds 4 > gpurun_out/final_bench_tenth.json 2> gpurun_out/final_bench_tenth.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof_gx1 -- python bench.py --steps 5 --warmup 1 $B > gpurun_out/final_prof_gx1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof_tenth -- python bench.py --workload tenth --steps 1 --warmup 0 $B > gpurun_out/final_prof_tenth.log 2>&1
for wl in gx1 tenth; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/final_pmc_${c}_$wl -- python bench.py --workload $wl --steps 1 --warmup 0 --no-thermo $B --calibrate > gpurun_out/final_pmc_${c}_$wl.log 2>&1
  done
done
echo final-done
