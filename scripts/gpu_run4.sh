set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/t4.log || true
: > gpurun_out/bench_sweep4.json
for wr in "8 1" "4 1" "16 1"; do set -- $wr; python bench.py --steps 20 --warmup 3 --waves $1 --rows $2 --no-thermo --no-cpu-baseline >> gpurun_out/bench_sweep4.json 2>> gpurun_out/b4.err || true; done
for wr in "8 2" "16 2" "4 2" "8 4"; do set -- $wr; python bench.py --workload tenth --steps 2 --warmup 1 --waves $1 --rows $2 --no-thermo --no-cpu-baseline >> gpurun_out/bench_sweep4.json 2>> gpurun_out/b4.err || true; done
# HBM traffic counters, separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_tenth -- python bench.py --workload tenth --steps 1 --warmup 0 --waves 16 --rows 2 --no-thermo --no-cpu-baseline --calibrate > gpurun_out/pmc_fetch_tenth.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_tenth -- python bench.py --workload tenth --steps 1 --warmup 0 --waves 16 --rows 2 --no-thermo --no-cpu-baseline --calibrate > gpurun_out/pmc_write_tenth.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_gx1 -- python bench.py --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --calibrate > gpurun_out/pmc_fetch_gx1.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_gx1 -- python bench.py --steps 1 --warmup 0 --no-thermo --no-cpu-baseline --calibrate > gpurun_out/pmc_write_gx1.log 2>&1 || true
ls gpurun_out/pmc_fetch_tenth/*/ | head
echo done
