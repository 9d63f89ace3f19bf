set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/t6full.log 2>&1 || true
grep -E "passed|failed|FAILED|Error" gpurun_out/t6full.log | head -20 > gpurun_out/t6.log || true
: > gpurun_out/bench_sweep6.json
for wr in "4 1" "8 1"; do set -- $wr; python bench.py --steps 20 --warmup 3 --waves $1 --rows $2 --no-thermo --no-cpu-baseline --no-dropin-timing >> gpurun_out/bench_sweep6.json 2>> gpurun_out/b6.err || true; done
python bench.py --steps 20 --warmup 3 --waves 4 --rows 1 --no-derive --no-thermo --no-cpu-baseline --no-dropin-timing >> gpurun_out/bench_sweep6.json 2>> gpurun_out/b6.err || true
for wr in "4 2" "8 2" "16 2" "4 4" "8 4" "4 1" "8 1"; do set -- $wr; python bench.py --workload tenth --steps 2 --warmup 1 --waves $1 --rows $2 --no-thermo --no-cpu-baseline --no-dropin-timing >> gpurun_out/bench_sweep6.json 2>> gpurun_out/b6.err || true; done
python bench.py --workload tenth --steps 2 --warmup 1 --waves 4 --rows 2 --no-derive --no-thermo --no-cpu-baseline --no-dropin-timing >> gpurun_out/bench_sweep6.json 2>> gpurun_out/b6.err || true
echo done
