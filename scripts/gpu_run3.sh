set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/t3.log || true
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_gx1.json 2> gpurun_out/bench_gx1.err || true
: > gpurun_out/bench_sweep.json
for wr in "8 1" "4 1" "4 2" "16 1" "8 2"; do set -- $wr; python bench.py --steps 10 --warmup 2 --waves $1 --rows $2 --no-thermo --no-cpu-baseline >> gpurun_out/bench_sweep.json 2>> gpurun_out/bench_gx1.err || true; done
python bench.py --steps 5 --warmup 2 --no-graph --no-thermo --no-cpu-baseline >> gpurun_out/bench_sweep.json 2>> gpurun_out/bench_gx1.err || true
: > gpurun_out/bench_tenth_sweep.json
for wr in "4 4" "4 8" "8 4" "8 2" "4 2" "16 2"; do set -- $wr; python bench.py --workload tenth --steps 2 --warmup 1 --waves $1 --rows $2 --no-thermo --no-cpu-baseline >> gpurun_out/bench_tenth_sweep.json 2>> gpurun_out/bench_tenth.err || true; done
echo done
