set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/t2.log || true
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || true
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_gx1.json 2> gpurun_out/bench_gx1.err || true
python bench.py --steps 5 --warmup 2 --no-graph --no-thermo --no-cpu-baseline > gpurun_out/bench_gx1_nograph.json 2>> gpurun_out/bench_gx1.err || true
for t in 8 16 32; do python bench.py --steps 10 --warmup 2 --tile-rows $t --no-thermo --no-cpu-baseline >> gpurun_out/bench_tiles.json 2>> gpurun_out/bench_gx1.err || true; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_gx1 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_gx1.log 2>&1 || true
python bench.py --workload tenth --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_tenth.json 2> gpurun_out/bench_tenth.err || true
echo done
