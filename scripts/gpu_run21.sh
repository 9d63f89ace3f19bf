cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/bench_slabs.json
B="--no-thermo --no-cpu-baseline --no-dropin-timing"
python bench.py --steps 10 --warmup 2 $B >> gpurun_out/bench_slabs.json 2>> gpurun_out/b21.err
for ov in 0 8; do
  python bench.py --steps 10 --warmup 2 --slabs 8 --overlap $ov $B >> gpurun_out/bench_slabs.json 2>> gpurun_out/b21.err
  CICE4_AMD_SELF_COMM=1 python bench.py --steps 10 --warmup 2 --slabs 8 --overlap $ov $B >> gpurun_out/bench_slabs.json 2>> gpurun_out/b21.err
  CICE4_AMD_SELF_COMM=1 CICE4_AMD_NO_COMM_GRAPH=1 python bench.py --steps 10 --warmup 2 --slabs 8 --overlap $ov $B >> gpurun_out/bench_slabs.json 2>> gpurun_out/b21.err
done
python bench.py --workload tenth --steps 2 --warmup 1 $B >> gpurun_out/bench_slabs.json 2>> gpurun_out/b21.err
CICE4_AMD_SELF_COMM=1 python bench.py --workload tenth --steps 2 --warmup 1 --slabs 8 --overlap 8 $B >> gpurun_out/bench_slabs.json 2>> gpurun_out/b21.err
CICE4_AMD_SELF_COMM=1 python bench.py --workload tenth --steps 2 --warmup 1 --slabs 8 --overlap 0 $B >> gpurun_out/bench_slabs.json 2>> gpurun_out/b21.err
echo done
