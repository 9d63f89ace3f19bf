set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/t5.log || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5 -- python bench.py --steps 3 --warmup 1 --no-thermo --no-cpu-baseline --calibrate > gpurun_out/prof5.log 2>&1 || true
echo done
