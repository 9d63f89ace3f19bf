set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_sq -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin-timing > gpurun_out/pmc_sq.log 2>&1 || true
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1 || true
echo done
