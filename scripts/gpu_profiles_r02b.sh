#!/bin/bash
# Round-2 evidence for the resident EVP loop (k_evp_resident): the default bench line, rocprofv3 kernel stats of the same
# command, separate PMC passes at gx1 (FETCH_SIZE / WRITE_SIZE with the calibration stream; SQ counters).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
O=gpurun_out/r02bprof
mkdir -p $O
B="--no-cpu-baseline --no-dropin-timing"
timeout -k 10 500 python bench.py > $O/bench_gx1.json 2> $O/bench_gx1.err
echo bench-done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python bench.py --steps 5 --warmup 1 $B > $O/stats_default.log 2>&1
echo stats-done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_gx1 -- python bench.py --no-tenth --steps 1 --warmup 0 $B --calibrate > $O/pmc_${c}_gx1.log 2>&1
done
echo pmc-done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq1_gx1 -- python bench.py --no-tenth --steps 1 --warmup 0 $B > $O/sq1_gx1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $O/sq2_gx1 -- python bench.py --no-tenth --steps 1 --warmup 0 $B > $O/sq2_gx1.log 2>&1
echo sq-done
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
