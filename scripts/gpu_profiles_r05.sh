#!/bin/bash
# Round-5 evidence for profiles/ (taken at the commit named in gpurun_out/r05prof/commit.txt; every CSV also carries the hash of
# the kernel sources it was taken from, bench.kernel_source_sha, which bench.py and tests/test_profiles.py check against the tree):
#   * the default bench line;
#   * rocprofv3 --kernel-trace --stats, ONE RUN PER WORKLOAD (gx1 / 0.1 degree full cover / 0.1 degree polar caps), so that the
#     dominant kernel's average duration is the headline workload's and no other's;
#   * separate PMC passes (FETCH_SIZE / WRITE_SIZE with the 8-byte-lane calibration stream; two SQ counter sets) for gx1 and 0.1 degree.
# usage: gpu_profiles_r05.sh [part]   part = all | stats | pmc | sq   (the parts fit one gpurun call each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PART=${1:-all}
O=gpurun_out/r05prof
mkdir -p $O
cp commit.txt $O/commit.txt 2>/dev/null || echo unknown > $O/commit.txt
python -c "import bench; print(bench.kernel_source_sha())" > $O/source_sha.txt
B="--no-cpu-baseline --no-dropin-timing"
if [ $PART = all ] || [ $PART = stats ]; then
  timeout -k 10 500 python bench.py > $O/bench_gx1.json 2> $O/bench_gx1.err || echo "bench failed"
  echo bench-done
  rm -rf $O/stats_gx1 $O/stats_tenth_full $O/stats_tenth_caps
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gx1 -- python bench.py --steps 5 --warmup 1 --no-tenth --no-thermo $B > $O/stats_gx1.log 2>&1 || echo "stats gx1 failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gx1_thermo -- python bench.py --steps 5 --warmup 1 --no-tenth $B > $O/stats_gx1_thermo.log 2>&1 || echo "stats gx1 thermo failed"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tenth_full -- python bench.py --workload tenth --steps 3 --warmup 1 --no-thermo $B > $O/stats_tenth_full.log 2>&1 || echo "stats tenth full failed"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_tenth_caps -- python bench.py --workload tenth --cover caps --steps 3 --warmup 1 --no-thermo $B > $O/stats_tenth_caps.log 2>&1 || echo "stats tenth caps failed"
  echo stats-done
fi
for wl in gx1 tenth; do
  X="--no-tenth"; [ $wl = tenth ] && X="--workload tenth"
  if [ $PART = all ] || [ $PART = pmc ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $O/pmc_${c}_$wl
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$wl -- python bench.py $X --steps 1 --warmup 0 --ramp-seconds 0 $B --calibrate > $O/pmc_${c}_$wl.log 2>&1 || echo "pmc $c $wl failed"
    done
    echo pmc-$wl-done
  fi
  if [ $PART = all ] || [ $PART = sq ]; then
    i=0
    for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"; do
      i=$((i+1))
      rm -rf $O/sq${i}_$wl
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq${i}_$wl -- python bench.py $X --steps 1 --warmup 0 --ramp-seconds 0 $B > $O/sq${i}_$wl.log 2>&1 || echo "sq$i $wl failed"
    done
    echo sq-$wl-done
  fi
done
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
