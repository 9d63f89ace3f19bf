cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/t9full.log 2>&1
grep -E "passed|failed|FAILED|Error|^[0-9.]+s " gpurun_out/t9full.log | head -30
