#!/bin/bash
# the driver's N > 1 command (torchrun) with both ranks on the one GPU of this box, shared-memory link instead of RCCL
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CICE4_AMD_BENCH_DEVICE=0 CICE4_AMD_BENCH_LINK=shm timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/torchrun2.json 2> gpurun_out/torchrun2.err
echo "rc=$?"
grep -a "cross-rank\|attempt\|verif" gpurun_out/torchrun2.err | tail -6
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/torchrun2.json") if l.startswith("{")][-1])
o = d.get("other_decomposition", {})
print("value %.0f  decomposition: %s" % (d["value"], d["config"]["decomposition"][:90]))
print("other %.0f  used=%s" % (o.get("value", 0), o.get("used")))
print("tenth %.0f (%s)" % (d["tenth"]["value"], d["tenth"]["config"]["decomposition"][:80]))
PY
