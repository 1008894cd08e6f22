#!/bin/bash
# Round 6: the multi-GPU path's baseline on the one-GPU box — two ranks sharing cuda:0 (`bench.py --gpus 2 --share-gpu`), every gather
# mode, with `gather_verified` and what the gather costs the step (against `--gather none`).  Two ranks on one device cannot form an RCCL communicator ("Duplicate GPU
# detected"): the collective runs over gloo here — plumbing and bookkeeping, not xGMI; libfmdgather.so's same-device rotation is covered by tests/test_multi_gpu_host.py below.  No scaling figure: the first real 8-GPU run has
# these lines to diverge from.  Output: gpurun_out/r6_two_rank/.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export GPU_MAX_HW_QUEUES=8 HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6_two_rank; rm -rf $O; mkdir -p $O
P=29540
for mode in none rotate root all; do
  for ch in 2048 4096; do
    P=$((P + 1))
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --share-gpu --backend gloo --channels $ch --gather $mode --steps 40 --warmup 3 \
      2> $O/err_${mode}_$ch.txt | tail -1 > $O/line_${mode}_$ch.json
    python3 - $O/line_${mode}_$ch.json $mode $ch <<'PY' | tee -a $O/summary.jsonl
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g = d.get("gather") or {}
    print(json.dumps({"gather": sys.argv[2], "channels_per_rank": int(sys.argv[3]), "n_gpus_claimed": d["n_gpus"], "value_msa_s": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4),
                      "gather_verified": d.get("gather_verified"), "bytes_per_rank_per_step": g.get("bytes_per_rank_per_step"), "format": g.get("format"),
                      "gb_per_s_per_link_at_this_rate": g.get("gb_per_s_per_link_at_this_rate"), "workload": d["config"]["workload"][:120]}))
except Exception as e:
    print(json.dumps({"gather": sys.argv[2], "channels_per_rank": int(sys.argv[3]), "error": str(e)[:200]}))
PY
  done
done
# the C++ host (libfmdgather.so, RCCL point-to-point / same-device copies): the multi-GPU host's own driver, ranks sharing the device
python3 -m pytest tests/test_multi_gpu_host.py -q -m gpu 2>&1 | tail -3 | tee $O/multi_gpu_host_tests.txt
rm -f $O/err_*.txt
