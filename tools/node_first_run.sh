#!/bin/bash
# The first run on a multi-GPU node (nothing here has ever crossed xGMI - the builder's boxes have one GPU).
# usage: tools/node_first_run.sh [N ...]      (default: 2 8; needs N visible MI355X)
# 1. tools/xchg_selftest.py under torch.distributed.run: the LDS-free peer exchange (hipIpc-mapped blocks) against the process group's own all_reduce;
# 2. bench.py --gpus N (the driver's form): config 4 = N x 4096 envs, DDPG with the per-update gradient all-reduce;
# and a table of what each line says about the exchange that ran, the replica synchronisation and the replicas' weight spread.
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/node_first_run; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
sizes="${@:-2 8}"
for n in $sizes; do
  echo "== $n ranks: peer-exchange self-test"
  KS_DIST_BACKEND=nccl python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) tools/xchg_selftest.py > $out/xchg_$n.log 2>&1
  echo "   rc $? : $(tail -1 $out/xchg_$n.log)"
  echo "== $n GPUs: bench.py (the driver's form)"
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 20 --warmup 5 > $out/bench_$n.log 2>&1
  echo "   rc $?"
done
python3 - "$out" $sizes <<'PY'
import json, sys
out, sizes = sys.argv[1], sys.argv[2:]
print(f"{'GPUs':>4} {'env-steps/s':>12} {'ms/step':>8}  exchange | replica_sync | launch_chunk | weight spread | dropped episodes")
for n in sizes:
    try:
        line = [l for l in open(f"{out}/bench_{n}.log") if l.startswith("{")][-1]
        d = json.loads(line); r = d.get("rccl") or {}
        print(f"{n:>4} {d['value']:12.0f} {d['ms_per_step']:8.3f}  {r.get('exchange')} | {r.get('replica_sync')} | {r.get('launch_chunk')} | "
              f"{d.get('replica_weight_checksum_spread')} | {(d['config'].get('free_running') or {}).get('episodes_dropped')}")
    except Exception as e:
        print(f"{n:>4} no bench line ({e}); see {out}/bench_{n}.log")
PY
