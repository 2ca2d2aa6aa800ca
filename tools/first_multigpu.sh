#!/bin/bash
# tools/first_multigpu.sh -- the FIRST run on a node with more than one GPU: everything of SURVEY.md 8(e) that has never executed
# on real peers (one GPU per lease until now), in one command.  Nothing here is a measurement the repository claims; it is the kit
# that turns the first multi-GPU lease into one.  Output: gpurun_out/multigpu/*.json|log and a summary on stdout.
#   1. the seven tests of tests/test_gpu_multidevice.py (skipped below 2 GPUs: peer loads of a range-split index, femto_amd_comm_gather
#      with N ranks, replicated / striped multi-device handles on distinct devices)
#   2. bench.py --gpus N for N in 1 2 4 8 (replicated index, RCCL gather; both gather forms and the no-gather step are extras of
#      every N > 1 run): T1 / (N x TN) per N, per-rank search / stall / payload
#   3. cfg 5 (8 GiB text) on N GPUs: --layout replicated vs split vs striped (direct peer loads over xGMI)
#   4. tools/exchange_bench.py: range-split locate by walker exchange vs the replicated handle (1 GiB and, with --cfg5, 8 GiB)
# Usage: bash tools/first_multigpu.sh [--cfg5] [--steps K]
set -u
CFG5=0; STEPS=20
while [ $# -gt 0 ]; do case "$1" in --cfg5) CFG5=1;; --steps) STEPS=$2; shift;; esac; shift; done
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/multigpu; mkdir -p $O
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NG"
[ "$NG" -lt 2 ] && echo "one GPU: the multi-device tests skip and every N > 1 run below shares this device over gloo (control flow only)"
python -m pytest tests/test_gpu_multidevice.py -q -m gpu -rs 2>&1 | tail -12 | tee $O/tests.log
run() { # N extra-args...
  N=$1; shift
  if [ "$N" -le 1 ]; then python bench.py --gpus 1 --steps $STEPS --warmup 5 --no-extra "$@"
  elif [ "$N" -le "$NG" ]; then python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps $STEPS --warmup 5 "$@"
  else  # fewer GPUs than ranks: the ranks share them, the gather goes through gloo, a 16 MiB text -- control flow only, never a measurement
    FEMTO_AMD_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 2 --warmup 1 --text-log2 24 --npats 100000 --cpu-sample 2000 "$@"
  fi
}
for N in 1 2 4 8; do
  run $N > $O/replicated_n$N.json 2> $O/replicated_n$N.err
  run $N --gather native > $O/native_n$N.json 2> $O/native_n$N.err
done
python - <<'PY'
import json, glob, os
O = "gpurun_out/multigpu"
def last(path):
    try:
        lines = [l for l in open(path) if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None
t1 = None
print("N  gather  G patterns/s  ms/step  T1/(N*TN)  verified  per-rank search ms / gather stall ms / payload MB")
for N in (1, 2, 4, 8):
    for g in ("replicated", "native"):
        d = last(f"{O}/{g}_n{N}.json")
        if not d:
            print(N, g, "no result line (see the .err file)")
            continue
        if N == 1 and g == "replicated":
            t1 = d["ms_per_step"]
        dry = "16 MiB" in json.dumps(d["config"].get("index", {})) or d["config"].get("text_bytes", 1 << 30) < (1 << 30)
        eff = (t1 / d["ms_per_step"]) if (t1 and not dry) else None      # weak scaling: per-GPU work fixed, so T1 / TN (dry runs on a shared GPU: no figure)
        pr = d["config"].get("per_rank") or []
        print(N, g, round(d["value"] / 1e9, 2), round(d["ms_per_step"], 3), None if eff is None else round(eff, 3), d["config"].get("gathered_results_verified"),
              [(round(r["search_ms_per_step"], 3), round(r["gather_stall_ms_per_step"], 3), round(r["gather_payload_bytes"] / 1e6, 1)) for r in pr])
PY
if [ "$NG" -ge 2 ]; then
  for N in 2 4 8; do [ "$N" -le "$NG" ] || continue
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700 + N)) tools/exchange_bench.py > $O/exchange_n$N.json 2> $O/exchange_n$N.err
    grep '^{' $O/exchange_n$N.json | tail -1 | cut -c1-600
  done
else
  python tools/exchange_bench.py --npats 500000 > $O/exchange_n1.json 2> $O/exchange_n1.err; grep '^{' $O/exchange_n1.json | tail -1 | cut -c1-600
fi
if [ "$CFG5" = 1 ]; then
  for L in replicated split striped; do
    for N in 2 4 8; do [ "$N" -le "$NG" ] || continue
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29800 + N)) bench.py --gpus $N --steps 6 --warmup 2 \
        --workload acgt_hit --text-log2 33 --layout $L --cpu-sample 0 > $O/cfg5_${L}_n$N.json 2> $O/cfg5_${L}_n$N.err
      python -c "
import json,sys
l=[x for x in open('$O/cfg5_${L}_n$N.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('cfg5', '$L', $N, 'GPUs:', None if not d else (round(d['value']/1e9,2), 'G patterns/s', round(d['ms_per_step'],2), 'ms/step'))"
    done
  done
  [ "$NG" -ge 2 ] && python -m torch.distributed.run --nnodes=1 --nproc-per-node $NG --master-addr 127.0.0.1 --master-port 29899 tools/exchange_bench.py --text-log2 33 > $O/exchange_cfg5.json 2> $O/exchange_cfg5.err
fi
echo "done: $O"
