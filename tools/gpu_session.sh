# tools/gpu_session.sh -- what the driver runs at the end of a round, in one gpurun call (run from the repo root on the GPU box):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_session.sh > gpurun_out/round_end.log 2>&1; tail -c 3000 gpurun_out/round_end.log'
set -u
SECONDS=0
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -12
echo "tests $SECONDS s"
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round_end_bench.json 2> gpurun_out/round_end_bench.err
tail -c 400 gpurun_out/round_end_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/round_end_bench.json") if l.startswith("{")][-1])
print("bench:", round(d["value"] / 1e9, 2), "G patterns/s", round(d["ms_per_step"], 4), "ms/step;", {k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, list, str))})
PY
echo "all $SECONDS s"
