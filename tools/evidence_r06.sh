# tools/evidence_r06.sh -- on the GPU box: the long randomised parity soak, the damaged-index fuzz and a same-box repeat of the
# headline bench line (how much one box varies between runs); the logs are promoted to profiles/r06_soak_fuzz.txt and
# profiles/r06_repeat.txt by hand.  About 20 minutes.
set -u
SECONDS=0
export TMPDIR=/tmp
O=gpurun_out/evidence_r06; mkdir -p $O
python - <<'PY' > $O/hash.txt
import benchlib.common as c; print("source hash", c.source_hash())
PY
timeout 1200 python tools/soak.py 160 100 > $O/soak_a.log 2>&1; echo "soak_a rc $? $SECONDS s"; tail -1 $O/soak_a.log
FEMTO_AMD_SOAK_ROWS=20000000,40000000 timeout 900 python tools/soak.py 6 5000 > $O/soak_big.log 2>&1; echo "soak_big rc $? $SECONDS s"; tail -1 $O/soak_big.log
for s in 11 12 13; do timeout 600 python tools/fuzz_gpu.py $s 60 > $O/fuzz_$s.log 2>&1; echo "fuzz $s rc $? $SECONDS s"; tail -1 $O/fuzz_$s.log | cut -c1-200; done
for i in 1 2 3; do python bench.py --steps 200 --warmup 10 --no-extra --cpu-sample 0 > $O/repeat_$i.json 2> $O/repeat_$i.err; echo "repeat $i $SECONDS s"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/evidence_r06/repeat_*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d["roofline"]
    print(f, round(d["value"] / 1e9, 3), "G/s", round(d["ms_per_step"], 4), "ms/step kernel", round(r["kernel_ms"], 4), "ms frac", round(r["frac"], 3))
PY
echo "all $SECONDS s"
