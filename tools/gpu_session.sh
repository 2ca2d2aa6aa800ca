set -u
SECONDS=0
export TMPDIR=/tmp
for i in 1 2; do
  echo "fair"; python tools/regexp_bench.py --which approx --reps 1 --concurrent 8 2>/dev/null | grep concurrent | cut -c1-330
  echo "nofair"; FEMTO_AMD_NFA_FAIR=0 python tools/regexp_bench.py --which approx --reps 1 --concurrent 8 2>/dev/null | grep concurrent | cut -c1-330
done
echo "ab $SECONDS s"
bash tools/final_round_r06.sh cfg5
echo "all $SECONDS s"
