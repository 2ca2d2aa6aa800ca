source tools/exp_variants.sh exp7 true
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not 8gib and not full_size_1gib" 2>&1 | grep -E "passed|failed|Error" | tail -4
run acgt --
run hit -- --workload acgt_hit
run eng -- --workload eng
run reads100 -- --workload acgt_hit --plen 100 --npats 4000000
export TMPDIR=/tmp
for w in eng; do
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats_$w -o stats -- python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-extra --pmc off --workload $w > $OUT/bench_stats_$w.json 2> $OUT/stats_$w.err
python - $w <<'PY'
import csv,glob,sys
for f in glob.glob(f"gpurun_out/exp7/stats_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        if "densify" in r["Name"] or "extract" in r["Name"] or "rocclr" in r["Name"]: continue
        print(r["Name"][:110], r["Calls"], r["AverageNs"])
PY
done
find $OUT -name "*.csv" -size +2M -delete
