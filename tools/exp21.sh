export TMPDIR=/tmp
for w in acgt_hit acgt; do
  rm -rf /tmp/tr_$w; rocprofv3 --kernel-trace -f csv -d /tmp/tr_$w -o tr -- python bench.py --steps 6 --warmup 2 --workload $w --no-extra --cpu-sample 0 --pmc off > /dev/null 2> /tmp/tr_$w.err
  echo "== $w"; python tools/trace_gaps.py /tmp/tr_$w 3 | cut -c1-150
done
