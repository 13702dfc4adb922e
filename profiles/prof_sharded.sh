#!/bin/bash
# usage: profiles/prof_sharded.sh <tag>  -> gpurun_out/<tag>_sharded_{kernel_stats.csv,pmc_FETCH_SIZE.csv,pmc_WRITE_SIZE.csv},
# gpurun_out/traffic_sharded_<tag>.json: the HBM traffic of one step of `bench.py --sharded-index` (every kernel of the
# step, summed), from separate --pmc passes like profiles/prof.sh
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/profs_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$tag/kt -o kt -- python bench.py --sharded-index --steps 2 --warmup 1 "$@" > gpurun_out/${tag}_sharded_bench_under_rocprof.json 2>/tmp/profs_$tag/kt_err.log
find /tmp/profs_$tag/kt -name '*kernel_stats.csv' -exec cp {} gpurun_out/${tag}_sharded_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/profs_$tag/$c -o p -- python bench.py --sharded-index --steps 1 --warmup 0 "$@" > /dev/null 2>/tmp/profs_$tag/${c}_err.log
  f=$(find /tmp/profs_$tag/$c -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python - "$f" "gpurun_out/${tag}_sharded_pmc_$c.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    key = (r.get("Kernel_Name", "")[:80], r.get("Counter_Name", ""))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += float(r.get("Counter_Value", 0) or 0)
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
for (k, c), (n, s) in agg.items():
    w.writerow([k, c, n, s, s / n])
PY
  else echo "no counter csv for $c"; tail -3 /tmp/profs_$tag/${c}_err.log; fi
done
python profiles/tools/make_traffic_sharded.py $tag 1 0 "$@" || true
