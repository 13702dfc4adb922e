#!/bin/bash
# usage: profiles/prof.sh <tag> [bench args...]   -> gpurun_out/<tag>_{kernel_stats.csv,pmc_*.csv,bench.json}
# PROF_PMC_ARGS: extra bench args of the counter passes (default: two steps, no warm-up: the means of the counter
# passes are over every dispatch of the run -- 10 of the hot kernel at the default 5 launches per step)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PMC_ARGS=${PROF_PMC_ARGS:---steps 2 --warmup 0}
mkdir -p gpurun_out /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/kt -o kt -- python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/${tag}_bench_under_rocprof.json 2>/tmp/prof_$tag/kt_err.log
find /tmp/prof_$tag/kt -name '*kernel_stats.csv' -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --output-format csv -d /tmp/prof_$tag/$n -o p -- python bench.py --no-cpu-baseline --no-extras "$@" $PMC_ARGS > /dev/null 2>/tmp/prof_$tag/${n}_err.log
  f=$(find /tmp/prof_$tag/$n -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python - "$f" "gpurun_out/${tag}_pmc_$n.csv" <<'PY'
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
  else echo "no counter csv for $c"; tail -3 /tmp/prof_$tag/${n}_err.log; fi
done
python profiles/tools/make_traffic.py $tag "$@" $PMC_ARGS || true
ls -la gpurun_out/
