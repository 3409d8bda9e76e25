cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/aggpmc; rm -rf $O; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p -- python tools/agg_fwd_pmc.py > $O/p$i.log 2>&1
  tail -1 $O/p$i.log
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").replace("txe::", "").split("(")[0][:60]
    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in per.items():
    if "gat_agg" in k:
        print(k, {n: sum(x) / len(x) for n, x in v.items()})
PY
done
python tools/agg_fwd_variants.py 2>&1 | head -3
python tools/kernel_times.py --steps 20 2>&1 | grep -E "^step|aggregate|zsum"
