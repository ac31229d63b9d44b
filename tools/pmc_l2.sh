#!/bin/bash
# L2 / HBM counters of the field kernels on the headline workload: tools/pmc_l2.sh <tag>   (env DINER_AMD_LIB selects the build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcl2_$1; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --cpu-rays 0 --no-modes --no-configs"
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$n -o pmc -- $CMD > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "h3n::k_field" in k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    hit = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1)
    print("$1", k, "L2 hit %.4f" % hit, "HBM read %.1f GB per frame (FETCH_SIZE x 2 KB)" % (v["FETCH_SIZE"] * 2 * 1024 / 1e9),
          "= %.1f KB per point" % (v["FETCH_SIZE"] * 2 * 1024 / (480000 * 128) / 1e3))
PY
