#!/bin/bash
# A/B of the LDE kernels on the GPU box: time + FETCH_SIZE per variant.  usage: tools/ab_lde.sh <outdir> [ENVVAR]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; var=$2
mkdir -p $out
python3 tools/lde_probe.py 22 43 5 > $out/new_time.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/new_fetch -o f -- python3 tools/lde_probe.py 22 43 1 > $out/new_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/new_write -o f -- python3 tools/lde_probe.py 22 43 1 > $out/new_write.log 2>&1
rc=$?
if [ -n "$var" ] && [ $rc -eq 0 ]; then
  export $var=1
  python3 tools/lde_probe.py 22 43 5 > $out/old_time.log 2>&1 && \
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/old_fetch -o f -- python3 tools/lde_probe.py 22 43 1 > $out/old_fetch.log 2>&1
  rc=$?
fi
cat $out/*_time.log
python3 - $out <<'PY'
import csv, glob, sys, collections, re
for tag in ("new_fetch", "new_write", "old_fetch"):
    for f in glob.glob(sys.argv[1] + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        tot, n = collections.defaultdict(float), collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("lcp2::", "")
            tot[name] += float(row["Counter_Value"]); n[name] += 1
        for k in tot:
            mult = 2.0 if "fetch" in tag else 1.0
            print(tag, k, "launches", n[k], "GB/launch %.3f" % (mult * tot[k] * 1024 / n[k] / 1e9))
PY
exit $rc
