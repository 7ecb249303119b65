#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sweep_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o run -- python3 $R/profiles/microbench/sweep_alone.py > $O/f.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o run -- python3 $R/profiles/microbench/sweep_alone.py > $O/w.log 2>&1 || exit 1
cd $R && python3 - <<PY
import csv, glob, collections
for d,name in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    f=glob.glob("gpurun_out/sweep_pmc/%s/*counter_collection.csv"%d)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "sweep3g" in k: agg[k].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(name, k, "launches", len(v), "median counter", sorted(v)[len(v)//2], "-> KB x1024 = %.3f GB" % (sorted(v)[len(v)//2]*1024/1e9))
PY
