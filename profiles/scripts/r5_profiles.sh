#!/bin/bash
# round 5: everything the committed profiles/r05_* come from, one call (the library is built before: rocprofv3 --pmc
# initialises the GPU before Python starts)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5final; rm -rf $O; mkdir -p $O
cd $R
# the counter passes first: the bench line then reads THIS build's traffic.json / describe_model.json
cd /tmp && export TMPDIR=/tmp
echo "== fir pmc"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/profiles/pmc_fir.py > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/profiles/pmc_fir.py > $O/pmc_write.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_pyr_fetch -- python3 $R/profiles/pmc_fir.py --pyramid > $O/pmc_pyr_fetch.log 2>&1 &&
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_pyr_write -- python3 $R/profiles/pmc_fir.py --pyramid > $O/pmc_pyr_write.log 2>&1
echo "== describe pmc"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $O/pmc_desc/a -o run --output-format csv -- python3 $R/profiles/describe_model.py --run > $O/pmc_desc_a.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_desc/b -o run --output-format csv -- python3 $R/profiles/describe_model.py --run > $O/pmc_desc_b.log 2>&1
cd $R
python3 profiles/pmc_fir.py --parse $O/pmc_fetch $O/pmc_write $O/pmc_pyr_fetch $O/pmc_pyr_write > $O/traffic.json 2>$O/traffic.err
python3 profiles/describe_model.py --parse $O/pmc_desc 2523709698 > $O/describe_model.json 2>$O/dm.err
head -c 1200 $O/describe_model.json; echo; python3 -c "
import json; t=json.load(open('$O/traffic.json'))
for k,v in sorted(t.items()): print(k, v['hbm_bytes'], v['ratio'])"
cp $O/traffic.json $O/describe_model.json $R/profiles/
echo "== bench"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); d.pop('details',None); print(json.dumps(d)[:3800])
PY
echo "== sharded N=1"; timeout -k 10 300 python bench.py --sharded --no-cpu --no-micro --no-host --steps 10 --warmup 3 > $O/bench_sharded1.json 2>/dev/null; tail -c 700 $O/bench_sharded1.json; echo
echo "== rehearsal 8 ranks (weak main line + strong_1024 sub-record)"; timeout -k 10 900 python bench.py --rehearse-threads 8 --steps 2 --warmup 1 --no-cpu > $O/rehearse8.json 2>$O/rehearse8.err; tail -c 1500 $O/rehearse8.json; echo; tail -3 $O/rehearse8.err
echo "== register"; timeout -k 10 400 python bench.py --register --steps 3 --warmup 1 > $O/bench_register.json 2>$O/reg.err; tail -c 400 $O/bench_register.json; echo
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats (in-step launches only)"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/kstats -o run --output-format csv -- python3 $R/bench.py --no-cpu --no-host --no-micro --no-strong-leg --no-pyramid-leg --steps 10 --warmup 3 > $O/kstats_bench.json 2> $O/kstats.err
tail -c 900 $O/kstats_bench.json; echo
echo "== the octave-0 extrema sweep alone"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/sweep -o run --output-format csv -- python3 $R/profiles/microbench/sweep_alone.py > $O/sweep.log 2>&1
python3 - <<PY > $O/sweep_alone.txt
import csv
print(open("$O/sweep.log").read().strip().splitlines()[-1])
for r in csv.DictReader(open("$O/sweep/run_kernel_stats.csv")):
    if "extrema" in r["Name"] or "dogmax" in r["Name"] or "zero_mask" in r["Name"]:
        print("%-64s calls %3s  avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cat $O/sweep_alone.txt
cd $R; ls $O/kstats
