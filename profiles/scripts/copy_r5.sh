#!/bin/bash
# copies what profiles/scripts/r5_profiles.sh left under gpurun_out/r5final into profiles/r05_* (run in the repo root)
O=gpurun_out/r5final; P=profiles
last() { python3 -c "import sys; print(open(sys.argv[1]).read().strip().splitlines()[-1])" "$1"; }
last $O/bench.json > $P/r05_bench_512.json
last $O/bench_sharded1.json > $P/r05_bench_sharded_n1.json
last $O/rehearse8.json > $P/r05_rehearsal_8ranks.json
last $O/bench_register.json > $P/r05_bench_register.json
last $O/kstats_bench.json > $P/r05_bench_step_profiled.json
cp $O/kstats/run_kernel_stats.csv $P/r05_kernel_stats_step.csv
cp $O/kstats/run_kernel_trace.csv $P/r05_kernel_trace_step.csv
cp $(ls -t $O/pmc_fetch/*/*_counter_collection.csv | head -1) $P/r05_pmc_fetch_size_fir512.csv
cp $(ls -t $O/pmc_write/*/*_counter_collection.csv | head -1) $P/r05_pmc_write_size_fir512.csv
cp $(ls -t $O/pmc_pyr_fetch/*/*_counter_collection.csv | head -1) $P/r05_pmc_fetch_size_pyramid512.csv
cp $(ls -t $O/pmc_pyr_write/*/*_counter_collection.csv | head -1) $P/r05_pmc_write_size_pyramid512.csv
cp $O/pmc_desc/a/run_counter_collection.csv $P/r05_pmc_describe_a.csv
cp $O/pmc_desc/b/run_counter_collection.csv $P/r05_pmc_describe_b.csv
cp $O/sweep_alone.txt $P/r05_sweep_alone.txt
cp $O/traffic.json $P/traffic.json
cp $O/describe_model.json $P/describe_model.json
ls -la $P/r05_* $P/traffic.json $P/describe_model.json
