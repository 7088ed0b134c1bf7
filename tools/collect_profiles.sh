#!/bin/bash
# Collects the round's final evidence in ONE session on an MI355X box (boxes differ by up to 10 %: numbers that are quoted
# together must come from the same box).  Run through gpurun from the repository root:
#     gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r04 > gpurun_out/collect.log 2>&1'
# and copy gpurun_out/final/<TAG>_* into profiles/ afterwards.
# The per-kernel evidence (rocprofv3 kernel stats, PMC traffic) is taken with `--serial-plan --no-legs`: every launch of a kernel
# then has one size and shares the GPU with nothing, which is also how bench.py itself measures its `roofline` object.
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SER="--serial-plan --no-legs --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k64 -o run -- python3 $R/bench.py $SER > $OUT/k64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k8 -o run -- python3 $R/bench.py --scaling strong --members 8 $SER > $OUT/k8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1 -o run -- python3 $R/bench.py --scaling strong --members 1 $SER > $OUT/k1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k32c5 -o run -- python3 $R/bench.py --config cfg5 $SER > $OUT/k32c5.log 2>&1
cp $OUT/k64/run_kernel_stats.csv $OUT/${TAG}_model_bench_kernel_stats.csv
cp $OUT/k8/run_kernel_stats.csv $OUT/${TAG}_model_bench_kernel_stats_8members.csv
cp $OUT/k1/run_kernel_stats.csv $OUT/${TAG}_model_bench_kernel_stats_1member.csv
cp $OUT/k32c5/run_kernel_stats.csv $OUT/${TAG}_model_bench_kernel_stats_cfg5.csv
echo "kernel traces done"
# PMC passes, each on its own (never together with a trace domain other than --kernel-trace)
PMC="--steps 36 --warmup 6 --regions 1 $SER"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -o run -- python3 $R/bench.py $PMC > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -o run -- python3 $R/bench.py $PMC > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc $OUT/${TAG}_pmc_model_step.json "python3 bench.py $PMC (64 members)" 8
# cfg 5 with the physics-only arrays as fp32 (the default) and as fp64 (round 3's layout): the column kernel's bytes
for st in 1 0; do
  export PYSPEEDY_AMD_PHYS_STORE32=$st
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc5_$st/fetch -o run -- python3 $R/bench.py --config cfg5 $PMC > $OUT/pmc5_fetch_$st.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc5_$st/write -o run -- python3 $R/bench.py --config cfg5 $PMC > $OUT/pmc5_write_$st.log 2>&1
  python3 $R/tools/pmc_summary.py $OUT/pmc5_$st $OUT/${TAG}_pmc_cfg5_storage32_$st.json "PYSPEEDY_AMD_PHYS_STORE32=$st python3 bench.py --config cfg5 $PMC (32 members)" 8
done
unset PYSPEEDY_AMD_PHYS_STORE32
# The bench lines AFTER the PMC passes: roofline.traffic of a line is the committed PMC figure, and the one this session has just
# taken (stamped with the sha of the device sources: traffic_stale = false) is put where bench.py looks for it -- on this box only;
# the copy that is committed comes back through gpurun_out/final like everything else.
mkdir -p $R/profiles && cp $OUT/${TAG}_pmc_model_step.json $OUT/${TAG}_pmc_cfg5_storage32_1.json $R/profiles/ 2>/dev/null || true
# the default line carries every BASELINE config since round 4 (cfg3, cfg4_shard8, cfg5, cfg2_transforms legs)
python3 $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_20.json 2>> $OUT/bench.err
python3 $R/bench.py --serial-plan --no-legs --no-cpu-baseline > $OUT/${TAG}_bench_serial_plan.json 2>> $OUT/bench.err
python3 $R/bench.py --config cfg5 --no-cpu-baseline --no-legs > $OUT/${TAG}_bench_cfg5.json 2>> $OUT/bench.err
python3 $R/bench.py --one-process --gpus 1 --no-cpu-baseline --steps 360 > $OUT/${TAG}_bench_one_process.json 2>> $OUT/bench.err
echo "bench lines done"
# the Legendre stage on its own at 16 384 fields (north_star's literal target): kernel stats, then the two PMC passes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/leg -o run -- python3 $R/tools/perf_legendre_only.py > $OUT/${TAG}_legendre_only.txt 2>&1
cp $OUT/leg/run_kernel_stats.csv $OUT/${TAG}_legendre_only_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcleg/fetch -o run -- python3 $R/tools/perf_legendre_only.py 16384 20 > $OUT/pmcleg_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmcleg/write -o run -- python3 $R/tools/perf_legendre_only.py 16384 20 > $OUT/pmcleg_write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmcleg $OUT/${TAG}_pmc_legendre_only.json "python3 tools/perf_legendre_only.py 16384 20 (16 384 fields per launch)" 5
rm -rf $OUT/leg/run_kernel_trace.csv $OUT/pmcleg
rm -rf $OUT/k64/run_kernel_trace.csv $OUT/k8/run_kernel_trace.csv $OUT/k1/run_kernel_trace.csv $OUT/k32c5/run_kernel_trace.csv $OUT/pmc/fetch $OUT/pmc/write $OUT/pmc5_0 $OUT/pmc5_1
python3 $R/tools/copy_rate.py > $OUT/${TAG}_device_copy_rate.txt 2>&1 || true
# round 6: the streaming ceiling with the library's own probe kernels, the column kernel at whole and fractional rounds of the
# machine, the facade's time loop (stretches, export) and the cost of a pause between two stretches
python3 $R/tools/stream_ceiling.py --json $OUT/${TAG}_stream_ceiling.json > $OUT/${TAG}_stream_ceiling.txt 2>&1 || true
python3 $R/tools/exp_tail.py > $OUT/${TAG}_column_tail.txt 2>&1 || true
python3 $R/tools/experiments/r06_facade_plans.py 32 64 256 > $OUT/${TAG}_facade_run.txt 2>&1 || true
python3 $R/tools/perf_facade_export.py 64 5 > $OUT/${TAG}_facade_export.txt 2>&1 || true
python3 $R/tools/experiments/r06_idle_gap.py > $OUT/${TAG}_idle_gap.txt 2>&1 || true
echo "all done"
ls -la $OUT
