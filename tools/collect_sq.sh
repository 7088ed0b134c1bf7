#!/bin/bash
# SQ / TCC counter passes over the 64-member model step in the serial plan (three separate rocprofv3 --pmc passes, nothing but
# --kernel-trace beside them), summarised by tools/sq_summary.py into gpurun_out/final/<TAG>_sq_model_step.txt.
#     gpurun --timeout 900 -- 'bash tools/collect_sq.sh r06 > gpurun_out/collect_sq.log 2>&1'
# PYSPEEDY_AMD_LIB selects another build of the library (A/B in one session: run twice with different TAGs).
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D=/tmp/sq_$TAG
rm -rf $D
PMC="--steps 36 --warmup 6 --regions 1 --serial-plan --no-legs --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
    --output-format csv -d $D/a -o run -- python3 $R/bench.py $PMC > $OUT/sq_a_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d $D/b -o run -- python3 $R/bench.py $PMC > $OUT/sq_b_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D/c -o run -- python3 $R/bench.py $PMC > $OUT/sq_c_$TAG.log 2>&1
{
  echo "SQ / TCC counters of the kernels of the 64-member model step (MI355X): rocprofv3 --kernel-trace --pmc, three separate passes"
  echo "(8 SQ counters; 8 SQ counters; TCC_HIT_sum TCC_MISS_sum) over python3 bench.py $PMC;"
  echo "library: ${PYSPEEDY_AMD_LIB:-pyspeedy_amd/libpyspeedy_amd.so}; mean per launch after the first 8 launches of each kernel; tools/sq_summary.py."
  echo
  python3 $R/tools/sq_summary.py $D 8 spec2grid grid2spec physics_kernel spectral_step geopotential
} > $OUT/${TAG}_sq_model_step.txt
rm -rf $D
echo done
