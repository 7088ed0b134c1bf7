# A/B of the LDS lane maps of round 6 (profiles/r06_lds_conflicts.txt).  The variants are built from the tree with
# tools/experiments/r06_lds_lane_maps.patch applied (git apply), one library per map:
#   for v in "v0 0 0" "v2 2 0" "v2p 2 1" "v4 4 0" "v8 8 0" "v1p 1 1"; do set -- $v
#     tools/build_variant.sh lds_$1 "" -- -DSPD_LDS_MAP=$2 -DSPD_LDS_PAIRS=$3; done
# then: gpurun -- 'bash tools/experiments/r06_ab_lds.sh'
for rep in 1 2; do
for v in v0 v2 v2p v4 v8 v1p; do
  PERF_LIB=build_variants/lib_lds_$v.so python tools/perf_transforms.py 4928 16384 2>&1 | grep -v "amdgpu.ids\|device copy" >> gpurun_out/ab_lds_micro.txt
done; done
for rep in 1 2; do
for v in v0 v2 v2p v4 v1p; do
  PYSPEEDY_AMD_LIB=$PWD/build_variants/lib_lds_$v.so python bench.py --no-legs --no-cpu-baseline > gpurun_out/ab_lds_${v}_$rep.json 2>/dev/null
done; done
echo finished
