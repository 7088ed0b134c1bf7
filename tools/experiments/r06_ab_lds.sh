for rep in 1 2; do
for v in v0 v2 v2p v4 v8 v1p; do
  PERF_LIB=build_variants/lib_lds_$v.so python tools/perf_transforms.py 4928 16384 2>&1 | grep -v "amdgpu.ids\|device copy" >> gpurun_out/ab_lds_micro.txt
done; done
for rep in 1 2; do
for v in v0 v2 v2p v4 v1p; do
  PYSPEEDY_AMD_LIB=$PWD/build_variants/lib_lds_$v.so python bench.py --no-legs --no-cpu-baseline > gpurun_out/ab_lds_${v}_$rep.json 2>/dev/null
done; done
echo finished
