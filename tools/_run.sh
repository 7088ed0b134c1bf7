cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t
for m in 1 8; do
python tools/exp_bound.py $m 2>/dev/null | cut -c1-220
PYSPEEDY_AMD_COUPLER_IN_SPECTRAL=0 python tools/exp_bound.py $m 2>/dev/null | sed 's/^/nocpl /' | cut -c1-240
PYSPEEDY_AMD_FOLD_GEO=0 python tools/exp_bound.py $m 2>/dev/null | sed 's/^/nofold /' | cut -c1-240
PYSPEEDY_AMD_FOLD_GEO=0 PYSPEEDY_AMD_COUPLER_IN_SPECTRAL=0 python tools/exp_bound.py $m 2>/dev/null | sed 's/^/neither /' | cut -c1-260
done | tee gpurun_out/r3t/small.txt
