cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
export SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_probes.so
for B in 4 88 352; do timeout 60 python tools/chain_trace.py --B $B; done > gpurun_out/r2a/chain.log 2>&1
cat gpurun_out/r2a/chain.log
