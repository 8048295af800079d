set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
for B in 4 88 176 352; do python tools/bench_sweep.py --T 1024 --B $B --ops fwd --n 20; done > gpurun_out/r2a/base.log 2>&1
export SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_probes.so
for B in 88 352; do python tools/bench_sweep.py --T 1024 --B $B --ops fwd --n 10 --flags 0,3,12,44; done > gpurun_out/r2a/abl.log 2>&1
for B in 4 88 352; do python tools/chain_trace.py --B $B; done > gpurun_out/r2a/chain.log 2>&1
cat gpurun_out/r2a/base.log gpurun_out/r2a/abl.log gpurun_out/r2a/chain.log
