cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
export SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_probes.so
export SEMICRF_HYBRID_START=0
L=gpurun_out/r2b/abl2.log
: > $L
timeout 100 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd --n 10 --flags 0,12,76,44,3 >> $L 2>&1
export SEMICRF_HYBRID_PANEL_WAVES=0
echo "no hybrid waves" >> $L
timeout 100 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd --n 10 --flags 0,12,76,44,3 >> $L 2>&1
grep -v amdgpu $L
