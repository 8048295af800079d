cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
L=gpurun_out/r2d/pace.log
: > $L
for v in hip pace1 pace2; do
for xr in 0 4; do
  echo "== $v XR=$xr" >> $L
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so SEMICRF_XR=$xr timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd --n 20 >> $L 2>&1
done
done
grep -v amdgpu $L
export SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_probes.so
SEMICRF_XR=0 timeout 60 python tools/chain_trace.py --B 352 2>&1 | grep -v amdgpu | tail -8
SEMICRF_XR=4 timeout 60 python tools/chain_trace.py --B 352 2>&1 | grep -v amdgpu | tail -8
