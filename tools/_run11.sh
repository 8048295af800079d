cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d; L=gpurun_out/r2d/ring.log
: > $L
export SEMICRF_XR=0
for v in ring5 ring6; do
  echo "== $v" >> $L
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd,bwd --n 20 >> $L 2>&1
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd --flags 3 --n 10 >> $L 2>&1
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 691 --B 360 --ops fwd --n 20 >> $L 2>&1
done
grep -v amdgpu $L
