cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
L=gpurun_out/r2d/recent.log
: > $L
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "persist_vs_oracle or full_size or repeatable or minimal or edge or model_shape" 2>&1 | tail -5
for xr in 0 2 3 4 5; do
  echo "== XR=$xr" >> $L
  SEMICRF_XR=$xr timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd,bwd --n 20 >> $L 2>&1
done
for T in 691 2048; do
for xr in 0 4; do
echo "== XR=$xr T=$T" >> $L
SEMICRF_XR=$xr timeout 120 python tools/bench_sweep.py --T $T --B 352 --ops fwd --n 10 >> $L 2>&1
done
done
echo "== XR=4 B=88" >> $L
timeout 120 python tools/bench_sweep.py --T 1024 --B 88 --ops fwd --n 20 >> $L 2>&1
grep -v amdgpu $L
export SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_probes.so
timeout 60 python tools/chain_trace.py --B 352 2>&1 | grep -v amdgpu
