cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
L=gpurun_out/r2c/diet.log
: > $L
SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_s0.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "persist_vs_oracle or full_size or repeatable or minimal or edge or model_shape" 2>&1 | tail -5
for v in s0d0 s0 s0nt640 s0nt768 hip; do
  echo "== $v" >> $L
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd,bwd --n 20 >> $L 2>&1
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 2048 --B 352 --ops fwd --n 10 >> $L 2>&1
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 691 --B 360 --ops fwd --n 20 >> $L 2>&1
done
grep -v amdgpu $L
