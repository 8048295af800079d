cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
L=gpurun_out/r2b/variants.log
: > $L
for v in hip early tpt32 early_tpt32; do
  echo "== variant $v" >> $L
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd,bwd --n 20 >> $L 2>&1
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 2048 --B 352 --ops fwd --n 10 >> $L 2>&1
  SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/bench_sweep.py --T 691 --B 360 --ops fwd,bwd --n 20 >> $L 2>&1
done
grep -v amdgpu.ids $L
SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_early.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "persist_vs_oracle or full_size or repeatable" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "segment or model_shape" 2>&1 | tail -15
timeout 600 python bench.py 2>gpurun_out/r2b/bench.err | tee gpurun_out/r2b/bench.json | cut -c1-3000
tail -20 gpurun_out/r2b/bench.err
