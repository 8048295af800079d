cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
L=gpurun_out/r2c/sched.log
: > $L
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "persist_vs_oracle or full_size or repeatable or minimal or edge" 2>&1 | tail -5
echo "== sched0" >> $L
SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_sched0.so timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd,bwd --n 20 >> $L 2>&1
for ra in 0 2 4 6 10; do
  echo "== sched1 RA=$ra" >> $L
  SEMICRF_RUN_AHEAD=$ra timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd,bwd --n 20 >> $L 2>&1
done
for T in 691 2048; do
echo "== sched0 T=$T" >> $L
SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_sched0.so timeout 120 python tools/bench_sweep.py --T $T --B 352 --ops fwd,bwd --n 10 >> $L 2>&1
echo "== sched1 T=$T" >> $L
timeout 120 python tools/bench_sweep.py --T $T --B 352 --ops fwd,bwd --n 10 >> $L 2>&1
done
echo "== sched1 B=88" >> $L
timeout 120 python tools/bench_sweep.py --T 1024 --B 88 --ops fwd --n 20 >> $L 2>&1
grep -v amdgpu $L
export SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_probes.so
timeout 60 python tools/chain_trace.py --B 352 2>&1 | grep -v amdgpu
