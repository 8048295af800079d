# The exact sequence behind profiles/r02_*: the GPU tests, bench.py, bench.py under rocprofv3 --kernel-trace --stats (headline
# workload only: --no-extra, so that every persist_sweep row is T=1024, NBatch=352), then separate --pmc passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass) for the forward sweep, and cache / stall counters of the gradient sweep.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.log
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_fwd_$c -- python $GRAFT_REPO_ROOT/tools/bench_sweep.py --ops fwd --n 5 > $OUT/pmc_fwd_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_bwd_$n -- python $GRAFT_REPO_ROOT/tools/bench_sweep.py --ops bwd --n 5 > $OUT/pmc_bwd_$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_grid.py > $OUT/grid.md 2>/dev/null
tail -1 $OUT/kt.log | cut -c1-400
