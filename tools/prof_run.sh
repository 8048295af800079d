set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py 2>&1 | tail -3
mkdir -p gpurun_out/prof
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/bench_sweep.py --ops fwd --n 5 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write -- python $GRAFT_REPO_ROOT/tools/bench_sweep.py --ops fwd --n 5 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*.csv" | head -30
tail -2 gpurun_out/prof/kt.log
