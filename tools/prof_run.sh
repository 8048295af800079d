# The exact sequence behind profiles/<tag>_*: GPU tests, bench.py, bench.py under rocprofv3 --kernel-trace --stats twice (headline
# workload only: --no-extra, every persist_sweep row is T=1024 x 352; and with the extras: every product kernel shows up), then
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass): HBM traffic of the forward and gradient sweeps, cache and
# stall counters of the gradient sweep, and matrix-pipe counters + busy cycles of the scorer kernels.  tools/collect_profiles.py
# turns the output into profiles/<tag>_*.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tee $OUT/pytest.log
timeout 900 python bench.py 2>$OUT/bench.err > $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-live-traffic > $OUT/kt.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_all -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic > $OUT/kt_all.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py > $OUT/kt_train.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train_bf16x3 -- python $GRAFT_REPO_ROOT/tools/train_step_probe.py bf16x3-train > $OUT/kt_train_bf16x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_scorer -- python $GRAFT_REPO_ROOT/tools/bench_scorer_all.py 20 > $OUT/kt_scorer.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_fwd_$c -- python $GRAFT_REPO_ROOT/tools/bench_sweep.py --ops fwd --n 5 > $OUT/pmc_fwd_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_bwd_$n -- python $GRAFT_REPO_ROOT/tools/bench_sweep.py --ops bwd --n 5 > $OUT/pmc_bwd_$n.log 2>&1
done
for c in "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_scorer_$n -- python $GRAFT_REPO_ROOT/tools/bench_scorer_all.py 3 big > $OUT/pmc_scorer_$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_grid.py > $OUT/grid.md 2>/dev/null
timeout 300 python tools/bench_shapes.py > $OUT/shapes.md 2>/dev/null
timeout 300 python tools/bench_scorer_all.py 10 > $OUT/scorer.txt 2>/dev/null
(timeout 300 python tools/bwd3_probe.py 5 --proj 2>/dev/null | grep -v amdgpu.ids) > $OUT/bwd3.txt
for c in fp32 bf16x3 bf16x3-bwd bf16x3-train bf16x3-all; do echo "train step, scorer.contraction = $c: $(timeout 200 python tools/train_step_probe.py $c 2>/dev/null | grep 'ms per step')"; done > $OUT/train_modes.txt
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*agent_info*" -delete
cat $OUT/pytest.log; cut -c1-300 $OUT/bench.json; cat $OUT/scorer.txt $OUT/bwd3.txt $OUT/train_modes.txt
