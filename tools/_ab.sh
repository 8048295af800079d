cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/bench_scorer.py 2>&1 | grep impl=0
timeout 300 python tools/bench_scorer.py --T 691 --C 360 2>&1 | grep impl=0
timeout 300 python tools/bench_scorer.py --T 2048 --C 88 2>&1 | grep impl=0
timeout 300 python tools/bench_scorer.py --T 512 --C 64 2>&1 | grep impl=0
timeout 300 python tools/bench_scorer.py --T 200 --C 352 2>&1 | grep impl=0
timeout 600 python tools/bench_fused.py 2>&1 | tail -8
