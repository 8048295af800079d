cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "scorer or score" 2>&1 | tail -8
for v in 128 32; do echo "variant $v"; SEMICRF_SCORE_VARIANT=$v timeout 300 python tools/bench_scorer.py 2>&1 | grep impl=0; done
timeout 300 python tools/bench_scorer.py --T 691 --C 360 2>&1 | grep impl=0
timeout 300 python tools/bench_scorer.py --T 2048 --C 88 2>&1 | grep impl=0
timeout 300 python tools/bench_scorer.py --T 512 --C 64 2>&1 | grep impl=0
