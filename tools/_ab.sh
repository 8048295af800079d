cd $GRAFT_REPO_ROOT
for v in hip nt hip nt; do echo $v; SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 300 python tools/bench_scorer.py 2>&1 | grep impl=0; done
