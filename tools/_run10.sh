cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d; L=gpurun_out/r2d/pw.log
: > $L
export SEMICRF_XR=0
for pw in 2 3 4; do for hp in 0 2; do
  echo "== PW=$pw HYB=$hp" >> $L
  SEMICRF_PANEL_WAVES=$pw SEMICRF_HYBRID_PANEL_WAVES=$hp timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd --n 20 >> $L 2>&1
done; done
for hs in 8 16 40; do
  echo "== HSTART=$hs" >> $L
  SEMICRF_HYBRID_START=$hs timeout 120 python tools/bench_sweep.py --T 1024 --B 352 --ops fwd --n 20 >> $L 2>&1
done
grep -v amdgpu $L
