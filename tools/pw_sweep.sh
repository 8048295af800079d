for n in 128 192 256 320 384 468; do echo "panel WGs $n:"; SEMICRF_PANEL_WGS=$n timeout 120 python tools/bench_sweep.py --ops fwd,bwd --n 20 2>&1 | tail -2; done
