#!/bin/bash
# development: time the diagonal phase (ev6->ev7 of block 32) for each ablation variant
for v in hip abl1 abl2 abl3 abl4 abl8 abl16 abl24 abl31; do
  out=$(SEMICRF_LIB=$PWD/transkun_amd/libsemicrf_$v.so timeout 120 python tools/spine_trace.py --flags 3 2>&1 | grep -E "k= +32 ")
  echo "$v: $out" | awk '{print $1, "diag16 =", $(NF)-$(NF-1), "cycles; block period (vs k=33 later)"}'
done
