# usage: bash tools/_cmp.sh OUTNAME variant1 variant2 ...   (shapes from $SHAPES)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT; shift
V=$PWD/transkun_amd/_variants
SH=${SHAPES:-"1024x88 691x96 691x384 1024x352 2048x352"}
for rep in 1 2; do
for v in "$@"; do
  export SEMICRF_LIB=$V/$v/libsemicrf_hip.so
  echo "== $v (round $rep)"
  timeout 300 python tools/bench_shapes.py $SH 2>&1 | grep "^| [0-9]"
done; done > $OUT/cmp.txt 2>&1
cat $OUT/cmp.txt
