# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the gradient sweep and of the T=2048 Viterbi sweep
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof/bwd_$c -- python $R/tools/bench_sweep.py --ops bwd --n 5 > $R/gpurun_out/prof/bwd_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof/vit_$c -- python $R/tools/bench_sweep.py --T 2048 --ops vit --n 3 > $R/gpurun_out/prof/vit_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for tag in ("bwd","vit"):
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        f=glob.glob(f'gpurun_out/prof/{tag}_{c}/*/*_counter_collection.csv')[0]
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'persist' in r['Kernel_Name']: agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
        for k,v in agg.items(): print(tag,c,k,len(v),round(sum(v)/len(v)))
PY
