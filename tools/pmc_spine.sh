cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/prof
for fl in 3 259; do
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_f$fl -- python $R/tools/bench_sweep.py --ops fwd --n 5 --flags $fl > $R/gpurun_out/prof/pmc_f$fl.log 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for fl in (3,259):
    f=glob.glob(f'gpurun_out/prof/pmc_f{fl}/*/*_counter_collection.csv')[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'persist' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print('flags',fl,k,len(v),sum(v)/len(v))
PY
