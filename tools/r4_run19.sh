cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
export DN_TINY_LEN=0
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU -d $O/sq1 -o run --output-format csv -- python3 tools/trace_stats.py 512 100 1200 > $O/sq1.log 2>&1; echo "sq1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $O/sq2 -o run --output-format csv -- python3 tools/trace_stats.py 512 100 1200 > $O/sq2.log 2>&1; echo "sq2 rc=$?"
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob('gpurun_out/r4s/sq*/run_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_baseline<10, 128>' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
for k in sorted(tot): print('%-22s %.3e' % (k, tot[k]))
PY
