#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out/k1split
timeout 900 python -m pytest tests/test_split_gpu.py -x -q -m gpu -k "stats_split or forward_site" 2>&1 | tail -5 | tee gpurun_out/k1split/pytest.txt
timeout 600 python tools/k1_split_variants.py 2>&1 | tee gpurun_out/k1split/variants.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k1split/stats -o s -- python3 $R/tools/k1_split_time.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/k1split/pmc -o s -- python3 $R/tools/k1_split_time.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/k1split/stats/*kernel_stats.csv'):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 5: print(r['Name'][:80], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
for f in glob.glob('gpurun_out/k1split/pmc/*counter_collection.csv'):
    acc = {}
    for r in csv.DictReader(open(f)):
        if 'xtx_split' in r['Kernel_Name'] or 'xty_f16x3' in r['Kernel_Name']:
            k = (r['Kernel_Name'][:40], r['Counter_Name']); acc.setdefault(k, []).append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print(k, sum(v) / len(v))
PY
