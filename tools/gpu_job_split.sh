#!/bin/bash
# round 3: parity of the pre-split path, variants timing, kernel trace + PMC of the default build
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
mkdir -p gpurun_out/split
timeout 900 python -m pytest tests/test_split_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/split/pytest.txt
cat gpurun_out/split/pytest.txt
timeout 600 python tools/split_run_variants.py 2>&1 | tee gpurun_out/split/variants.txt
timeout 300 python tools/k3_split_time.py 2>&1 | tee gpurun_out/split/time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/split/stats -o s -- python3 $R/tools/k3_split_time.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/split/pmc_sq2 -o s -- python3 $R/tools/k3_split_time.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/split/stats/*kernel_stats.csv'):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 8: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
for f in glob.glob('gpurun_out/split/pmc_sq2/*counter_collection.csv'):
    acc = {}
    for r in csv.DictReader(open(f)):
        if 'apply_split' in r['Kernel_Name'] or 'affine_ring' in r['Kernel_Name']:
            k = (r['Kernel_Name'][:40], r['Counter_Name']); acc.setdefault(k, []).append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print(k, sum(v) / len(v))
PY
