#!/bin/bash
# A/B of two builds of K2's fused launch by rocprofv3 kernel durations: $1 = extra flags of variant B (variant A = defaults)
R=$PWD
mkdir -p gpurun_out
run() {  # tag
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k2ab_$1 -o s -- python3 $R/tools/factor_only.py 256 1 > /dev/null 2>&1
    cd $R
    python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/k2ab_$1/s_kernel_trace.csv')))
d=sorted((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'cholesky_fused' in r['Kernel_Name'])
print('$1', len(d), 'min %.1f med %.1f max %.1f' % (d[0], d[len(d)//2], d[-1]))
PY
}
python -m wc_gan_amd.build --force > /dev/null 2>&1; run A1
WC_EXTRA_FLAGS="$1" python -m wc_gan_amd.build --force > /dev/null 2>&1; run B1
python -m wc_gan_amd.build --force > /dev/null 2>&1; run A2
WC_EXTRA_FLAGS="$1" python -m wc_gan_amd.build --force > /dev/null 2>&1; run B2
python -m wc_gan_amd.build --force > /dev/null 2>&1
