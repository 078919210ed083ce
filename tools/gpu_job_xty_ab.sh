#!/bin/bash
# A/B of two builds of the K1/K4 kernels by rocprofv3 kernel durations: $1 = extra flags of variant B (variant A = defaults)
R=$PWD
mkdir -p gpurun_out
run() {  # tag
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xtyab_$1 -o s -- python3 $R/tools/xty_only.py 30 > $R/gpurun_out/xtyab_$1.log 2>&1
    cd $R
    python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/xtyab_$1/s_kernel_trace.csv')))
for key in ('xty_f16x3_kernel<256, false', 'xty_f16x3_kernel<256, true'):
    d=sorted((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if key in r['Kernel_Name'])
    print('$1', key, len(d), 'min %.1f med %.1f max %.1f' % (d[0], d[len(d)//2], d[-1]))
PY
}
python -m wc_gan_amd.build --force > /dev/null 2>&1; run A1
WC_EXTRA_FLAGS="$1" python -m wc_gan_amd.build --force > /dev/null 2>&1; run B1
python -m wc_gan_amd.build --force > /dev/null 2>&1; run A2
WC_EXTRA_FLAGS="$1" python -m wc_gan_amd.build --force > /dev/null 2>&1; run B2
python -m wc_gan_amd.build --force > /dev/null 2>&1
