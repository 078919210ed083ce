#!/bin/bash
# development, timing only: the one-pass K6 on planes with the gradient's chunks taken from planes as well (-DWC_K6_ABL=512) against the shipped kernel
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for V in base gypl; do
  WC_LIB=$R/wc_gan_amd/csrc/build/var/lib_$V.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/k6v_$V -o s -- python3 $R/tools/stage_only.py 20 k6xsplit > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for v in "base gypl".split():
    f = glob.glob(f'gpurun_out/k6v_{v}/**/*kernel_trace.csv', recursive=True)
    if not f: print(v, 'no trace'); continue
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f[0])) if 'onepass_ring_kernel' in r['Kernel_Name'])
    print(f"{v:6s} K6 onepass_ring_kernel<false,true,true> n={len(d)} min {d[0]:.1f} med {d[len(d)//2]:.1f} max {d[-1]:.1f}")
PY
