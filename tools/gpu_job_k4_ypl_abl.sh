cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for V in base ypl; do
  WC_LIB=$R/wc_gan_amd/csrc/build/var/lib_$V.so rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/k4v_$V -o s -- python3 $R/tools/stage_only.py 20 k4xsplit > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for v in "base ypl".split():
    f = glob.glob(f'gpurun_out/k4v_{v}/**/*kernel_trace.csv', recursive=True)
    if not f: print(v, 'no trace'); continue
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f[0])) if 'xty_f16x3_kernel<256, true' in r['Kernel_Name'])
    print(f"{v:6s} K4 xty_f16x3_kernel<256,true,2,true> n={len(d)} min {d[0]:.1f} med {d[len(d)//2]:.1f} max {d[-1]:.1f}")
PY
