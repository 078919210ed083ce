cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for V in base nostore nogather nomfma nostoregather; do
  WC_LIB=$R/wc_gan_amd/csrc/build/var/lib_$V.so rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/nfv_$V -o s -- python3 $R/tools/narrow_fwd_only.py 20 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for v in "base nostore nogather nomfma nostoregather".split():
    f = glob.glob(f'gpurun_out/nfv_{v}/**/*kernel_trace.csv', recursive=True)
    if not f: print(v, 'no trace'); continue
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f[0])) if 'conv_fwd_narrow' in r['Kernel_Name'])
    print(f"{v:14s} n={len(d)} min {d[0]:.1f} med {d[len(d)//2]:.1f}")
PY
