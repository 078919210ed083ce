# usage: gpu_job_kvar.sh <stage_only mode> <kernel-name substring> <variant tags...>: the kernel's durations under rocprofv3 for csrc/build/var/lib_<tag>.so
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
MODE=$1; KN=$2; shift 2
for V in "$@"; do
  WC_LIB=$R/wc_gan_amd/csrc/build/var/lib_$V.so rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kvar_$V -o s -- python3 $R/tools/stage_only.py 20 $MODE > /dev/null 2>&1
done
cd $R
python - "$KN" "$@" <<'PY'
import csv, glob, sys
kn = sys.argv[1]
for v in sys.argv[2:]:
    f = glob.glob(f'gpurun_out/kvar_{v}/**/*kernel_trace.csv', recursive=True)
    if not f: print(v, 'no trace'); continue
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f[0])) if kn in r['Kernel_Name'])
    print(f"{v:10s} {kn} n={len(d)} min {d[0]:.1f} med {d[len(d)//2]:.1f} max {d[-1]:.1f}" if d else f"{v} none")
PY
