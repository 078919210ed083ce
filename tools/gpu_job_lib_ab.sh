# usage: gpu_job_lib_ab.sh <rounds> <tag...>: bench.py's step (no CPU baseline) on the in-tree library ("main") and on csrc/build/var/lib_<tag>.so, alternating
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out
N=$1; shift
for r in $(seq 1 $N); do
  for V in main "$@"; do
    L=$R/wc_gan_amd/libwc_hip.so; [ $V != main ] && L=$R/wc_gan_amd/csrc/build/var/lib_$V.so
    python tools/bench_with_lib.py $L --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$V', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.1f' % d['value'])"
  done
done
