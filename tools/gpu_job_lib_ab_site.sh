# usage: gpu_job_lib_ab_site.sh <rounds> <tag...>: bench.py's site figures (producer, forward site, site + producer, K3 in flow) on the in-tree library and on csrc/build/var/lib_<tag>.so
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out
N=$1; shift
for r in $(seq 1 $N); do
  for V in main "$@"; do
    L=$R/wc_gan_amd/libwc_hip.so; [ $V != main ] && L=$R/wc_gan_amd/csrc/build/var/lib_$V.so
    python tools/bench_with_lib.py $L --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$V', 'step %.3f ms' % d['ms_per_step'], '| producer', r['producer_us']['residual add as the layers run it'], '| forward site', r['forward_site_us'], '| site + producer', r['forward_site_plus_producer_us']['planes'],
              '| K3 launch', r['launch_us'], 'in flow', r['in_flow_us'], 'back to back', r['back_to_back_us'], '| stage producer', [v['us'] for k, v in r['site_stages'].items() if k.startswith('producer: residual add -> pre-split planes + K1')])"
  done
done
