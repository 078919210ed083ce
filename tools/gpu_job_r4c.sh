#!/bin/bash
# round 4: the whole GPU suite, then the bench line (hipGraph step + roofline of the layers' kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r4c_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err
tail -5 gpurun_out/r4c_tests.txt; tail -c 3000 gpurun_out/r4c_bench.err; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r4c_bench.json').read().strip().splitlines()[-1])
    r = d['roofline']
    print("value", d['value'], "ms", d['ms_per_step'], "eager", d.get('eager_launch'))
    for k in ('kernel', 'launch_us', 'frac', 'frac_of_stream_copy', 'back_to_back_us', 'in_flow_us', 'forward_site_us', 'forward_site_fp32_input_us', 'producer_us', 'forward_site_plus_producer_us', 'stream_copy_GBs', 'error'):
        print(k, r.get(k))
    for k, v in r.get('k3_kernels', {}).items(): print("  ", v['launch_us'], v['frac_of_stream_copy'], k[:70])
    for k, v in r.get('site_stages', {}).items(): print("  ", v, k[:80])
except Exception as e:
    print("no bench line", e)
PY
