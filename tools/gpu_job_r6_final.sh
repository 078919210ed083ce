# Round 6 evidence on one box: the GPU suite, K2's pipe check (both kernels + stage boundaries), every site under rocprofv3, the bench line.
cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|version\|Hostname\|Librccl" | tail -12 > gpurun_out/r6_gpu_suite.txt
{ echo "K2 per call (tools/k2_pipe_check.py: ops.factor = K1 tail + prepare + factor + inverse; W L = I residual over 10 repeats; best of 5 back-to-back loops of 30 calls), round 6 default: cholesky_phased_kernel, a relay of three factorising workgroups at C = 256"
  python tools/k2_pipe_check.py 10 2>&1 | grep -v amdgpu.ids
  echo; echo "the same with WC_K2_FUSED_R5=1: round 5's one-workgroup factorisation (cholesky_fused_kernel, with this round's row-major panel)"
  WC_K2_FUSED_R5=1 python tools/k2_pipe_check.py 10 2>&1 | grep -v amdgpu.ids
  echo; echo "relay stage boundaries (WC_K2_BOUNDS, tools/k2_time_only.py, C = 256, one matrix):"
  for B in 16 6 4,9 5,10 3,6,10 4,8,12; do echo -n "bounds $B: "; WC_K2_BOUNDS=$B python tools/k2_time_only.py wc_gan_amd/libwc_hip.so 2>&1 | grep -v amdgpu.ids | tail -1; done
} > gpurun_out/r6_k2_pipe_check.txt
bash tools/gpu_job_sites_all.sh > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 2>gpurun_out/r6_bench.err | tail -1 > gpurun_out/r6_bench_line.json
tail -3 gpurun_out/r6_gpu_suite.txt; head -12 gpurun_out/r6_k2_pipe_check.txt | cut -c1-160; head -12 gpurun_out/r6_sites_all.txt | cut -c1-150; python -c "
import json; d=json.load(open('gpurun_out/r6_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_us'], d['roofline']['forward_site_plus_producer_us'])"
