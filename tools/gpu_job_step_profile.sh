R=$PWD
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/step_prof_v4 -o s -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/step_prof_v4.log 2>&1
cd $R
python tools/summarize_profile.py gpurun_out/step_prof_v4 gpurun_out/step_prof_v4.md >/dev/null; grep -E "sn_|cholesky|gemv|NormTwo" gpurun_out/step_prof_v4.md | cut -c1-200
