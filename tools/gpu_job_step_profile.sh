R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/step_prof_v19 -o s -- python3 $R/tools/step_only.py 3 > $R/gpurun_out/step_prof_v19.log 2>&1
cd $R
tail -1 gpurun_out/step_prof_v19.log
python tools/summarize_profile.py gpurun_out/step_prof_v19 gpurun_out/step_prof_v19.md gap >/dev/null; head -60 gpurun_out/step_prof_v19.md | cut -c1-150
