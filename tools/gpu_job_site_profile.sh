R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/site_prof -o s -- python3 $R/tools/kernel_bench.py > $R/gpurun_out/site_prof.log 2>&1
cd $R
cat gpurun_out/site_prof.log | grep -v "^W2\|^E2\|^I2" | tail -14
python tools/summarize_profile.py gpurun_out/site_prof gpurun_out/site_prof.md >/dev/null; head -40 gpurun_out/site_prof.md | cut -c1-150
