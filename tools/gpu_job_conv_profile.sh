R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/conv_prof -o c -- python3 $R/tools/conv_bench.py > $R/gpurun_out/conv_prof.log 2>&1
cd $R
python tools/summarize_profile.py gpurun_out/conv_prof gpurun_out/conv_prof.md >/dev/null; head -30 gpurun_out/conv_prof.md | cut -c1-160
grep -A1 " N" gpurun_out/conv_prof.log | grep -v Warn | head -40
