set -x
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -4
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r1c.json 2> gpurun_out/bench_r1c.err; tail -1 gpurun_out/bench_r1c.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k3_stats -o k3 -- python3 $R/tools/apply_only.py 20 > $R/gpurun_out/k3_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/k3_pmc_fetch -o k3 -- python3 $R/tools/apply_only.py 6 > $R/gpurun_out/k3_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/k3_pmc_write -o k3 -- python3 $R/tools/apply_only.py 6 > $R/gpurun_out/k3_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/k3_pmc_sq -o k3 -- python3 $R/tools/apply_only.py 6 > $R/gpurun_out/k3_pmc_sq.log 2>&1
cd $R
python tools/summarize_profile.py gpurun_out/k3_stats gpurun_out/k3_stats.md; head -12 gpurun_out/k3_stats.md | cut -c1-160
ls gpurun_out/k3_pmc_fetch gpurun_out/k3_pmc_sq
