R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "256 0" "512 0" "256 1"; do
  set -- $cfg
  export WC_WRW_TARGET=$1
  if [ "$2" = "1" ]; then export WC_WRW_NOSWAP=1; else unset WC_WRW_NOSWAP; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wrw_$1_$2 -o s -- python3 $R/tools/wrw_only.py 128 32 256 256 same > /dev/null 2>&1
  echo "target $1 noswap $2"; grep -E "wrw" $R/gpurun_out/wrw_$1_$2/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
