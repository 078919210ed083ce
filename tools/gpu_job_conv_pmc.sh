R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/conv_stats -o c -- python3 $R/tools/conv_only.py 20 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/conv_pmc_sq -o c -- python3 $R/tools/conv_only.py 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/conv_pmc_lds -o c -- python3 $R/tools/conv_only.py 6 > /dev/null 2>&1
# (a FETCH_SIZE / WRITE_SIZE pass of this job ran into the 900 s limit on the box: left out)
cd $R
python - <<'PY'
import csv, glob, collections
def counters(d):
    fs = glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in fs:
        for r in csv.DictReader(open(f)):
            if 'conv_f16x3' in r['Kernel_Name']:
                acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}
for d in ('conv_pmc_sq', 'conv_pmc_lds', 'conv_pmc_mem'):
    print(d, counters(d))
f = glob.glob('gpurun_out/conv_stats/**/*kernel_trace.csv', recursive=True)[0]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'conv_f16x3' in r['Kernel_Name']]
print('durations us: n=%d mean %.1f min %.1f max %.1f' % (len(dur), sum(dur) / len(dur), min(dur), max(dur)))
PY
