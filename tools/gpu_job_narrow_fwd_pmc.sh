cd "$GRAFT_REPO_ROOT"; R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nf_stats -o s -- python3 $R/tools/narrow_fwd_only.py 20 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/nf_pmc -o s -- python3 $R/tools/narrow_fwd_only.py 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/nf_pmc2 -o s -- python3 $R/tools/narrow_fwd_only.py 6 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/nf_stats/**/*kernel_trace.csv', recursive=True)[0]
d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'conv_fwd_narrow' in r['Kernel_Name'])
print('narrow fwd us: n=%d min %.1f med %.1f max %.1f' % (len(d), d[0], d[len(d)//2], d[-1]))
for dd in ('nf_pmc', 'nf_pmc2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f'gpurun_out/{dd}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_fwd_narrow' in r['Kernel_Name']:
                acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    print({k: round(sum(v.values()) / len(v)) for k, v in acc.items()})
PY
