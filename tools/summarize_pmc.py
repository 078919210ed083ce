"""Folds the rocprofv3 --pmc passes of the K3 apply kernel (separate FETCH_SIZE / WRITE_SIZE / SQ runs) and the
--kernel-trace --stats pass into profiles/<name>.json.  Corrections as MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE counts a 128-B request as 64 B for 16 B/lane streaming reads (x2); WRITE_SIZE is exact; both in KiB units... 
usage: python tools/summarize_pmc.py gpurun_out profiles/r1_apply_k3_pmc.json"""
import collections, csv, glob, json, sys
root, dst = sys.argv[1], sys.argv[2]
KERNEL = 'affine_ring_kernel'
def counters(d):
    f = glob.glob(f'{root}/{d}/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if KERNEL in r['Kernel_Name']:
            acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}
def durations(d):
    f = glob.glob(f'{root}/{d}/**/*kernel_trace.csv', recursive=True)[0]
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if KERNEL in r['Kernel_Name']]
fetch = counters('k3_pmc_fetch')['FETCH_SIZE']; write = counters('k3_pmc_write')['WRITE_SIZE']; sq = counters('k3_pmc_sq')
dur = durations('k3_stats')
M, C = 128 * 32 * 32, 256
alg = 2 * M * C * 4 + (C * C + C) * 4
rd, wr = fetch * 1024 * 2, write * 1024
out = {
 "kernel": "affine_ring_kernel<256,false> (wc_apply_f32 with plan), 128x32x32x256 fp32, MI355X",
 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters (separate passes) -- python3 tools/apply_only.py 6; "
           "averages over the 6 dispatches; kernel time from --kernel-trace --stats over 20 dispatches (tools/gpu_job_r1c.sh)",
 "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
 "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> x2 for a 16 B/lane streaming read (LDS-DMA alike); WRITE_SIZE exact",
 "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
 "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4),
 "kernel_avg_us_rocprof_stats": sum(dur) / len(dur), "kernel_min_us_rocprof_stats": min(dur), "kernel_max_us_rocprof_stats": max(dur),
 "sq_counters": sq,
 "derived": {
   "mfma_busy_cycles_per_simd": sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024,
   "wave_wait_any_frac": round(sq.get('SQ_WAIT_ANY', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1), 3),
   "wave_wait_inst_frac": round(sq.get('SQ_WAIT_INST_ANY', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1), 3),
   "wave_cycles_per_wave_x4": sq.get('SQ_WAVE_CYCLES', 0) * 4 / 2048,
 },
}
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out, indent=1))
