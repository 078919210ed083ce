"""Folds the rocprofv3 passes of tools/gpu_job_pmc.sh into one JSON per kernel under profiles/.
usage: python tools/summarize_pmc2.py gpurun_out <tag> <kernel substring> <algorithmic bytes per launch> profiles/<name>.json ["note"]
Corrections as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB) counts a 128-B request as 64 B for 16 B/lane
streaming reads -> x2; WRITE_SIZE (KiB) is exact."""
import collections, csv, glob, json, sys
root, tag, kern, alg, dst = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
note = sys.argv[6] if len(sys.argv) > 6 else ""
def counters(d):
    fs = glob.glob(f'{root}/{tag}_{d}/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in fs:
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}
def durations():
    f = glob.glob(f'{root}/{tag}_stats/**/*kernel_trace.csv', recursive=True)[0]
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if kern in r['Kernel_Name']]
fetch = counters('pmc_fetch').get('FETCH_SIZE'); write = counters('pmc_write').get('WRITE_SIZE')
sq = counters('pmc_sq'); sq.update(counters('pmc_sq2'))
dur = durations()
rd, wr = fetch * 1024 * 2, write * 1024
avg = sum(dur) / len(dur)
out = {"kernel": kern, "note": note,
       "source": f"rocprofv3 --kernel-trace --stats (20 dispatches) and separate --pmc passes FETCH_SIZE / WRITE_SIZE / SQ (6 dispatches each), tools/gpu_job_pmc.sh, tag {tag}",
       "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
       "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE x2 for 16 B/lane streaming reads on gfx950; WRITE_SIZE exact",
       "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4),
       "kernel_avg_us_rocprof_stats": avg, "kernel_min_us_rocprof_stats": min(dur), "kernel_max_us_rocprof_stats": max(dur),
       "achieved_GBs_algorithmic": round(alg / avg / 1e3, 1), "frac_of_8TBs": round(alg / avg / 1e3 / 8000.0, 4),
       "sq_counters": sq,
       "derived": {"mfma_busy_cycles_per_simd": sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024,
                   "wave_wait_any_frac": round(sq.get('SQ_WAIT_ANY', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1), 3),
                   "wave_wait_inst_frac": round(sq.get('SQ_WAIT_INST_ANY', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1), 3),
                   "valu_per_mfma": round(sq.get('SQ_INSTS_VALU', 0) / max(sq.get('SQ_INSTS_MFMA', 1), 1), 2)}}
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out, indent=1))
