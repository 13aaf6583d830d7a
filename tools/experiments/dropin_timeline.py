"""one frame of the per-call loop from a rocprofv3 kernel trace: start offset, duration, queue, kernel
(rocprofv3 --kernel-trace -d gpurun_out/dtl -o s --output-format csv -- python3 tools/dropin_probe.py 257 2000;
python3 tools/experiments/dropin_timeline.py gpurun_out/dtl/s_kernel_trace.csv [frame])"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'ransac_refit' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
a, b = idx[k], idx[k + 1]
t0 = int(rows[a + 1]['Start_Timestamp'])
for r in rows[a + 1:b + 2]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f %7.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0][:40]))
spans = [int(rows[idx[i + 1]]['End_Timestamp']) - int(rows[idx[i]]['End_Timestamp']) for i in range(20, len(idx) - 1)]
spans.sort()
print("frame period (refit end to refit end): median %.1f us, p10 %.1f, p90 %.1f" % (spans[len(spans) // 2] / 1e3, spans[len(spans) // 10] / 1e3, spans[9 * len(spans) // 10] / 1e3))
