"""One step of a bench.py run from its rocprofv3 kernel trace: the launches in start order with the gap in front of each.
usage: python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv
import sys

rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))), key=lambda e: e[0])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marks = [i for i, e in enumerate(rows) if e[2].startswith(('void k_group_mid', 'void k_front_', 'void k_group_small'))]
lo, hi = marks[-back - 1], marks[-back]
step = rows[lo:hi]
t0 = step[0][0]
prev_end = t0
gaps = 0.0
busy = 0.0
for s, e, n in step:
    gap = (s - prev_end) / 1e3
    if gap > 0:
        gaps += gap
    print('%9.1f us  +%6.1f gap  %7.1f us  %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, n[:110]))
    prev_end = max(prev_end, e)
    busy += (e - s) / 1e3
print('step: %.1f us from first start to last end, %d launches, kernel time %.1f us, gaps %.1f us' % ((prev_end - t0) / 1e3, len(step), busy, gaps))
