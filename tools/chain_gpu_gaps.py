#!/usr/bin/env python3
"""The chain step's timeline on the GPU from a rocprofv3 kernel trace (`rocprofv3 --kernel-trace -d DIR -o ch -- python3
bench.py --workload c3-chain ...`, rocpd database output): every kernel of the second-to-last step with its start, duration
and the idle gap before it, up to the next step's first kernel.  (The last step of a profiled run is followed by the
profiler's own flush and is left out.)   usage: chain_gpu_gaps.py DIR/.../ch_results.db"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [r for r in cur.execute("select name, start, end from kernels order by start") if 'sb::' in r[0]]
first = [i for i, r in enumerate(rows) if 'exonbin_kernel' in r[0]]
first = [i - 1 if i and 'iso_masks_kernel' in rows[i - 1][0] else i for i in first]   # (an annotation that is not pinned: its masks are made first)
i0, i1 = first[-2], first[-1]
t0 = rows[i0][1]
prev_end, busy = t0, 0
for n, s, e in rows[i0:i1 + 1]:
    print("%-62s start %8.3f ms  dur %7.3f ms  gap before %7.3f ms" % (n.split('(')[0][-62:], (s - t0) / 1e6, (e - s) / 1e6, max(0, s - prev_end) / 1e6))
    if n is not rows[i1][0] or s != rows[i1][1]:
        busy += max(0, e - max(s, prev_end))
        prev_end = max(prev_end, e)
print("step to step %.3f ms; kernels busy %.3f ms; idle inside the step %.3f ms; between the steps %.3f ms" % (
    (rows[i1][1] - t0) / 1e6, busy / 1e6, (prev_end - t0 - busy) / 1e6, (rows[i1][1] - prev_end) / 1e6))
