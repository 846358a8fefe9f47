"""Timeline of one sweep of a side-by-side batch from a rocprofv3 kernel trace (t_kernel_trace.csv of
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 profiles/batch_run.py N p q B`): the kernels between the
last-but-one and the last k_prep_nodes launch -- start (us from the first), duration, kernel, grid in workgroups -- and, per
tile step, when each of its diagonal blocks started.
usage: python profiles/batch_timeline.py t_kernel_trace.csv [sweeps back, default 2]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                     int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])),
                     int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y'])), int(r['Queue_Id'])))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
preps = [i for i, r in enumerate(rows) if r[2].startswith('k_prep_nodes')]
i0, i1 = preps[-back - 1], preps[-back]
t0 = rows[i0][0]
print('one sweep: %.1f us' % ((rows[i1][0] - t0) * 1e-3))
for s, e, name, gx, gy, qid in rows[i0:i1]:
    short = name.replace('void ', '')
    short = short[:short.index('(')] if '(' in short else short
    print('%9.1f %8.1f  q%-2d %-44s %5d x %d' % ((s - t0) * 1e-3, (e - s) * 1e-3, qid, short[:44], gx, gy))
