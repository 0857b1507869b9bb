"""Scratch: from a rocprofv3 kernel trace of tools/quick_build.py, the timeline of the last few build calls (start and end of every
kernel relative to the call's first kernel, us)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'ndt::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
calls, cur = [], []
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ndt::', '')
    first = name.startswith('k_bounds_parts') or (name.startswith('k_chunk_sort<4>') and (not cur or not cur[-1][0].startswith('k_bounds')))
    if first and cur:
        calls.append(cur); cur = []
    cur.append((name, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
calls.append(cur)
for c in calls[8:11] + calls[-3:]:
    t0 = c[0][1]
    print(' | '.join(f"{n} {1e-3 * (s - t0):.1f}-{1e-3 * (e - t0):.1f}" for n, s, e in c))
