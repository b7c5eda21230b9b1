"""Per-frame kernel breakdown of a rocprofv3 kernel trace of tools/prof_sift.py: python tools/sift_frames.py <kernel_trace.csv>"""
import csv, collections, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
frames = []; cur = None
for r in tr:
    n = r['Kernel_Name'].split('(')[0].replace('uvo::', '').replace('void ', '')
    if n == 'k_sift_resize2x':
        cur = collections.defaultdict(lambda: [0.0, 0]); frames.append([int(r['Grid_Size_X']), cur, int(r['Start_Timestamp']), 0])
    if cur is not None:
        cur[n][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cur[n][1] += 1
        frames[-1][3] = int(r['End_Timestamp'])
seen = set()
for g, f, a, b in frames[2:]:
    if g in seen: continue
    seen.add(g)
    print(f"frame with doubled width {g}: span {(b - a) / 1e3:.1f} us, kernel sum {sum(v[0] for v in f.values()):.1f} us")
    for k, v in sorted(f.items(), key=lambda x: -x[1][0]): print('   %-44s %9.1f us  %4d launches' % (k[:44], v[0], v[1]))
