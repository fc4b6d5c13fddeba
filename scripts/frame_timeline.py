"""Print the kernel timeline of one frame from a rocprofv3 kernel trace: python scripts/frame_timeline.py <trace.csv> [frame]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'preprocess_kernel' in r['Kernel_Name']]
f = int(sys.argv[2]) if len(sys.argv) > 2 else -3
i0, i1 = idx[f], idx[f + 1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void olsr::', '')[:40]
    print(f"{name:42s} start {(s-t0)/1e3:8.1f}us dur {(e-s)/1e3:7.1f}us gap {(s-prev_end)/1e3:6.1f}us grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Workgroup_Size_X']} lds {r['LDS_Block_Size']} vgpr {r['VGPR_Count']}")
    prev_end = e
print(f"frame total {(prev_end - t0)/1e3:.1f} us")
