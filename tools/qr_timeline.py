"""Kernel timeline of the last CholeskyQR2 factor() in a rocprofv3 kernel trace of tools/qr_only.py (3 calls)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = [i for i, r in enumerate(rows) if 'gram_ts_kernel' in r['Kernel_Name']]
a = g[-2]                                   # the last call's first Gram kernel
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:]:
    st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if st > 25e6: break
    print(f"+{st / 1e3:9.1f} us  {(en - st) / 1e3:8.1f} us  {r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70]}")
