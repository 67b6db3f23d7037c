import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemv_q_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["K=8192 q4 f32+norm", "K=8192 q4 f32", "K=8192 q4 preq", "K=8192 q4 pair+norm", "K=28672 q4 f32", "K=8192 q6 f32+norm"]
rows = rows[-240:]
for i in range(12):
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[20 * i:20 * i + 20])
    print(f"N={256 if i < 6 else 4096:5d} {names[i % 6]:22s} median {d[10] / 1e3:6.2f} us  min {d[0] / 1e3:6.2f}")
