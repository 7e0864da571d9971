import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "chol_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
last = rows[-n:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
gaps = [(int(last[i + 1]["Start_Timestamp"]) - int(last[i]["End_Timestamp"])) / 1e3 for i in range(n - 1)]
key = [k for k in last[0] if "Grid" in k and "X" in k.upper()]
grid = [int(r[key[0]]) // 256 for r in last] if key else []
print("dur us:", [round(x, 1) for x in d])
print("gaps us:", [round(x, 1) for x in gaps], "\nsum dur", round(sum(d)), "sum gaps", round(sum(gaps)))
print("grid wgs:", grid)
