import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    # find launches: index of the hipModuleLaunch / hipLaunchKernel calls
    big = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 300000]
    print(f, len(rows), "calls;", len(big), "longer than 300 us")
    for r in big[-40:]:
        print("  %10.3f ms  %-40s %8.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e6, r["Function"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
