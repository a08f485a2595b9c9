import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    if "gather_gemm" not in n and "wgrad" not in n: continue
    key = n.split("::")[1].split("(")[0] + " grid" + r["Grid_Size"]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(key, r["Counter_Name"])] += 1
for k, v in agg.items():
    n = cnt[(k, "SQ_WAVE_CYCLES")] or 1
    wc = v.get("SQ_WAVE_CYCLES", 1)
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 16
    print(k, "launches", n)
    for c in sorted(v):
        print(f"   {c:28s} {v[c]/n:14.0f}  {v[c]/wc:6.3f} of wave quad-cycles" + (f"  {v[c]/mf:6.2f} per MFMA" if mf and "INSTS" in c else ""))
