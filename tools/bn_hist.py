import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
steps=26
acc=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "bn_" in n and "fv2p" in n:
        k=n.split("(")[0].replace("void ","")
        acc[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in acc.items():
    v.sort()
    big=[x for x in v if x>40]
    print(f"{k:40s} n/step {len(v)/steps:5.1f} total/step {sum(v)/steps:7.1f} us  median {v[len(v)//2]:6.1f}  p90 {v[int(len(v)*0.9)]:6.1f} max {v[-1]:6.1f}  >40us: {len(big)/steps:4.1f}/step {sum(big)/steps:7.1f} us/step")
