import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks=[i for i,r in enumerate(rows) if "fps_wave_k" in r[2]]
print(len(rows), "kernels,", len(marks), "steps")
import collections
acc=collections.defaultdict(list)
for a,b in zip(marks[len(marks)//2:-1], marks[len(marks)//2+1:]):
    seg=[r for r in rows[a:b] if "conv_rows_ksplit<128, false, 64, 1" in r[2]]
    for k,r in enumerate(seg):
        acc[(len(seg),k)].append((r[1]-r[0])/1e3)
for (n,k),v in sorted(acc.items()):
    print("launches/step %d  #%d  avg %.1f us  (n=%d)"%(n,k,sum(v)/len(v),len(v)))
