import csv, sys, collections
f, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if "k_conv_rowsr" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
if not acc:
    print(v, "no k_conv_rowsr rows in", f)
for k in sorted(acc):
    print("%-8s %-40s %16.0f per launch (%d launches)" % (v, k, acc[k] / n[k], n[k]))
