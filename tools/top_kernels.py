import csv, glob
f = glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:78].replace("(anonymous namespace)::", ""), r["Calls"], r["AverageNs"], r["Percentage"])
