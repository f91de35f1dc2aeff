import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Name", "")
    if any(k in n for k in sys.argv[2].split(",")):
        print("%-42s calls %s avg %.1f us" % (n.split("(")[0][-42:], r.get("Calls"), float(r.get("AverageNs", r.get("Average", 0))) / 1e3))
