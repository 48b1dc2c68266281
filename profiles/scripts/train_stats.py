import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
one=[r for r in rows if 'k_adam' in r['Name']][0]
steps=int(one['Calls'])
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("steps", steps, "total per step us %.0f"%(tot/steps/1e3), "launches/step %.0f"%(sum(int(r['Calls']) for r in rows)/steps))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls'])/steps:7.1f} {float(r['AverageNs'])/1e3:8.1f} {int(r['TotalDurationNs'])/steps/1e3:8.1f}")
