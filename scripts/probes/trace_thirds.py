import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# group consecutive sarl_reg_kernel calls into chunks of 90
names=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].split('(')[0][-40:]
    names[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,v in names.items():
    if len(v)>=180:
        k=len(v)//3
        print(n.ljust(42), len(v), [round(sum(v[i*k:(i+1)*k])/k,1) for i in range(3)])
