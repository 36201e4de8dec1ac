import csv,glob,sys,collections
d=sys.argv[1]
dur={}
for f in glob.glob(d+'/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'match_encode_kernel' in r['Kernel_Name']:
            dur[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
cnt=collections.defaultdict(dict)
for f in glob.glob(d+'/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'match_encode_kernel' in r['Kernel_Name']:
            cnt[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
rows=[]
for k in sorted(dur,key=int):
    c=cnt.get(k,{})
    rows.append((int(k),dur[k],c))
for k,du,c in rows[:3]+rows[-8:]:
    print(k,du,{n:round(v) for n,v in c.items()}, {n:round(v/du,3) for n,v in c.items()})
