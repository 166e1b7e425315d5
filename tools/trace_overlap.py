import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'conv_f32' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ov=0; tot=0
for a,b in zip(rows,rows[1:]):
    e=int(a['End_Timestamp']); s=int(b['Start_Timestamp'])
    tot+=1
    if s<e: ov+=1
print('conv dispatches',len(rows),'overlapping consecutive pairs',ov,'of',tot)
t0=int(rows[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in rows)
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
print('span ms',(t1-t0)/1e6,'sum kernel ms',busy/1e6, 'queues', collections.Counter(r['Queue_Id'] for r in rows))
