import csv,glob,sys
import numpy as np
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
fw=[r for r in rows if 'lsmr_fwd' in r[2]]
st=np.array([r[0] for r in fw]); d=np.diff(st)
print('fwd start-to-start: median %.1f us, p10 %.1f, p90 %.1f, n=%d'%(np.median(d)/1e3,np.percentile(d,10)/1e3,np.percentile(d,90)/1e3,len(d)))
# durations and true gaps inside an iteration
names=['lsmr_fwd','lsmr_adj','lsmr_update3']
idx=[i for i,r in enumerate(rows) if 'lsmr_fwd' in r[2]]
g=[];du=[]
for i in idx[100:1100]:
    if i+3<len(rows) and 'lsmr_adj' in rows[i+1][2] and 'lsmr_update3' in rows[i+2][2] and 'lsmr_fwd' in rows[i+3][2]:
        du.append([rows[i+k][1]-rows[i+k][0] for k in range(3)])
        g.append([rows[i+k+1][0]-rows[i+k][1] for k in range(3)])
print('dur us', np.median(np.array(du),0)/1e3, 'gap us', np.median(np.array(g),0)/1e3)
