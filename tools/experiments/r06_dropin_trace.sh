cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dropin_trace; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/dropin_trace
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/dropin_trace -- python tools/bench_orbmatcher_dropin.py > gpurun_out/dropin_trace/bench.txt 2>&1
python - <<'P'
import csv,glob
k=glob.glob('gpurun_out/dropin_trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(k)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]
for r in rows:
    n=r['Kernel_Name']; n=n.split('(')[0].replace('void orbs::','').replace('orbs::','')
    out.append((n,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Workgroup_Size_X'] if 'Workgroup_Size_X' in r else r.get('Workgroup_Size','')))
# group runs of 7
i=0
while i<len(out):
    j=i
    while j<len(out) and out[j][0]==out[i][0] and out[j][2]==out[i][2]: j+=1
    d=[x[1] for x in out[i:j]]
    print(f"{out[i][0][:60]:60s} wg {out[i][2]:>5} x{j-i:3d}  min {min(d):8.1f} us  med {sorted(d)[len(d)//2]:8.1f}")
    i=j
m=glob.glob('gpurun_out/dropin_trace/**/*memory_copy_trace.csv',recursive=True)
if m:
    rows=list(csv.DictReader(open(m[0])))
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
    print('copies',len(d),'median us',sorted(d)[len(d)//2], 'max', max(d))
P
cat gpurun_out/dropin_trace/bench.txt | tail -14
