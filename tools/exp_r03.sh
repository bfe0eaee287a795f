mkdir -p gpurun_out
repo=$PWD
out=$repo/gpurun_out/tl20
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o bench -- python3 "$repo/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --repeats 3 --no-other-configs --profile-sweeps 2 > "$out/trace.log" 2>&1
cd $repo
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tl20/trace/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'pa::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
# split into calls by gaps > 40us with no kernel running
calls=[];cur=[rows[0]];end=int(rows[0]['End_Timestamp'])
for r in rows[1:]:
    if int(r['Start_Timestamp'])-end>30000: calls.append(cur);cur=[]
    cur.append(r); end=max(end,int(r['End_Timestamp']))
calls.append(cur)
for c in calls:
    s=int(c[0]['Start_Timestamp']);e=max(int(r['End_Timestamp']) for r in c)
    print(len(c),'kernels span',(e-s)/1e3,'us start',(s-t0)/1e3)
c=[c for c in calls if 80<len(c)<95][-1]
s0=int(c[0]['Start_Timestamp'])
for r in c[:10]+c[-12:]:
    s,e=int(r['Start_Timestamp'])-s0,int(r['End_Timestamp'])-s0
    print(f"{s/1e3:9.2f} -> {e/1e3:9.2f} ({(e-s)/1e3:6.2f}) q{r['Queue_Id']} {r['Kernel_Name'].replace('void ','').split('(')[0][:46]}")
PY
