cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -- python /root/repo/tools/time_eval.py --L ${1:-200} --N ${2:-10000} --q ${3:-21} --seed ${4:-12345} --reps 6 > /tmp/cs.log 2>&1
f=$(ls /tmp/cs/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last evaluation: from the last plm_expand on
idx=[i for i,r in enumerate(rows) if 'plm_expand' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp']); prev=None
for r in rows[idx:idx+14]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'^void ','',n)[:44]
    print("%8.1f  gap %5.1f  dur %7.1f  grid %7s x %4s  %s"%((s-t0)/1e3,(s-prev)/1e3 if prev else 0,(e-s)/1e3,int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),r['Workgroup_Size_X'],n))
    prev=e
PY
