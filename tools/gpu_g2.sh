set -u
O=gpurun_out/g2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -k "not cfg2" 2>&1 | tail -12 > $O/graph_tests.log; tail -6 $O/graph_tests.log
timeout 600 python tools/time_closed_loop.py 50 40 50 24 $O/closed_cfg2.json > $O/closed_cfg2.log 2>&1; tail -22 $O/closed_cfg2.log | cut -c1-250
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cl -o p -- python $GRAFT_REPO_ROOT/tools/time_closed_loop.py 50 40 50 24 > $GRAFT_REPO_ROOT/$O/prof_cl.log 2>&1)
cp $(find /tmp/prof_cl -name "*kernel_stats.csv" | head -1) $O/kernel_stats_closed_cfg2.csv
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_cl/**/*kernel_trace.csv', recursive=True)
if f:
    rows=list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    # split into steps by k_update_aabb launches
    steps=[]; cur=None
    for r in rows:
        n=r['Kernel_Name']
        if 'k_update_aabb' in n:
            cur={}; steps.append(cur)
        if cur is None: continue
        key=n.split('(')[0].replace('void ','').replace('avn::','')[:40]
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        cur[key]=cur.get(key,0)+d
    for i in (4,8,12,20):
        if i<len(steps):
            s=steps[i]; tot=sum(s.values())
            print('step',i,'kernel time total %.0f us'%tot, sorted(((round(v),k) for k,v in s.items()),reverse=True)[:14])
PY
head -25 $O/kernel_stats_closed_cfg2.csv | cut -c1-200
