"""Forward and backward phase of a replayed train step from a rocprofv3 kernel trace: per phase the main queue's busy time and
kernel-boundary gaps, and every kernel name by queue with count / total / average (in-step durations: beside the other stream).
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --reps 1 --steps 5 --warmup 2 --no-cpu-baseline
    python tools/phase_breakdown.py DIR/**/t_kernel_trace.csv [rows per phase]"""
import csv, sys, collections, re
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
opt=[r for r in rows if "adam_multi_kernel" in r[3]]
a,b=opt[-2][1],opt[-1][1]
step=[r for r in rows if a<=r[0]<b]
# forward = until the first kernel whose name contains 'bwd' or 'wgrad'
t_bwd=next(r[0] for r in step if ('bwd' in r[3] or 'wgrad' in r[3]))
print('step ms',(b-a)/1e6,'forward ms',(t_bwd-a)/1e6)
def short(n):
    n=n.replace('(anonymous namespace)::','').replace('void ','')
    n=re.sub(r'\(.*','',n)
    return n[:80]
for phase,lo,hi in (('forward',a,t_bwd),('backward',t_bwd,b)):
    ks=[r for r in step if lo<=r[0]<hi]
    g=collections.defaultdict(lambda:[0,0.0])
    for r in ks:
        g[(r[2],short(r[3]))][0]+=1; g[(r[2],short(r[3]))][1]+=(r[1]-r[0])/1e3
    # idle on main queue
    mainq=collections.Counter(r[2] for r in ks).most_common(1)[0][0]
    mk=sorted([r for r in ks if r[2]==mainq])
    idle=sum(max(0,mk[i+1][0]-mk[i][1]) for i in range(len(mk)-1))/1e3
    print(f'== {phase}: {len(ks)} kernels, main queue {mainq}: {len(mk)} kernels busy {sum(r[1]-r[0] for r in mk)/1e6:.2f} ms, gaps {idle/1e3:.2f} ms')
    for (q,n),(c,t) in sorted(g.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
        print(f'  q{q} {t/1e3:7.3f} ms {c:4d}x {t/c:7.1f} us  {n}')
