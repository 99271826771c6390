"""Scheduling model of the rounds of one 1024-problem step on the measured per-problem QP chains (profiles/r02_qp_iters_1024.npy,
written by scripts/gpu_dump_traces.py): 8 XCDs of 32 CUs, workgroup g on XCD g mod 8, one workgroup per CU, in-order dispatch.
Compares lock-step rounds with whole-pass selection under several priorities.  No GPU needed."""
import heapq, numpy as np
US=1.014e-3; OVH=0.25; CUS=256; NX=8; CPX=CUS//NX
a=np.load('profiles/r02_qp_iters_1024.npy')
chains=[[int(v) for v in row[1:] if v>0] for row in a]
B=len(chains)
tot=sum(sum(c) for c in chains); print("ideal",tot*US/CUS)
def round_time(run, left, base):
    tmax=0.0
    for c in range(NX):
        durs=[min(base,left[b])*US for b in run if b%NX==c]
        if not durs: continue
        h=[0.0]*CPX; heapq.heapify(h)
        for d in durs: heapq.heappush(h, heapq.heappop(h)+d)   # in-order dispatch to the earliest free CU
        tmax=max(tmax,max(h))
    return tmax
def sim(policy, base=6250, perclass=True, extra=0):
    qi=[0]*B; left=[c[0] for c in chains]; served=[0]*B
    active=set(range(B)); t=0.0; rounds=0
    while active:
        ids=sorted(active); A=len(ids)
        if policy=="all" or A<=CUS: run=ids
        else:
            P=A//CUS
            if policy=="rr": key=lambda b:(served[b], b)
            elif policy=="stage": key=lambda b:(qi[b], served[b], b)
            elif policy=="oracle": key=lambda b:(-(left[b]+sum(chains[b][qi[b]+1:])), b)
            if perclass:
                run=[]
                for c in range(NX):
                    cl=sorted([b for b in ids if b%NX==c], key=key)
                    run+=cl[:CPX*P]
            else:
                run=sorted(ids,key=key)[:P*CUS]
        t+=round_time(run,left,base)+OVH; rounds+=1
        for b in run:
            it=min(base,left[b]); left[b]-=it; served[b]+=1
            if left[b]<=0:
                qi[b]+=1
                if qi[b]>=len(chains[b]): active.discard(b)
                else: left[b]=chains[b][qi[b]]
    return t,rounds
print("all           -> %.0f ms, %d rounds"%sim("all"))
for pol in ("stage","rr","oracle"):
    print(pol,"global    -> %.0f ms, %d rounds"%sim(pol,perclass=False))
    print(pol,"per class -> %.0f ms, %d rounds"%sim(pol,perclass=True))
for base in (3125, 12500):
    print("rr per class base",base,"-> %.0f ms, %d rounds"%sim("rr",base=base))

def sim2(keyname, base=6250, MAXIT=100000):
    qi=[0]*B; left=[c[0] for c in chains]; served=[0]*B; done=[0]*B
    active=set(range(B)); t=0.0; rounds=0
    while active:
        ids=sorted(active); A=len(ids)
        if A<=CUS: run=ids
        else:
            P=A//CUS
            if keyname=="est": key=lambda b:(-((MAXIT-done[b])//base + (MAXIT//base if qi[b]==0 else 0)), b)
            elif keyname=="est_rr": key=lambda b:(-((MAXIT-done[b])//base + (MAXIT//base if qi[b]==0 else 0)), served[b], b)
            run=[]
            for c in range(NX):
                cl=sorted([b for b in ids if b%NX==c], key=key)
                run+=cl[:CPX*P]
        t+=round_time(run,left,base)+OVH; rounds+=1
        for b in run:
            it=min(base,left[b]); left[b]-=it; served[b]+=1; done[b]+=it
            if left[b]<=0:
                qi[b]+=1; done[b]=0
                if qi[b]>=len(chains[b]): active.discard(b)
                else: left[b]=chains[b][qi[b]]
    return t,rounds
print("est per class -> %.0f ms, %d rounds"%sim2("est"))
print("est_rr per class -> %.0f ms, %d rounds"%sim2("est_rr"))
