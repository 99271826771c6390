"""First-light check of the QP kernels on a real MI355X (dev script, not a test)."""
import ctypes as C, sys, os, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import osqp_ref as o
from sco_py_amd import _lib as L

lib = C.CDLL(L.lib_path())
for name in ("sco_last_error","sco_qp_default_settings","sco_qp_create","sco_qp_destroy","sco_qp_load","sco_qp_solve","sco_qp_info","sco_qp_last_timing"):
    fn = getattr(lib, name); fn.restype, fn.argtypes = L.ABI[name]
L._lib = lib   # bind the subset for BatchedQP

def penalty_qp(rng, T, d, r, pc=10.0):
    nx = T*d; ns = T*r; n = nx+ns
    Q = np.zeros((n,n))
    for t in range(T-1):
        for j in range(d):
            a, b = t*d+j, (t+1)*d+j
            Q[a,a]+=2; Q[b,b]+=2; Q[a,b]-=2; Q[b,a]-=2
    x0 = rng.standard_normal(nx)*0.3
    rows=[]; lo=[]; hi=[]
    for j in range(d):
        e=np.zeros(n); e[j]=1; rows.append(e); v=x0[j]+0.1*rng.standard_normal(); lo.append(v); hi.append(v)
    for t in range(T):
        for k in range(r):
            e=np.zeros(n); e[t*d:(t+1)*d]=rng.standard_normal(d); e[nx+t*r+k]=-1
            rows.append(e); lo.append(-np.inf); hi.append(rng.standard_normal())
    for j in range(n):
        e=np.zeros(n); e[j]=1; rows.append(e)
        if j<nx: lo.append(x0[j]-1); hi.append(x0[j]+1)
        else: lo.append(0.0); hi.append(np.inf)
    q=np.zeros(n); q[nx:]=pc
    return Q,q,np.array(rows),np.array(lo),np.array(hi)

def run(T,d,r,B,seed,weights=False):
    rng=np.random.default_rng(seed)
    probs=[penalty_qp(rng,T,d,r) for _ in range(B)]
    P0,q0,A0,l0,u0=probs[0]
    n=len(q0); m=len(l0)
    # shared pattern = union over the batch (all structurally identical here)
    Pu=sp.triu(sp.csc_matrix(P0),format='csc'); Pu.sort_indices(); Ac=sp.csc_matrix(A0!=0,dtype=float); Ac=sp.csc_matrix(Ac); Ac.sort_indices()
    Pp,Pi,Ap,Ai=Pu.indptr,Pu.indices,Ac.indptr,Ac.indices
    rows_of=np.asarray(Ai); cols_of=np.repeat(np.arange(n),np.diff(Ap))
    prow=np.asarray(Pi); pcol=np.repeat(np.arange(n),np.diff(Pp))
    Pval=np.stack([p[0][prow,pcol] for p in probs]); Aval=np.stack([p[2][rows_of,cols_of] for p in probs])
    q=np.stack([p[1] for p in probs]); l=np.stack([p[3] for p in probs]); u=np.stack([p[4] for p in probs])
    w=None
    if weights:
        w=np.ones((B,m),dtype=np.int32); w[:,d:d+T*r]=rng.integers(1,5,size=(B,1))
    qp=L.BatchedQP(B,n,m,Pp,Pi,Ap,Ai)
    print("info",qp.info(),"n",n,"m",m)
    qp.load(Pval,q,Aval,l,u,w)
    t=time.time(); x,y,st,it,res=qp.solve(); dt=time.time()-t
    print("solve wall %.3fs"%dt, qp.last_timing())
    worst=0; wy=0; bad=0
    for b in range(min(B,16)):
        ro=o.solve(probs[b][0],probs[b][1],probs[b][2],probs[b][3],probs[b][4],w=None if w is None else w[b])
        dx=np.abs(x[b]-ro.x).max(); dyv=np.abs(y[b]-ro.y).max()
        worst=max(worst,dx); wy=max(wy,dyv)
        if st[b]!=ro.info.status_val or it[b]!=ro.info.iter: bad+=1; print("  mismatch b",b,st[b],ro.info.status_val,it[b],ro.info.iter)
    print("T,d,r,B",T,d,r,B,"max|dx|",worst,"max|dy|",wy,"status/iter mismatches",bad,"iters",it[:8],"status",st[:8])
    qp.close()
    return worst

print("tiny"); run(3,1,1,2,0)
print("small"); run(5,3,4,8,1)
print("weights"); run(5,3,4,8,2,weights=True)
print("7x20"); run(20,7,10,64,3,weights=True)
print("7x20 B=1024"); run(20,7,10,1024,4)
