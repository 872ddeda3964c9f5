import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import crux_jl_amd as crux
from crux_jl_amd import _lib as L
ctx=crux.default_context(); rng=np.random.default_rng(0); PER=os.environ.get("PER","1")=="1"
def chain(d,a): return crux.Chain(*[crux.Dense(d[i],d[i+1],a[i]) for i in range(len(a))])
def setup():
    N,B=4000,128
    S,A=crux.ContinuousSpace(8),crux.DiscreteSpace(4)
    r=np.random.default_rng(1)
    buf=crux.ExperienceBuffer(S,A,N,prioritized=PER); D=crux.buffer_like(buf,capacity=B)
    a_id=r.integers(0,4,N)
    buf.push_({"s":r.normal(0,1,(8,N)).astype(np.float32),"a":np.eye(4,dtype=bool)[:,a_id],"sp":r.normal(0,1,(8,N)).astype(np.float32),"r":r.normal(0,1,(1,N)).astype(np.float32),"done":r.random((1,N))<0.01,"episode_end":np.zeros((1,N),bool)})
    if PER: buf.update_priorities_(np.arange(1,N+1),(np.abs(r.normal(0,1,N))+1e-3).astype(np.float32))
    q=crux.DiscreteNetwork(chain([8,256,256,4],["relu","relu","identity"]),[1,2,3,4],seed=1); qm=crux.clone_policy(q); q.attach_optimizer(crux.Adam(np.float32(1e-3)))
    return buf,D,q,qm
def run(fused, K):
    if fused: os.environ.pop("CRUX_NO_FUSED_EPOCH",None)
    else: os.environ["CRUX_NO_FUSED_EPOCH"]="1"
    buf,D,q,qm=setup(); out=[]
    raw=np.zeros(L.INFO_N,np.float32)
    for k in range(K):
        ctx.check(ctx.lib.crux_dqn_epoch(q.h,qm.h,buf.h,D.h,0.99,1 if PER else 0,0.5,k+1,raw.ctypes.data_as(L.vp)))
        out.append((D.indices[:128].copy(), q.get_params(), (buf.priority_params()["priorities"].copy() if PER else np.zeros(1)), raw.copy(), D["s"].copy()))
    return out
a=run(True,6); b=run(False,6)
for k,(x,y) in enumerate(zip(a,b)):
    print(k, "ids equal", np.array_equal(x[0],y[0]), "params maxdiff", np.abs(x[1]-y[1]).max(), "prio maxdiff", np.abs(x[2]-y[2]).max(), "info", x[3][:3], y[3][:3], "batch s equal", np.array_equal(x[4],y[4]))
