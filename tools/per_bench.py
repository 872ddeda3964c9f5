import sys, time, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import crux_jl_amd as crux
ctx=crux.default_context(); rng=np.random.default_rng(0)
N,B=int(os.environ.get("PER_N", 1_000_000)),128
S,A=crux.ContinuousSpace(8),crux.DiscreteSpace(4)
buf=crux.ExperienceBuffer(S,A,N,prioritized=True); D=crux.buffer_like(buf,capacity=B)
chunk=min(100_000, N)
for _ in range(N//chunk):
    a_id=rng.integers(0,4,chunk)
    buf.push_({"s":rng.normal(0,1,(8,chunk)).astype(np.float32),"a":np.eye(4,dtype=bool)[:,a_id],"sp":rng.normal(0,1,(8,chunk)).astype(np.float32),"r":rng.normal(0,1,(1,chunk)).astype(np.float32),"done":rng.random((1,chunk))<0.01,"episode_end":np.zeros((1,chunk),bool)})
buf.update_priorities_(np.arange(1,N+1),(np.abs(rng.normal(0,1,N))+1e-3).astype(np.float32))
de=ctx.alloc(4*B); ctx.h2d(de, (np.abs(rng.normal(0,1,B))+1e-3).astype(np.float32))
k=[0]
def step():
    k[0]+=1; crux.prioritized_sample_(D,buf,i=k[0])
    ctx.check(ctx.lib.crux_per_update_device(buf.h, ctx.lib.crux_buffer_indices_ptr(D.h), de, B))
for _ in range(5): step()
ctx.sync(); t0=time.perf_counter()
for _ in range(300): step()
ctx.sync(); dt=(time.perf_counter()-t0)/300
print("PER sample+update per step: %.1f us"%(dt*1e6))
def t(fn, n=200):
    for _ in range(5): fn()
    ctx.sync(); t0=time.perf_counter()
    for _ in range(n): fn()
    ctx.sync(); return (time.perf_counter()-t0)/n*1e6
def samp():
    k[0]+=1; crux.prioritized_sample_(D,buf,i=k[0])
def upd():
    ctx.check(ctx.lib.crux_per_update_device(buf.h, ctx.lib.crux_buffer_indices_ptr(D.h), de, B))
print("sample only (cumsum valid): %.1f us; update only: %.1f us" % (t(samp), t(upd)))
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(100): step()
ctx.prof_enable(False)
for s in ("per_scan","per_search","gather"):
    ms,n=ctx.prof_get(s); print(s, "%.2f us x %d" % (1e3*ms/max(1,n), n))
