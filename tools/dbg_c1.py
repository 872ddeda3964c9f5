import os, sys, time, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import crux_jl_amd as crux
ctx=crux.default_context()
def chain(d,a): return crux.Chain(*[crux.Dense(d[i],d[i+1],a[i]) for i in range(len(a))])
N=int(os.environ.get("N","20000"))
q=crux.DiscreteNetwork(chain([2,8,4],["relu","identity"]),[1,2,3,4],seed=1)
sv=crux.DQN(q,crux.ContinuousSpace(2),N=N,dN=4,max_steps=100,c_opt={"batch_size":128})
ctx.prof_reset(); ctx.prof_enable(True)
t0=time.perf_counter(); crux.solve(sv,crux.SimpleGridWorld(n_envs=1,seed=0)); ctx.sync(); t=time.perf_counter()-t0
ctx.prof_enable(False)
ms,n=ctx.prof_get("td_step")
print("N=%d wall %.3f s; solve kernel %.3f s in %d launches -> %.2f us per gradient step" % (N,t,ms/1e3,n,ms*1e3/(N-200)))
