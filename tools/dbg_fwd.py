import os, sys, ctypes as C, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import crux_jl_amd as crux
from crux_jl_amd import _lib as L
ctx=crux.default_context(); lib=ctx.lib
lib.crux_debug_exec_forward.restype=C.c_int32; lib.crux_debug_exec_forward.argtypes=[C.c_void_p,C.c_void_p,C.c_int64,C.c_void_p,C.c_int32,C.c_void_p]
def chain(d,a): return crux.Chain(*[crux.Dense(d[i],d[i+1],a[i]) for i in range(len(a))])
q=crux.ContinuousNetwork(chain([8,256,256,4],["relu","relu","identity"]),seed=1)
B=128; rng=np.random.default_rng(0)
x=np.asfortranarray(rng.normal(0,1,(8,B)).astype(np.float32)); dy=np.asfortranarray(rng.normal(0,1,(4,B)).astype(np.float32))
dx=ctx.alloc(x.nbytes); ctx.h2d(dx,x); ddy=ctx.alloc(dy.nbytes); ctx.h2d(ddy,dy); dout=ctx.alloc(4*4*B)
g=lib.crux_mlp_grads_ptr(q.h)
def grads():
    a=np.empty(q.n_params,np.float32); ctx.d2h(C.c_void_p(g),a); return a
ctx.check(lib.crux_mlp_forward_cached(q.h,dx,B,dout)); y0=ctx.d2h(dout,np.empty((4,B),np.float32,order="F")).copy()
ctx.check(lib.crux_mlp_backward(q.h,dx,B,ddy,1.0,1,None)); ctx.sync(); g0=grads()
ctx.check(lib.crux_debug_exec_forward(q.h,dx,B,dout,1,ddy)); y1=ctx.d2h(dout,np.empty((4,B),np.float32,order="F")).copy(); g1=grads()
print("forward equal:", np.array_equal(y0,y1), np.abs(y0-y1).max(), "grads equal:", np.array_equal(g0,g1), np.abs(g0-g1).max())
d=q.network.dims; off=0
for l in range(3):
    n=d[l+1]*d[l]; print("layer",l,"W maxdiff",np.abs(g0[off:off+n]-g1[off:off+n]).max(),"b maxdiff",np.abs(g0[off+n:off+n+d[l+1]]-g1[off+n:off+n+d[l+1]]).max()); off+=n+d[l+1]
