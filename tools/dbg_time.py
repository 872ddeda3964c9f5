import os, sys, time, ctypes as C, numpy as np
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
def T(fn,n=200):
    for _ in range(10): fn()
    ctx.sync(); t0=time.perf_counter()
    for _ in range(n): fn()
    ctx.sync(); return (time.perf_counter()-t0)/n*1e6
print("eager fwd (3 gemm launches, no sync): %.1f us" % T(lambda: ctx.check(lib.crux_mlp_forward_cached(q.h,dx,B,None))))
def eb():
    ctx.check(lib.crux_mlp_forward_cached(q.h,dx,B,None)); ctx.check(lib.crux_mlp_backward(q.h,dx,B,ddy,1.0,1,None))
print("eager fwd+bwd (8 launches): %.1f us" % T(eb))
print("exec fwd (3 ops, sync+copy): %.1f us" % T(lambda: ctx.check(lib.crux_debug_exec_forward(q.h,dx,B,dout,0,None))))
print("exec fwd+bwd (8 ops): %.1f us" % T(lambda: ctx.check(lib.crux_debug_exec_forward(q.h,dx,B,dout,1,ddy))))
lib.crux_debug_exec_nops.restype=C.c_int32; lib.crux_debug_exec_nops.argtypes=[C.c_void_p,C.c_int32,C.c_int32]
print("exec 1 nop (launch+sync floor): %.1f us" % T(lambda: ctx.check(lib.crux_debug_exec_nops(ctx.h,1,1))))
print("exec 8 nops: %.1f us" % T(lambda: ctx.check(lib.crux_debug_exec_nops(ctx.h,8,1))))
print("exec 8 nops x 64 blocks: %.1f us" % T(lambda: ctx.check(lib.crux_debug_exec_nops(ctx.h,8,64))))
lib.crux_debug_exec_forward_reps.restype=C.c_int32; lib.crux_debug_exec_forward_reps.argtypes=[C.c_void_p,C.c_void_p,C.c_int64,C.c_int32]
for dims in ([8,256,256,4],[256,256],[8,256],[256,4]):
    n=crux.ContinuousNetwork(chain(dims,["relu"]*(len(dims)-2)+["identity"]),seed=1)
    xx=np.asfortranarray(rng.normal(0,1,(dims[0],B)).astype(np.float32)); dxx=ctx.alloc(xx.nbytes); ctx.h2d(dxx,xx)
    t1=T(lambda: ctx.check(lib.crux_debug_exec_forward_reps(n.h,dxx,B,1)),50); t50=T(lambda: ctx.check(lib.crux_debug_exec_forward_reps(n.h,dxx,B,50)),20)
    print(dims, "exec per forward pass: %.2f us (%d gemm ops)" % ((t50-t1)/49, len(dims)-1), " eager: %.2f us" % T(lambda: ctx.check(lib.crux_mlp_forward_cached(n.h,dxx,B,None)),200))
