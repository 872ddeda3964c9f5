import os, sys, time, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import crux_jl_amd as crux
ctx=crux.default_context(); lib=ctx.lib
lib.crux_debug_exec_nops.restype=C.c_int32; lib.crux_debug_exec_nops.argtypes=[C.c_void_p,C.c_int32,C.c_int32]
for n in (1, 100, 1000):
    for _ in range(3): ctx.check(lib.crux_debug_exec_nops(ctx.h,n,1))
    t0=time.perf_counter()
    for _ in range(10): ctx.check(lib.crux_debug_exec_nops(ctx.h,n,1))
    dt=(time.perf_counter()-t0)/10
    print("n=%d ops: %.1f us per launch, %.2f us per op" % (n, dt*1e6, dt*1e6/n))
