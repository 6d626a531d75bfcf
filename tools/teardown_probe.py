#!/usr/bin/env python3
"""tools/teardown_probe.py -- GPU-box probe: how long does the OS take to tear down a process that holds X GB of
device memory (and Y GB of page-locked host memory)?  Parent-observed time minus the child's own run time."""
import subprocess, sys, time
child = r'''
import ctypes as C, os, sys, time
t0=time.time()
hip=C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes=[C.POINTER(C.c_void_p),C.c_size_t]; hip.hipMemset.argtypes=[C.c_void_p,C.c_int,C.c_size_t]
hip.hipHostMalloc.argtypes=[C.POINTER(C.c_void_p),C.c_size_t,C.c_uint]
gb=float(sys.argv[1]); touch=int(sys.argv[2]); pinned=float(sys.argv[3])
ps=[]
n=int(gb*4) 
for i in range(n):
    p=C.c_void_p(); assert hip.hipMalloc(C.byref(p), 256<<20)==0; ps.append(p)
    if touch: hip.hipMemset(p,1,256<<20)
if pinned>0:
    q=C.c_void_p(); assert hip.hipHostMalloc(C.byref(q), int(pinned*(1<<30)), 0)==0
hip.hipDeviceSynchronize()
sys.stderr.write("child_s=%.3f\n"%(time.time()-t0)); sys.stderr.flush()
os._exit(0)
'''
for gb, touch, pinned in ((0, 0, 0), (8, 0, 0), (24, 0, 0), (24, 1, 0), (48, 1, 0), (0, 0, 0.25), (0, 0, 1.0)):
    t = time.time()
    pr = subprocess.run([sys.executable, "-c", child, str(gb), str(touch), str(pinned)], capture_output=True, text=True)
    wall = time.time() - t
    cs = float(pr.stderr.strip().split("child_s=")[-1]) if "child_s=" in pr.stderr else float("nan")
    print(f"device {gb:4} GB touched={touch} pinned {pinned} GB: parent wall {wall:.3f} s, child body {cs:.3f} s, outside {wall - cs:.3f} s", flush=True)
