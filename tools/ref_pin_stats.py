import sys, os, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import orc
REF="/root/repo/oracle/_ref"
lib=C.CDLL(REF+"/libref_probe.so"); lib.ref_probe_last_error.restype=C.c_char_p
rng=np.random.default_rng(5); n=20000
cases=np.zeros((n,14),np.float32)
cases[:,0:4]=rng.random((n,4)); cases[:,4:7]=(rng.random((n,3))-0.5)*rng.choice([4.0,60.0,3000.0],size=(n,1))
cases[:,7:11]=rng.random((n,4))*np.array([0.05,0.05,0.05,0.5]); cases[:,11:14]=(rng.random((n,3))-0.5)*rng.choice([4.0,60.0,3000.0],size=(n,1))
mask=np.zeros((n,3),np.int32); axis=rng.integers(0,3,n); mask[np.arange(n),axis]=rng.choice([-1,1],n)
mine=np.stack([orc.view_light(c[0:4],c[4:7],c[7:11],c[11:14],m) for c,m in zip(cases,mask)])
f=lambda a:a.ctypes.data_as(C.POINTER(C.c_float)); i=lambda a:a.ctypes.data_as(C.POINTER(C.c_int32))
for name in ("ref_probe_gfx950.co","ref_probe_gfx950_strict.co"):
    out=np.zeros((n,4),np.float32)
    rc=lib.ref_probe_view_light((REF+"/"+name).encode(),f(cases),i(mask),f(out),n); assert rc==0, lib.ref_probe_last_error()
    rel=np.abs(mine-out)/np.maximum(np.abs(out),1e-6)
    ulp=np.abs(mine.view(np.int32).astype(np.int64)-out.view(np.int32).astype(np.int64))
    print(name,"max rel",rel.max(),"p99.9",np.quantile(rel,0.999),"median",np.median(rel),"frac<=1e-5",(rel<=1e-5).mean(),"max ulp",ulp.max(),"bit-identical",(ulp==0).mean())
    w=np.unravel_index(rel.argmax(),rel.shape); print(" worst case",cases[w[0]],mask[w[0]],"ref",out[w[0]],"mine",mine[w[0]])
