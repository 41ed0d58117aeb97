import sys, ctypes as C, numpy as np
sys.path[:0]=["zorakaudio-experimental-plugins_amd","."]
import zabatch
class HS(C.Structure):
    _fields_=[("spl",C.c_void_p),("sliders",C.c_void_p),("vars",C.c_void_p),("mem",C.c_void_p),("mem_n",C.c_int64),("pm",C.c_void_p),("mt",C.c_void_p),("mti",C.c_void_p),("vm",C.c_void_p),("vi",C.c_void_p)]
for variant in ("numpy","ctypes"):
    e = zabatch.Engine("DPT", 1, mem_cap=65536, max_block=8192)
    L = e.L
    L.zab_state_download.argtypes=[C.c_void_p, C.c_int32, C.POINTER(HS)]
    L.zab_state_upload.argtypes=[C.c_void_p, C.c_int32, C.POINTER(HS)]
    L.zab_run_section.argtypes=[C.c_void_p, C.c_int32, C.c_int32]
    bufs = {"spl":np.zeros(64),"sliders":np.array(zabatch.leaf_meta("DPT")["default_sliders"]),"vars":np.zeros(e.nvars),"pm":np.zeros(3,np.int64),"mt":np.zeros(624,np.uint32),"mti":np.zeros(1,np.uint32),"vm":np.zeros(1,np.int64),"vi":np.zeros(1,np.int32)}
    h=HS()
    for k,v in bufs.items(): setattr(h,k,v.ctypes.data)
    if variant=="numpy":
        m=np.zeros(65536); h.mem=m.ctypes.data
    else:
        m=(C.c_double*65536)(); h.mem=C.addressof(m)
    h.mem_n=65536
    print(variant, "upload", L.zab_state_upload(e.h,0,C.byref(h)), L.zab_last_error())
    print(variant, "init", L.zab_run_section(e.h,0,0), L.zab_last_error())
    print(variant, "download", L.zab_state_download(e.h,0,C.byref(h)), L.zab_last_error())
    print(variant, "slider", L.zab_run_section(e.h,1,0), L.zab_last_error())
    print(variant, "download", L.zab_state_download(e.h,0,C.byref(h)), L.zab_last_error())
    e.close()
print("---- process sequence")
e = zabatch.Engine("DPT", 1, mem_cap=65536, max_block=8192)
L = e.L
e.set_sliders(zabatch.leaf_meta("DPT")["default_sliders"]); e.prepare()
m=np.zeros(65536); h=HS(); h.mem=m.ctypes.data; h.mem_n=65536
for n in (512,512,176):
    y = e.process_host(np.zeros((1,2,n),np.float32), block=n)
    print(n, "download", L.zab_state_download(e.h,0,C.byref(h)), L.zab_last_error())
    print(n, "upload", L.zab_state_upload(e.h,0,C.byref(h)), L.zab_last_error())
