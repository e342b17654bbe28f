import ctypes, os
os.environ["GPT_EDGE_STRESS_DEBUG"] = "1"
lib = ctypes.CDLL("/root/repo/gptools_amd/csrc/build/libedge_stress.so")
lib.edge_stress_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
out = (ctypes.c_longlong * 8)()
print("rc", lib.edge_stress_run(0, 10, 1 << 16, 8, out), list(out))
