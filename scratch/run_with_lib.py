"""python scratch/run_with_lib.py <lib.so> <script.py> [args...]: run a scratch script against another library build."""
import sys, os, ctypes, runpy
import torch  # (before any HIP library is loaded: torch brings its own runtime)
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
probe = ctypes.CDLL(_lib.LIB_PATH)
for name in list(_lib.SIGNATURES):
    if not hasattr(probe, name): del _lib.SIGNATURES[name]
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
