#!/bin/bash
# K-build time of several library builds: kb_ab.sh <workload> lib...
wl=$1; shift
for l in "$@"; do
python3 - "$wl" "$l" <<'P'
import sys, os, ctypes
sys.path.insert(0, '/root/repo')
from gptools_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[2])
probe = ctypes.CDLL(_lib.LIB_PATH)
for name in list(_lib.SIGNATURES):
    if not hasattr(probe, name): del _lib.SIGNATURES[name]
lib = sys.argv[2]
sys.argv = ['kb_time.py', sys.argv[1], '12']
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf): runpy.run_path('/root/repo/scratch/kb_time.py', run_name='__main__')
print("%-28s %s" % (lib, buf.getvalue().strip()))
P
done
