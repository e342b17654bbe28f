"""Import harness for the *reference* gptools (container-only; never runs on the GPU box).

Used only by ``gen_golden.py`` to produce the committed ``*.npz`` fixtures.  It
(1) restores the numpy aliases that modern scipy no longer re-exports (the
reference calls ``scipy.asarray``, ``scipy.tile`` ... everywhere; ``eigh(eigvals=...)``), (2) builds the
reference's own Cython/C Matern-5/2 extension *out of tree* in a temp dir from
the sources where they lie under /root/reference, and (3) imports
``/root/reference/gptools`` unchanged.  Nothing from the reference is copied
into this repository.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import warnings

REF_ROOT = os.environ.get("GPTOOLS_REFERENCE", "/root/reference")


def _build_matern(tmp):
    kd = os.path.join(REF_ROOT, "gptools", "kernel")
    for rel in ("_matern.pyx", os.path.join("src", "matern.c"), os.path.join("include", "matern.h")):
        shutil.copy(os.path.join(kd, rel), tmp)
    with open(os.path.join(tmp, "setup_ref.py"), "w") as f:
        f.write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "import numpy\n"
            "setup(ext_modules=cythonize([Extension('_matern', ['_matern.pyx', 'matern.c'],"
            " include_dirs=[numpy.get_include(), '.'])], language_level=3))\n"
        )
    subprocess.run(
        [sys.executable, "setup_ref.py", "build_ext", "--inplace"],
        cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
    )


def import_reference():
    """Return the imported reference ``gptools`` module."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import numpy
    import scipy
    import scipy.special, scipy.linalg, scipy.stats, scipy.optimize, scipy.interpolate  # noqa
    warnings.simplefilter("ignore")
    sys.dont_write_bytecode = True
    for name in dir(numpy):
        if not name.startswith("_") and not hasattr(scipy, name):
            try:
                setattr(scipy, name, getattr(numpy, name))
            except Exception:
                pass
    # scipy >= 1.14 dropped eigh's ``eigvals=(lo, hi)`` keyword, which draw_sample(method='eig') passes
    # (gaussian_process.py:1304-1307); same meaning under its new name
    _eigh = scipy.linalg.eigh

    def eigh_compat(a, *args, **kw):
        if "eigvals" in kw:
            kw["subset_by_index"] = kw.pop("eigvals")
        return _eigh(a, *args, **kw)
    scipy.linalg.eigh = eigh_compat
    tmp = tempfile.mkdtemp(prefix="gptools_ref_matern_")
    _build_matern(tmp)
    sys.path.insert(0, tmp)
    import _matern
    sys.modules["gptools.kernel._matern"] = _matern
    sys.path.insert(0, REF_ROOT)
    import matplotlib
    matplotlib.use("Agg")
    import gptools
    return gptools
