#!/usr/bin/env python3
"""Build the REAL reference (rlangefe/pygemma) for use as the parity oracle — in this container only.

TEST INFRASTRUCTURE. Nothing in the product imports this.

The reference is Python + one Cython module.  It is compiled *from where it lies* under
/root/reference (setup.py:11-16 recipe: cythonize pygemma_model/pygemma_model.pyx, -O3,
numpy include dir) into a scratch directory OUTSIDE the repository (default
/tmp/pygemma_ref).  No reference source, generated C or bytecode is ever written into
/root/repo: the only things that enter the repo are the small golden vectors that
tests/golden/make_golden.py emits by *calling* the built reference.

Layout produced (importable with PYTHONPATH=/tmp/pygemma_ref):
    /tmp/pygemma_ref/pygemma/__init__.py          (empty, ours)
    /tmp/pygemma_ref/pygemma/pygemma_model*.so    (built from the reference .pyx)
    /tmp/pygemma_ref/pygemma/lmm.py -> /root/reference/lmm/lmm.py   (symlink, not a copy)

Usage:  python oracle/build_ref.py [--dest /tmp/pygemma_ref]
"""
import argparse
import os
import sys

REF = "/root/reference"


def build(dest: str) -> str:
    import numpy
    from setuptools import Extension
    from setuptools.dist import Distribution
    from Cython.Build import cythonize

    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not present: the reference oracle can only be built in the build container")
    pkg = os.path.join(dest, "pygemma")
    os.makedirs(pkg, exist_ok=True)
    open(os.path.join(pkg, "__init__.py"), "a").close()
    link = os.path.join(pkg, "lmm.py")
    if not os.path.islink(link):
        os.symlink(os.path.join(REF, "lmm", "lmm.py"), link)

    ext = Extension(
        "pygemma.pygemma_model",
        [os.path.join(REF, "pygemma_model", "pygemma_model.pyx")],
        include_dirs=[numpy.get_include()],
        extra_compile_args=["-O3", "-w"],
    )
    build_tmp = os.path.join(dest, "_build")
    exts = cythonize([ext], build_dir=build_tmp, quiet=True,
                     compiler_directives={"language_level": "3"})
    dist = Distribution({"ext_modules": exts})
    cmd = dist.get_command_obj("build_ext")
    cmd.build_lib = dest
    cmd.build_temp = build_tmp
    cmd.ensure_finalized()
    cmd.run()
    return dest


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dest", default="/tmp/pygemma_ref")
    a = ap.parse_args()
    d = build(a.dest)
    sys.path.insert(0, d)
    from pygemma import lmm  # noqa: E402  (prints the reference's import banner)
    print("reference oracle importable from", d, "->", lmm.__file__)
