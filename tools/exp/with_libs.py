#!/usr/bin/env python
"""Run a script once per experiment library build/exp/lib_*.so, each swapped in as rllab_amd/librllab_amd.so in a child
process (GPU box scratch copy only): python tools/exp/with_libs.py tools/exp/csplit_time.py <args>"""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
target = os.path.join(ROOT, "rllab_amd", "librllab_amd.so")
shutil.copy(target, target + ".orig")
try:
    for lib in sorted(glob.glob(os.path.join(ROOT, "build", "exp", "lib_*.so"))):
        shutil.copy(lib, target)
        os.utime(target, None)
        print("==", os.path.basename(lib), flush=True)
        subprocess.call([sys.executable] + sys.argv[1:], cwd=ROOT)
finally:
    shutil.copy(target + ".orig", target)
    os.utime(target, None)
