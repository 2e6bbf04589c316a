#!/usr/bin/env python
"""Run the same commands against the production library and every experiment build build/exp/lib_*.so (the library file
is swapped in place and restored).  GPU box only.   tools/exp/lib_ab.py '<shell command>' ['<shell command>' ...]"""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
target = os.path.join(ROOT, "rllab_amd", "librllab_amd.so")
shutil.copy2(target, target + ".orig")
try:
    for lib in [target + ".orig"] + sorted(glob.glob(os.path.join(ROOT, "build", "exp", "lib_*.so"))) + [target + ".orig"]:
        shutil.copy2(lib, target)
        os.utime(target, None)          # newer than every source: build() leaves it alone
        for cmd in sys.argv[1:]:
            print("==", os.path.basename(lib), "::", cmd, flush=True)
            subprocess.call(cmd, shell=True, cwd=ROOT)
finally:
    shutil.copy2(target + ".orig", target)
    os.utime(target, None)
    os.remove(target + ".orig")
