#!/usr/bin/env python
"""Time loss / grad / FVP passes (tools/kernel_bench.py) for every experiment library build/exp/lib_*.so, each swapped
in as rllab_amd/librllab_amd.so in a child process (GPU box scratch copy only)."""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
target = os.path.join(ROOT, "rllab_amd", "librllab_amd.so")
shutil.copy(target, target + ".orig")
try:
    for lib in [target + ".orig"] + sorted(glob.glob(os.path.join(ROOT, "build", "exp", "lib_*.so"))):
        shutil.copy(lib, target)
        os.utime(target, None)
        print("==", os.path.basename(lib), flush=True)
        subprocess.call([sys.executable, os.path.join(ROOT, "tools", "kernel_bench.py")] + sys.argv[1:])
finally:
    shutil.copy(target + ".orig", target)
