"""GPU box: configs[4] on one GPU (bench.run_mixed: one fused call per mixed batch, 4 in flight) for several library builds in one session.
usage: python tools_dev/mixed_libs.py libA.so libB.so ...   (each library runs in a child process: RPCC_HIP_LIB is read at import)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = """
import sys, types, torch
sys.path.insert(0, %r)
import rpcc_amd, bench
a = types.SimpleNamespace(accuracy=0.02)
for slots in (1, 4):
    for rep in range(2):
        r = bench.run_mixed(a, dict(dev=torch.device("cuda:0")), per=85, reps=24, slots=slots)
        print("%%-40s slots %%d  %%8.0f frames/s  %%.3f ms per mixed batch  verified %%s" %% (sys.argv[1], slots, r["value"], r["ms_per_mixed_batch"], r["verified"]), flush=True)
""" % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, lib], env=dict(os.environ, RPCC_HIP_LIB=os.path.join(ROOT, lib)), check=False)
