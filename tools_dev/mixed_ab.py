import os, sys, types, subprocess
for lib in sys.argv[1:]:
    env = dict(os.environ, RPCC_HIP_LIB=os.path.abspath(lib))
    code = ("import os,sys,types,torch\nsys.path.insert(0,os.getcwd())\nimport rpcc_amd,bench\n"
            "for rep in range(3):\n r=bench.run_mixed(types.SimpleNamespace(accuracy=0.02),dict(dev=torch.device('cuda:0')),per=85,reps=24,slots=1)\n print('%s',r['value'],r['ms_per_mixed_batch'],r['verified'],flush=True)\n" % lib)
    subprocess.run([sys.executable, "-c", code], env=env)
