import sys, os, subprocess
if len(sys.argv) == 1:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for lib in ["", "super_resolution_amd/variants/lib_NOPREFETCH.so", "super_resolution_amd/variants/lib_NOBIASTAP.so"]:
        env = dict(os.environ)
        if lib:
            env["HAT_MI355X_LIB"] = os.path.join(root, lib)
        print("=====", lib or "default", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag_ffn.py")).read().split('print("---- single taps')[0])
