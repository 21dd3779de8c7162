import os, sys, runpy
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/geometrics_amd") else ".")
sys.argv = ["time_deform_layer.py"] + sys.argv[1:]
if "--no-tail" in sys.argv:
    from geometrics_amd import deform
    deform._tail_tables = lambda csr: (None, None, None, None)
    sys.argv.remove("--no-tail")
runpy.run_path("tools/time_deform_layer.py", run_name="__main__")
