import sys, runpy
sys.path.insert(0, "/root/repo")
from tulip_amd.engine import TulipEngine
TulipEngine.embed_fold_on_chain = (sys.argv[1] == "1")
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--no-secondary", "--no-reference-loop", "--steps", "100", "--warmup", "20"]
runpy.run_path("/root/repo/bench.py", run_name="__main__")
