"""A few batches of one bench workload, for ad-hoc profiling (profiles/pmc_pass.sh <tag> profiles/run_workload.py "<counters>"):
WORKLOAD=crt-pi (bench.py --workload names), STEPS=2."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py", "--workload", os.environ.get("WORKLOAD", "crt-pi"), "--steps", os.environ.get("STEPS", "2"), "--warmup", "1", "--no-cpu-baseline"]
sys.path.insert(0, R)
import runpy
runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
