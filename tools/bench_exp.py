"""bench.py against the experiments build (make -C recommendersystems_amd/csrc exp): tuning knobs (RWR_TUNE_ENV names) are read
only by that build.   python tools/bench_exp.py --config C5 --steps 3 ..."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from recommendersystems_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("librwr.so", "librwr_exp.so")
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
