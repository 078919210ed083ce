"""Development: bench.py on another build of the library (tools/build_var.py).  usage: bench_with_lib.py <lib.so> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wc_gan_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
bench.main(sys.argv[2:])
