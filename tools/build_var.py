"""Development: variants of ONE source of libwc_hip.so -> wc_gan_amd/csrc/build/var/lib_<tag>.so (the other objects are the in-tree build's).
usage: tools/build_var.py wc_split base= STAMPS=-DWC_SPLIT_STAMPS=1 ...      (tag=flags; flags may be empty)"""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wc_gan_amd", "csrc"); OUT = os.path.join(CSRC, "build", "var"); os.makedirs(OUT, exist_ok=True)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
SRC = sys.argv[1]
others = [o for o in glob.glob(os.path.join(CSRC, "build", "*.o")) if os.path.basename(o) != SRC + ".o"]
def one(spec):
    tag, _, fl = spec.partition("=")
    obj = os.path.join(OUT, f"{SRC}_{tag}.o"); lib = os.path.join(OUT, f"lib_{tag}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + fl.split() + ["-I", os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, SRC + ".hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others)
    return lib
with ThreadPoolExecutor(4) as ex:
    for l in ex.map(one, sys.argv[2:]): print(l)
