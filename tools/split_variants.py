"""Development: libwc_hip variants that differ in the flags of ONE source (default wc_split.hip) -> wc_gan_amd/csrc/build/var/lib_<tag>.so
usage: python tools/split_variants.py [--src wc_split] "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the defaults)"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wc_gan_amd import build as B
args = sys.argv[1:]
SRC = "wc_split"
if args and args[0] == "--src":
    SRC = args[1]; args = args[2:]
B.build(verbose=False)
CSRC = B.CSRC; OUT = os.path.join(CSRC, "build", "var"); os.makedirs(OUT, exist_ok=True)
others = [os.path.join(CSRC, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != SRC + ".hip"]
def one(v):
    tag = v.replace("=", "_").replace("-D", "").replace(" ", "_") or "base"
    obj = os.path.join(OUT, f"{SRC}_{tag}.o"); lib = os.path.join(OUT, f"lib_{tag}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + [f for f in B.FLAGS if f] + v.split() + ["-c", os.path.join(CSRC, SRC + ".hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others)
    return lib
with ThreadPoolExecutor(4) as ex:
    for l in ex.map(one, args): print(l)
