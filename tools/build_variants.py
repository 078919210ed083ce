"""Development: builds libwc_hip variants with -DWC_ABL=<bits> (wc_fast.hip only) into wc_gan_amd/csrc/build/abl/."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wc_gan_amd", "csrc"); OUT = os.path.join(CSRC, "build", "abl"); os.makedirs(OUT, exist_ok=True)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
ALL = ("wc_rows", "wc_fast", "wc_fast_xty", "wc_small", "wc_sn", "wc_abi")
SRC = os.environ.get("WC_VARIANT_SRC", "wc_fast")
others = [os.path.join(CSRC, "build", f + ".o") for f in ALL if f != SRC]
def one(v):
    tag = v.replace("=", "_").replace("-D", "").replace(" ", "_") or "base"
    obj = os.path.join(OUT, f"wc_fast_{tag}.o"); lib = os.path.join(OUT, f"lib_{tag}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + v.split() + ["-c", os.path.join(CSRC, SRC + ".hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others)
    return lib
with ThreadPoolExecutor(4) as ex:
    for l in ex.map(one, sys.argv[1:]): print(l)
