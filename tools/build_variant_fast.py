"""Tuning helper: rebuild only the named sources with extra -D flags and link them with the objects of the regular build
into libcstark_hip_<name>.so.  usage: build_variant_fast.py name src1.hip[,src2.hip] D1 D2 ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
B = importlib.import_module("certificate_stark_amd.build")
name, srcs, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
objs = []
for f in sorted(os.listdir(B.CSRC)):
    if not f.endswith(".hip"):
        continue
    o = os.path.join(B.OBJ, f[:-4] + ".o")
    if f in srcs:
        o = os.path.join(B.OBJ, f[:-4] + "_" + name + ".o")
        subprocess.check_call([B.hipcc()] + B.FLAGS + ["-D" + d for d in defs] + ["-c", os.path.join(B.CSRC, f), "-o", o])
    objs.append(o)
out = os.path.join(B.HERE, "libcstark_hip_%s.so" % name)
subprocess.check_call([B.hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-o", out] + objs)
print(out)
