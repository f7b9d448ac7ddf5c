"""TEST INFRASTRUCTURE ONLY -- builds the oracle's native pieces.

* build_oracle_c(): gcc-compiles oracle/nms_oracle.c (our C restatement of the 1-D NMS)
  into oracle/_build/libnms_oracle.so.
* build_ref_nms(): when /root/reference exists (build container only), compiles the
  reference's OWN libs/utils/csrc/nms_cpu.cpp *where it lies* into oracle/_ref/
  (git-ignored, travels to the GPU box as a prebuilt .so).  No reference source is copied.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/libs/utils/csrc/nms_cpu.cpp"


def build_oracle_c():
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(HERE, "nms_oracle.c")
    out = os.path.join(out_dir, "libnms_oracle.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", out, src, "-lm"])
    return out


def build_ref_nms():
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    if glob.glob(os.path.join(out_dir, "nms_1d_cpu*.so")):
        return out_dir
    if not os.path.exists(REF_SRC):
        return None
    from torch.utils.cpp_extension import load
    load(name="nms_1d_cpu", sources=[REF_SRC], build_directory=out_dir, verbose=False)
    # keep only the shared object (ninja files / objects are build debris)
    for f in os.listdir(out_dir):
        if not f.endswith(".so"):
            p = os.path.join(out_dir, f)
            if os.path.isfile(p):
                os.remove(p)
    return out_dir


if __name__ == "__main__":
    print(build_oracle_c())
    print(build_ref_nms())
