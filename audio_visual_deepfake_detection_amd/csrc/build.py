"""Builds libavdf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python audio_visual_deepfake_detection_amd/csrc/build.py [--force]
(run it by path: importing the package needs an up-to-date .so)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libavdf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
SOURCES = ["gemm_conv.hip", "norm.hip", "attn.hip", "pointwise.hip", "smallconv.hip", "loss.hip",
           "decode_nms.hip", "interp.hip", "optim.hip", "audio.hip", "emotion.hip", "video.hip", "chain.hip", "glue.hip", "fpn.hip", "api.cpp"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on"]


# AVDF_PW_EXPERIMENTAL=1: also build the wave-specialised pointwise forward core of tools/experimental/pw_gemm/ (round-3
# experiment, +0.25 % in situ: not part of the product library or of include/avdf.h) and its plan codes >= 8
PW_EXP = os.environ.get("AVDF_PW_EXPERIMENTAL") == "1"
PW_DIR = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools", "experimental", "pw_gemm")


def _stale(obj, deps):
    return (not os.path.exists(obj)) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def build(force=False, verbose=False):
    obj_dir = os.path.join(HERE, "build_pwexp" if PW_EXP else "build")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(HERE, "common.h"), os.path.join(HERE, "gemm_tuned.inc"),
            os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "avdf.h")]
    sources = [os.path.join(HERE, s) for s in SOURCES]
    flags = list(FLAGS)
    if PW_EXP:
        sources.append(os.path.join(PW_DIR, "pw_gemm.hip"))
        hdrs.append(os.path.join(PW_DIR, "pw_gemm.h"))
        flags += ["-DAVDF_PW_EXPERIMENTAL", "-I", PW_DIR, "-I", HERE]
    jobs = []
    for src in sources:
        s = os.path.basename(src)
        obj = os.path.join(obj_dir, s.rsplit(".", 1)[0] + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + flags + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, os.path.basename(s).rsplit(".", 1)[0] + ".o") for s in sources]
    if force or jobs or _stale(OUT, objs):
        run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
