#!/usr/bin/env python3
"""Build libyololite_hip.so (gfx950) in-tree with hipcc.  No cmake, no torch extension machinery:
ten translation units (three of them compiled three times: fp32, bf16-MFMA and fp16-MFMA builds), one shared library with a plain C ABI (include/yololite_hip.h).

    python yololite-official-repo_amd/csrc/build.py [--force | --asan]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libyololite_hip.so")
OBJ = os.path.join(HERE, "_obj")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]
UNITS = [   # (source, extra flags, object name)
    ("yl_api.hip", [], "yl_api.o"),
    ("yl_conv.hip", [], "yl_conv.o"),
    ("yl_stemblock.hip", [], "yl_stemblock.o"),
    ("yl_convc.hip", [], "yl_convc.o"),
    ("yl_dpp.hip", [], "yl_dpp.o"),
    ("yl_se.hip", [], "yl_se.o"),
    # element-wise / reduction ops of the hgnetv2 / convnextv2 backbones: one rounding per operation (LayerNorm, GRN)
    ("yl_ops.hip", ["-ffp-contract=off"], "yl_ops.o"),
    ("yl_convc.hip", ["-DYL_BF16=1"], "yl_convc_bf16.o"),
    # bf16-MFMA inference mode: the same two units compiled again under distinct symbol names (yl_dev.h)
    ("yl_conv.hip", ["-DYL_BF16=1"], "yl_conv_bf16.o"),
    ("yl_stemblock.hip", ["-DYL_BF16=1"], "yl_stemblock_bf16.o"),
    # fp16-MFMA inference mode (the reference's fp16 autocast, scripts/helpers/evaluate.py:399,415): a third compilation
    ("yl_convc.hip", ["-DYL_BF16=1", "-DYL_F16=1"], "yl_convc_f16.o"),
    ("yl_conv.hip", ["-DYL_BF16=1", "-DYL_F16=1"], "yl_conv_f16.o"),
    ("yl_stemblock.hip", ["-DYL_BF16=1", "-DYL_F16=1"], "yl_stemblock_f16.o"),
    # fp16-STORAGE mode (option "store_f16", round 6): fp16 operands and fp16 activation tensors in HBM -- a fourth compilation
    ("yl_convc.hip", ["-DYL_BF16=1", "-DYL_F16=1", "-DYL_F16S=1"], "yl_convc_f16s.o"),
    ("yl_conv.hip", ["-DYL_BF16=1", "-DYL_F16=1", "-DYL_F16S=1"], "yl_conv_f16s.o"),
    ("yl_stemblock.hip", ["-DYL_BF16=1", "-DYL_F16=1", "-DYL_F16S=1"], "yl_stemblock_f16s.o"),
    # reference-exact fp32 arithmetic in decode/NMS: no fused multiply-add contraction
    ("yl_post.hip", ["-ffp-contract=off"], "yl_post.o"),
    ("yl_pre.hip", ["-ffp-contract=off"], "yl_pre.o"),
    # evaluation consumers: python-float / numpy-float32 exact IoU arithmetic
    ("yl_eval.hip", ["-ffp-contract=off"], "yl_eval.o"),
    # tracker: float32 scalar arithmetic of the reference's bbox conversions / IoU, op by op
    ("yl_track.hip", ["-ffp-contract=off"], "yl_track.o"),
]
DEPS = ["yl_internal.h", "yl_dev.h", "yl_lp.h", "yl_epi.h", "yl_decode.h", os.path.join("..", "..", "include", "yololite_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_asan(verbose=True):
    """AddressSanitizer build of the HOST side: libyololite_hip_asan.so next to the normal library --
    every translation unit compiled with -fsanitize=address on the host pass only (-fno-gpu-sanitize: the gfx950 code
    objects are the production ones).  Run the host-logic / symbol tests under it with
        LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) \
        ASAN_OPTIONS=detect_leaks=0 YOLOLITE_HIP_LIB=.../libyololite_hip_asan.so python -m pytest tests -m "not gpu"
    (round 3: 11 host tests clean).  Meant for the host logic on a CPU box: with the HIP runtime live under the ASAN
    interceptors the -m gpu suite crawls (4 tests in 20 minutes on the MI355X box) -- do not spend GPU time on it.
    Delete the library afterwards (28 MB; it would travel with every gpurun snapshot)."""
    hipcc = _hipcc()
    obj = os.path.join(HERE, "_obj_asan")
    os.makedirs(obj, exist_ok=True)
    out = os.path.join(os.path.dirname(HERE), "libyololite_hip_asan.so")
    san = ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-fno-omit-frame-pointer", "-g"]
    cmds = [[hipcc] + COMMON + extra + san + ["-c", os.path.join(HERE, src), "-o", os.path.join(obj, oname)]
            for src, extra, oname in UNITS]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, cmds))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan",
         "-o", out] + [os.path.join(obj, oname) for _, _, oname in UNITS])
    return out


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    deps = [os.path.join(HERE, d) for d in DEPS] + [os.path.abspath(__file__)]
    jobs = []
    for src, extra, oname in UNITS:
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ, oname)
        if force or _stale(o, [s] + deps):
            jobs.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, oname) for _, _, oname in UNITS]
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build_asan() if "--asan" in sys.argv else build(force="--force" in sys.argv))
