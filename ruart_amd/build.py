"""Build libruart_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m ruart_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(HERE, "..", "include")
LIB = os.path.join(HERE, "libruart_hip.so")
SOURCES = ["gemm.hip", "gemm_corr.hip", "gemm_tn.hip", "bert_kernels.hip", "bert_train_kernels.hip", "bert_train_attn.hip", "bert_forward.hip", "sdnet_attention.hip", "sdnet_lstm.hip", "sdnet_gemm.hip", "sdnet_optim.hip", "sdnet_scorer.hip", "phoc.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", INC, "-I", CSRC, "-Wno-unused-result", "-Wno-pass-failed"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_shared.h"), os.path.join(INC, "ruart_hip.h")]
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = [s[:-4] + ".o" for s in srcs]

    def cc(pair):
        src, obj = pair
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return True
        return False

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        changed = list(ex.map(cc, zip(srcs, objs)))
    if force or any(changed) or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
