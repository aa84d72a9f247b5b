"""Build librsparse_wrmf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

In-tree output (rsparse_amd/lib/) so the .so travels with the repo snapshot to the GPU box.
"""
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
SRC = [PKG / "csrc" / "wrmf_kernels.hip", PKG / "csrc" / "wrmf_cgq.hip", PKG / "csrc" / "wrmf_ne.hip", PKG / "csrc" / "wrmf_chol.hip", PKG / "csrc" / "wrmf_chol_lr.hip", PKG / "csrc" / "wrmf_topk.hip", PKG / "csrc" / "wrmf_ingest.hip", PKG / "csrc" / "wrmf_nnls.hip", PKG / "csrc" / "wrmf_bias.hip",
       PKG / "csrc" / "wrmf_capi.cpp"]
DEPS = SRC + [PKG / "csrc" / "wrmf_internal.h", PKG / "csrc" / "wrmf_device.h",
              PKG.parent / "include" / "rsparse_wrmf_hip.h"]
OUT = PKG / "lib" / "librsparse_wrmf_hip.so"


def build(force=False, verbose=False):
    OUT.parent.mkdir(exist_ok=True)
    if not force and OUT.exists() and all(OUT.stat().st_mtime >= d.stat().st_mtime for d in DEPS):
        return OUT
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           *map(str, SRC), "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
