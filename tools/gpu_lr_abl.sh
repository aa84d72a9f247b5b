#!/bin/bash
# timing-only ablations of the low-rank Cholesky kernel (RSP_LR_ABL builds): user half of config 4 through one WRMF half-iteration
# usage: tools/gpu_lr_abl.sh "" _lr4 ...   ("" = the product library; _lrN = rsparse_amd/lib/librsparse_wrmf_hip_lrN.so built with
# -DRSP_LR_ABL=N, bits as listed in wrmf_chol_lr.hip)
for n in "$@"; do
  lib=$PWD/rsparse_amd/lib/librsparse_wrmf_hip$n.so
  RSPARSE_HIP_LIB=$lib timeout 300 python - <<PY
import torch, time
from rsparse_amd import synth
from rsparse_amd.engine import HipBackend, ShardedALS
be = HipBackend()
d = synth.make_dataset(10_000_000, 1_000_000, device=be.device, feedback="implicit")
als = ShardedALS(be, 10_000_000, 1_000_000, 128, d["c_ui"], d["c_iu"], d["nnz"], feedback="implicit", lambda_=0.1)
g = torch.Generator(device=be.device).manual_seed(11)
U = torch.randn(10_000_000, 128, generator=g, device=be.device) * 0.01
V = torch.randn(1_000_000, 128, generator=g, device=be.device) * 0.01
G = als.gramian(V, als.lay_item)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    als.half_iteration("users", U, V, 0, G=G, want_loss=False)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
print("lib [$n] users half %.1f ms" % (1e3 * t))
be.profile(True)
als.half_iteration("users", U, V, 0, G=G, want_loss=False)
torch.cuda.synchronize()
ms = be.profile_last(); names = be.profile_last_names()
print("lib [$n] per launch:", "; ".join("%s %.1f" % (nm[:44], t) for nm, t in zip(names, ms) if t > 0.05))
be.profile(False)
if hasattr(be.lib, "rsparse_hip_dev_lrw_prof"):   # -DRSP_LRW_PROF builds: ticks per phase and class of the wave-per-pass kernel
    import ctypes
    buf = (ctypes.c_ulonglong * 32)()
    be.lib.rsparse_hip_dev_lrw_prof(buf, 1)
    als.half_iteration("users", U, V, 0, G=G, want_loss=False)
    torch.cuda.synchronize()
    be.lib.rsparse_hip_dev_lrw_prof(buf, 1)
    for c, nm in enumerate(("49..64", "33..48", "17..32", "<=16")):
        v = list(buf[8 * c: 8 * c + 8]); n = max(v[7], 1)
        print("lrw prof %-7s passes %8d  ticks/pass: issue %.0f  gather-wait %.0f  split+V' %.0f  terms+T %.0f  solve %.0f  P+y %.0f  rest %.0f  | total %.0f" % (nm, v[7], *[x / n for x in v[:7]], sum(v[:7]) / n))
PY
done
