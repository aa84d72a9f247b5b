#!/bin/bash
# ingest tests + timing of the on-device transposition at config 3 scale
OUT=gpurun_out/${1:-ing}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ingest.py tests/test_top_product.py tests/test_hip_parity.py -m gpu -q --timeout=600 -p no:cacheprovider -x -k "ingest or transpose or predict or golden or fit_transform" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-300
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/timing.txt
import time, torch, sys
sys.path.insert(0, ".")
from rsparse_amd import synth
from rsparse_amd.engine import HipBackend
be = HipBackend(0)
for nu, ni in ((1_000_000, 100_000), (10_000_000, 1_000_000)):
    d = synth.make_dataset(nu, ni, device=be.device)
    p, i, x = d["c_iu"]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pt, it, xt = be.transpose_csc(ni, nu, p, i, x)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = torch.equal(pt, d["c_ui"][0]) and torch.equal(it, d["c_ui"][1]) and torch.equal(xt, d["c_ui"][2])
    nnz = int(i.numel())
    print(dict(users=nu, items=ni, nnz=nnz, ms=round(dt * 1e3, 2), GBps_at_12B_x_2_per_pass=round(nnz * 24e-9 / dt, 1), identical_to_generator=ok))
PY
