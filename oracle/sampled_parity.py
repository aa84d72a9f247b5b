"""ORACLE -- TEST INFRASTRUCTURE ONLY (see wrmf_oracle.cpp header; PARITY UNPINNED).

Full-size sampled parity: after one half-iteration of the device path at BASELINE.json's full sizes, a
sample of the solved rows is re-solved by the fp64 oracle from the SAME inputs (the row's non-zeros, the
fixed side's factor vectors it touches, its warm start, the Gramian) and compared row by row.

Why sampled: the oracle needs ~1 s per 10^5 non-zeros at rank 128; the full 5e8-nnz half-iteration would
take hours, while the paths that only exist at full size (large per-team row quotas, the 100+-chunk rows, the
350k-nnz item, int64 scratch offsets) are all reached by picking rows per launch bucket.

Row choice per launch bucket (the length classes of wrmf_cgq.hip's launch table): the longest and the shortest
row of the class, rows at both class edges, the first and the last row of the class in schedule order, and a
seeded random fill up to `per_bucket`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch

from . import wrmf_oracle as O

# (lo, hi] length classes = launch buckets of the CG path; the exact solvers use one kernel for every length but
# the same classes keep the sample spread over the length distribution
LENGTH_CLASSES = ((0, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 1 << 30))


def pick_rows(col_ptrs, per_bucket=64, seed=0):
    """col_ptrs: int32 tensor (device or host).  Returns a sorted int64 tensor of row ids (on col_ptrs' device)."""
    lens = torch.diff(col_ptrs.to(torch.int64))
    g = torch.Generator(device="cpu").manual_seed(seed)
    picked = []
    for lo, hi in LENGTH_CLASSES:
        cand = torch.nonzero((lens > lo) & (lens <= hi)).flatten()
        if cand.numel() == 0:
            continue
        cl = lens[cand]
        fixed = [cand[torch.argmax(cl)], cand[torch.argmin(cl)], cand[0], cand[-1]]
        for edge in (lo + 1, hi):
            e = cand[cl == edge]
            if e.numel():
                fixed += [e[0], e[-1]]
        fixed = torch.stack(fixed)
        n_rand = max(0, per_bucket - fixed.numel())
        r = cand[torch.randint(0, cand.numel(), (n_rand,), generator=g).to(cand.device)] if n_rand else cand[:0]
        picked.append(torch.cat([fixed, r]))
    if not picked:
        return torch.zeros(0, dtype=torch.int64, device=col_ptrs.device)
    return torch.unique(torch.cat(picked))


def gramian64(F, lam):
    """fp64 Gramian of the fixed side + fl(lambda) I (R/model_WRMF.R:474-486), on F's device, in row blocks."""
    n, k = F.shape
    G = torch.zeros((k, k), dtype=torch.float64, device=F.device)
    step = 1 << 20
    for r0 in range(0, n, step):
        b = F[r0:r0 + step].double()
        G += b.T @ b
    G += torch.eye(k, dtype=torch.float64, device=F.device) * float(np.float32(lam))
    return G


def check(csc, F, S_before, S_after, rows, lam, solver, cg_steps, implicit, dynamic_lambda=True, G64=None,
          n_threads=32, yardstick=False):
    """csc = (p int32, i int32, x f32) tensors of the solved side (columns = solved rows); F: (n_fixed, k) fixed-side
    factors; S_before / S_after: (len(rows), k) warm starts / device results of the sampled rows (any device).
    Returns dict(rows_checked, max_row_err, fro_err, worst_row, worst_len, per_class).  yardstick: also solve the sample
    with the oracle in float (the reference's precision = "float" arithmetic) and report ITS distance from the fp64
    solution as max_row_err_f32_oracle -- what fp32 arithmetic costs on these very systems."""
    p, i, x = csc
    rows = rows.to(p.device)
    p64 = p.to(torch.int64)
    starts, ends = p64[rows], p64[rows + 1]
    lens = ends - starts
    tot = int(lens.sum())
    # positions of the sampled rows' non-zeros
    off = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=p.device)
    torch.cumsum(lens, 0, out=off[1:])
    pos = torch.arange(tot, dtype=torch.int64, device=p.device)
    rid = torch.repeat_interleave(torch.arange(rows.numel(), device=p.device), lens)
    pos = pos - off[rid] + starts[rid]
    idx = i[pos].to(torch.int64)
    val = x[pos].double().cpu().numpy()
    uniq, inv = torch.unique(idx, return_inverse=True)
    Xs = np.asfortranarray(F[uniq].double().cpu().numpy().T)          # k x n_uniq, fp64 image of the fp32 factors
    k = Xs.shape[0]
    ps = off.cpu().numpy().astype(np.int32)
    isub = inv.cpu().numpy().astype(np.int32)
    Y = np.asfortranarray(S_before.double().cpu().numpy().T).copy(order="F")
    if implicit:
        G = np.asfortranarray((G64 if G64 is not None else gramian64(F, lam)).cpu().numpy())
        O.als_implicit(ps, isub, val, Xs, Y, G, lam, solver, cg_steps, n_threads=n_threads)
    else:
        cnt = np.zeros(Xs.shape[1])
        O.als_explicit(ps, isub, val, Xs, Y, cnt, lam, solver, cg_steps, dynamic_lambda, n_threads=n_threads)
    err32 = None
    if yardstick:
        Y32 = np.asfortranarray(S_before.float().cpu().numpy().T).copy(order="F")
        Xs32 = np.asfortranarray(Xs, dtype=np.float32)
        if implicit:
            O.als_implicit(ps, isub, val, Xs32, Y32, np.asfortranarray(G, dtype=np.float32), lam, solver, cg_steps,
                           n_threads=n_threads)
        else:
            O.als_explicit(ps, isub, val, Xs32, Y32, np.zeros(Xs.shape[1], dtype=np.float32), lam, solver, cg_steps,
                           dynamic_lambda, n_threads=n_threads)
        err32 = np.linalg.norm(Y32.astype(np.float64) - Y, axis=0) / np.maximum(np.linalg.norm(Y, axis=0), 1e-300)
    got = S_after.double().cpu().numpy().T
    den = np.maximum(np.linalg.norm(Y, axis=0), 1e-300)
    err = np.linalg.norm(got - Y, axis=0) / den
    lens_h = lens.cpu().numpy()
    w = int(np.argmax(err)) if err.size else 0
    per_class = {}
    for lo, hi in LENGTH_CLASSES:
        m = (lens_h > lo) & (lens_h <= hi)
        if m.any():
            per_class["%d-%s" % (lo + 1, hi if hi < (1 << 30) else "max")] = {
                "rows": int(m.sum()), "max_row_err": float(err[m].max()),
                "fro_err": float(np.linalg.norm(got[:, m] - Y[:, m]) / max(np.linalg.norm(Y[:, m]), 1e-300)),
                "longest": int(lens_h[m].max())}
    extra = {} if err32 is None else {"max_row_err_f32_oracle": float(err32.max()) if err32.size else 0.0}
    return {**extra, "rows_checked": int(rows.numel()), "nnz_checked": tot,
            "max_row_err": float(err.max()) if err.size else 0.0,
            "fro_err": float(np.linalg.norm(got - Y) / max(np.linalg.norm(Y), 1e-300)),
            "worst_row": int(rows[w]) if err.size else -1, "worst_len": int(lens_h[w]) if err.size else 0,
            "per_class": per_class}


def half_iteration_with_check(als, side, U, V, solver, per_bucket=64, seed=0, n_threads=32, yardstick=False):
    """Run one half-iteration of a single-rank rsparse_amd.engine.ShardedALS and check a sample of its rows.
    Returns (loss, report)."""
    if side == "items":
        F, nF, S, nS, csc = U, als.n_user, V, als.n_item, als.csc_items
    else:
        F, nF, S, nS, csc = V, als.n_item, U, als.n_user, als.csc_users
    p, i, x = csc.keep
    rows = pick_rows(p, per_bucket, seed)
    before = S[rows].clone()
    assert als.ws == 1, "sampled parity runs on the single-rank layout (storage order == global order)"
    G64 = gramian64(F[:nF], als.lambda_) if als.implicit else None
    loss = als.half_iteration(side, U, V, solver)
    rep = check((p, i, x), F[:nF], before, S[rows], rows, als.lambda_, solver, als.cg_steps, als.implicit,
                als.dynamic_lambda, G64=G64, n_threads=n_threads, yardstick=yardstick)
    rep["side"] = side
    return loss, rep
