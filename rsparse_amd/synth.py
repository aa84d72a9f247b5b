"""Synthetic user x item interaction matrices for the BASELINE configs.

Counter-based: every quantity is a pure function of (seed, user) or (seed, user, item) through a
64-bit mix hash, so a block of users can be generated on any rank / device and is identical
everywhere (torch CPU and torch ROCm run the same integer code).

  degree      d_u = clip(round(LogNormal(mu, sigma=1)), 1, d_max), mu chosen so the mean is `mean_deg`
  items       stratified draws u_j = (j + r)/d_u mapped through n_items * u^2 (Zipf-like popularity),
              made strictly increasing, then scattered over the id space by a fixed hash permutation
  values      implicit: 1 + Geometric(1/2) counts, used raw as confidence (R/model_WRMF.R:51-53);
              explicit: ratings 1..5 with the movielens100k histogram

Both orientations are produced the way the reference holds them (R/model_WRMF.R:184-191):
  c_ui  CSC of users x items (columns = items)   c_iu  CSC of its transpose (columns = users),
each with sorted row indices inside a column, int32 indices, float32 values.
"""
import math

import torch

_M1 = -49064778989728563      # 0xff51afd7ed558ccd as signed 64-bit
_M2 = -4265267296055464877    # 0xc4ceb9fe1a85ec53
_GOLD = -7046029254386353131  # 0x9e3779b97f4a7c15
_RATING_CDF = (0.06110, 0.17480, 0.44625, 0.78799, 1.0)   # movielens100k value histogram


def _lsr(x, s):
    """logical shift right on int64 tensors"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def mix64(x):
    """murmur3 fmix64 on int64 tensors (wrapping arithmetic)."""
    x = x ^ _lsr(x, 33)
    x = x * _M1
    x = x ^ _lsr(x, 33)
    x = x * _M2
    x = x ^ _lsr(x, 33)
    return x


def uniform(counter, seed, stream):
    """U[0,1) float64 from (counter, seed, stream); 53 random bits."""
    h = mix64(counter * _GOLD + (int(seed) * 1000003 + int(stream) * 7919 + 12345))
    return _lsr(h, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def degrees(user_ids, seed, mean_deg, d_max, n_items):
    mu = math.log(mean_deg) - 0.5
    u1 = uniform(user_ids, seed, 1).clamp_min(1e-300)
    u2 = uniform(user_ids, seed, 2)
    z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
    d = torch.round(torch.exp(mu + z)).to(torch.int64)
    return d.clamp(1, min(d_max, n_items))


def item_permutation(n_items, seed, device):
    ids = torch.arange(n_items, dtype=torch.int64, device=device)
    return torch.argsort(mix64(ids * _GOLD + int(seed) * 31 + 777), stable=True)


def generate_user_block(u0, u1, n_items, seed=20250222, mean_deg=50.0, d_max=5000, feedback="implicit",
                        device="cpu", perm=None):
    """CSR rows [u0, u1) of the users x items matrix: (indptr int64 [u1-u0+1], item int64, value f32),
    items sorted inside each row."""
    dev = torch.device(device)
    users = torch.arange(u0, u1, dtype=torch.int64, device=dev)
    deg = degrees(users, seed, mean_deg, d_max, n_items)
    indptr = torch.zeros(u1 - u0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=indptr[1:])
    nnz = int(indptr[-1])
    row = torch.repeat_interleave(users, deg)
    first = torch.repeat_interleave(indptr[:-1], deg)
    j = torch.arange(nnz, dtype=torch.int64, device=dev) - first          # position inside the row
    d = torch.repeat_interleave(deg, deg)
    r = uniform(row * 8192 + j, seed, 3)
    u = (j.to(torch.float64) + r) / d.to(torch.float64)
    raw = torch.floor(n_items * u * u).to(torch.int64).clamp_(0, n_items - 1)
    # strictly increasing inside a row: item_j = j + cummax_j(raw_j - j), then cap from the right
    a = raw - j
    # segmented cummax via a global cummax on (row_rank * BIG + a): rows are contiguous and ordered
    big = n_items + d_max + 1
    rr = row - u0
    a = torch.cummax(rr * (2 * big) + (a + big), 0).values - rr * (2 * big) - big
    item = a + j
    item = torch.minimum(item, n_items - d + j)
    if perm is None:
        perm = item_permutation(n_items, seed, dev)
    item = perm[item]
    # canonical order inside each row
    key = torch.sort(row * n_items + item).values
    row_s = key // n_items
    item_s = key - row_s * n_items
    val = values(row_s, item_s, seed, feedback)
    return indptr, item_s, val


def values(row, item, seed, feedback):
    u = uniform(row * 1048573 + item, seed, 4)
    if feedback == "implicit":
        g = torch.floor(-torch.log2((1.0 - u).clamp_min(1e-12))).clamp_(0, 30)
        return (1.0 + g).to(torch.float32)
    v = torch.ones_like(u)
    for c in _RATING_CDF[:-1]:
        v = v + (u >= c).to(u.dtype)
    return v.to(torch.float32)


def transpose_to_csc(n_users, n_items, indptr_u, item, val, u0=0):
    """users-major CSR block -> CSC by item of the same block (row ids are global user ids)."""
    dev = item.device
    deg = (indptr_u[1:] - indptr_u[:-1])
    row = torch.repeat_interleave(torch.arange(u0, u0 + deg.numel(), dtype=torch.int64, device=dev), deg)
    key, order = torch.sort(item * n_users + row)
    item_s = key // n_users
    row_s = key - item_s * n_users
    counts = torch.bincount(item_s, minlength=n_items)
    p = torch.zeros(n_items + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=p[1:])
    return p, row_s, val[order]


def make_dataset(n_users, n_items, seed=20250222, mean_deg=50.0, d_max=5000, feedback="implicit", device="cpu",
                 block=2_000_000):
    """Full matrix in both orientations.  Returns a dict with
       c_iu = (p int32 [n_users+1], i int32 (items), x f32)   columns = users
       c_ui = (p int32 [n_items+1], i int32 (users), x f32)   columns = items
    Generated in user blocks to bound temporary memory."""
    dev = torch.device(device)
    perm = item_permutation(n_items, seed, dev)
    ptrs, items, vals = [], [], []
    base = 0
    for u0 in range(0, n_users, block):
        u1 = min(n_users, u0 + block)
        ip, it, v = generate_user_block(u0, u1, n_items, seed, mean_deg, d_max, feedback, dev, perm)
        ptrs.append(ip[:-1] + base)
        base += int(ip[-1])
        items.append(it.to(torch.int32))
        vals.append(v)
    if base >= 2 ** 31:
        raise ValueError("nnz >= 2^31 does not fit the reference's 32-bit index layout")
    p_u = torch.cat(ptrs + [torch.tensor([base], dtype=torch.int64, device=dev)])
    item = torch.cat(items)
    val = torch.cat(vals)
    del ptrs, items, vals
    p_i, row_i, val_i = transpose_to_csc(n_users, n_items, p_u, item.to(torch.int64), val)
    return {
        "n_users": n_users, "n_items": n_items, "nnz": base,
        "c_iu": (p_u.to(torch.int32), item, val),
        "c_ui": (p_i.to(torch.int32), row_i.to(torch.int32), val_i),
    }


def make_shard(n_users, n_items, world_size, rank, bounds_fn, seed=20250222, mean_deg=50.0, d_max=5000,
               feedback="implicit", device="cpu", block=2_000_000, be=None, group=None):
    """One rank's share of the matrix `make_dataset` produces.  NO rank generates more than its own users' rows (round 6;
    through round 5 every rank streamed over the WHOLE matrix twice -- 2 x 2.6 s of generation that nothing shared at
    N = 8, and what made eight processes on one device look hung):

      1. the per-user counts are the degrees, known without generating anything -> the user bounds;
      2. the rank generates the rows of ITS user block (the generator is counter-based: a block is identical whoever makes it)
         = its c_iu block (columns = my users);
      3. the per-item counts are one all-reduce of the ranks' bincounts -> the item bounds;
      4. its c_ui block (columns = my items) is built among the ranks exactly as a sharded `WRMF.fit_transform` builds it:
         `engine.item_block_from_user_blocks` -- three all-to-alls of the matrix's bytes + one on-device transposition.

    `be` (a backend with `transpose_csc`: engine.HipBackend, or the tests' CPU stand-in) and `group` (torch.distributed, None =
    the default group) are needed for steps 3-4 when world_size > 1.  `bounds_fn(cnt_user, cnt_item) -> (user_bounds,
    item_bounds)` decides the blocks (engine.ShardedALS.layouts).  Returns dict(nnz, cnt_user, cnt_item, user_bounds,
    item_bounds, c_iu, c_ui) with the CSC blocks re-based to 0 and GLOBAL row indices, bit-identical to slices of
    `make_dataset`'s matrix."""
    dev = torch.device(device)
    perm = item_permutation(n_items, seed, dev)
    cnt_user = degrees(torch.arange(n_users, dtype=torch.int64, device=dev), seed, mean_deg, d_max, n_items)
    ub, _ = bounds_fn(cnt_user, None)   # the user bounds only need the degrees
    u0, u1 = ub[rank]
    keep_p, keep_i, keep_x = [], [], []
    base = 0
    cnt_item = torch.zeros(n_items, dtype=torch.int64, device=dev)
    for b0 in range(u0, u1, block):
        b1 = min(u1, b0 + block)
        ip, it, v = generate_user_block(b0, b1, n_items, seed, mean_deg, d_max, feedback, dev, perm)
        cnt_item += torch.bincount(it, minlength=n_items)
        keep_p.append(ip[:-1] + base)
        base += int(ip[-1])
        keep_i.append(it.to(torch.int32))
        keep_x.append(v)
    if base >= 2 ** 31:
        raise ValueError("a rank's block holds >= 2^31 non-zeros: more than the reference's 32-bit index layout addresses")
    p_u = torch.cat(keep_p + [torch.tensor([base], dtype=torch.int64, device=dev)]) if keep_p else \
        torch.zeros(1, dtype=torch.int64, device=dev)
    c_iu = (p_u.to(torch.int32), torch.cat(keep_i) if keep_i else torch.zeros(0, dtype=torch.int32, device=dev),
            torch.cat(keep_x) if keep_x else torch.zeros(0, dtype=torch.float32, device=dev))
    del keep_p, keep_i, keep_x
    if world_size > 1:
        from . import engine
        if be is None:
            raise ValueError("make_shard at world_size > 1 needs a backend (transpose_csc) and a process group")
        engine.all_reduce_any(cnt_item, group)
    nnz = int(cnt_item.sum())
    if nnz >= 2 ** 31:
        raise ValueError("nnz >= 2^31 does not fit the reference's 32-bit index layout")
    ub, ib = bounds_fn(cnt_user, cnt_item)
    if world_size > 1:
        c_ui = engine.item_block_from_user_blocks(be, group, world_size, rank, n_users, ub, ib, *c_iu)
    else:
        p_i, row_i, val_i = transpose_to_csc(n_users, n_items, p_u, c_iu[1].to(torch.int64), c_iu[2])
        c_ui = (p_i.to(torch.int32), row_i.to(torch.int32), val_i)
    return {"n_users": n_users, "n_items": n_items, "nnz": nnz, "cnt_user": cnt_user, "cnt_item": cnt_item,
            "user_bounds": ub, "item_bounds": ib, "c_iu": c_iu, "c_ui": c_ui}
