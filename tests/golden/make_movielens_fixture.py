#!/usr/bin/env python3
"""Convert the reference's bundled data set into a small binary fixture.

Source (data, not code): /root/reference/data/movielens100k.RData -- the dgCMatrix
`movielens100k` (943 x 1682, 100 000 ratings) that every reference WRMF test runs on
(tests/testthat.R:9, tests/testthat/test-wrmf.R:6-7).

The .RData container is bzip2 -> R serialisation format "RDX2" / XDR.  This script is a
minimal pure-Python XDR reader (only the SEXP types that occur in that file) and writes
`movielens100k_csc.npz` with the dgCMatrix slots exactly as R holds them
(src/utils.cpp:69-78 reads the same slots):

    Dim (int32[2]), p (int32[ncol+1]), i (int32[nnz], 0-based), x (float64[nnz])

Run once in the build container (needs /root/reference); the .npz is committed.
"""
import bz2
import struct
import sys
from pathlib import Path

import numpy as np


class XDR:
    def __init__(self, buf):
        self.b = buf
        self.o = 0
        self.refs = []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]
        self.o += 4
        return v

    def raw(self, n):
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        is_obj = bool(flags & (1 << 8))
        if t == 254:  # NILVALUE
            return None
        if t == 253:  # R_GlobalEnv
            return "<globalenv>"
        if t == 242:  # R_EmptyEnv
            return "<emptyenv>"
        if t == 241:  # BASEENV
            return "<baseenv>"
        if t == 255:  # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:  # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if t == 9:  # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            return self.raw(n).decode("utf-8", "replace")
        if t == 2:  # LISTSXP (pairlist)
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                nflags = struct.unpack_from(">i", self.b, self.o)[0]
                nt = nflags & 0xFF
                if nt == 254:
                    self.o += 4
                    break
                if nt != 2:
                    raise ValueError("unexpected pairlist tail type %d" % nt)
                self.o += 4
                has_attr = bool(nflags & (1 << 9))
                has_tag = bool(nflags & (1 << 10))
            return out
        if t in (10, 13):  # LGLSXP / INTSXP
            n = self.i32()
            v = np.frombuffer(self.raw(4 * n), dtype=">i4").astype(np.int32)
            attr = self.item() if has_attr else None
            return {"v": v, "attr": attr} if attr else v
        if t == 14:  # REALSXP
            n = self.i32()
            v = np.frombuffer(self.raw(8 * n), dtype=">f8").astype(np.float64)
            attr = self.item() if has_attr else None
            return {"v": v, "attr": attr} if attr else v
        if t == 16:  # STRSXP
            n = self.i32()
            v = [self.item() for _ in range(n)]
            attr = self.item() if has_attr else None
            return {"v": v, "attr": attr} if attr else v
        if t == 19:  # VECSXP
            n = self.i32()
            v = [self.item() for _ in range(n)]
            attr = self.item() if has_attr else None
            return {"v": v, "attr": attr} if attr else v
        if t == 25:  # S4SXP: only attributes
            attr = self.item() if has_attr else None
            return {"S4": True, "attr": attr}
        raise ValueError("unsupported SEXP type %d at offset %d" % (t, self.o))


def val(x):
    return x["v"] if isinstance(x, dict) and "v" in x else x


def main():
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/movielens100k.RData")
    dst = Path(__file__).with_name("movielens100k_csc.npz")
    buf = bz2.decompress(src.read_bytes())
    assert buf[:5] == b"RDX2\n", buf[:5]
    x = XDR(buf)
    x.o = 5
    assert x.raw(2) == b"X\n"
    x.i32(); x.i32(); x.i32()  # format version 2, writer R version, min reader version
    top = x.item()  # pairlist of (symbol, value)
    (name, obj), = top
    assert name == "movielens100k", name
    slots = dict(obj["attr"])
    Dim = val(slots["Dim"]).astype(np.int32)
    p = val(slots["p"]).astype(np.int32)
    i = val(slots["i"]).astype(np.int32)
    xv = val(slots["x"]).astype(np.float64)
    dn = val(slots["Dimnames"])
    rown = val(dn[0]) if dn and dn[0] is not None else None
    coln = val(dn[1]) if dn and len(dn) > 1 and dn[1] is not None else None
    assert tuple(Dim) == (943, 1682) and len(xv) == 100000 and len(p) == 1683
    assert p[-1] == 100000 and i.min() == 0 and i.max() == 942
    # rows sorted within each column, as Matrix guarantees for dgCMatrix
    for c in range(Dim[1]):
        seg = i[p[c]:p[c + 1]]
        assert np.all(np.diff(seg) > 0)
    hist = {int(v): int((xv == v).sum()) for v in np.unique(xv)}
    assert hist == {1: 6110, 2: 11370, 3: 27145, 4: 34174, 5: 21201}, hist
    np.savez_compressed(dst, Dim=Dim, p=p, i=i, x=xv,
                        rownames=np.array(rown if rown is not None else [], dtype="U16"),
                        colnames=np.array(coln if coln is not None else [], dtype="U16"))
    print("wrote", dst, dst.stat().st_size, "bytes; value histogram", hist)


if __name__ == "__main__":
    main()
