"""Build librsparse_wrmf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

One object per source (compiled in parallel, rebuilt only when the source or a shared header changed), then one link.
In-tree output (rsparse_amd/lib/) so the .so travels with the repo snapshot to the GPU box; the objects live in
rsparse_amd/lib/obj/ (git-ignored like the .so).

    python -m rsparse_amd.build [--force] [-D NAME[=VALUE] ...] [--out other.so]

-D / --out are for dev builds (in-kernel profilers, ablations: tools/build_prof.sh, tools/build_abl.sh); such builds use
their own object directory.
"""
import hashlib
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
SRC = [CSRC / n for n in ("wrmf_kernels.hip", "wrmf_cgq.hip", "wrmf_cgp.hip", "wrmf_ne.hip", "wrmf_chol.hip", "wrmf_chol_wave.hip", "wrmf_chol_mf.hip", "wrmf_cg_mf.hip", "wrmf_chol_lr.hip",
                          "wrmf_topk.hip", "wrmf_ingest.hip", "wrmf_nnls.hip", "wrmf_bias.hip", "wrmf_lu.hip",
                          "wrmf_f64.hip", "wrmf_wide.hip", "wrmf_wide_cg.hip", "wrmf_ctx_kernels.hip", "wrmf_capi.cpp", "wrmf_f64_capi.cpp", "wrmf_ctx.cpp")]
HEADERS = [CSRC / "wrmf_chol_mf.attrs.csv", CSRC / "wrmf_mf.h", CSRC / "wrmf_internal.h", CSRC / "wrmf_device.h", CSRC / "wrmf_ldlt.h", CSRC / "wrmf_f64.h", PKG.parent / "include" / "rsparse_wrmf_hip.h"]
DEPS = SRC + HEADERS
OUT = PKG / "lib" / "librsparse_wrmf_hip.so"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


# wrmf_chol_mf.hip names its accumulator registers itself (a0..a159, inline asm) and needs hipcc to stay out of the accumulator
# file: LLVM's function attribute "amdgpu-agpr-alloc"="0" has no source spelling, the ForceFunctionAttrs pass applies it from a
# csv (function,attribute=value).  The listing of that very compilation is then audited (tools/dbg/acc_audit.py: no
# compiler-generated accumulator-file instruction, nothing touches a register an asm load has in flight, two waves per SIMD);
# if the audit fails -- another compiler, a lost flag -- the file is rebuilt with -DMF_SAFE (tiles above hipcc's own share of the
# accumulator file, one wave per SIMD) and audited again; a build that passes neither is an error, never a silent corruption.
EXTRA_FLAGS = {n: ["-mllvm", "-forceattrs-csv-path=" + str(CSRC / "wrmf_chol_mf.attrs.csv"), "-fno-slp-vectorize"] for n in ("wrmf_chol_mf.hip", "wrmf_cg_mf.hip")}
AUDITED = {"wrmf_chol_mf.hip", "wrmf_cg_mf.hip"}
REG_LIMIT = {"wrmf_cg_mf.hip": 512}   # (one wave per SIMD by design: 320 accumulator registers per row)


def audit_listing(src, extra, defines, obj):
    sys.path.insert(0, str(PKG.parent / "tools" / "dbg"))
    import acc_audit
    lst = obj.with_suffix(".s")

    def listing(flags, floor):
        cmd = ["hipcc", *FLAGS, *flags, *["-D" + d for d in defines], "-S", "--cuda-device-only", str(src), "-o", str(lst)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed on %s:\n%s" % (src.name, r.stderr[-3000:]))
        acc, flight, _ = acc_audit.audit(str(lst), quiet=True, acc_floor=floor)
        return acc, flight, lst.read_text()
    acc, flight, text = listing(extra, 0)
    import re
    # vector registers + the 160 named accumulator registers <= 256: two waves per SIMD.  Demanded of the kernels that run in
    # the normal case; the `_any` instantiations (some confidence below 1: two operand sets) may take a few registers more
    nfree = re.findall(r"\.amdhsa_kernel (\w+)[\s\S]*?\.amdhsa_next_free_vgpr (\d+)", text)
    limit = REG_LIMIT.get(src.name, 256)
    two_waves = bool(nfree) and all(int(v) <= limit for name, v in nfree if "_any" not in name)
    if acc == 0 and flight == 0 and two_waves:
        return
    print("  %s: audit of the two-waves-per-SIMD build failed (accumulator-file %d, in-flight %d, layout ok %s): rebuilding with -DMF_SAFE"
          % (src.name, acc, flight, two_waves), flush=True)
    safe = ["-DMF_SAFE"]
    cmd = ["hipcc", *FLAGS, *safe, *["-D" + d for d in defines], "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s (MF_SAFE):\n%s" % (src.name, r.stderr[-3000:]))
    # (MF_SAFE: hipcc may spill to a0..a95, the tiles are a96..a255: a compiler access at or above a96 breaks rule 1)
    acc, flight, text = listing(safe, 96)
    if acc or flight:
        raise RuntimeError("%s: the MF_SAFE build fails the audit too (accumulator-file %d, in-flight %d)" % (src.name, acc, flight))


def build(force=False, verbose=False, defines=(), out=None):
    out = Path(out) if out else OUT
    out.parent.mkdir(exist_ok=True)
    import os
    # dev builds only: RSPARSE_HIPCC_EXTRA="wrmf_cg_mf.hip=-fno-slp-vectorize;other.hip=-flag1,-flag2" (compiler switches of one source)
    per_file = {}
    for item in filter(None, os.environ.get("RSPARSE_HIPCC_EXTRA", "").split(";")):
        name, _, fl = item.partition("=")
        per_file[name] = [f for f in fl.split(",") if f]
    if per_file and out == OUT:
        raise RuntimeError("RSPARSE_HIPCC_EXTRA is for dev builds (--out)")
    tag = hashlib.sha1((" ".join(sorted(defines)) + repr(sorted(per_file.items()))).encode()).hexdigest()[:8] if (defines or per_file) else "release"
    objdir = OUT.parent / "obj" / tag
    objdir.mkdir(parents=True, exist_ok=True)
    src = [s for s in SRC if s.exists()]
    hdr_m = max(h.stat().st_mtime for h in HEADERS)
    todo, objs = [], []
    names = [d.split("=")[0] for d in defines]
    hdr_text = "".join(h.read_text() for h in HEADERS if h.suffix == ".h")
    for s in src:
        o = objdir / (s.stem + ".o")
        if (defines or per_file) and s.name not in per_file and not any(n in s.read_text() or n in hdr_text for n in names):
            rel = OUT.parent / "obj" / "release" / (s.stem + ".o")   # a dev define this source never mentions: the release object
            if rel.exists() and rel.stat().st_mtime >= max(s.stat().st_mtime, hdr_m):
                objs.append(rel)
                continue
        objs.append(o)
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr_m):
            todo.append((s, o))
    if not todo and out.exists() and all(out.stat().st_mtime >= o.stat().st_mtime for o in objs):
        return out

    def compile_one(so):
        s, o = so
        t0 = time.time()
        extra = list(EXTRA_FLAGS.get(s.name, [])) + per_file.get(s.name, [])
        cmd = ["hipcc", *FLAGS, *extra, *["-D" + d for d in defines], "-c", str(s), "-o", str(o)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s.name, r.stderr[-6000:]))
        if s.name in AUDITED:
            audit_listing(s, extra, defines, o)
        if verbose:
            print("  %-20s %.1f s" % (s.name, time.time() - t0), flush=True)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(compile_one, todo))
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *map(str, objs), "-ldl", "-lpthread", "-o", str(out)]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    defs, outp, i = [], None, 0
    while i < len(argv):
        if argv[i] == "-D":
            defs.append(argv[i + 1]); i += 2
        elif argv[i].startswith("-D"):
            defs.append(argv[i][2:]); i += 1
        elif argv[i] == "--out":
            outp = argv[i + 1]; i += 2
        else:
            i += 1
    t0 = time.time()
    print(build(force="--force" in argv, verbose=True, defines=tuple(defs), out=outp), "%.1f s" % (time.time() - t0))
