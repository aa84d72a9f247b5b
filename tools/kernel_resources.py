#!/usr/bin/env python3
"""Static resource usage of every kernel of the library (no GPU needed): registers, spills, scratch, LDS and the occupancy
the compiler settles on, from `hipcc -Rpass-analysis=kernel-resource-usage` over rsparse_amd/csrc/*.hip with the flags of
rsparse_amd/build.py.

    python tools/kernel_resources.py                 # markdown: kernels that spill, then one line per kernel family
    python tools/kernel_resources.py --all           # every instantiation
    python tools/kernel_resources.py --json out.json

A spill is not a verdict -- a scratch reload in a prologue costs nothing, one inside a sweep does -- but it says where to look
(`llvm-objdump -d` of the object, `scratch_load` inside the loop labels).  profiles/r04/kernel_resource_usage.md is this
script's output at the end of round 4."""
import json
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from rsparse_amd import build as B   # noqa: E402


def usage_of(src, tmp):
    r = subprocess.run(["hipcc", *B.FLAGS, *B.EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(Path(tmp) / (src.stem + ".o")),
                        "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src.name, r.stderr[-3000:]))
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"file": src.name, "mangled": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif ":" in t and cur is not None:
            k, v = t.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def main():
    argv = sys.argv[1:]
    srcs = [s for s in B.SRC if s.suffix == ".hip"]
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=8) as ex:
        rows = [r for rs in ex.map(lambda s: usage_of(s, tmp), srcs) for r in rs]
    # (binutils' c++filt does not know DF16_ = _Float16; Dh = half demangles and reads the same)
    names = subprocess.run(["c++filt"], input="\n".join(r["mangled"].replace("DF16_", "Dh") for r in rows),
                           stdout=subprocess.PIPE, text=True).stdout
    for r, d in zip(rows, names.splitlines()):
        d = d.replace("rsparse_hip::(anonymous namespace)::", "").replace("rsparse_hip::", "")
        r["kernel"] = re.sub(r"^void ", "", re.sub(r"\((AlsArgs|F64Args|WideArgs|int|float|double|unsigned|long|void|bool|char|half|rsparse|hip).*$", "", d))
    rows = [r for r in rows if "rocprim" not in r["kernel"]]   # the library sort of the ingest: not ours to tune
    if "--json" in argv:
        Path(argv[argv.index("--json") + 1]).write_text(json.dumps(rows, indent=1))

    def num(r, k):
        return int(r.get(k, "0"))

    def line(r):
        return "| `%s` | %s | %d | %d | %d | %d | %d | %s | %s |" % (
            r["kernel"][:96], r["file"], num(r, "VGPRs"), num(r, "AGPRs"), num(r, "VGPRs Spill"), num(r, "SGPRs Spill"),
            num(r, "ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]", "0"), r.get("Occupancy [waves/SIMD]", "?"))
    head = ("| kernel | source | VGPRs | AGPRs | VGPR spills | SGPR spills | scratch B/lane | static LDS B | waves/SIMD |\n"
            "|---|---|---|---|---|---|---|---|---|")
    print("%d kernels in %d sources (gfx950, %s)\n" % (len(rows), len(srcs), " ".join(B.FLAGS)))
    spill = [r for r in rows if num(r, "VGPRs Spill") or num(r, "ScratchSize [bytes/lane]")]
    print("### kernels with vector spills or scratch (%d)\n\n%s" % (len(spill), head))
    for r in sorted(spill, key=lambda r: -num(r, "VGPRs Spill")):
        print(line(r))
    if "--all" in argv:
        print("\n### every kernel\n\n" + head)
        for r in rows:
            print(line(r))
        return
    print("\n### per kernel family: instantiations, VGPR range, worst spill\n")
    print("| family | source | instantiations | VGPRs (min–max) | with vector spills | waves/SIMD |\n|---|---|---|---|---|---|")
    fam = {}
    for r in rows:
        fam.setdefault((re.sub(r"<.*$", "", r["kernel"]), r["file"]), []).append(r)
    for (f, src), rs in fam.items():
        v = [num(r, "VGPRs") + num(r, "AGPRs") for r in rs]
        occ = sorted({r.get("Occupancy [waves/SIMD]", "?") for r in rs}, key=lambda x: int(x) if x.isdigit() else 0)
        print("| `%s` | %s | %d | %d–%d | %d | %s |" % (f, src, len(rs), min(v), max(v),
                                                       sum(1 for r in rs if num(r, "VGPRs Spill")), "–".join(occ[:1] + occ[-1:]) if len(occ) > 1 else occ[0]))


if __name__ == "__main__":
    main()
