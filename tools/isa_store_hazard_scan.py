"""Scan the gfx950 code object of a built library for the pattern behind round 5's "masked store" corruption:

    buffer_store_dwordx3/x4 v[a:b], vaddr, s[..], sN offen        (more than 64 bits of data, SGPR offset)
    v_<op> vK, ...          with a <= K <= b                      (VALU write of a data register in the very next issue slot)

gfx9's rule "one wait state between a VMEM store of more than 64 bits and a VALU write of its data registers" is applied by LLVM's
hazard recogniser only when the store has no SGPR offset; on gfx950 the pair above, back to back, stores the LATER value under
memory back-pressure (tools/store_hazard_probe.hip, profiles/r06_store_hazard.txt).  The compiler can emit it whenever it
schedules address arithmetic for the next store into a freed data register, so the library is scanned after every build
(tests/test_host.py::test_no_kernel_has_the_store_data_hazard).
Usage: python tools/isa_store_hazard_scan.py [lib.so]   -> exit code 1 and the offending sites if any"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from codeobj_hash import code_object  # noqa: E402

STORE = re.compile(r"^\s*buffer_store_(?:dwordx[34]|format_xyzw?|format_d16_xyzw)\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
GSTORE = re.compile(r"^\s*(?:global|flat|scratch)_store_dwordx[34]\s+\S+,\s*v\[(\d+):(\d+)\]")
VDEST = re.compile(r"^\s*(v_[a-z0-9_]+)\s+(v\[(\d+):(\d+)\]|v(\d+))\b")


def _valu_writes(insn, a, b):
    d = VDEST.match(insn)
    if not d or d.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return False
    lo, hi = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
    return lo <= b and hi >= a


def _wait_states(insn):
    """issue slots an instruction occupies as far as this hazard goes: s_nop N = N + 1, anything else 1"""
    m = re.match(r"^\s*s_nop\s+(\d+)", insn)
    return int(m.group(1)) + 1 if m else 1


def scan(so):
    """-> (hits, stores of more than 64 bits, those with an SGPR offset).  Required distance to a VALU write of the data registers,
    as measured on gfx950 (tools/store_hazard_probe.hip): ONE wait state behind a buffer store with an SGPR offset (the case LLVM
    does not pad), TWO behind a buffer store with a literal offset and behind global / flat stores (LLVM pads those; checked anyway)."""
    elf = code_object(so)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
    hits, n_stores, n_soff = [], 0, 0
    kernel = "?"
    lines = [l.split("//")[0] for l in txt.split("\n")]
    for i, ln in enumerate(lines):
        m = re.match(r"^[0-9a-f]{16} <(.+)>:", ln)
        if m:
            kernel = m.group(1)
            continue
        s_ = STORE.match(ln)
        g = None if s_ else GSTORE.match(ln)
        if not s_ and not g:
            continue
        n_stores += 1
        if s_:
            a, b, soff = int(s_.group(1)), int(s_.group(2)), s_.group(4)
            sgpr = bool(re.match(r"^s\d+$|^m0$|^vcc_(lo|hi)$|^ttmp\d+$", soff))
            need = 1 if sgpr else 2
            n_soff += 1 if sgpr else 0
        else:
            a, b, need = int(g.group(1)), int(g.group(2)), 2
        gap = 0
        for nxt in lines[i + 1:i + 6]:
            if not nxt.strip() or re.match(r"^[0-9a-f]{16} <", nxt):
                break
            if gap >= need:
                break
            if _valu_writes(nxt, a, b):
                hits.append((kernel, ln.strip(), nxt.strip(), gap))
                break
            gap += _wait_states(nxt)
    return hits, n_stores, n_soff


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pero_ocr_amd", "libpocr_hip.so")
    hits, n_stores, n_soff = scan(so)
    print(f"{os.path.relpath(so)}: {n_stores} stores of more than 64 bits (buffer / global / flat), {n_soff} buffer stores with an SGPR offset, "
          f"{len(hits)} followed too closely by a VALU write of their data registers")
    for k, a, b, gap in hits[:40]:
        print(f"  {k[:110]}\n      {a}\n      {b}      ({gap} wait state(s) between them)")
    sys.exit(1 if hits else 0)
