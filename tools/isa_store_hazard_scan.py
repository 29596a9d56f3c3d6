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
VDEST = re.compile(r"^\s*(v_[a-z0-9_]+)\s+(v\[(\d+):(\d+)\]|v(\d+))\b")


def scan(so):
    elf = code_object(so)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
    hits, n_stores, n_soff = [], 0, 0
    kernel = "?"
    lines = txt.split("\n")
    for i, ln in enumerate(lines):
        m = re.match(r"^[0-9a-f]{16} <(.+)>:", ln)
        if m:
            kernel = m.group(1)
            continue
        s = STORE.match(ln.split("//")[0])
        if not s:
            continue
        n_stores += 1
        a, b, soff = int(s.group(1)), int(s.group(2)), s.group(4)
        if not re.match(r"^s\d+$|^m0$|^vcc_(lo|hi)$|^ttmp\d+$", soff):          # literal / inline-constant offset: LLVM pads this case itself
            continue
        n_soff += 1
        nxt = next((l.split("//")[0] for l in lines[i + 1:i + 3] if l.strip() and not re.match(r"^[0-9a-f]{16} <", l)), "")
        d = VDEST.match(nxt)
        if not d or d.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            continue
        lo, hi = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)), int(d.group(5)))
        if lo <= b and hi >= a:
            hits.append((kernel, ln.strip(), nxt.strip()))
    return hits, n_stores, n_soff


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pero_ocr_amd", "libpocr_hip.so")
    hits, n_stores, n_soff = scan(so)
    print(f"{os.path.relpath(so)}: {n_stores} buffer stores of more than 64 bits, {n_soff} with an SGPR offset, {len(hits)} followed at once by a VALU write of their data registers")
    for k, a, b in hits[:40]:
        print(f"  {k[:110]}\n      {a}\n      {b}")
    sys.exit(1 if hits else 0)
