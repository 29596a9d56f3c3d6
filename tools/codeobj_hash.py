"""sha256 of the gfx950 code object embedded in a built library (whole ELF and its .text section): two builds whose device code is
byte-identical compute the same bits.  Usage: python tools/codeobj_hash.py [path/to/lib.so ...]"""
import hashlib, os, struct, subprocess, sys, tempfile


def code_object(so):
    data = open(so, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = data.find(magic)
    n = struct.unpack_from("<Q", data, at + len(magic))[0]
    p = at + len(magic) + 8
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", data, p); p += 24
        triple = data[p:p + tlen].decode(); p += tlen
        if "gfx950" in triple:
            return data[at + off:at + off + size]
    raise SystemExit(f"{so}: no gfx950 code object")


def text_section(elf):
    with tempfile.NamedTemporaryFile(suffix=".co") as f, tempfile.NamedTemporaryFile(suffix=".text") as t:
        f.write(elf); f.flush()
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.text", f.name, t.name], check=True)
        return open(t.name, "rb").read()


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for so in (sys.argv[1:] or [os.path.join(here, "..", "pero_ocr_amd", "libpocr_hip.so")]):
        elf = code_object(so)
        txt = text_section(elf)
        print(f"{os.path.relpath(so)}: code object {len(elf)} B sha256 {hashlib.sha256(elf).hexdigest()[:16]}; .text {len(txt)} B sha256 {hashlib.sha256(txt).hexdigest()[:16]}")


def kernels(so):
    """name -> sha256 of the kernel's machine code (symbol table of the code object: FUNC symbols of .text)."""
    elf = code_object(so)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf); f.flush()
        syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--syms", "--wide", f.name], capture_output=True, text=True, check=True).stdout
        secs = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", "--wide", f.name], capture_output=True, text=True, check=True).stdout
    import re
    m = re.search(r"\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
    addr, off = int(m.group(1), 16), int(m.group(2), 16)
    out = {}
    for ln in syms.split("\n"):
        p = ln.split()
        if len(p) >= 8 and p[3] == "FUNC" and p[6] != "UND":
            a, n = int(p[1], 16), int(p[2])
            out[p[7]] = hashlib.sha256(elf[off + a - addr: off + a - addr + n]).hexdigest()[:16]
    return out
