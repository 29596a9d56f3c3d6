"""Register / scratch / LDS figures of every kernel in the built library (from the gfx950 code object embedded in libpocr_hip.so).
Usage: python tools/kernel_meta.py [substring filter]"""
import os, re, struct, subprocess, sys, tempfile
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pero_ocr_amd", "libpocr_hip.so")
if len(sys.argv) > 2: so = sys.argv[2]
data = open(so, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
at = data.find(magic)
n = struct.unpack_from("<Q", data, at + len(magic))[0]
p, elf = at + len(magic) + 8, None
for _ in range(n):
    off, size, tlen = struct.unpack_from("<QQQ", data, p); p += 24
    triple = data[p:p + tlen].decode(); p += tlen
    if "gfx950" in triple: elf = data[at + off:at + off + size]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(elf); f.flush()
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True, check=True).stdout
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s*(\S+)", blk)
    nm = g("name").group(1)
    try:
        nm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", nm], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    if flt in nm:
        print(f"vgpr {g('vgpr_count').group(1):>4} agpr {blk.split()[0]:>3} sgpr {g('sgpr_count').group(1):>4} scratch {g('private_segment_fixed_size').group(1):>4} lds {g('group_segment_fixed_size').group(1):>6}  {nm[:150]}")
