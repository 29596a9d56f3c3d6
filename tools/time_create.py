import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pero_ocr_amd import netspec, _native
for arch, kw in (("vgg_blstm_ctc", {}), ("vgg_sa_s2s", {"dec_layers": 3})):
    spec = netspec.NetSpec(num_classes=233, arch=arch, **kw)
    t0 = time.perf_counter(); w = netspec.generate_weights(spec, 1); t1 = time.perf_counter()
    flat = netspec.pack_weights(spec, w); t2 = time.perf_counter()
    e = _native.NativeEngine(spec, flat, 0); t3 = time.perf_counter()
    print(arch, f"generate {t1-t0:.2f}s pack {t2-t1:.2f}s create {t3-t2:.2f}s floats {flat.size/1e6:.1f}M")
    e.close()
