"""What the range guard's fall-back costs a caller: time of pocr_create, of the first re-run (which
creates the bf16x3 engine and its buffers) and of a later one, against a launch that stays in range.
    python tools/fallback_cost.py [lines] [w_pad]"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_ocr_amd import _native, netspec, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    w_pad = int(sys.argv[2]) if len(sys.argv) > 2 else 576
    chars = synth.make_charset(99)
    spec = netspec.NetSpec(num_classes=len(chars) + 1)
    blob = netspec.pack_weights(spec, netspec.generate_weights(spec, 7))
    crops = synth.make_crops(3, [w_pad - 64] * n)
    pool = np.concatenate([c.reshape(-1) for c in crops])
    offs = np.concatenate([[0], np.cumsum([c.size for c in crops])[:-1]]).astype(np.int64)
    widths = np.array([c.shape[1] for c in crops], np.int32)

    def launch(eng, slot=0):
        t = time.perf_counter()
        eng.slot_stage_ragged(slot, pool, offs, widths, [w_pad] * n, 32)
        eng.slot_launch(slot, want_logits=False, want_argmax=True)
        eng.slot_collect(slot)
        return (time.perf_counter() - t) * 1e3

    t = time.perf_counter(); eng = _native.NativeEngine(spec, blob, 0); t_create = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); eng2 = _native.NativeEngine(spec, blob, 0); t_create2 = (time.perf_counter() - t) * 1e3
    eng2.close()
    runs = [launch(eng) for _ in range(4)]
    if os.environ.get("POCR_FORCE_RANGE_FALLBACK") == "1":
        print("  same, every launch re-run on bf16x3 (ms):       " + " ".join(f"{x:.1f}" for x in runs)
              + f"   fall-backs {eng.range_fallbacks()}")
    else:
        print(f"{n} lines @ W_pad {w_pad}: pocr_create {t_create:.0f} ms (first of the process), {t_create2:.0f} ms (second)")
        print("  stage + launch + collect, in range (ms):        " + " ".join(f"{x:.1f}" for x in runs), flush=True)
        # the library reads the switch once, at its first guard: the forced half is a process of its own
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, POCR_FORCE_RANGE_FALLBACK="1"), check=True)
    eng.close()


if __name__ == "__main__":
    main()
