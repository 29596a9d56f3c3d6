"""CPU tests of the product's host logic and of the C-ABI library surface
(no compute calls: there is no GPU here)."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

from conftest import REPO, gpu_available
from oracle import engine_oracle
from pero_ocr_amd import _native, netspec
from pero_ocr_amd.ocr_engine import line_ocr_engine
from pero_ocr_amd.ocr_engine.softmax import softmax


def test_plan_chunks_matches_oracle_random():
    rng = np.random.RandomState(3)
    for trial in range(200):
        n = rng.randint(1, 60)
        widths = rng.randint(1, 2500, size=n).tolist()
        if trial % 5 == 0:
            widths = [widths[0]] * n                      # all ties
        bs = int(rng.choice([1, 2, 8, 16, 32, 274]))
        got = line_ocr_engine.plan_chunks(widths, 480 * bs)
        ref = engine_oracle.chunk_plan(widths, 480 * bs)
        assert [(c.line_ids, c.max_width) for c in got] == [(ids, mw) for ids, mw in ref]
        for c in got:
            assert c.w_pad == min(c.max_width + 64, 480 * bs)


def test_plan_chunks_golden(golden):
    for name in ("c1", "ragged", "c2"):
        g = golden(name)
        got = line_ocr_engine.plan_chunks(g.widths, 480 * g.batch_size)
        assert [[c.line_ids, c.max_width] for c in got] == g.plan


def test_zero_width_line_raises_like_reference():
    # only when the empty line opens a chunk (the reference divides by ceil32(0), line_ocr_engine.py:81-87)
    with pytest.raises(ZeroDivisionError):
        line_ocr_engine.plan_chunks([0], 3840)
    with pytest.raises(ZeroDivisionError):
        engine_oracle.chunk_plan([0], 3840)
    assert [c.line_ids for c in line_ocr_engine.plan_chunks([100, 0], 3840)] == [[0, 1]]


def test_softmax_matches_oracle():
    x = np.random.RandomState(0).randn(7, 13, 50).astype(np.float32) * 8
    assert np.array_equal(softmax(x, axis=2), engine_oracle.softmax(x.reshape(-1, 50), axis=1).reshape(x.shape))
    assert softmax(x[0, 0]).shape == (50,)


def test_library_exports_every_header_symbol():
    """include/pocr.h is the contract: every function it declares must be exported."""
    hdr = open(os.path.join(REPO, "include", "pocr.h")).read()
    declared = set(re.findall(r"\b(pocr_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    lib = ctypes.CDLL(_native.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.load().pocr_abi_version() == _native.ABI_VERSION


def test_weight_count_agrees_with_library():
    lib = _native.load()
    for spec in (netspec.NetSpec(num_classes=100), netspec.NetSpec(num_classes=232, height=48, conv_out=256,
                                                                    lstm_hidden=128, lstm_layers=3)):
        cfg = _native.PocrConfig(_native.ABI_VERSION, spec.height, spec.num_classes, spec.conv_out,
                                 spec.lstm_hidden, spec.lstm_layers)
        assert lib.pocr_num_weight_floats(ctypes.byref(cfg)) == netspec.num_weight_floats(spec)


def test_create_rejects_bad_config_and_wrong_blob_size():
    lib = _native.load()
    h = ctypes.c_void_p()
    w = np.zeros(10, np.float32)
    bad = _native.PocrConfig(_native.ABI_VERSION, 41, 100, 512, 256, 2)
    assert lib.pocr_create(ctypes.byref(bad), w.ctypes.data_as(_native._f32p), w.size, 0, ctypes.byref(h)) != 0
    assert b"height" in lib.pocr_last_error()
    ok = _native.PocrConfig(_native.ABI_VERSION, 40, 100, 512, 256, 2)
    assert lib.pocr_create(ctypes.byref(ok), w.ctypes.data_as(_native._f32p), w.size, 0, ctypes.byref(h)) != 0
    assert b"floats" in lib.pocr_last_error()
    old = _native.PocrConfig(99, 40, 100, 512, 256, 2)
    assert lib.pocr_create(ctypes.byref(old), w.ctypes.data_as(_native._f32p), w.size, 0, ctypes.byref(h)) != 0
    assert b"ABI" in lib.pocr_last_error()


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback(golden, tmp_path):
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
    g = golden("c1")
    path = g.write_engine_json(tmp_path)

    class Dev:
        type, index = "cuda", 0
    with pytest.raises(RuntimeError, match="no HIP device|CPU fallback"):
        PytorchEngineLineOCR(path, Dev())


def test_cpu_device_is_refused(golden, tmp_path):
    from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import PytorchEngineLineOCR
    g = golden("c1")
    path = g.write_engine_json(tmp_path)

    class Cpu:
        type, index = "cpu", None
    with pytest.raises(RuntimeError, match="no CPU path"):
        PytorchEngineLineOCR(path, Cpu())


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under pero_ocr_amd/ may reference it."""
    pkg = os.path.join(REPO, "pero_ocr_amd")
    for root, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(root, f), encoding="utf8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(root, f)
                assert "/root/reference" not in src, os.path.join(root, f)
    # outside the package: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use the oracle
    for f in os.listdir(os.path.join(REPO, "tools")):
        if f.endswith(".py"):
            src = open(os.path.join(REPO, "tools", f), encoding="utf8").read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
            assert "/root/reference" not in src, f
    bench = open(os.path.join(REPO, "bench.py"), encoding="utf8").read()
    assert len(re.findall(r"^\s*(?:from|import)\s+oracle\b", bench, re.M)) == 1      # inside cpu_baseline()
    assert "/root/reference" not in bench


def test_export_weights_roundtrip_from_torch_state_dict(tmp_path):
    """tools/export_weights.py: a torch state_dict with the engine's topology -> identical blob."""
    import importlib.util
    from oracle import model_oracle
    spec = netspec.NetSpec(num_classes=37, conv_out=64, lstm_hidden=32, lstm_layers=2)
    w = netspec.generate_weights(spec, 11)
    net = model_oracle.OracleNet(spec, w)
    sp = importlib.util.spec_from_file_location("export_weights", os.path.join(REPO, "tools", "export_weights.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    spec2, w2 = mod.state_to_weights(net.state_dict(), height=40)
    assert spec2 == spec
    assert all(np.array_equal(w[k], w2[k]) for k in w), [k for k in w if not np.array_equal(w[k], w2[k])]


def test_embeddings_layer_is_reachable_like_in_the_reference():
    """user_scripts/select_embed_id.py:114-120: `for name, child in engine.model.named_modules()` -> the module named
    "embeddings_layer" with original_name "Embedding", `next(child.parameters()).cpu().detach().numpy()` = the table."""
    from pero_ocr_amd.ocr_engine import pytorch_ocr_engine as pe
    w = np.arange(24, dtype=np.float32).reshape(4, 6)
    view = pe._EmbeddingView(w)
    assert view.original_name == "Embedding" and view.weight.shape[0] - 1 == 3          # get_mean_embed_id's expression
    assert np.array_equal(next(view.parameters()).cpu().detach().numpy(), w)


def test_plan_launches_are_balanced():
    """every launch within one chunk's work of the mean - no shortfall piling up in the last launch (it sets the
    activation high-water mark of a slot and the pipeline's tail)."""
    from pero_ocr_amd import synth
    for n in (47, 300, 2048, 8192):
        widths = synth.make_widths(33, n)
        chunks = line_ocr_engine.plan_chunks(widths, 480 * 8)
        for target in (line_ocr_engine.LAUNCH_WORK_TARGET, 40000, 10 ** 9):
            launches = line_ocr_engine.plan_launches(chunks, target)
            assert [c for l in launches for c in l.chunks] == list(chunks)          # order kept, nothing lost
            works = [l.work for l in launches]
            total, biggest = sum(works), max(len(c.line_ids) * c.w_pad for c in chunks)
            n_l = max(1, -(-total // target))
            assert len(launches) <= n_l
            assert max(works) <= total / n_l + biggest, (n, target, max(works), total / n_l, biggest)


def test_bench_json_shape_for_rccl_and_gloo_fallback():
    """a figure whose exchange ran over the gloo fallback must not be readable as the RCCL result."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    res = {"value": 123.0}
    ok = bench.shape_rccl_fields(res, {"rccl_ranks": 8}, "rccl (pocr_allgather_labels, C ABI)")
    assert ok == {"rccl_ranks": 8}
    fb = bench.shape_rccl_fields(res, {"rccl_ranks": 0}, "gloo FALLBACK (RCCL communicator not available on every rank; this rank: x)")
    assert fb["value"] is None and fb["value_gloo_fallback"] == 123.0 and fb["rccl_ranks"] == 0
    single = bench.shape_rccl_fields(res, {"rccl_ranks": 0}, "none")
    assert single == {"rccl_ranks": 0}
    assert bench.conv_peak_tflops(2) == pytest.approx(2500.0 / 3) and bench.conv_peak_tflops(3) == pytest.approx(2500.0 / 6)
    assert bench.conv_peak_tflops(0) == 157.3


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_rank_launch_plan_for_eight_gpus(tmp_path):
    """what `bench.py --gpus 8` starts when no launcher did - one process per GPU behind
    torch.distributed.run, rendezvous on 127.0.0.1 at a free port, dmabuf IPC kept for RCCL - and, executed for real with a
    stand-in script, the environment the eight ranks see (RANK / LOCAL_RANK 0..7, WORLD_SIZE 8, the same MASTER_* everywhere)."""
    import subprocess
    bench = _load_bench()
    cmd, env = bench.rank_launch_plan(8, ["--gpus", "8", "--steps", "3"], environ={"PATH": os.environ.get("PATH", "")})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 <= port <= 65535
    assert cmd[-5:] == [os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                    # set when the caller's environment lacks it
    _c, env1 = bench.rank_launch_plan(8, [], environ={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert env1["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"                   # ... and kept when it has it
    _c, _e = bench.rank_launch_plan(2, [], port=29555)
    assert _c[_c.index("--master-port") + 1] == "29555"
    # the same plan with a stand-in for bench.py: every rank writes what it was given
    probe = tmp_path / "probe.py"
    probe.write_text("import json, os, sys\n"
                     "keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')\n"
                     "json.dump({k: os.environ.get(k) for k in keys} | {'argv': sys.argv[1:]},\n"
                     "          open(os.path.join(os.path.dirname(__file__), 'rank%s.json' % os.environ['RANK']), 'w'))\n")
    cmd, env = bench.rank_launch_plan(8, ["--gpus", "8"])
    cmd[cmd.index(os.path.join(REPO, "bench.py"))] = str(probe)
    assert subprocess.call(cmd, env=dict(env, OMP_NUM_THREADS="1"), timeout=300) == 0
    seen = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(8)]
    assert [s["RANK"] for s in seen] == [str(r) for r in range(8)]
    assert [s["LOCAL_RANK"] for s in seen] == [str(r) for r in range(8)]
    assert {s["WORLD_SIZE"] for s in seen} == {"8"} and {s["MASTER_ADDR"] for s in seen} == {"127.0.0.1"}
    assert {s["MASTER_PORT"] for s in seen} == {cmd[cmd.index("--master-port") + 1]}
    assert {s["HSA_ENABLE_IPC_MODE_LEGACY"] for s in seen} == {os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    assert all(s["argv"] == ["--gpus", "8"] for s in seen)


def test_bench_refuses_a_world_of_another_size():
    """`--gpus N` under a launcher of another world size must not print a line (a 4-rank number read as the 8-GPU result)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr and not p.stdout.strip()


def test_resize_area_sparse_equals_dense_definition():
    """The fractional INTER_AREA host resample (every page after the first takes it once the adaptive factor is remembered):
    the separable sparse form equals the dense area-weight definition, and a 2000 x 1500 page takes well under a second."""
    import time
    from pero_ocr_amd.layout_engines.torch_parsenet import resize_area
    rng = np.random.RandomState(3)

    def dense(img, ds):
        h, w = img.shape[:2]
        oh, ow = int(np.rint(h / ds)), int(np.rint(w / ds))

        def weights(n_in, n_out):
            scale = n_in / n_out
            m = np.zeros((n_out, n_in))
            for o in range(n_out):
                a, b = o * scale, min((o + 1) * scale, n_in)
                for i in range(int(np.floor(a)), min(int(np.ceil(b)), n_in)):
                    m[o, i] = min(b, i + 1) - max(a, i)
                m[o] /= m[o].sum()
            return m
        out = np.einsum("oh,hwc->owc", weights(h, oh), img.astype(np.float64))
        out = np.einsum("pw,owc->opc", weights(w, ow), out)
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)
    for h, w, ds in ((120, 200, 3.3), (97, 131, 1.7), (64, 64, 1.01), (201, 77, 4.6)):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        assert np.array_equal(resize_area(img, ds), dense(img, ds)), (h, w, ds)
    page = rng.randint(0, 256, (1500, 2000, 3)).astype(np.uint8)
    t0 = time.perf_counter()
    out = resize_area(page, 3.3)
    assert out.shape == (455, 606, 3)
    assert time.perf_counter() - t0 < 3.0          # (was 4.9 s with dense [n_out, n_in] matrices; ~0.2 s now)


def test_lazy_crop_quacks_like_the_numpy_crop():
    """`line.crop` of the resident cropper is a LazyCrop: shape / ndim / dtype / size like the numpy array the reference
    stores there, and a numpy copy the moment somebody reads pixels (np.asarray, indexing, copy) - fetched ONCE."""
    reads = []

    class FakeOwner:                      # stands in for _native.ResidentCrops (no GPU here)
        device_id = 0

        def read(self, offset, nbytes):
            reads.append((offset, nbytes))
            return (np.arange(nbytes, dtype=np.int64) + offset).astype(np.uint8)
    c = _native.LazyCrop(FakeOwner(), 1000, (40, 7, 3))
    assert c.shape == (40, 7, 3) and c.ndim == 3 and c.dtype == np.uint8 and c.size == 840 and len(c) == 40 and c._host is None
    a = np.asarray(c)
    assert a.shape == (40, 7, 3) and a.dtype == np.uint8 and a[0, 0, 0] == 1000 % 256 and reads == [(1000, 840)]
    assert np.array_equal(c[3], a[3]) and np.array_equal(c.copy(), a) and c.astype(np.float32).dtype == np.float32
    assert reads == [(1000, 840)] and c._host is not None            # one fetch, then the host copy is kept
    assert np.array_equal(np.ascontiguousarray(c, dtype=np.uint8).reshape(-1), a.reshape(-1))     # what _pack_lines does


def test_labels_to_strings_vectorised_equals_per_symbol_join():
    """The one-pass decoder must give exactly what the reference's per-symbol join gives (pytorch_ocr_engine.py:29-32),
    for empty lines, the blank symbol, non-BMP code points and character sets holding multi-code-point entries."""
    from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import labels_to_strings
    rng = np.random.default_rng(5)
    chars = [chr(0x21 + i) for i in range(90)] + ["\U0001F600", "ř", "​"]
    lab = rng.integers(0, len(chars), (64, 37)).astype(np.int32)
    lens = rng.integers(0, 38, 64).astype(np.int32)
    lens[:3] = (0, 37, 1)
    want = ["".join(chars[c] for c in lab[i, :lens[i]]) for i in range(64)]
    assert labels_to_strings(lab, lens, chars) == want
    assert labels_to_strings(lab, lens, chars) == want          # cached table
    multi = list(chars)
    multi[4] = "ch"
    assert labels_to_strings(lab, lens, multi) == ["".join(multi[c] for c in lab[i, :lens[i]]) for i in range(64)]
    assert labels_to_strings(np.zeros((0, 4), np.int32), np.zeros(0, np.int32), chars) == []


def test_no_kernel_of_the_library_spills():
    """Every kernel of the shipped library fits its register budget: `.private_segment_fixed_size` (scratch bytes per lane)
    is 0 for all of them (one layout-network tile spilled 484 B per lane).  Read from the gfx950 code
    object embedded in libpocr_hip.so (clang offload bundle) with llvm-readelf; skipped where the ROCm tool is absent."""
    import re
    import struct
    import subprocess
    import tempfile
    from pero_ocr_amd import _native
    tool = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    so = os.path.join(os.path.dirname(_native.__file__), _native.LIB_NAME)
    if not os.path.exists(tool) or not os.path.exists(so):
        pytest.skip("llvm-readelf or the built library is not here")
    data = open(so, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = data.find(magic)
    assert at >= 0
    n = struct.unpack_from("<Q", data, at + len(magic))[0]
    p, elf = at + len(magic) + 8, None
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", data, p)
        p += 24
        triple = data[p:p + tlen].decode()
        p += tlen
        if "gfx950" in triple:
            elf = data[at + off:at + off + size]
    assert elf, "no gfx950 code object in the library"
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        notes = subprocess.run([tool, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    names = re.findall(r"\.name:\s*(\S+)", notes)
    scratch = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", notes)]
    assert len(scratch) >= 100, len(scratch)
    bad = [(nm, s) for nm, s in zip([x for x in names if x.startswith("_Z")], scratch) if s]
    assert max(scratch) == 0, bad[:5]


def test_no_kernel_has_the_store_data_hazard():
    """Round 5's "masked store" corruption, root-caused in round 6 (profiles/r06_store_hazard.txt): on gfx950 a
    `buffer_store_dwordx4 ... sN offen` followed in the very next issue slot by a VALU write of one of its data registers can store the
    LATER value; LLVM's hazard recogniser pads that pair only when the store has no SGPR offset.  The shipped library must not contain
    the pair anywhere (the compiler emits it or not depending on how it schedules the address arithmetic of the next store)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_scan", os.path.join(REPO, "tools", "isa_store_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hits, n_stores, n_soff = mod.scan(_native.lib_path())
    assert n_stores >= 40 and n_soff >= 40, (n_stores, n_soff)            # (the scanner still sees the staged epilogues' stores)
    assert not hits, hits[:4]


def test_chunk_plan_is_a_lazy_sequence_with_the_plan_as_arrays():
    """plan_chunks returns the plan as arrays (what a rank of a sharded pass reads) and makes `Chunk` objects on demand:
    the arrays and the objects must say the same, slicing / iteration / equality behave like a list's, and the plan equals
    the oracle's for widths with ties, a single line and more lines than a chunk holds."""
    rng = np.random.RandomState(11)
    for widths, limit in (([777], 3840), ([64] * 200, 3840), (rng.randint(1, 4000, 500).tolist(), 480 * 8), ([5000, 10, 10], 3840)):
        plan = line_ocr_engine.plan_chunks(widths, limit)
        ref = engine_oracle.chunk_plan(widths, limit)
        assert len(plan) == len(ref) and [(c.line_ids, c.max_width) for c in plan] == [(i, m) for i, m in ref]
        assert plan.sizes.tolist() == [len(c.line_ids) for c in plan] and plan.w_pads.tolist() == [c.w_pad for c in plan]
        assert plan[-1] is plan[len(plan) - 1] and plan[0:2] == list(plan)[0:2] and plan == list(plan)
        assert sorted(i for c in plan for i in c.line_ids) == list(range(len(widths)))
    assert len(line_ocr_engine.plan_chunks([], 3840)) == 0
    big = line_ocr_engine.plan_chunks([70000, 3, 100000], 3840)          # widths beyond 16 bits take the comparison sort
    assert [c.line_ids for c in big] == [[2], [0], [1]]


def test_labels_to_strings_paths():
    """The vectorised decode (one take / UTF-32 decode / split) against the per-symbol loop: ragged lengths incl. empty lines and
    padding rows (length -1), a row selection in another order, symbols starting at a column offset, multi-code-point entries
    (loop path) and a character set that contains NUL (slice path)."""
    from pero_ocr_amd.ocr_engine.pytorch_ocr_engine import labels_to_strings
    rng = np.random.RandomState(5)
    chars = [chr(0x100 + i) for i in range(40)]
    labels = rng.randint(0, 40, size=(30, 17)).astype(np.int32)
    lens = rng.randint(0, 18, size=30).astype(np.int32)
    lens[3] = 0
    want = ["".join(chars[c] for c in labels[i, :lens[i]]) for i in range(30)]
    assert labels_to_strings(labels, lens, chars) == want
    rows = np.array([7, 0, 29, 3], np.int64)
    assert labels_to_strings(labels, lens, chars, rows=rows) == [want[i] for i in rows]
    padded = np.full((32, 20), -1, np.int32)
    padded[:30, 3:] = labels
    plen = np.concatenate([lens, [-1, -1]]).astype(np.int32)
    assert labels_to_strings(padded, plen, chars, col0=3) == want + ["", ""]
    assert labels_to_strings(padded[:, ::1][:30], lens, chars, col0=3) == want
    multi = chars[:-1] + ["ch"]                                         # an entry of two code points: the loop path
    assert labels_to_strings(labels, lens, multi) == ["".join(multi[c] for c in labels[i, :lens[i]]) for i in range(30)]
    nul = ["\\x00"] + chars[1:]                                          # NUL in the set: no split on it
    assert labels_to_strings(labels, lens, nul) == ["".join(nul[c] for c in labels[i, :lens[i]]) for i in range(30)]
    assert labels_to_strings(labels[:0], lens[:0], chars) == []
    # rows of zero columns, lengths beyond the row (clipped, never read from the next row), all lines empty,
    # rows that already hold code points (sharding._CodePoints) incl. a NUL among them
    from pero_ocr_amd.sharding import _CodePoints
    assert labels_to_strings(np.zeros((3, 0), np.int32), np.array([0, 2, 0], np.int32), chars) == ["", "", ""]
    assert labels_to_strings(labels, np.zeros(30, np.int32), chars) == [""] * 30
    over = lens.copy(); over[5] = 40; over[29] = 1000
    want_over = ["".join(chars[c] for c in labels[i, :min(over[i], 17)]) for i in range(30)]
    assert labels_to_strings(labels, over, chars) == want_over
    assert labels_to_strings(labels, over, multi) == ["".join(multi[c] for c in labels[i, :min(over[i], 17)]) for i in range(30)]
    cp = (labels + 0x100).astype(np.int32)
    assert labels_to_strings(cp, lens, _CodePoints()) == want
    cp0 = cp.copy(); cp0[2, 1] = 0; cp0[7, 0] = 0
    l0 = lens.copy(); l0[2] = max(l0[2], 2); l0[7] = max(l0[7], 1)                 # (both NULs inside their lines)
    want0 = ["".join(chr(c) for c in cp0[i, :l0[i]]) for i in range(30)]
    assert "\x00" in want0[2] and want0[7].startswith("\x00")
    assert labels_to_strings(cp0, l0, _CodePoints()) == want0
    assert labels_to_strings(np.zeros((2, 0), np.int32), np.array([1, 0], np.int32), _CodePoints()) == ["", ""]
