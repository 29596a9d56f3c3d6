"""Sequence-to-sequence (transformer) engine, SURVEY.md section 8 row f-3.

CPU part: the oracle restatement against the fixtures written from the reference's own
TransformerEngineLineOCR run (oracle/gen_golden_s2s.py), and the host logic (batch plan, part
splitting, merging) against the oracle.  GPU part: the HIP engine through the reference-shaped
TransformerEngineLineOCR against the same fixtures.
"""
import json
import os
import random

import numpy as np
import pytest

from pero_ocr_amd import netspec
from pero_ocr_amd.ocr_engine import line_ocr_engine as host
from pero_ocr_amd.ocr_engine import transformer_ocr_engine as tengine
from oracle import s2s_oracle

from conftest import gpu_available

LOGIT_TOL = 1e-3            # north_star: logits within 1e-3 (fp32)
SAFE_MARGIN = 2e-4          # a reference top-2 margin below this may legitimately flip under fp32 re-association


# ------------------------------------------------------------------------------------------- CPU

def test_edit_distance_and_overlap_match_oracle():
    rnd = random.Random(5)
    for _ in range(400):
        a = "".join(rnd.choice("abcd") for _ in range(rnd.randint(0, 14)))
        b = "".join(rnd.choice("abcd") for _ in range(rnd.randint(0, 14)))
        assert host.levenshtein_distance(list(a), list(b)) == s2s_oracle.edit_distance(a, b)
        assert host.find_best_overlap(a, b) == s2s_oracle.best_overlap(a, b)          # native (pocr_best_overlap)
        assert host.find_best_overlap_py(a, b) == s2s_oracle.best_overlap(a, b)       # numpy
    long1 = "".join(rnd.choice("abcdefgh ") for _ in range(270))
    long2 = long1[-60:] + "".join(rnd.choice("abcdefgh ") for _ in range(200))
    assert host.find_best_overlap(long1, long2) == s2s_oracle.best_overlap(long1, long2) == 60
    assert host.levenshtein_distance("kitten", "sitting") == 3
    assert host.find_best_overlap("hello wor", "o world") == 5      # "o wor" == "o wor"


def test_merge_known_answers():
    lg = lambda t: np.arange(len(t) * 2, dtype=np.float32).reshape(len(t), 2)
    # overlap 5 ("o wor"): left keeps [:-3], right drops its first 2
    text, logits = host.merge_transcriptions_and_logits(["hello wor", "o world"], [lg("hello wor"), lg("o world")])
    assert text == "hello world" and logits.shape == (len(text), 2)
    # no overlap below CER 1 -> overlap 0 -> the reference's [: -0 // 2] = [:0] drops the whole left side
    text, logits = host.merge_transcriptions_and_logits(["abc", "xyz"], [lg("abc"), lg("xyz")])
    assert text == "xyz" and logits.shape == (3, 2)
    # logits of a single part are cut to the length of its transcription
    text, logits = host.merge_transcriptions_and_logits(["ab"], [np.zeros((7, 3), np.float32)])
    assert text == "ab" and logits.shape == (2, 3)
    for parts in (["abcab", "cabca", "bcabc"], ["", "abc"], ["abc", ""], ["aaaa", "aaaa"]):
        ls = [lg(p) for p in parts]
        t1, l1 = host.merge_transcriptions_and_logits(parts, ls)
        t2, l2 = s2s_oracle.merge_parts(parts, ls)
        assert t1 == t2 and np.array_equal(l1, l2)


def test_split_spans():
    assert tengine.split_spans(1024, 1024) == [(0, 1024)]
    assert tengine.split_spans(1025, 1024) == [(0, 1024), (768, 1025)]
    assert tengine.split_spans(2100, 1024) == [(0, 1024), (768, 1792), (1536, 2100)]
    assert tengine.split_spans(5, 1e10) == [(0, 5)]
    for w in (1, 700, 1024, 1025, 1792, 1793, 4000):
        assert tengine.split_spans(w, 1024) == s2s_oracle.split_line(w, 1024)


@pytest.mark.parametrize("name", ["s2s_ragged", "s2s_c32"])
def test_batch_plan_matches_reference_run(golden, name):
    g = golden(name)
    batches = tengine.plan_batches(g.widths, 480 * g.batch_size, g.max_line_width)
    assert [[b.line_ids, b.max_width, b.spans] for b in batches] == g.plan
    for b in batches:
        assert b.w_pad >= 1088 and b.w_pad % 4 == 0
        assert b.pad_left == 32 + (1088 - b.w_batch) // 2 if b.w_batch < 1088 else b.pad_left == 32


def check_against_golden(g, texts, logits, coords, steps=None):
    assert texts == g.transcriptions
    assert coords == g.logit_coords
    worst = 0.0
    for i in range(g.n):
        ref_rows, ref = g.arrays[f"rows_{i}"], g.arrays[f"dense_{i}"]
        got = np.asarray(logits[i].todense()) if hasattr(logits[i], "todense") else np.asarray(logits[i])
        assert got.shape == (len(g.transcriptions[i]), len(g.characters))
        if got.shape[0] == 0:
            continue
        assert np.array_equal(np.argmax(got, axis=1), g.arrays[f"argmax_{i}"].astype(np.int64))
        worst = max(worst, float(np.max(np.abs(got[ref_rows] - ref))))
        l2 = float(np.sqrt(np.sum(got.astype(np.float64) ** 2)))
        assert abs(l2 - float(g.arrays[f"l2_{i}"][0])) <= 1e-4 * max(1.0, l2)
    assert worst < LOGIT_TOL, worst
    return worst


def test_oracle_reproduces_reference_s2s_ragged(golden):
    g = golden("s2s_ragged")
    assert g.min_top2_margin > SAFE_MARGIN
    model = s2s_oracle.OracleS2S(g.spec(), g.weights())
    texts, logits, coords, extras = s2s_oracle.process_lines(model, g.crops(), g.characters, g.height,
                                                             480 * g.batch_size, g.max_line_width)
    check_against_golden(g, texts, logits, coords)
    assert extras["steps"] == g.steps
    lens = [len(t) for t in texts]
    assert min(lens) <= 8 and max(lens) > 272           # early finishers, limit hitters and merged over-long lines


def test_weight_table_s2s_roundtrip(tmp_path):
    spec = netspec.NetSpec(num_classes=13, arch=netspec.ARCH_S2S, dec_layers=1, sa_layers=1, sa_ff=64, conv_out=64, sa_heads=2)
    w = netspec.generate_weights(spec, 3)
    assert w["dec.out.bias"][-2] == np.float32(36.0) and w["dec.embed.weight"].shape == (13, 64)
    path = str(tmp_path / "m.pocrw")
    netspec.save_blob(path, spec, w)
    spec2, w2 = netspec.load_blob(path)
    assert spec2 == spec and all(np.array_equal(w[k], w2[k]) for k in w)


def test_export_transformer_state_dict():
    """tools/export_weights.py --transformer: the reference's TransformerOCR parameter names -> blob order."""
    import importlib.util
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("export_weights", os.path.join(here, "tools", "export_weights.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    spec = netspec.NetSpec(num_classes=13, arch=netspec.ARCH_S2S, dec_layers=2, sa_layers=1, sa_ff=64, conv_out=64, sa_heads=2)
    w = netspec.generate_weights(spec, 11)
    state = {}
    conv_slots = [0, 2, 5, 7, 10, 12, 14]                  # VGG16 .features indices of the first seven convs
    for i in range(1, 10):
        pre = (f"encoder_frontend.blocks_2d.blocks_2d.{conv_slots[i - 1]}." if i <= 7 else
               f"encoder_frontend.blocks_2d.blocks_2d.19.{(i - 8) * 2}.")
        state[pre + "weight"], state[pre + "bias"] = w[f"conv{i}.weight"], w[f"conv{i}.bias"]
    bn = "encoder_frontend.blocks_2d.blocks_2d.20."
    state[bn + "weight"], state[bn + "bias"], state[bn + "running_mean"], state[bn + "running_var"] = \
        w["bn.gamma"], w["bn.beta"], w["bn.mean"], w["bn.var"]
    state[bn + "num_batches_tracked"] = np.zeros((), np.int64)
    state["encoder_frontend.aggregation_conv.0.weight"], state["encoder_frontend.aggregation_conv.0.bias"] = w["agg.weight"], w["agg.bias"]
    state["encoder.input_norm.weight"], state["encoder.input_norm.bias"] = w["sa.norm.weight"], w["sa.norm.bias"]
    names = (("lin1", "linear1"), ("lin2", "linear2"), ("norm1", "norm1"), ("norm2", "norm2"), ("norm3", "norm3"))
    p = "encoder.trans_encoder.layers.0."
    state[p + "self_attn.in_proj_weight"], state[p + "self_attn.in_proj_bias"] = w["sa0.in_proj.weight"], w["sa0.in_proj.bias"]
    state[p + "self_attn.out_proj.weight"], state[p + "self_attn.out_proj.bias"] = w["sa0.out_proj.weight"], w["sa0.out_proj.bias"]
    for ours, theirs in names[:4]:
        state[p + theirs + ".weight"], state[p + theirs + ".bias"] = w[f"sa0.{ours}.weight"], w[f"sa0.{ours}.bias"]
    for l in range(2):
        p = f"trans_decoder.layers.{l}."
        for ours, theirs in (("self", "self_attn"), ("cross", "multihead_attn")):
            state[p + theirs + ".in_proj_weight"], state[p + theirs + ".in_proj_bias"] = w[f"dec{l}.{ours}.in_proj.weight"], w[f"dec{l}.{ours}.in_proj.bias"]
            state[p + theirs + ".out_proj.weight"], state[p + theirs + ".out_proj.bias"] = w[f"dec{l}.{ours}.out_proj.weight"], w[f"dec{l}.{ours}.out_proj.bias"]
        for ours, theirs in names:
            state[p + theirs + ".weight"], state[p + theirs + ".bias"] = w[f"dec{l}.{ours}.weight"], w[f"dec{l}.{ours}.bias"]
    state["dec_embeder.weight"] = w["dec.embed.weight"]
    state["dec_out_proj.weight"], state["dec_out_proj.bias"] = w["dec.out.weight"], w["dec.out.bias"]
    spec2, w2 = mod.transformer_state_to_weights(state, height=40, heads=2)
    assert spec2 == spec
    assert np.array_equal(netspec.pack_weights(spec2, w2), netspec.pack_weights(spec, w))


# ------------------------------------------------------------------------------------------- GPU

def make_engine(g, tmp_path, batch_size=None):
    cfg = {"line_px_height": g.height, "line_vertical_scale": 1.0, "checkpoint": "absent.pocrw",
           "characters": g.characters[:-2], "net_name": g.net_name, "max_line_width": g.max_line_width,
           "net": {"weight_seed": g.weight_seed, "boundary_bias": g.boundary_bias}}
    path = os.path.join(str(tmp_path), f"{g.name}.json")
    with open(path, "w", encoding="utf8") as f:
        json.dump(cfg, f)
    import torch
    return tengine.TransformerEngineLineOCR(path, torch.device("cuda:0"), batch_size=batch_size or g.batch_size)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["s2s_ragged", "s2s_c32"])
def test_gpu_s2s_golden(golden, tmp_path, name):
    assert gpu_available(), "GPU test on a box without a HIP device"
    g = golden(name)
    eng = make_engine(g, tmp_path)
    assert eng.characters == g.characters and eng.sentence_boundary_ind == len(g.characters) - 2
    crops = g.crops()
    texts, logits, coords = eng.process_lines([c.copy() for c in crops], sparse_logits=False)
    worst = check_against_golden(g, texts, logits, coords)
    print(f"[{name}] max |dlogit| vs reference rows = {worst:.2e}")
    t2, l2, c2 = eng.process_lines(crops)
    assert t2 == texts and c2 == coords
    for i in range(g.n):                                   # CSC built on the GPU == sparsified dense logits
        assert l2[i].format == "csc" and l2[i].dtype == np.float32 and l2[i].shape == np.asarray(logits[i]).shape
        assert abs(int(l2[i].nnz) - g.nnz_sparse[i]) <= max(2, g.nnz_sparse[i] // 200)
        d = np.asarray(l2[i].todense())
        keep = d != 0
        assert np.array_equal(d[keep], np.asarray(logits[i])[keep])
    t3, l3, c3 = eng.process_lines(crops, no_logits=True)
    assert t3 == texts and all(x is None for x in l3) and all(x is None for x in c3)
    with pytest.raises(AttributeError):
        eng.process_lines(crops[:2], tight_crop_logits=True)


@pytest.mark.gpu
def test_gpu_s2s_run_ocr_seam_and_launch_independence(golden, tmp_path):
    """run_ocr on a hand-assembled batch equals the oracle; recognising the batches of a page one by one
    gives bit-identical results to the merged launches process_lines uses."""
    g = golden("s2s_ragged")
    eng = make_engine(g, tmp_path)
    crops = g.crops()
    model = s2s_oracle.OracleS2S(g.spec(), g.weights())
    ids = [1, 5, 7]                                       # 17, 1 and 96 px wide
    batch = np.zeros((len(ids), g.height, 96 + 64, 3), np.uint8)
    for row, i in zip(batch, ids):
        row[:, 32:32 + crops[i].shape[1]] = crops[i]
    t_ref, l_ref = model.run_ocr(batch, g.characters)
    t_gpu, l_gpu = eng.run_ocr(batch)
    assert t_gpu == t_ref and l_gpu.shape == l_ref.shape
    keep = max(len(t) for t in t_ref) + 1
    assert float(np.max(np.abs(l_gpu[:, :keep] - l_ref[:, :keep]))) < LOGIT_TOL
    # per-batch launches vs merged launches
    merged = eng.process_lines(crops, sparse_logits=False)
    old = tengine.LAUNCH_MAX_LINES
    try:
        tengine.LAUNCH_MAX_LINES = 1
        single = eng.process_lines(crops, sparse_logits=False)
    finally:
        tengine.LAUNCH_MAX_LINES = old
    assert merged[0] == single[0] and merged[2] == single[2]
    for a, b in zip(merged[1], single[1]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_s2s_shard_adapter_and_degenerate_inputs(golden, tmp_path):
    g = golden("s2s_ragged")
    eng = make_engine(g, tmp_path)
    crops = g.crops()
    from pero_ocr_amd import sharding
    rec = sharding.seq2seq_recogniser(eng)
    batches = tengine.plan_batches(g.widths, eng.max_input_horizontal_pixels, eng.max_line_width)
    mine = sharding.assign_by_cost([len(b.parts) * b.w_pad for b in batches], 2)
    got = {}
    for share in mine:                                    # the two ranks' shares, run one after the other
        got.update(rec(crops, [batches[i] for i in share]))
    assert [got[i] for i in range(g.n)] == g.transcriptions
    # degenerate inputs
    assert eng.process_lines([]) == ([], [], [])
    with pytest.raises(ValueError):
        eng.process_lines([np.zeros((g.height + 1, 20, 3), np.uint8)])
    with pytest.raises(ZeroDivisionError):               # the reference divides by ceil32(0) (line_ocr_engine.py:87)
        eng.process_lines([np.zeros((g.height, 0, 3), np.uint8)])
    # a failed crop (float64 zeros, page_parser.py:390-391) behaves like its uint8 cast
    a = eng.process_lines([np.zeros((g.height, g.height, 3))], sparse_logits=False)
    b = eng.process_lines([np.zeros((g.height, g.height, 3), np.uint8)], sparse_logits=False)
    assert a[0] == b[0] and np.array_equal(a[1][0], b[1][0])
    # the CTC entry points refuse a sequence-to-sequence engine
    with pytest.raises(RuntimeError):
        eng.net.run_batch(np.zeros((1, g.height, 64, 3), np.uint8))


def test_batch_plan_properties_random():
    """plan_batches vs the oracle's restatement on random width sets, plus structural invariants."""
    rnd = random.Random(11)
    for _ in range(200):
        n = rnd.randint(1, 40)
        widths = [rnd.choice([1, 5, 31, 32, 33, 300, 640, 1023, 1024, 1025, 1500, 2049, 3000, 5000]) if rnd.random() < 0.5
                  else rnd.randint(1, 2600) for _ in range(n)]
        bs = rnd.choice([1, 2, 4, 8, 35])
        mlw = rnd.choice([512, 1024, 1e10])
        ours = tengine.plan_batches(widths, 480 * bs, mlw)
        ref = s2s_oracle.plan_batches(widths, 480 * bs, mlw)
        assert [(b.line_ids, b.max_width) for b in ours] == [(list(ids), mw) for ids, mw in ref]
        seen = sorted(i for b in ours for i in b.line_ids)
        assert seen == list(range(n))                                   # every line exactly once
        for b in ours:
            assert b.w_pad >= tengine.MIN_INPUT_WIDTH and b.w_batch <= 480 * bs
            assert sum(b.spans) == len(b.parts)
            for (i, a, e_), in zip(b.parts):
                assert 0 <= a < e_ <= widths[i] and e_ - a <= max(mlw, 1)
            for i, span in zip(b.line_ids, b.spans):
                assert [p[1:] for p in b.parts if p[0] == i][:span] == s2s_oracle.split_line(widths[i], mlw)
        groups = tengine.plan_launches(ours)
        assert [b for g_ in groups for b in g_] == ours                  # launches keep the batch order


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(conv_out=256, sa_heads=4, sa_ff=512, sa_layers=1, dec_layers=1),      # LayerNorm prologue instance for E = 256
    dict(conv_out=512, sa_heads=16, sa_ff=1024, sa_layers=1, dec_layers=2),    # head dim 32
    dict(conv_out=384, sa_heads=3, sa_ff=768, sa_layers=1, dec_layers=1),      # head dim 128, E without a fused-LN instance
    dict(conv_out=512, sa_heads=8, sa_ff=2048, sa_layers=1, dec_layers=1, height=32),
])
def test_gpu_s2s_other_geometries_against_oracle(kw, tmp_path):
    """Other widths / head dims / heights than the fixtures use: HIP engine vs the (reference-pinned) oracle."""
    chars = [chr(0x61 + i) for i in range(20)]
    height = kw.pop("height", 40)
    spec = netspec.NetSpec(num_classes=len(chars) + 2, arch=netspec.ARCH_S2S, height=height, **kw)
    weights = netspec.generate_weights(spec, 99, boundary_bias=30.0)
    path = os.path.join(str(tmp_path), "m.pocrw")
    netspec.save_blob(path, spec, weights)
    cfg = {"line_px_height": height, "line_vertical_scale": 1.0, "checkpoint": path, "characters": chars, "max_line_width": 1024,
           "net_name": {"dim_model": spec.conv_out, "dim_ff": spec.sa_ff, "heads": spec.sa_heads, "encoder_layers": spec.sa_layers,
                        "decoder_layers": spec.dec_layers, "conv_subsampling": [8, 4]}}
    jpath = os.path.join(str(tmp_path), "e.json")
    with open(jpath, "w", encoding="utf8") as f:
        json.dump(cfg, f)
    import torch
    from pero_ocr_amd import synth
    eng = tengine.TransformerEngineLineOCR(jpath, torch.device("cuda:0"), batch_size=4)
    crops = synth.make_crops(31, [200, 64, 333, 500, 90], height)
    model = s2s_oracle.OracleS2S(spec, weights)
    want_t, want_l, want_c, _ = s2s_oracle.process_lines(model, crops, eng.characters, height, 480 * 4, 1024)
    got_t, got_l, got_c = eng.process_lines(crops, sparse_logits=False)
    assert got_t == want_t and got_c == want_c
    worst = max([float(np.max(np.abs(a - b))) for a, b in zip(got_l, want_l) if a.size] or [0.0])
    assert worst < LOGIT_TOL, worst


@pytest.mark.gpu
def test_gpu_s2s_bench_configuration_against_oracle(tmp_path):
    """The configuration tools/s2s_bench.py measures (E 512, 8 heads, 2 encoder / 3 decoder layers, 233 classes) at a boundary bias
    where random weights run most lines into the 272-step limit: strings and logits of the HIP engine against the oracle over
    the whole decoding loop."""
    import torch
    from pero_ocr_amd import synth
    chars = synth.make_charset(231)
    net = {"dim_model": 512, "dim_ff": 2048, "heads": 8, "encoder_layers": 2, "decoder_layers": 3, "conv_subsampling": [8, 4]}
    path = os.path.join(str(tmp_path), "ocr.json")
    with open(path, "w", encoding="utf8") as f:
        json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": "absent", "characters": chars, "net_name": net,
                   "max_line_width": 1024, "net": {"weight_seed": 20261002, "boundary_bias": 18.0}}, f)
    eng = tengine.TransformerEngineLineOCR(path, torch.device("cuda:0"), batch_size=4)
    crops = synth.make_crops(602, [512] * 4, 40)
    got_t, got_l, _ = eng.process_lines(crops, sparse_logits=False)
    spec = eng.net_spec
    model = s2s_oracle.OracleS2S(spec, netspec.generate_weights(spec, 20261002, boundary_bias=18.0))
    want_t, want_l, _c, _ = s2s_oracle.process_lines(model, crops, eng.characters, 40, 480 * 4, 1024)
    assert got_t == want_t and max(len(t) for t in got_t) >= 272
    worst = max(float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) for a, b in zip(got_l, want_l))
    assert worst < LOGIT_TOL, worst


@pytest.mark.gpu
def test_gpu_s2s_range_guard_reruns_on_bf16x3(tmp_path):
    """The sequence-to-sequence engine's encoder runs in the f16x2 arithmetic: fp32's precision, f16's range; the reference
    computes in plain fp32 (pero_ocr/ocr_engine/transformer_ocr_engine.py:32-47).  A network whose activations leave f16's range
    (conv4 x 2^17, conv5 x 2^-17: the same function) must decode like the oracle on the rescaled weights with no action by the
    caller: pocr_s2s_decode runs encoder and decoding loop again on the bf16x3 engine and the slot's reads go there - dense,
    sparse and text-only calls; a network in range never takes the fall-back."""
    from pero_ocr_amd import _native, synth
    if _native.conv_split() != 2:
        pytest.skip("the range guard belongs to the f16x2 arithmetic")
    import torch
    chars = [chr(0x61 + i) for i in range(20)]
    spec = netspec.NetSpec(num_classes=len(chars) + 2, arch=netspec.ARCH_S2S, conv_out=256, sa_heads=4, sa_ff=512, sa_layers=1, dec_layers=1)
    base = netspec.generate_weights(spec, 99, boundary_bias=30.0)
    w = dict(base)
    w["conv4.weight"] = base["conv4.weight"] * np.float32(2.0 ** 17)
    w["conv4.bias"] = base["conv4.bias"] * np.float32(2.0 ** 17)
    w["conv5.weight"] = base["conv5.weight"] * np.float32(2.0 ** -17)
    crops = synth.make_crops(31, [200, 64, 333, 500, 90, 310, 40], 40)
    net_cfg = {"dim_model": spec.conv_out, "dim_ff": spec.sa_ff, "heads": spec.sa_heads, "encoder_layers": spec.sa_layers,
               "decoder_layers": spec.dec_layers, "conv_subsampling": [8, 4]}
    fallbacks = {}
    for name, weights in (("rescaled", w), ("in range", base)):
        path = os.path.join(str(tmp_path), f"{name[:2]}.pocrw")
        netspec.save_blob(path, spec, weights)
        jpath = os.path.join(str(tmp_path), f"{name[:2]}.json")
        with open(jpath, "w", encoding="utf8") as f:
            json.dump({"line_px_height": 40, "line_vertical_scale": 1.0, "checkpoint": path, "characters": chars, "max_line_width": 1024,
                       "net_name": net_cfg}, f)
        eng = tengine.TransformerEngineLineOCR(jpath, torch.device("cuda:0"), batch_size=4)
        model = s2s_oracle.OracleS2S(spec, weights)
        want_t, want_l, want_c, _ = s2s_oracle.process_lines(model, crops, eng.characters, 40, 480 * 4, 1024)
        got_t, got_l, got_c = eng.process_lines(crops, sparse_logits=False)
        assert got_t == want_t and got_c == want_c, name
        worst = max([float(np.max(np.abs(a - b))) for a, b in zip(got_l, want_l) if a.size] or [0.0])
        print(f"[s2s range guard, {name}] max |dlogit| vs the oracle {worst:.3e}, fall-backs {eng.net.range_fallbacks()}")
        assert worst < LOGIT_TOL, (name, worst)
        t2, l2, _c2 = eng.process_lines(crops)                      # sparse (CSC built on the device that decoded)
        assert t2 == want_t
        for a, b in zip(l2, got_l):
            d = np.asarray(a.todense()); keep = d != 0
            assert np.array_equal(d[keep], np.asarray(b)[keep])
        t3, _l3, _c3 = eng.process_lines(crops, no_logits=True)
        assert t3 == want_t
        fallbacks[name] = eng.net.range_fallbacks()
    assert fallbacks["rescaled"] >= 3 and fallbacks["in range"] == 0, fallbacks
