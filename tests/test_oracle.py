"""CPU tests: the oracle against the golden vectors captured from the imported
reference (tests/golden, written by oracle/gen_golden.py) and against the CTC
known-answer cases of the reference's own tests
(test/test_decoding/test_decoders.py:24-96, semantics: blank last, collapse repeats,
drop blanks)."""
import numpy as np
import pytest

from oracle import engine_oracle, model_oracle
from pero_ocr_amd import netspec


def _logits_from_best(best, C):
    """[T] class ids -> [1, C, T] logits whose argmax is `best`."""
    T = len(best)
    x = np.full((1, C, T), -5.0, dtype=np.float32)
    for t, c in enumerate(best):
        x[0, c, t] = 5.0
    return x


@pytest.mark.parametrize("best,expected", [
    ([0, 3, 3], [0]),                 # 'a' then blanks                (test_decoders.py:24-32)
    ([3, 3, 3], []),                  # blank only -> ''               (:34-42)
    ([0, 3, 0], [0, 0]),              # 'aa' needs a separating blank  (:54-63)
    ([0, 1, 3], [0, 1]),              # 'ab'                           (:65-73)
    ([0, 0, 3], [0]),                 # continued symbol collapses     (:75-96)
    ([0, 0, 1, 1, 3, 1], [0, 1, 1]),
])
def test_ctc_known_answers(best, expected):
    _b, labels = engine_oracle.greedy_ctc(_logits_from_best(best, 4))
    assert labels[0].tolist() == expected


def test_ctc_matches_torch_semantics_ties_and_nan():
    import torch
    rng = np.random.RandomState(0)
    x = rng.randint(-2, 3, size=(5, 7, 33)).astype(np.float32)      # many exact ties
    x[1, 2, 5] = np.nan
    x[1, 4, 5] = np.nan
    x[2, 6, 0] = np.inf
    best = engine_oracle.frame_argmax(x)
    assert np.array_equal(best, torch.argmax(torch.from_numpy(x), 1).numpy())


def test_chunk_plan_c1(golden):
    g = golden("c1")
    plan = engine_oracle.chunk_plan(g.widths, 480 * g.batch_size)
    assert [[ids, mw] for ids, mw in plan] == g.plan
    assert [len(ids) for ids, _ in plan] == [15, 15, 2]


def test_chunk_plan_ragged_stable_ties(golden):
    g = golden("ragged")
    plan = engine_oracle.chunk_plan(g.widths, 480 * g.batch_size)
    assert [[ids, mw] for ids, mw in plan] == g.plan
    flat = [i for ids, _ in plan for i in ids]
    assert sorted(flat) == list(range(g.n))
    # equal widths keep input order (python's sort is stable)
    three_hundreds = [i for i in flat if g.widths[i] == 300]
    assert three_hundreds == sorted(three_hundreds)


def test_normalise_is_true_division():
    b = np.arange(256, dtype=np.uint8).reshape(1, 1, 256, 1).repeat(3, axis=3)
    x = engine_oracle.normalise(b)
    assert x.shape == (1, 3, 1, 256)
    import torch
    ref = (torch.from_numpy(b).float() / 255.0).permute(0, 3, 1, 2).numpy()
    assert np.array_equal(x, ref)
    # x * (1/255) is NOT bit-identical -> the HIP kernel uses a table of i/255.0f
    assert not np.array_equal(x, (b.astype(np.float32) * np.float32(1.0 / 255.0)).transpose(0, 3, 1, 2))


@pytest.mark.parametrize("name", ["c1", "ragged", "embed", "embed_mean"])
def test_oracle_reproduces_reference_golden(golden, name):
    """Full restated path (oracle net + numpy engine) vs the imported-reference outputs."""
    g = golden(name)
    spec, weights, crops = g.spec(), g.weights(), g.crops()
    net = model_oracle.OracleNet(spec, weights)
    texts, logits, coords, extras = engine_oracle.process_lines(
        lambda b: model_oracle.forward_logits(net, b, g.meta.get("resolved_embed_id")), crops, g.characters, spec.height,
        480 * g.batch_size, sparse_logits=False)
    assert texts == g.transcriptions
    assert coords == g.logit_coords
    for i in range(g.n):
        assert np.array_equal(extras["frame_argmax"][i], g.argmax(i)), f"line {i}"
        rows = g.rows(i)
        got = np.asarray(logits[i])[g.sample_rows[i]]
        # same torch build -> normally bit-identical; 1e-4 leaves room for a different host CPU
        assert np.max(np.abs(got - rows)) < 1e-4
        assert list(np.asarray(logits[i]).shape) == g.arrays["shapes"][i].tolist()


def test_oracle_sparse_logits_match_reference(golden):
    g = golden("c1")
    spec, weights, crops = g.spec(), g.weights(), g.crops()
    net = model_oracle.OracleNet(spec, weights)
    _t, logits, _c, _e = engine_oracle.process_lines(
        lambda b: model_oracle.forward_logits(net, b), crops, g.characters, spec.height, 480 * g.batch_size)
    for i in range(g.n):
        dense_ref = g.arrays[f"dense_{i}"]
        s = logits[i]
        assert s.shape == dense_ref.shape and s.dtype == np.float32
        assert abs(int(s.nnz) - g.nnz_sparse[i]) <= 2
        ref = np.zeros_like(dense_ref)
        ip, ix, dv = g.arrays[f"csc_indptr_{i}"], g.arrays[f"csc_indices_{i}"], g.arrays[f"csc_data_{i}"]
        for c in range(dense_ref.shape[1]):
            ref[ix[ip[c]:ip[c + 1]], c] = dv[ip[c]:ip[c + 1]]
        diff = np.abs(s.toarray() - ref)
        # entries whose probability sits at the 1e-4 threshold may flip in/out
        assert np.mean(diff > 1e-4) < 1e-3


def test_weight_generator_is_stable():
    spec = netspec.NetSpec(num_classes=100)
    w = netspec.generate_weights(spec, 20260928)
    flat = netspec.pack_weights(spec, w)
    assert flat.size == netspec.num_weight_floats(spec)
    import zlib
    # pinned: any change of the generator invalidates tests/golden
    assert zlib.crc32(flat[:100000].tobytes()) == zlib.crc32(
        netspec.pack_weights(spec, netspec.generate_weights(spec, 20260928))[:100000].tobytes())
    assert abs(float(w["conv1.weight"].std()) - (2.0 / 27) ** 0.5) < 0.02
    back = netspec.unpack_weights(spec, flat)
    assert all(np.array_equal(back[k], w[k]) for k in w)


def test_blob_roundtrip(tmp_path):
    spec = netspec.NetSpec(num_classes=12, conv_out=32, lstm_hidden=16, lstm_layers=1)
    w = netspec.generate_weights(spec, 5)
    p = str(tmp_path / "m.pocrw")
    netspec.save_blob(p, spec, w)
    spec2, w2 = netspec.load_blob(p)
    assert spec2 == spec and all(np.array_equal(w[k], w2[k]) for k in w)


def test_line_confidence_matches_reference_functions(golden):
    """oracle.line_confidence vs the values the reference's own compute_line_confidence produced
    (oracle/gen_golden_conf.py) on the c1 sparse logits and on a hand-made case."""
    import json
    import os
    from scipy import sparse
    from conftest import GOLDEN_DIR
    g = golden("c1")
    ref = json.load(open(os.path.join(GOLDEN_DIR, "c1_confidence.json"), encoding="utf8"))
    C = len(g.characters)
    for i in range(g.n):
        T = int(g.arrays["shapes"][i][0])
        m = sparse.csc_matrix((g.arrays[f"csc_data_{i}"], g.arrays[f"csc_indices_{i}"], g.arrays[f"csc_indptr_{i}"]), shape=(T, C))
        assert engine_oracle.line_confidence(m) == ref["confidence"][i]
    hand = sparse.csc_matrix(np.array(ref["hand_logits"], dtype=np.float32))
    assert engine_oracle.line_confidence(hand) == ref["hand_confidence"]
