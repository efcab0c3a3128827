"""Pin the numpy oracle against fixtures produced by the imported reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import tt_oracle as O

TOL = 1e-5      # CPU restatement vs reference fp32


def test_encoder_layers(golden):
    z, sd = golden
    x = z["inputs"]
    for i in range(O.n_layers(sd, "encoder.")):
        x, _ = O.layer_fwd(x, O.layer_params(sd, "encoder.", i), None)
        assert rel_err(x, z["enc_layer%d" % i]) < TOL


def test_label_encoder_masked_and_unmasked(golden):
    z, sd = golden
    tg = np.concatenate([np.zeros((2, 1), dtype=np.int64), z["targets"]], 1)
    y, _ = O.decoder_fwd(tg, sd, O.look_ahead_mask(tg.shape[1])[:, :, None])
    assert rel_err(y, z["dec_masked"]) < TOL
    y, _ = O.decoder_fwd(tg, sd, None)
    assert rel_err(y, z["dec_unmasked"]) < TOL


@pytest.mark.parametrize("tag", ["full", "ragged"])
def test_logits_loss_and_all_grads(golden, tag):
    z, sd = golden
    r = O.transducer_loss_and_grads(z["inputs"], z["targets"], z[tag + "/act_lens"], z[tag + "/label_lens"], sd)
    assert rel_err(r["logits"], z["logits"]) < TOL
    assert rel_err(r["costs"], z[tag + "/costs"]) < TOL
    assert abs(r["loss"] - z[tag + "/loss"]) / abs(z[tag + "/loss"]) < TOL
    assert rel_err(r["dinputs"], z[tag + "/dinputs"]) < 5 * TOL
    keys = [k for k in z.files if k.startswith(tag + "/grad/")]
    assert len(keys) == len(r["grads"])
    for k in keys:
        name = k[len(tag) + 6:]
        assert rel_err(r["grads"][name], z[k]) < 5 * TOL, name


@pytest.mark.parametrize("mname", ["band_10_2", "left_8_0", "chunk_8_16"])
def test_encoder_streaming_masks(golden, mname):
    z, sd = golden
    mask = z["mask/%s/mask" % mname]
    if mname == "band_10_2":
        assert np.array_equal(mask != 0, O.context_mask(mask.shape[0], 10, 2))
    if mname == "chunk_8_16":
        assert np.array_equal(mask != 0, O.chunk_mask(mask.shape[0], 8, 16))
    y, caches = O.encoder_fwd(z["inputs"], sd, mask[:, :, None])
    assert rel_err(y, z["mask/%s/enc_out" % mname]) < TOL
    grads = {}
    dx = O.stack_bwd(z["enc_cotangent"], caches, sd, "encoder.", grads)
    assert rel_err(dx, z["mask/%s/dinputs" % mname]) < 5 * TOL
    for k in z.files:
        if k.startswith("mask/%s/grad/" % mname):
            name = k.split("/grad/")[1]
            assert rel_err(grads[name], z[k]) < 5 * TOL, name


def test_greedy_decode_tokens(golden):
    z, sd = golden
    hyp = O.recognize(z["inputs"], z["greedy/lens"], sd)
    for b, h in enumerate(hyp):
        assert h == z["greedy/tokens%d" % b].tolist()


def test_float64_oracle_close_to_float32(golden):
    z, sd = golden
    sd64 = {k: v.astype(np.float64) if v.dtype == np.float32 else v for k, v in sd.items()}
    r32 = O.transducer_loss_and_grads(z["inputs"], z["targets"], z["full/act_lens"], z["full/label_lens"], sd)
    r64 = O.transducer_loss_and_grads(z["inputs"].astype(np.float64), z["targets"], z["full/act_lens"],
                                      z["full/label_lens"], sd64)
    assert abs(r32["loss"] - r64["loss"]) / abs(r64["loss"]) < 1e-6
    for k in r64["grads"]:
        assert rel_err(r32["grads"][k], r64["grads"][k]) < 2e-5, k


def test_flat_view_shift_equals_index_form():
    rng = np.random.default_rng(0)
    for L in (1, 2, 5, 17):
        G = rng.normal(size=(2, 3, L, L))
        row, col, valid = O.rel_shift_index(L)
        assert np.array_equal(O.rel_shift_flat(G), G[:, :, row, col] * valid)
        dBD = rng.normal(size=(2, 3, L, L))
        dG = np.zeros_like(G)
        np.add.at(dG, (slice(None), slice(None), row, col), dBD * valid)
        assert np.allclose(O.rel_shift_flat_bwd(dBD), dG, atol=0, rtol=0)


def test_sparse_greedy_tokens(golden, request):
    """the reference's recognize() on the same nets with the blank logit raised (tools/gen_golden_r2.py): 5-20 % of the frames emit, so the
    blank branch of tt/model.py:76-83 carries most frames and the label histories outgrow both table lengths"""
    import os
    from conftest import GOLDEN
    z, sd = golden
    name = request.node.callspec.params["golden"]
    g = np.load(os.path.join(GOLDEN, "greedy_sparse.npz"))
    sd = dict(sd)
    sd["joint.project_layer.bias"] = sd["joint.project_layer.bias"].copy()
    sd["joint.project_layer.bias"][0] += float(g[name + "/blank_bias"])
    hyp = O.recognize(g[name + "/inputs"], g[name + "/lens"], sd)
    for b, h in enumerate(hyp):
        assert h == g["%s/tokens%d" % (name, b)].tolist()
    assert 0 < sum(len(h) for h in hyp) < 0.25 * int(g[name + "/lens"].sum())


def test_beam_search_tokens(golden, request):
    """the reference's recognize_beam_search (tt/model.py:110-198, width 5) on the same nets (tests/golden/greedy_sparse.npz)"""
    import os
    from conftest import GOLDEN
    z, sd = golden
    name = request.node.callspec.params["golden"]
    g = np.load(os.path.join(GOLDEN, "greedy_sparse.npz"))
    sd = dict(sd)
    sd["joint.project_layer.bias"] = sd["joint.project_layer.bias"].copy()
    sd["joint.project_layer.bias"][0] += float(g[name + "/blank_bias"])
    hyp = O.recognize_beam_search(g[name + "/inputs"][:, :48], g[name + "/beam_lens"], sd)
    for b, h in enumerate(hyp):
        assert h == g["%s/beam_tokens%d" % (name, b)].tolist(), b
