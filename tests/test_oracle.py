"""The oracle (oracle/pf_oracle.py) against outputs of the reference itself.

Goldens were produced by oracle/gen_golden.py, which imports /root/reference on
CPU in the build container; they are data (inputs + expected outputs) only.
"""
import os

import numpy as np
import pytest

from oracle import pf_oracle as O
from phyloformer_amd.fasta import load_alignment

# fp32 re-association noise of the reference itself is ~1e-6 relative on the
# residual stream (|x| up to ~130) — see BASELINE.md §2 (fp32 vs fp64: 7e-7..1.9e-6)
TOL_DIST = 2e-5


def test_taps_tiny_every_sub_block(golden, weights):
    g = golden("taps_tiny.npz")
    taps = {}
    d = O.forward(weights("pf").tensors, g["idx"], tap=lambda k, v: taps.__setitem__(k, v))
    assert np.array_equal(taps["embed"], g["embed"])  # table lookup + one add: bit-exact
    for b in range(6):
        for sub in ("row", "col", "ffn"):
            k = f"block{b}.{sub}"
            scale = np.abs(g[k]).max()
            assert np.abs(taps[k] - g[k]).max() <= 1e-5 * scale, k
    assert np.abs(taps["logits"] - g["logits"]).max() <= 2e-4   # logits reach 23 here
    assert np.abs(d - g["dist"]).max() <= 5e-5                  # distances reach 12 here


def test_batch_and_squeeze_semantics(golden, weights):
    g = golden("batch_small.npz")
    d = O.forward_batch(weights("pf").tensors, g["idx"])
    assert d.shape == g["dist"].shape == (2, 15)
    assert np.abs(d - g["dist"]).max() <= TOL_DIST
    d2 = O.forward(weights("pf").tensors, g["idx_n2"])
    assert g["dist_n2"].shape == ()          # torch.squeeze → 0-dim for N == 2 (model.py:185)
    assert abs(float(d2[0]) - float(g["dist_n2"])) <= TOL_DIST


def test_config_c2_synthetic(golden, weights):
    g = golden("configs.npz")
    for a, ref in zip(g["c2_idx"], g["c2_dist"]):
        d = O.forward(weights("pf").tensors, a)
        assert np.abs(d - ref).max() <= TOL_DIST


@pytest.mark.parametrize("ckpt,stem", [("pf", "0_20_tips"), ("pf", "1_30_tips"),
                                       ("pf_base", "0_20_tips"), ("pf_indel", "2_20_tips"),
                                       ("pf_cherry", "3_20_tips"), ("pf_selreg", "4_20_tips")])
def test_e2e_reference_msas(golden, weights, repo, ckpt, stem):
    g = golden("e2e_testdata.npz")
    idx, _ids = load_alignment(os.path.join(repo, "data", "testdata", "msas", f"{stem}.fa"))
    d = O.forward(weights(ckpt).tensors, idx)
    assert np.abs(d - g[f"{ckpt}/{stem}"]).max() <= TOL_DIST


def test_survey_spot_values(golden):
    # SURVEY.md §4 item 1, measured during the survey with the reference
    g = golden("e2e_testdata.npz")
    d = g["pf/0_20_tips"]
    assert np.allclose(d[:4], [0.3306444, 0.3026287, 0.3003316, 0.2963659], atol=5e-7)
    assert abs(d.sum() - 113.761261) < 1e-3
    assert abs(g["pf_selreg/0_20_tips"].sum() - 18.061365) < 1e-3


def test_site_shards_match_unsharded(golden, weights):
    g = golden("configs.npz")
    a = g["c2_idx"][0]
    d1 = O.forward(weights("pf").tensors, a)
    for shards in (2, 8, 7):
        ds = O.forward(weights("pf").tensors, a, shards=shards)
        assert np.abs(ds - d1).max() <= 1e-5
    assert np.abs(O.forward(weights("pf").tensors, a, shards=8) - g["c2_dist"][0]).max() <= TOL_DIST


def test_collapsed_algebra_matches_reference_form(weights):
    """The device formulation (V projection / out_proj pulled out of the sums) in fp32."""
    w = weights("pf").tensors
    rng = np.random.default_rng(0)
    x = rng.standard_normal((12, 40, 64)).astype(np.float32) * 3
    for pre_norm, pre, axis, count in (("row_norm", "row_attention.", 1, 40), ("col_norm", "col_attention.", 0, 12)):
        p = "attention_blocks.2."
        xn = O.layer_norm(x, w[p + pre_norm + ".weight"], w[p + pre_norm + ".bias"])
        q, s_q, s_k, s_kv = O.attention_stats(xn, w, p + pre, axis, 4)
        ref = O.attention_apply(q, s_q, s_k, s_kv, w, p + pre, count)
        q2, a, k_, z = O.collapsed_stats(xn, w, p + pre, axis)
        M = O.collapsed_mix(a, k_, z, w, p + pre, count, 4)
        M = np.expand_dims(M, axis)
        got = np.einsum("...h,...hc->...c", q2, np.broadcast_to(M, q2.shape + (64,))) + w[p + pre + "out_proj.bias"]
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_seq_cap_and_bad_input(weights):
    w = weights("pf").tensors
    with pytest.raises(ValueError, match="n_seqs must be smaller or equal to 200"):
        O.forward(w, np.zeros((201, 4), np.uint8))
    with pytest.raises(ValueError):
        O.forward(w, np.full((3, 4), 22, np.uint8))


def test_torch_port_matches_goldens(golden, weights, repo):
    """oracle/pf_oracle_torch.py is the `cpu_baseline` of bench.py: torch CPU ops in the reference's op order
    and layout.  It must reproduce the reference's own outputs (tests/golden, written by oracle/gen_golden.py
    from /root/reference) - otherwise the baseline number would describe some other computation."""
    import torch
    from oracle import pf_oracle_torch
    torch.set_num_threads(4)
    z = golden("configs.npz")
    for k in range(2):
        got = pf_oracle_torch.forward(weights("pf").tensors, z["c2_idx"][k])
        assert np.abs(got - z["c2_dist"][k]).max() <= 2e-6
    g = golden("e2e_testdata.npz")
    for ckpt, stem in (("pf_base", "0_20_tips"), ("pf_indel", "1_30_tips"), ("pf", "0_40_tips")):
        idx, _ids = load_alignment(os.path.join(repo, "data", "testdata", "msas", f"{stem}.fa"))
        got = pf_oracle_torch.forward(weights(ckpt).tensors, idx)
        assert np.abs(got - g[f"{ckpt}/{stem}"]).max() <= 2e-6
