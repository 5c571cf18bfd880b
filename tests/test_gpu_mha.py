"""-m gpu: softmax MultiHeadAttention kernels (csrc/pf_mha.hip.h) against the reference class
(goldens) and the oracle.  Tolerance: 5e-5 absolute on outputs of magnitude ~1 — every contraction
is split-fp16 x3 (two fp16 limbs per operand, 2^-22 relative; rounds 1-5: bf16 limbs, with three terms /
six passes on the Q-K path) with fp32 softmax; the reference's own fp32 evaluation differs from fp64 by
up to 1e-5 on the peaked case."""
import numpy as np
import pytest

from oracle.mha_oracle import mha_forward

pytestmark = pytest.mark.gpu
TOL = 5e-5


def case(golden, name):
    z = golden("mha.npz")
    return {k.split("/", 1)[1]: z[k] for k in z.keys() if k.startswith(name + "/")}


@pytest.fixture(scope="module")
def mha_cls():
    from phyloformer_amd.attention import MultiHeadAttention
    return MultiHeadAttention


@pytest.mark.parametrize("name", ["small", "ragged", "peaky", "one_key"])
def test_mha_matches_reference_class(golden, mha_cls, name):
    g = case(golden, name)
    m = mha_cls(4, 64).load_state_dict(g)
    y = m(g["x"])
    assert y.shape == g["y"].shape and y.dtype == np.float32
    err = np.abs(y - g["y"]).max()
    print(name, "max abs err", err)
    assert err <= TOL
    m.close()


@pytest.mark.parametrize("shape,gain", [((1, 8, 500), 2.0), ((2, 3, 129), 4.0), ((1, 2, 1770), 1.0),
                                        ((1, 5, 31), 6.0), ((3, 1, 32), 1.0)])
def test_mha_matches_oracle_on_random_shapes(golden, mha_cls, shape, gain):
    rng = np.random.default_rng(sum(shape))
    g = case(golden, "small")
    w = {k: v.copy() for k, v in g.items() if k not in ("x", "y")}
    w["q_proj.weight"] *= gain
    w["k_proj.weight"] *= gain
    x = (rng.standard_normal(shape + (64,)) * 1.5).astype(np.float32)
    m = mha_cls(4, 64).load_state_dict(w)
    y = m(x)
    ref = mha_forward(w, x)
    err = np.abs(y - ref).max()
    print(shape, "max abs err", err, "max|y|", np.abs(ref).max())
    assert err <= TOL * max(1.0, np.abs(ref).max())
    # rows are independent: a second call on a permutation of the rows gives the permuted result
    perm = rng.permutation(shape[1])
    y2 = m(x[:, perm])
    assert np.array_equal(y2, y[:, perm])
    m.close()


def test_mha_error_behaviour(mha_cls, golden):
    with pytest.raises(ValueError):
        mha_cls(3, 64)                                   # attention.py:27-31
    g = case(golden, "small")
    m = mha_cls(4, 64)
    with pytest.raises(RuntimeError):
        m(g["x"])                                        # no weights yet
    m.load_state_dict(g)
    with pytest.raises(ValueError):
        m(g["x"][..., :32])
    m.close()
    with pytest.raises(ValueError):
        mha_cls(2, 64).load_state_dict(g)                # kernels are specialised for 4 heads
