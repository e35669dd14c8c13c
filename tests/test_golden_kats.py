"""Committed oracle-generated golden vectors for the [K2] path (tests/golden/oracle_kats.json):
the oracle must keep reproducing them (CPU), and the HIP path must reproduce the c1 trajectory (GPU)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_oracle_vectors as mk          # noqa: E402

K = json.load(open(os.path.join(HERE, "golden", "oracle_kats.json")))


def test_oracle_reproduces_committed_kats():
    assert abs(K["zero_weight_loss_V17"] - 2.833213344) < 1e-9
    for cell, ref in K["single_batch"].items():
        got = mk.single_batch_case(cell)
        assert abs(got["loss"] - ref["loss"]) < 1e-12
        for k, (s, a) in ref["grad_sum_abs"].items():
            assert abs(got["grad_sum_abs"][k][0] - s) < 1e-10 and abs(got["grad_sum_abs"][k][1] - a) < 1e-10
    h = mk.c1_trajectory()
    np.testing.assert_allclose(h["loss"], K["c1_trajectory"]["loss"], rtol=1e-6)
    np.testing.assert_allclose(h["val_loss"], K["c1_trajectory"]["val_loss"], rtol=1e-6)


@pytest.mark.gpu
def test_hip_path_reproduces_committed_c1_trajectory():
    """config c1 through engine.Engine on the GPU vs the committed fp32-oracle trajectory (1e-3 rel)."""
    E = importlib.import_module("seq-recommendations_amd.engine")
    Bt = importlib.import_module("seq-recommendations_amd.batching")
    tr, va = mk.c1_data()
    cfg, p = mk.c1_params()
    eng = E.Engine(E.NetConfig(cell="lstm", act="relu", H=64, V_in=17, V_out=17, input="onehot", output="full",
                               use_bias=True, out_bias=False))
    for k, v in p.items():
        eng.set_param(k, v)
    order_rng = np.random.default_rng(5)
    B = 100
    loss, val = [], []
    for _ in range(3):
        index = order_rng.permutation(len(tr))
        tot = 0.0
        for s0 in range(0, len(tr), B):
            idx = index[s0:s0 + B]
            d = eng.upload(Bt.pack_sessions([tr[i] for i in idx]))
            tot += eng.train_step(d, lr=0.01, eps=1e-8, clipnorm=1.0).item() * len(idx)
        loss.append(tot / len(tr))
        v = 0.0
        for s0 in range(0, len(va), B):
            chunk = va[s0:s0 + B]
            v += eng.eval_loss(eng.upload(Bt.pack_sessions(chunk))).item() * len(chunk)
        val.append(v / len(va))
    np.testing.assert_allclose(loss, K["c1_trajectory"]["loss"], rtol=1e-3)
    np.testing.assert_allclose(val, K["c1_trajectory"]["val_loss"], rtol=1e-3)
