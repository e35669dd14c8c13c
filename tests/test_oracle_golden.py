"""Pin oracle.metrics against outputs of the reference's own utils.py/sampler.py
(tests/golden/reference_utils_sampler.json, made by make_reference_vectors.py)."""
import json
import os

import numpy as np

from oracle import metrics as om

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_utils_sampler.json")))["cases"]


def test_survey_kats_are_what_the_reference_returns():
    c = G["model_py_example"]
    np.testing.assert_allclose(c["multinomial_k1"][0], [0.07407407, 0.33333333, 0.25925926, 0.33333333], atol=1e-8)
    assert abs(c["compute_likelihood"] - 1.056365689274534) < 1e-12
    np.testing.assert_allclose(c["compute_likelihood_cut_0.7"], [1.075128343669345, 1.0140985565638092], rtol=1e-12)


def test_count_models_match_reference():
    c = G["model_py_example"]
    seqs, n = c["seqs"], c["n"]
    np.testing.assert_array_equal(om.multinomial_probabilities(seqs, n, 1.0), np.array(c["multinomial_k1"]))
    np.testing.assert_array_equal(om.multinomial_probabilities(seqs, n, 1.0, normalize=False), np.array(c["multinomial_k1_unnorm"]))
    A, g = om.transition_matrix(seqs, n, k=10e-7, freq=False, end_state=False)
    np.testing.assert_array_equal(A, np.array(c["transition_k1e-6"]["alpha"]))
    np.testing.assert_array_equal(g, np.array(c["transition_k1e-6"]["gamma"]))
    A2, g2 = om.transition_matrix(seqs, n, k=1.0, freq=True, end_state=True)
    np.testing.assert_array_equal(A2, np.array(c["transition_counts_end_state"]["alpha"]))
    np.testing.assert_array_equal(g2, np.array(c["transition_counts_end_state"]["gamma"]))
    preds = om.markov_predict(A, g, seqs)
    for a, b in zip(preds, c["markov_preds"]):
        np.testing.assert_array_equal(a, b)


def test_nll_metrics_match_reference():
    c = G["model_py_example"]
    preds = c["markov_preds"]
    assert om.compute_likelihood(preds, count_first_prob=False) == c["compute_likelihood"]
    assert om.compute_likelihood(preds, count_first_prob=True) == c["compute_likelihood_first"]
    tr, va = om.compute_likelihood_cut(preds, 0.7)
    assert [tr, va] == c["compute_likelihood_cut_0.7"]
    o = c["compute_likelihood_cut_orig_lengths"]
    tr, va = om.compute_likelihood_cut(o["padded"], 0.7, orig_lengths=o["lengths"])
    assert [tr, va] == o["out"]
    assert om.neg_log_likelihood(np.array([0.5, 0.25, 0.125])) == c["neg_log_likelihood"]


def test_random_count_models_and_metrics_match_reference():
    """Randomized cases (incl. the MSNBC vocabulary size 17) produced by the reference's own utils.py:
    every (k, freq, end_state) variant of transition_matrix, multinomial_probabilities, the Markov
    prediction rule, compute_likelihood (both flags) and compute_likelihood_cut (two cut points and the
    orig_lengths branch) -- bit for bit."""
    for c in G["random_count_models"]:
        seqs, n = c["seqs"], c["n"]
        for v in c["variants"]:
            A, g = om.transition_matrix(seqs, n, k=v["k"], freq=v["freq"], end_state=v["end_state"])
            np.testing.assert_array_equal(A, np.array(v["alpha"]))
            np.testing.assert_array_equal(g, np.array(v["gamma"]))
        np.testing.assert_array_equal(om.multinomial_probabilities(seqs, n, 0.5), np.array(c["multinomial_k0.5"]))
        A, g = om.transition_matrix(seqs, n, k=0.5, freq=False, end_state=False)
        preds = om.markov_predict(A, g, seqs)
        for a, b in zip(preds, c["markov_preds"]):
            np.testing.assert_array_equal(a, b)
        assert [om.compute_likelihood(preds, count_first_prob=False), om.compute_likelihood(preds, count_first_prob=True)] == c["compute_likelihood"]
        for tp, ref in c["cut"].items():
            assert list(om.compute_likelihood_cut(preds, float(tp))) == ref
        o = c["cut_orig_lengths"]
        assert list(om.compute_likelihood_cut(o["padded"], 0.7, orig_lengths=o["lengths"])) == o["out"]
