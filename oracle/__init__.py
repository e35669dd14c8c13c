"""CPU oracle for the seq-recommendations hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the arithmetic that the reference
(efikarra/seq-recommendations, Python 2 + Keras 2.0.x + Theano) performs on
its one hot path: ``model.py``'s RNN next-item models trained through
``experiments_methods.run_model`` (Masking -> SimpleRNN/LSTM[/GRU] -> Dense ->
softmax -> masked categorical cross-entropy -> BPTT -> clipnorm + Adagrad).

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker* or the *timed CPU
baseline*, never as the product.  Nothing under ``seq-recommendations_amd/``
imports this package; the product path fails loudly when the HIP library is
missing.

Pinning status:

* ``oracle.metrics`` (count models + NLL metric definitions, reference
  ``utils.py:79-178`` and ``model.py:127-167``) is PINNED: checked in
  ``tests/test_oracle_golden.py`` against ``tests/golden/
  reference_utils_sampler.json``, which was produced by running the
  reference's own ``utils.py``/``sampler.py`` in the build container
  (``tests/golden/make_reference_vectors.py``).
* ``oracle.nn`` (the Keras/Theano layer semantics, marked [K2] in SURVEY.md)
  is **PARITY UNPINNED**: the arithmetic lives in the un-vendored, un-pinned
  third-party dependency Keras 2.0.x / Theano 0.9 (``README.md:7-10``), which
  is absent from this image, and the reference ships no tests, golden vectors
  or result files for it.  The restatement follows the published Keras 2.0
  layer definitions at the reference's call sites (``model.py:241-258``,
  ``model.py:322-403``, ``experiments_methods.py:19-50``) and is held together
  by known-answer and property tests only (zero-weight loss = ln V,
  pad-invariance, finite differences, torch-autograd twin, ...).
* Sampled softmax, GRU, Recall@K and the counter RNG are extensions that the
  reference does not contain; the oracle is their specification.
"""
