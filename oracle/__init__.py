"""CPU oracle for the segmentation-network hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package may import this directory: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use
it, and only as the checker.

PARITY UNPINNED: the reference (/root/reference, utils/model.py:145-236)
delegates every arithmetic step of this path to third-party packages that are
neither vendored nor installed (segmentation-models 1.0.1, image-classifiers
1.0.0, keras-applications 1.0.8, tensorflow 2.10; environment.yml:37,43,84,90)
and ships no tests, golden vectors or model files.  This oracle restates the
published algorithms of those pinned versions.  What pins it instead:

* framework-free known-answer tests (parameter counts, DenseNet skip indices,
  closed-form loss values) in ``tests/test_oracle_kat.py``;
* two independent op back-ends for the same graph code, ``ops_np`` (plain
  numpy loops/matmul) and ``ops_torch`` (torch.nn.functional on CPU), which
  must agree; golden vectors produced by the torch back-end are committed under
  ``tests/golden`` and are labelled "torch-CPU second opinion", never "Keras".

Layout is NHWC everywhere; kernels are HWIO, transpose-conv kernels are
(kh, kw, Cout, Cin) as Keras stores them.
"""
