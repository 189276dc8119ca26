"""CPU oracle for the Stage-2 distillation step (TEST INFRASTRUCTURE ONLY).

This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the reference's
hot-path arithmetic.  It exists so that the hand-written HIP kernels in
``speech_distill_amd`` have something independent to be checked against on the
GPU box, where ``/root/reference`` does not exist.

Rules (enforced by ``tests/test_cabi.py::test_product_never_imports_the_oracle``):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import anything from here;
  * nothing in ``speech_distill_amd/`` imports it, and the product path raises
    when the HIP library is missing instead of falling back to this code.

Pinning: the reference has no tests and no golden vectors of its own
(SURVEY.md section 8c), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py``
(imports ``/root/reference/distillation_loss.py``, ``train.DistillationTrainer``
and ``data.ProcessedDataCollator`` and the installed HF ``Qwen3ForCausalLM``) and
committed as small fixtures under ``tests/golden/``.
"""
