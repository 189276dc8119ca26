"""speech_distill_amd: the Stage-2 distillation hot path of indiejoseph/speech-distill, MI355X-native.

Hand-written HIP (gfx950) kernels behind the reference's own plug-in surface:
``DistillationLoss.forward`` (distillation_loss.py:14-128) and
``DistillationTrainer.compute_loss`` (train.py:43-116).  There is NO CPU fallback: every compute
entry point raises if ``libsd_hip.so`` is missing or a tensor is not on a GPU.
"""
from ._lib import lib_path, load_lib, SdHipError  # noqa: F401
from .distillation_loss import DistillationLoss  # noqa: F401
from .qwen3 import HipQwen3ForCausalLM, Qwen3Dims  # noqa: F401

__all__ = ["DistillationLoss", "HipQwen3ForCausalLM", "Qwen3Dims", "load_lib", "lib_path", "SdHipError"]
