"""Drop-in for the reference's ``DistillationLoss`` (distillation_loss.py:6-128) on the HIP kernels.

Same constructor, same ``forward`` signature, same 4-tuple ``(total, task, distill, teacher_task)``,
same ``ValueError`` when neither teacher form is given.  Differences, all documented in DESIGN.md:
  * no shifted / boolean-compacted copies of the logits are made (distillation_loss.py:31-45): the
    kernel evaluates the valid-row predicate per row;
  * arithmetic is fp32 inside the kernel whatever the logits dtype (the reference runs log-softmax
    in the logits' dtype), outputs are fp32 0-d tensors;
  * N == 0 returns zeros that ARE connected to autograd (reference: constants, distillation_loss.py:47-53);
  * sub-losses are returned detached (the reference only ever ``.item()``s them, train.py:108-114).
"""
import torch
import torch.nn as nn

from .ops import KDLossFn, KDLossRowsFn


class DistillationLoss(nn.Module):
    def __init__(self, temperature=2.0, alpha=0.5, inplace_grad=False):
        super().__init__()
        self.temperature = temperature
        self.alpha = alpha
        # write d loss / d logits over the logits buffer itself (saves V*B*T*2 bytes); only safe
        # when nobody reads the logits after the loss -- the trainer turns it on for training steps
        self.inplace_grad = inplace_grad

    def forward(self, student_logits, labels, teacher_logits=None, teacher_top_k_v=None, teacher_top_k_i=None,
                speech_token_mask=None):
        if teacher_logits is None and (teacher_top_k_v is None or teacher_top_k_i is None):
            raise ValueError("Either teacher_logits or top_k must be provided")
        if not student_logits.is_contiguous():
            student_logits = student_logits.contiguous()
        if teacher_logits is not None:
            teacher_logits = teacher_logits.detach()
        total, out = KDLossFn.apply(student_logits, labels, teacher_logits, teacher_top_k_v, teacher_top_k_i,
                                    speech_token_mask, self.temperature, self.alpha,
                                    self.inplace_grad and student_logits.requires_grad)
        return total, out[1], out[2], out[3]

    def forward_rows(self, student_logits, row_labels, teacher_logits=None, teacher_top_k_v=None, teacher_top_k_i=None):
        """Same 4-tuple for rows already shifted and selected by ``ops.loss_rows`` (what distillation_loss.py:31-45
        does with copies): ``student_logits`` [R,V], ``row_labels`` [R], teacher form [R,V] or ([R,K], [R,K])."""
        if teacher_logits is None and (teacher_top_k_v is None or teacher_top_k_i is None):
            raise ValueError("Either teacher_logits or top_k must be provided")
        if teacher_logits is not None:
            teacher_logits = teacher_logits.detach()
        total, out = KDLossRowsFn.apply(student_logits.contiguous(), row_labels, teacher_logits, teacher_top_k_v,
                                        teacher_top_k_i, self.temperature, self.alpha,
                                        self.inplace_grad and student_logits.requires_grad)
        return total, out[1], out[2], out[3]
