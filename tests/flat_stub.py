"""A CPU model that speaks the flat-gradient protocol of ``HipQwen3ForCausalLM`` (test infrastructure only).

The HIP student cannot run on the CPU (no fallback by design), so the CPU tests of the *host* logic around it --
``DistillationTrainer._wrap_model`` / ``ddp.HipDataParallel`` / ``FlatGradAllReduce`` under HF Trainer with gloo --
use this stand-in.  It keeps exactly the parts of the protocol that logic touches:

  * ONE flat parameter buffer and ONE flat gradient buffer, HF-style ``nn.Parameter`` views into them,
    ``layer_ranges`` / ``embed_range`` / ``norm_range`` / ``numel_flat`` / ``dims.tie_word_embeddings``;
  * gradients are written into ``flat_grad`` BY HAND in a custom autograd Function's backward (they never pass
    autograd hooks, which is why torch DDP cannot serve such a model), accumulating when ``.grad`` is live;
  * the backward reports finished stages (head, layer L-1 .. 0, embedding) to ``_stage_cb`` and drives
    ``_reducer`` the way ``HipQwen3ForCausalLM._run_backward`` does (begin_step, split tied embedding, finish).

The arithmetic is a toy (embedding -> L residual linear layers -> gain -> tied head) written as the pure function
``stub_logits`` so a test can recompute any gradient with plain autograd.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from speech_distill_amd.ddp import STAGE_EMBED, STAGE_HEAD


def stub_logits(x0, Ws, gain, E_head):
    x = x0
    for W in Ws:
        x = x + torch.tanh(x @ W.t())
    return (x * gain) @ E_head.t()


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, ids, model):
        ctx.model, ctx.ids = model, ids
        with torch.no_grad():
            return stub_logits(model.embed.detach()[ids], [w.detach() for w in model.Ws], model.gain.detach(),
                               model.embed.detach())

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._run_backward(ctx.ids, dlogits)
        return torch.zeros(()), None, None


class FlatStubLM(nn.Module):
    def __init__(self, V=32, h=8, L=2, seed=0):
        super().__init__()
        self.dims = SimpleNamespace(vocab_size=V, hidden_size=h, tie_word_embeddings=True)
        g = torch.Generator().manual_seed(seed)
        off = V * h
        self.embed_range = (0, off)
        self.layer_ranges = []
        for _ in range(L):
            self.layer_ranges.append((off, off + h * h))
            off += h * h
        self.norm_range = (off, off + h)
        self.numel_flat = off + h
        self.flat = torch.randn(self.numel_flat, generator=g) * 0.2
        self.flat[self.norm_range[0]:] = 1.0
        self.flat_grad = None
        self._grads_live = False
        self.embed = nn.Parameter(self.flat[:V * h].view(V, h))
        self.Ws = nn.ParameterList([nn.Parameter(self.flat[a:b].view(h, h)) for a, b in self.layer_ranges])
        self.gain = nn.Parameter(self.flat[self.norm_range[0]:self.norm_range[1]])
        self._anchor = torch.zeros((), requires_grad=True)
        self._stage_cb = None
        self.seen = []       # (input_ids, labels) of every forward, for the test's reference computation
        self.micro_log = []  # per backward: number of buckets the reducer issued
        self.grad_log = []   # per backward: the flat gradient buffer when the backward returns

    def _apply(self, fn, recurse=True):
        # as HipQwen3ForCausalLM._apply: the parameters must stay views of ONE flat buffer when HF Trainer calls
        # ``model.to(args.device)`` (nn.Module._apply would re-allocate every parameter separately)
        assert fn(self.flat).device == self.flat.device
        return self

    def _views(self, flat):
        V, h = self.dims.vocab_size, self.dims.hidden_size
        return ([flat[:V * h].view(V, h)] + [flat[a:b].view(h, h) for a, b in self.layer_ranges]
                + [flat[self.norm_range[0]:self.norm_range[1]]])

    def forward(self, input_ids=None, attention_mask=None, labels=None, **kwargs):
        self.seen.append((input_ids.clone(), None if labels is None else labels.clone()))
        if torch.is_grad_enabled():
            logits = _Fn.apply(self._anchor, input_ids, self)
        else:
            logits = stub_logits(self.embed[input_ids], list(self.Ws), self.gain, self.embed)
        return type("O", (dict,), {"logits": property(lambda s: s["logits"])})(logits=logits)

    def zero_grad(self, set_to_none=True):
        self._grads_live = False
        for p in self.parameters():
            p.grad = None

    def _run_backward(self, ids, dlogits):
        if self.flat_grad is None:
            self.flat_grad = torch.zeros_like(self.flat)
        params = [self.embed] + list(self.Ws) + [self.gain]
        accumulate = self._grads_live
        if not accumulate:
            for p, v in zip(params, self._views(self.flat_grad)):
                p.grad = v
            self._grads_live = True
        # local gradients with plain autograd on detached leaves (tied embedding: head part and lookup part apart)
        E_head = self.embed.detach().clone().requires_grad_(True)
        x0 = self.embed.detach()[ids].clone().requires_grad_(True)
        Ws = [w.detach().clone().requires_grad_(True) for w in self.Ws]
        gain = self.gain.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            out = stub_logits(x0, Ws, gain, E_head)
        gE, gx0, gg, *gW = torch.autograd.grad(out, [E_head, x0, gain] + Ws, dlogits)

        red = getattr(self, "_reducer", None)
        fg = self.flat_grad
        dx0 = None
        if red is not None:
            red.begin_step()
            if red.wants_split_embedding():
                dx0 = torch.empty_like(gx0.reshape(-1, gx0.shape[-1]))
                red.set_embedding_exchange(ids.reshape(-1), dx0, self.embed.grad)
        cb = self._stage_cb or (lambda stage: None)

        def put(rng, g):
            if accumulate:
                fg[rng[0]:rng[1]] += g.reshape(-1)
            else:
                fg[rng[0]:rng[1]] = g.reshape(-1)
        put(self.norm_range, gg)
        put(self.embed_range, gE)
        cb(STAGE_HEAD)
        for l in range(len(self.Ws) - 1, -1, -1):
            put(self.layer_ranges[l], gW[l])
            cb(l)
        if dx0 is not None:
            dx0.copy_(gx0.reshape(dx0.shape))
        else:
            self.embed.grad.index_add_(0, ids.reshape(-1), gx0.reshape(-1, gx0.shape[-1]))
        cb(STAGE_EMBED)
        if red is not None:
            red.finish()
            self.micro_log.append(len(red.issued))
        self.grad_log.append(fg.clone())
