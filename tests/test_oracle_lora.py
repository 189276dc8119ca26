"""CPU tests of oracle/lora.py (the checker of the LoRA student; PARITY UNPINNED -- peft is absent) and of the host-side
configuration object of speech_distill_amd/lora.py."""
import math

import pytest
import torch

from oracle import lora as OL
from oracle import qwen3 as Q
from oracle import step as S

SHAPE = Q.Qwen3Shape(640, 128, 256, 2, 2, 1)


def _batch():
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 640, (2, 24), generator=g)
    lab = ids.clone()
    lab[:, :5] = -100
    return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": lab}


def test_scaling_and_targets():
    assert OL.scaling(32, 64, True) == 64 / math.sqrt(32) and OL.scaling(32, 64, False) == 2.0   # train.py:183-184, 198
    names = OL.target_names(SHAPE)
    assert len(names) == 7 * 2 and names[0] == "model.layers.0.self_attn.q_proj.weight"
    assert names[6] == "model.layers.0.mlp.down_proj.weight"


@pytest.mark.parametrize("init", ["pissa", "pissa_niter_4", "gaussian", True])
def test_init_rules(init):
    """peft reset_lora_parameters / pissa_init: B = 0 and the base untouched for the random inits; PiSSA splits W exactly
    (fp32) into the residual and s B A with B A of rank r carrying the top singular directions."""
    w = Q.init_weights(SHAPE, seed=0)
    base, lora = OL.attach(w, SHAPE, r=8, lora_alpha=16, init=init, base_dtype=None)
    s = OL.scaling(8, 16)
    assert torch.equal(base["lm_head.weight"], w["model.embed_tokens.weight"])       # the untied saved copy
    for name, (A, B) in lora.items():
        out_f, in_f = w[name].shape
        assert A.shape == (8, in_f) and B.shape == (out_f, 8)
        if init in ("gaussian", True):
            assert torch.equal(base[name], w[name]) and float(B.abs().max()) == 0
        else:
            assert float((base[name] + s * B @ A - w[name]).abs().max()) < 1e-5
            sv = torch.linalg.svdvals(w[name])
            got = torch.linalg.svdvals(s * B @ A)[:8]
            tol = 1e-4 if init == "pissa" else 0.1   # (randomized SVD of a matrix without a spectral gap)
            assert float((got - sv[:8]).abs().max()) <= tol * float(sv[0])
            if init == "pissa":   # the residual has lost exactly those directions
                assert abs(float(torch.linalg.svdvals(base[name])[0]) - float(sv[8])) < 1e-4


def test_merged_form_equals_the_adapter_form():
    """y = x W^T + s (x A^T) B^T (peft Linear.forward) is x (W + s B A)^T: on one projection, and on the whole model's
    loss (lora_step evaluates the merged form)."""
    g = torch.Generator().manual_seed(1)
    W, A, B = torch.randn(48, 32, generator=g), torch.randn(4, 32, generator=g), torch.randn(48, 4, generator=g)
    x = torch.randn(5, 32, generator=g)
    assert torch.allclose(x @ W.t() + 1.7 * (x @ A.t()) @ B.t(), x @ (W + 1.7 * B @ A).t(), atol=1e-4)
    w = Q.init_weights(SHAPE, seed=0)
    base, lora = OL.attach(w, SHAPE, r=8, lora_alpha=16, init="pissa", base_dtype=None)
    tshape = Q.Qwen3Shape(640, 256, 512, 2, 4, 2, tie_word_embeddings=False)
    tw = Q.init_weights(tshape, seed=1)
    plain = S.distill_step(w, SHAPE, tw, tshape, _batch(), top_k=16)
    lo = OL.lora_step(base, lora, SHAPE, tw, tshape, _batch(), r=8, lora_alpha=16, top_k=16)
    assert abs(float(plain["total"]) - float(lo["total"])) < 1e-5       # PiSSA at init IS the original model


def test_adapter_gradients_are_the_chain_rule_through_the_merge():
    """dA = s B^T dW, dB = s dW A^T with dW the plain model's weight gradient at the merged weights; the saved modules
    get the plain embedding / head gradients, split (the untied copies)."""
    w = Q.init_weights(SHAPE, seed=0)
    base, lora = OL.attach(w, SHAPE, r=8, lora_alpha=16, init="gaussian", base_dtype=None)
    g = torch.Generator().manual_seed(2)
    lora = {k: (A, torch.randn(B.shape, generator=g) * 0.02) for k, (A, B) in lora.items()}
    s = OL.scaling(8, 16)
    tshape = Q.Qwen3Shape(640, 256, 512, 2, 4, 2, tie_word_embeddings=False)
    tw = Q.init_weights(tshape, seed=1)
    lo = OL.lora_step(base, lora, SHAPE, tw, tshape, _batch(), r=8, lora_alpha=16, top_k=16)
    merged = {k: v.detach() for k, v in OL.merged_weights(base, lora, s).items()}
    untied = Q.Qwen3Shape(**{**SHAPE.__dict__, "tie_word_embeddings": False})
    plain = S.distill_step(merged, untied, tw, tshape, _batch(), top_k=16)
    assert set(lo["grads"]) == {n[:-6] + x for n in lora for x in ("lora_A", "lora_B")} | {"model.embed_tokens.weight", "lm_head.weight"}
    for name, (A, B) in lora.items():
        dW = plain["grads"][name]
        assert torch.allclose(lo["grads"][name[:-6] + "lora_A"], s * B.t() @ dW, rtol=1e-4, atol=1e-7)
        assert torch.allclose(lo["grads"][name[:-6] + "lora_B"], s * dW @ A.t(), rtol=1e-4, atol=1e-7)
    for k in ("model.embed_tokens.weight", "lm_head.weight"):
        assert torch.allclose(lo["grads"][k], plain["grads"][k], rtol=1e-4, atol=1e-7)


def test_bf16_storage_mode_is_noise_on_the_same_step():
    w = {k: v.bfloat16().float() for k, v in Q.init_weights(SHAPE, seed=0).items()}
    base, lora = OL.attach(w, SHAPE, r=8, lora_alpha=16, init="pissa")
    a = OL.lora_step(base, lora, SHAPE, None, None, dict(_batch(), teacher_top_k_v=torch.full((2, 24, 4), -1.4),
                                                        teacher_top_k_i=torch.arange(4).expand(2, 24, 4)), r=8, lora_alpha=16)
    b = OL.lora_step(base, lora, SHAPE, None, None, dict(_batch(), teacher_top_k_v=torch.full((2, 24, 4), -1.4),
                                                        teacher_top_k_i=torch.arange(4).expand(2, 24, 4)), r=8, lora_alpha=16,
                     storage="bf16")
    assert abs(float(a["total"]) - float(b["total"])) < 2e-2 * float(a["total"])
    for k, g in a["grads"].items():
        err = float((b["grads"][k] - g).norm() / g.norm())
        assert 1e-4 < err < 0.1, (k, err)


def test_lora_config_validation():
    from speech_distill_amd.lora import LoraConfig, SAVED, TARGETS
    c = LoraConfig()
    c.validate()
    assert (c.r, c.lora_alpha, c.use_rslora, c.init_lora_weights) == (32, 64, True, "pissa")     # train.py:474-487
    assert tuple(c.target_modules) == TARGETS and tuple(c.modules_to_save) == SAVED              # train.py:185-194
    assert LoraConfig(use_rslora=False).scaling == 2.0
    for bad in (dict(r=0), dict(r=129), dict(lora_dropout=0.1), dict(bias="all"), dict(target_modules=("qkv",)),
                dict(modules_to_save=("norm",)), dict(init_lora_weights="olora")):
        with pytest.raises((ValueError, NotImplementedError)):
            LoraConfig(**bad).validate()
