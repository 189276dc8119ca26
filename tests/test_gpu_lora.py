"""-m gpu tests of the LoRA student (train.py:180-202; speech_distill_amd/lora.py, csrc/sd_lora.hip).

PARITY UNPINNED (peft is absent from the reference tree and from the image): the checker is oracle/lora.py, the
restatement of peft's published LoRA layer.  Kernels are checked against fp64 torch arithmetic on the same bf16 inputs,
the assembled step against the oracle with the error budget of tests/gpu_util.py.
"""
import ctypes as C
import math
import os
import tempfile

import pytest
import torch

from gpu_util import assert_grad_budget, check_close, dev, record, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sda():
    import speech_distill_amd as m
    m.load_lib()
    return m


# ------------------------------------------------------------------------------------------------------ kernels
def _plan(targets, r_pad):
    """targets: list of dicts of GPU tensors (w_res, w_out, w_grad, a_sh, a_sc, b_sc, d_a, d_b)."""
    from speech_distill_amd import _lib
    lib = _lib.load_lib()
    arr = (_lib.LoraTarget * len(targets))()
    for t, s in zip(arr, targets):
        t.w_res, t.w_out, t.w_grad = s["w_res"].data_ptr(), s["w_out"].data_ptr(), s["w_grad"].data_ptr()
        t.a_shadow, t.a_scaled, t.b_scaled = s["a_sh"].data_ptr(), s["a_sc"].data_ptr(), s["b_sc"].data_ptr()
        t.d_a, t.d_b = s["d_a"].data_ptr(), s["d_b"].data_ptr()
        t.out_features, t.in_features = s["w_res"].shape
    nb = lib.sd_lora_plan_bytes(len(targets))
    host = C.create_string_buffer(nb)
    _lib.check(lib.sd_lora_plan_build(arr, len(targets), r_pad, host, nb), "plan")
    devp = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev())
    return lib, host, devp


@pytest.mark.parametrize("r_pad", [32, 64, 128])
def test_lora_merge_and_project_kernels(sda, r_pad):
    """One launch over targets of mixed shapes (the student's seven projection shapes at 1/4 width, a 32-row one, one
    whose rows are not a multiple of the 256-row merge chunk): merge and both projections against fp64 arithmetic on
    the same bf16 operands, to bf16 rounding of the result."""
    from speech_distill_amd import _lib
    g = torch.Generator().manual_seed(r_pad)
    shapes = [(512, 256), (256, 256), (256, 256), (256, 512), (768, 256), (768, 256), (256, 768), (32, 128), (416, 384)]
    if r_pad == 32:   # the student's real projection shapes (q, k/v, gate/up, down, o)
        shapes += [(2048, 1024), (1024, 1024), (3072, 1024), (1024, 3072), (1024, 2048)]
    scale = 64 / math.sqrt(r_pad)
    T = []
    for out_f, in_f in shapes:
        bf = lambda *s, std=1.0: to_dev((torch.randn(*s, generator=g) * std).bfloat16())  # noqa: E731
        A, B = torch.randn(r_pad, in_f, generator=g) * 0.05, torch.randn(out_f, r_pad, generator=g) * 0.05
        T.append(dict(w_res=bf(out_f, in_f, std=0.05), w_out=torch.zeros(out_f, in_f, dtype=torch.bfloat16, device=dev()),
                      w_grad=bf(out_f, in_f, std=0.01), a_sh=to_dev(A.bfloat16()), a_sc=to_dev((A * scale).bfloat16()),
                      b_sc=to_dev((B * scale).bfloat16()),
                      d_a=torch.full((r_pad, in_f), 7.0, dtype=torch.bfloat16, device=dev()),
                      d_b=torch.full((out_f, r_pad), 7.0, dtype=torch.bfloat16, device=dev())))
    lib, host, devp = _plan(T, r_pad)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.sd_lora_merge(devp.data_ptr(), host, st), "merge")
    _lib.check(lib.sd_lora_project(devp.data_ptr(), host, st), "project")
    torch.cuda.synchronize()
    for i, t in enumerate(T):
        d = lambda k: t[k].double().cpu()  # noqa: E731
        w_ref = d("w_res") + d("b_sc") @ d("a_sh")
        check_close(f"lora_merge_r{r_pad}_{i}", t["w_out"], w_ref, 2 ** -7.5, 2 ** -8.5)
        # the fp32 accumulator of 32..128 products rounds differently from fp64 only near ties
        assert float((t["w_out"].cpu() != w_ref.bfloat16()).float().mean()) <= 5e-3
        check_close(f"lora_dA_r{r_pad}_{i}", t["d_a"], d("b_sc").t() @ d("w_grad"), 2 ** -7.5, 2 ** -8.5)
        check_close(f"lora_dB_r{r_pad}_{i}", t["d_b"], d("w_grad") @ d("a_sc").t(), 2 ** -7.5, 2 ** -8.5)


def test_lora_plan_refuses_shapes_it_cannot_tile(sda):
    from speech_distill_amd import _lib
    lib = _lib.load_lib()
    z = torch.zeros(64 * 192, dtype=torch.bfloat16, device=dev())
    arr = (_lib.LoraTarget * 1)()
    for f in ("w_res", "w_out", "w_grad", "a_shadow", "a_scaled", "b_scaled", "d_a", "d_b"):
        setattr(arr[0], f, z.data_ptr())
    host = C.create_string_buffer(lib.sd_lora_plan_bytes(1))
    arr[0].out_features, arr[0].in_features = 64, 192          # in % 128 != 0
    assert lib.sd_lora_plan_build(arr, 1, 32, host, len(host)) == -3
    arr[0].out_features, arr[0].in_features = 48, 128          # out % 32 != 0
    assert lib.sd_lora_plan_build(arr, 1, 32, host, len(host)) == -3
    arr[0].out_features, arr[0].in_features = 64, 128
    assert lib.sd_lora_plan_build(arr, 1, 48, host, len(host)) == -3   # r_pad
    assert lib.sd_lora_plan_build(arr, 1, 32, host, 8) == -5
    assert lib.sd_lora_plan_build(arr, 1, 32, host, len(host)) == 0


def test_adamw_f32_shadow_matches_torch_adamw(sda):
    """Five steps of the fp32-master AdamW (bf16 gradient, clip coefficient from device memory) against
    torch.optim.AdamW on the same fp32 values; shadows = bf16(p), bf16(scale p)."""
    from speech_distill_amd import ops
    g = torch.Generator().manual_seed(0)
    n = 4 * 1000 + 4 * 37
    p0 = torch.randn(n, generator=g) * 0.1
    p = to_dev(p0.clone())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    s1, s2 = torch.zeros(n, dtype=torch.bfloat16, device=dev()), torch.zeros(n, dtype=torch.bfloat16, device=dev())
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    ss = torch.zeros(1, device=dev())
    for step in range(1, 6):
        gr = (torch.randn(n, generator=g) * (3.0 if step == 2 else 0.01)).bfloat16()
        ss.zero_()
        ops.sumsq(to_dev(gr), ss)
        ops.adamw_f32_shadow_(p, to_dev(gr), m, v, s1, s2, 11.3125, 1e-2, 0.9, 0.999, 1e-8, 0.1, step, ss, 1.0)
        ref.grad = gr.float()
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
    check_close("adamw_f32_shadow_p", p, ref.detach(), 1e-5)
    assert torch.equal(s1, p.bfloat16()) and torch.equal(s2, (p * 11.3125).bfloat16())


# --------------------------------------------------------------------------------------------------- the student
ST, TE = (640, 128, 256, 2, 2, 1), (640, 256, 512, 2, 4, 2)


def _students(sda, init, r=8, seed=3, tied=True):
    from oracle import qwen3 as Q
    from speech_distill_amd import lora as L
    shp = Q.Qwen3Shape(*ST, tie_word_embeddings=tied)
    w = {k: v.bfloat16().float() for k, v in Q.init_weights(shp, seed=1, norm_jitter=0.1).items()}
    model = sda.HipQwen3ForCausalLM(sda.Qwen3Dims(*ST, tie_word_embeddings=tied), device=dev(), init_std=0)
    model.load_hf_state_dict(w)
    cfg = L.LoraConfig(r=r, lora_alpha=16, init_lora_weights=init)
    return shp, w, L.get_lora_model(model, cfg, seed=seed), cfg


def _oracle_inputs(model):
    """The oracle's (base_w, lora) from what the HIP model holds: residual base, fp32 masters."""
    st = model._lora
    base = {k: p.detach().float().cpu() for k, p in model._params.items()}
    lora = {}
    for i, (name, _, _) in enumerate(st.targets):
        base[name] = st.base_view(i).float().cpu()
        lora[name] = (st.a_view(i)[:st.r].cpu().clone(), st.b_view(i)[:, :st.r].cpu().clone())
    return base, lora


@pytest.mark.parametrize("init", ["pissa", "gaussian", True])
def test_lora_init_matches_peft_rules(sda, init):
    """PiSSA: W_res + s B A reproduces W (to the bf16 rounding of the residual), B A = the rank-r truncation of W / s
    (the oracle's SVD on the CPU; the factors themselves are sign-ambiguous); gaussian / default: B = 0, A with the
    stated spread, base untouched; trainable set and parameter count as peft reports them; an untied copy is made."""
    from oracle import lora as OL
    shp, w, model, cfg = _students(sda, init, r=8)
    st = model._lora
    assert not model.dims.tie_word_embeddings and "lm_head.weight" in model._params      # modules_to_save untie the pair
    assert torch.equal(model._params["lm_head.weight"], model._params["model.embed_tokens.weight"])
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert sorted(names) == sorted(["model.embed_tokens.weight", "lm_head.weight"] +
                                   [t[0][:-6] + f"lora_{x}.weight" for t in st.targets for x in "AB"]), names
    assert len(st.targets) == 7 * 2
    train, total = st.trainable_parameters()
    assert train == 2 * 640 * 128 + sum(8 * (o + i) for _, o, i in st.targets)
    s = cfg.scaling
    assert s == 16 / math.sqrt(8)
    for i, (name, out_f, in_f) in enumerate(st.targets):
        A, B = st.a_view(i)[:8].cpu(), st.b_view(i)[:, :8].cpu()
        W = w[name]
        if init == "pissa":
            check_close(f"pissa_identity_{name}", st.base_view(i).float().cpu() + s * B @ A, W, 2 ** -6.5, 2e-3)
            Ao, Bo, _ = OL.init_pair(W, 8, s, "pissa", None)
            check_close(f"pissa_lowrank_{name}", B @ A, Bo @ Ao, 2e-3)
        else:
            assert torch.equal(st.base_view(i).float().cpu(), W) and float(B.abs().max()) == 0.0
            if init == "gaussian":
                assert abs(float(A.std()) - 1 / 8) < 0.02
            else:
                assert float(A.abs().max()) <= 1 / math.sqrt(in_f) and float(A.std()) > 0.4 / math.sqrt(in_f)
    # padding rows / columns (r = 8 inside r_pad = 32) are zero
    assert float(st.a_view(0)[8:].abs().max()) == 0.0 and float(st.b_view(0)[:, 8:].abs().max()) == 0.0
    # a model that is NOT tied keeps its own head (no copy)
    _, _, m2, _ = _students(sda, "gaussian", tied=False)
    assert m2._lora is not None and "lm_head.weight" in m2._params


def _batch(seed=0, B=3, T=40):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 600, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    am[1, 29:] = 0
    lab = ids.clone()
    lab[:, :6] = -100
    lab[am == 0] = -100
    return {"input_ids": ids, "attention_mask": am, "labels": lab}


def _hip_step(sda, model, teacher, batch, top_k=16):
    loss_fn = sda.DistillationLoss(temperature=2.0, alpha=0.5)
    ids, am, lab = (to_dev(batch[k]) for k in ("input_ids", "attention_mask", "labels"))
    with torch.no_grad():
        tl = teacher(input_ids=ids, attention_mask=am).logits
    from speech_distill_amd import ops
    kv, ki = ops.logsoftmax_topk(tl, top_k, vocab_size=640)
    out = model(input_ids=ids, attention_mask=am)
    total, task, distill, tloss = loss_fn(student_logits=out.logits, labels=lab, teacher_top_k_v=kv, teacher_top_k_i=ki)
    total.backward()
    return total, kv, ki


@pytest.mark.parametrize("init", ["pissa", "gaussian"])
def test_lora_step_matches_oracle(sda, init):
    """One distillation micro-step of the LoRA student (teacher forward, top-K, KL + CE, backward, projection) against
    oracle.lora.lora_step on the same residual base / A / B: the loss, and every trainable gradient (dA, dB of all 14
    targets, embed_tokens, lm_head) within 1.5 x the bf16-storage oracle's own error; then a second micro-batch
    accumulates, and zero_grad starts over."""
    from oracle import lora as OL, qwen3 as Q
    shp, w, model, cfg = _students(sda, init, r=8)
    st = model._lora
    if init == "gaussian":      # B = 0 would make dA vanish: give the adapter a life
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for k, p in st.params.items():
                if k.endswith("lora_B.weight"):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    tshp = Q.Qwen3Shape(*TE)
    tw = {k: v.bfloat16().float() for k, v in Q.init_weights(tshp, seed=2).items()}
    teacher = sda.HipQwen3ForCausalLM(sda.Qwen3Dims(*TE), device=dev(), init_std=0)
    teacher.load_hf_state_dict(tw)
    teacher.eval().requires_grad_(False)
    batch = _batch()
    total, kv, ki = _hip_step(sda, model, teacher, batch)
    model.finalize_grads()
    hip = {k[:-len(".weight")] if "lora_" in k else k: v.float().cpu() for k, v in st.grads().items()}
    base, lora = _oracle_inputs(model)
    ob = dict(batch, teacher_top_k_v=kv.cpu(), teacher_top_k_i=ki.cpu())
    kw = dict(r=8, lora_alpha=16, top_k=16)
    o32 = OL.lora_step(base, lora, shp, None, None, ob, **kw)
    o16 = OL.lora_step(base, lora, shp, None, None, ob, storage="bf16", **kw)
    rel = abs(float(total) - float(o32["total"])) / float(o32["total"])
    record("lora_step_loss", init=init, hip=float(total), oracle=float(o32["total"]), rel=rel)
    assert rel <= 5e-3
    assert set(hip) == set(o32["grads"]), set(hip) ^ set(o32["grads"])
    assert_grad_budget(f"lora_step_{init}", hip, o32["grads"], o16["grads"])
    for k, gr in hip.items():
        assert float(gr.norm()) > 0, k
    # accumulation: a second, different micro-batch adds its dW; the projection is taken from the sum
    g1 = {k: v.clone() for k, v in hip.items()}
    b2 = _batch(seed=1)
    _hip_step(sda, model, teacher, b2)
    model.finalize_grads()
    acc = {k[:-len(".weight")] if "lora_" in k else k: v.float().cpu() for k, v in st.grads().items()}
    o2 = OL.lora_step(base, lora, shp, tw, tshp, b2, **kw)
    for k in ("model.layers.0.self_attn.q_proj.lora_A", "model.layers.1.mlp.down_proj.lora_B", "lm_head.weight"):
        want = o32["grads"][k] + o2["grads"][k]
        err = float((acc[k] - want).norm() / want.norm())
        record("lora_accumulate", tensor=k, rel_l2=err)
        assert err <= 6e-2, (k, err)
    model.zero_grad()
    _hip_step(sda, model, teacher, batch)
    model.finalize_grads()
    again = st.grads()["model.layers.0.self_attn.q_proj.lora_A.weight"].float().cpu()
    assert torch.equal(again, g1["model.layers.0.self_attn.q_proj.lora_A"])


def test_lora_optimizer_steps_train_the_adapter_only(sda):
    """FlatAdamW on a LoRA student: A / B / embed / head move, the residual base and the norm gains do not; the decoder's
    weights equal W_res + s B A after every step (the merge follows the update); the loss goes down; the moments of the
    adapter are fp32; the gradient norm is the norm over exactly the trainable tensors."""
    from oracle import qwen3 as Q
    from speech_distill_amd.optim import FlatAdamW
    shp, w, model, cfg = _students(sda, "pissa", r=8)
    st = model._lora
    teacher = sda.HipQwen3ForCausalLM(sda.Qwen3Dims(*TE), device=dev(), seed=2)
    teacher.eval().requires_grad_(False)
    opt = FlatAdamW(model, lr=2e-3, clip=1.0)
    assert opt.exp_avg32.dtype == torch.float32 and opt.exp_avg32.numel() == st.master.numel()
    assert opt.exp_avg.numel() == 2 * 640 * 128
    batch = _batch()
    base0, gains0 = st.base.clone(), model._params["model.layers.0.input_layernorm.weight"].clone()
    a0 = st.master.clone()
    losses = []
    for it in range(6):
        total, _, _ = _hip_step(sda, model, teacher, batch)
        losses.append(float(total))
        if it == 0:
            n = opt.grad_norm(1.0)
            want = math.sqrt(sum(float(v.double().pow(2).sum()) for v in st.grads().values()))
            assert abs(float(n) - want) <= 1e-3 * want
        opt.step()
        opt.zero_grad()
        model.zero_grad()
    record("lora_training_losses", losses=losses)
    assert losses[-1] < losses[0] - 0.05, losses
    assert torch.equal(st.base, base0) and torch.equal(model._params["model.layers.0.input_layernorm.weight"], gains0)
    assert not torch.equal(st.master, a0)
    with torch.no_grad():
        model(input_ids=to_dev(batch["input_ids"]))       # runs the pending merge
    for i, (name, _, _) in enumerate(st.targets):
        A, B = st.a_view(i).double(), st.b_view(i).double()
        want = st.base_view(i).double() + (B * st.scale).bfloat16().double() @ A.bfloat16().double()
        check_close(f"merged_after_steps_{name}", model._params[name], want.cpu(), 2 ** -7, 2 ** -8.5)
    assert float(st.a_view(0)[8:].abs().max()) == 0.0 and float(st.b_view(0)[:, 8:].abs().max()) == 0.0


def test_lora_checkpoint_round_trip_and_merge(sda):
    """save_pretrained writes peft's adapter layout; loading it into a freshly attached adapter over the same base
    reproduces the logits bit for bit; merge_and_unload gives a plain model with the same logits."""
    import json
    from safetensors.torch import load_file
    from speech_distill_amd import lora as L
    shp, w, model, cfg = _students(sda, "gaussian", r=8)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for p in model._lora.params.values():
            p.copy_(torch.randn(p.shape, generator=g) * 0.03)
        model._params["lm_head.weight"].add_(0.01)
    ids = to_dev(_batch()["input_ids"])
    with torch.no_grad():
        ref = model(input_ids=ids).logits
    d = tempfile.mkdtemp()
    model.save_pretrained(d)
    c = json.load(open(os.path.join(d, "adapter_config.json")))
    assert c["peft_type"] == "LORA" and c["r"] == 8 and c["use_rslora"] and sorted(c["modules_to_save"]) == ["embed_tokens", "lm_head"]
    sd = load_file(os.path.join(d, "adapter_model.safetensors"))
    assert "base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight" in sd
    assert tuple(sd["base_model.model.model.layers.1.mlp.down_proj.lora_B.weight"].shape) == (128, 8)
    assert "base_model.model.lm_head.weight" in sd and "base_model.model.model.embed_tokens.weight" in sd
    assert len(sd) == 2 * 14 + 2 and sd["base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"].dtype == torch.float32
    _, _, fresh, _ = _students(sda, "gaussian", r=8, seed=99)
    with torch.no_grad():
        assert not torch.equal(fresh(input_ids=ids).logits, ref)
    fresh._lora.load_adapter(d)
    with torch.no_grad():
        assert torch.equal(fresh(input_ids=ids).logits, ref)
    plain = L.merge_and_unload(fresh)
    assert plain._lora is None and all(p.requires_grad for p in plain.parameters())
    assert not any("lora" in n for n, _ in plain.named_parameters())
    with torch.no_grad():
        assert torch.equal(plain(input_ids=ids).logits, ref)


def test_lora_through_the_hf_trainer_with_the_references_arguments(sda):
    """train.py:180-202 + :331-420 on the tiny config with the reference's TrainingArguments (eval and checkpoint every
    epoch, load_best_model_at_end, save_total_limit 3, gradient accumulation, clipping): the LoRA student trains through
    the fused optimizer, every checkpoint is a peft-style adapter directory, and after train() the student holds the best
    checkpoint's adapter bit for bit and evaluates to that checkpoint's eval loss."""
    import numpy as np
    from safetensors.torch import load_file
    from transformers import TrainingArguments
    import test_gpu_model as M
    from speech_distill_amd import lora as L
    from speech_distill_amd.collator import ProcessedDataCollator
    from speech_distill_amd.optim import FlatAdamW
    from speech_distill_amd.trainer import DistillationTrainer
    z, st, te, sw, tw, feats, pad, bos = M._c1(sda)
    student, teacher = M._build(sda, st, sw), M._build(sda, te, tw)
    teacher.eval().requires_grad_(False)
    student = L.get_lora_model(student, L.LoraConfig(r=8, lora_alpha=16, init_lora_weights="pissa"))
    out = tempfile.mkdtemp()
    args = TrainingArguments(
        output_dir=out, per_device_train_batch_size=4, per_device_eval_batch_size=4, gradient_accumulation_steps=2,
        num_train_epochs=4, learning_rate=2e-3, logging_steps=1, eval_strategy="epoch", save_strategy="epoch",
        load_best_model_at_end=True, save_total_limit=3, gradient_checkpointing=True, report_to=[],
        remove_unused_columns=False, label_names=["labels"], seed=42, data_seed=42, lr_scheduler_type="constant",
        warmup_steps=0, weight_decay=0.0, max_grad_norm=1.0, dataloader_num_workers=0, bf16=True)

    class DS(torch.utils.data.Dataset):
        def __init__(self, rows):
            self.rows = rows

        def __len__(self):
            return len(self.rows)

        def __getitem__(self, i):
            return dict(self.rows[i])
    coll = ProcessedDataCollator(M._Tok(pad, bos), pad_token_id=pad)
    tr = DistillationTrainer(model=student, args=args, train_dataset=DS(feats), eval_dataset=DS(feats[:4]),
                             data_collator=coll, teacher_model=teacher, temperature=2.0, alpha=0.5, top_k=16)
    tr.train()
    assert isinstance(getattr(tr.optimizer, "optimizer", tr.optimizer), FlatAdamW)
    hist = [h for h in tr.state.log_history if "loss" in h and "grad_norm" in h]
    assert len(hist) == 4 and all(math.isfinite(h["grad_norm"]) and h["grad_norm"] > 0 for h in hist)
    assert hist[-1]["loss"] < hist[0]["loss"], hist
    evals = [h["eval_loss"] for h in tr.state.log_history if "eval_loss" in h]
    assert len(evals) == 4 and all(np.isfinite(evals)) and evals[-1] < evals[0]
    ckpts = sorted(d for d in os.listdir(out) if d.startswith("checkpoint-"))
    assert len(ckpts) == 3, ckpts
    for c in ckpts:
        files = set(os.listdir(os.path.join(out, c)))
        assert {"adapter_config.json", "adapter_model.safetensors", "optimizer.pt", "trainer_state.json"} <= files
        assert "model.safetensors" not in files
    best = tr.state.best_model_checkpoint
    assert best is not None and os.path.basename(best) in ckpts
    sd = load_file(os.path.join(best, "adapter_model.safetensors"))
    for k, v in student.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    ev = tr.evaluate()
    record("lora_trainer", losses=[h["loss"] for h in hist], evals=evals, final_eval=ev["eval_loss"])
    assert abs(ev["eval_loss"] - min(evals)) <= 1e-6 + 1e-3 * abs(min(evals))
