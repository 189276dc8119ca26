"""Helpers shared by the -m gpu parity tests (HIP path vs oracle / golden fixtures)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity_log.jsonl")


def dev():
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16)


def to_dev(x):
    return x.to(dev()).contiguous()


def rel_err(got, ref):
    """max |got-ref| / max|ref| and the rms ratio, computed in fp64 on the CPU."""
    g = got.detach().double().cpu()
    r = ref.detach().double().cpu()
    scale = max(float(r.abs().max()), 1e-30)
    mx = float((g - r).abs().max()) / scale
    rms = float((g - r).pow(2).mean().sqrt()) / max(float(r.pow(2).mean().sqrt()), 1e-30)
    return mx, rms


def record(name, **kw):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in kw.items()}}) + "\n")


def check_close(name, got, ref, max_tol, rms_tol=None):
    mx, rms = rel_err(got, ref)
    record(name, max_rel=mx, rms_rel=rms, max_tol=max_tol)
    assert np.isfinite(mx), f"{name}: non-finite result"
    assert mx <= max_tol, f"{name}: max rel err {mx:.3e} > {max_tol:.1e} (rms {rms:.3e})"
    if rms_tol is not None:
        assert rms <= rms_tol, f"{name}: rms rel err {rms:.3e} > {rms_tol:.1e}"
    return mx, rms


def assert_grad_budget(name, hip_grads, fp32_grads, bf16_grads, factor=1.5, names=None):
    """Gradient parity WITH an error budget: per tensor, the relative L2 error of the HIP gradient
    against the fp32 oracle must not exceed `factor` x the error of the bf16-storage oracle (oracle/qwen3.py,
    storage="bf16": the same step with bf16 rounding at the HIP path's storage points) against the same fp32 oracle.
    A wrong 3 % term passes a bare "cosine >= 0.99"; it does not pass this."""
    from oracle.step import grad_error_budget
    rows = grad_error_budget(hip_grads, fp32_grads, bf16_grads, names)
    worst = max(rows.items(), key=lambda kv: kv[1]["ratio"])
    record(name + "_grad_budget", worst_tensor=worst[0], worst_ratio=worst[1]["ratio"],
           worst_err_hip=worst[1]["err_hip"], worst_err_bf16_oracle=worst[1]["err_bf16_oracle"], factor=factor,
           n_tensors=len(rows))
    bad = {k: v for k, v in rows.items() if not v["err_hip"] <= factor * v["err_bf16_oracle"]}
    assert not bad, f"{name}: HIP gradient error exceeds {factor} x the bf16 storage noise: {bad}"
    return rows
