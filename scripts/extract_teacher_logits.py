#!/usr/bin/env python3
"""Offline teacher top-K extraction on the MI355X path: counterpart of the reference's
``extract_teacher_logits.py`` (:17-146).  Teacher-only forward (BASELINE config 5: batch 64, seq_len 512)
-> log-softmax -> top-K -> per-sample UNPADDED fp16 / int32 arrays ``teacher_top_k_v`` / ``teacher_top_k_i``
(extract_teacher_logits.py:120-141), written next to the input columns with ``save_to_disk``."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def extract(teacher, batches, top_k, device):
    """batches: iterable of dicts with (teacher_)input_ids / attention_mask (right padded). Yields per-sample arrays."""
    from speech_distill_amd import ops
    all_v, all_i = [], []
    with torch.no_grad():
        for b in batches:
            ids = b.get("teacher_input_ids", b["input_ids"]).to(device)                       # :105
            am = b.get("teacher_attention_mask", b["attention_mask"]).to(device)
            logits = teacher(input_ids=ids, attention_mask=am).logits                         # :110
            v, i = ops.logsoftmax_topk(logits, top_k)                                         # :114-117
            lens = am.sum(1).cpu().tolist()
            v, i = v.cpu(), i.cpu()
            for r, n in enumerate(lens):                                                      # :120-129
                all_v.append(v[r, : int(n)].numpy())
                all_i.append(i[r, : int(n)].numpy())
    return all_v, all_i


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--teacher_model_path", required=True)   # extract_teacher_logits.py:154
    ap.add_argument("--dataset_path", required=True)
    ap.add_argument("--output_path", required=True)
    ap.add_argument("--top_k", type=int, default=100)        # :170
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--pad_token_id", type=int, default=153478)
    cfg = ap.parse_args()
    from datasets import load_from_disk
    from transformers import AutoModelForCausalLM
    import speech_distill_amd as sda
    from speech_distill_amd.collator import ProcessedDataCollator
    dev = torch.device("cuda:0")
    hf = AutoModelForCausalLM.from_pretrained(cfg.teacher_model_path, torch_dtype=torch.bfloat16)
    c = hf.config
    teacher = sda.HipQwen3ForCausalLM(
        sda.Qwen3Dims(c.vocab_size, c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                      c.num_key_value_heads, getattr(c, "head_dim", 128), c.rms_norm_eps, getattr(c, "rope_theta", 1e6),
                      c.tie_word_embeddings), device=dev, config=c, init_std=0)
    teacher.load_hf_state_dict(hf.state_dict())
    teacher.eval().requires_grad_(False)
    ds = load_from_disk(cfg.dataset_path)

    class _T:
        pad_token, pad_token_id = "<|semantic_token_end|>", cfg.pad_token_id

        def encode(self, *a, **k):
            return []
    loader = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=False,
                                         collate_fn=ProcessedDataCollator(_T(), pad_token_id=cfg.pad_token_id))
    v, i = extract(teacher, loader, cfg.top_k, dev)
    assert len(v) == len(ds)                                                                   # :133-137
    ds = ds.add_column("teacher_top_k_v", v).add_column("teacher_top_k_i", i)                  # :140-141
    ds.save_to_disk(cfg.output_path)                                                           # :145


if __name__ == "__main__":
    main()
