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


def nested_column(arrays, arrow_type, max_chunk_elems=1 << 30):
    """Per-sample [len_i, K] numpy arrays -> an arrow ``list<list<T>>`` column (chunked so that 32-bit list offsets never
    overflow).  The reference hands the list of arrays to ``Dataset.add_column`` (extract_teacher_logits.py:140-141) and
    lets arrow infer exactly this type; current pyarrow refuses 2-D numpy values there, so it is built explicitly --
    same on-disk type (halffloat / int32), same nesting."""
    import numpy as np
    import pyarrow as pa
    chunks, cur, n_el = [], [], 0

    def flush():
        if not cur:
            return
        K = cur[0].shape[1]
        flat = pa.array(np.concatenate([a.reshape(-1) for a in cur]), type=arrow_type)
        inner = pa.ListArray.from_arrays(pa.array(np.arange(0, len(flat) + 1, K, dtype=np.int32)), flat)
        rows = np.cumsum([0] + [a.shape[0] for a in cur]).astype(np.int32)
        chunks.append(pa.ListArray.from_arrays(pa.array(rows), inner))

    for a in arrays:
        if cur and n_el + a.size > max_chunk_elems:
            flush()
            cur, n_el = [], 0
        cur.append(a)
        n_el += a.size
    flush()
    return pa.chunked_array(chunks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--teacher_model_path", required=True)   # extract_teacher_logits.py:154
    ap.add_argument("--dataset_path", required=True)
    ap.add_argument("--output_path", required=True)
    ap.add_argument("--dataset_split", default="train")       # :162 (used when the directory holds a DatasetDict)
    ap.add_argument("--top_k", type=int, default=100)        # :170
    ap.add_argument("--batch_size", type=int, default=64, help=":167 (default there: 4; 288 GB of HBM take 64 x 512)")
    ap.add_argument("--max_length", type=int, default=None, help=":173; pre-processed rows longer than this are cut")
    # token strings (:175-217) are resolved through the tokenizer in --teacher_model_path when there is one
    ap.add_argument("--teacher_prefix", default="<|task_podcast|><|SPEAKER_0|>")
    ap.add_argument("--text_prefix", default='{"en": "", "zh": "", "yue": "<|Yue|>"}')
    ap.add_argument("--text_bos", default="<|text_start|>")
    ap.add_argument("--text_eos", default="<|text_end|>")
    ap.add_argument("--speech_bos", default="<|semantic_token_start|>")
    ap.add_argument("--speech_eos", default="<|semantic_token_end|>")
    ap.add_argument("--pad_token", default="<|semantic_token_end|>")
    ap.add_argument("--pad_token_id", type=int, default=153478, help="(+) when the model directory holds no tokenizer")
    cfg = ap.parse_args()
    from datasets import DatasetDict, load_from_disk
    import speech_distill_amd as sda
    from speech_distill_amd.collator import ProcessedDataCollator
    dev = torch.device("cuda:0")
    teacher = sda.HipQwen3ForCausalLM.from_pretrained(cfg.teacher_model_path, device=dev)       # :28-40
    teacher.eval().requires_grad_(False)
    ds = load_from_disk(cfg.dataset_path)
    if isinstance(ds, DatasetDict):
        ds = ds[cfg.dataset_split]
    if "teacher_input_ids" not in ds.column_names and "input_ids" not in ds.column_names:
        raise SystemExit("this path takes PRE-PROCESSED rows (data.py:124-141: teacher_input_ids / teacher_attention_mask); "
                         "the on-the-fly text+audio processing of extract_teacher_logits.py:44-79 is outside the hot path")
    if cfg.max_length is not None:
        cols = [c for c in ds.column_names if c.endswith("input_ids") or c.endswith("attention_mask")]
        ds = ds.map(lambda ex: {c: ex[c][:cfg.max_length] for c in cols})

    class _T:
        pad_token, pad_token_id = cfg.pad_token, cfg.pad_token_id

        def encode(self, *a, **k):
            return []
    tok = _T()
    if any(os.path.exists(os.path.join(cfg.teacher_model_path, f)) for f in ("tokenizer.json", "tokenizer_config.json")):
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(cfg.teacher_model_path)
        tok.pad_token = cfg.pad_token
    pad_id = tok.pad_token_id
    loader = torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=False,
                                         collate_fn=ProcessedDataCollator(tok, speech_bos=cfg.speech_bos, pad_token_id=pad_id))
    v, i = extract(teacher, loader, cfg.top_k, dev)
    assert len(v) == len(ds)                                                                   # :133-137
    import pyarrow as pa
    ds = ds.add_column("teacher_top_k_v", nested_column(v, pa.float16()))                      # :140-141
    ds = ds.add_column("teacher_top_k_i", nested_column(i, pa.int32()))
    ds.save_to_disk(cfg.output_path)                                                           # :145


if __name__ == "__main__":
    main()
