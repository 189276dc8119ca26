"""Vectorised counterpart of the reference's ``ProcessedDataCollator`` (data.py:201-387), the
component that defines the hot path's input contract.

Same constructor and output keys/dtypes: right-padded ``input_ids`` / ``attention_mask`` (pad =
``pad_token_id``), ``labels`` = ids with pad -> -100 and every position before the first
``speech_bos`` -> -100 (a row without ``speech_bos`` is all -100), teacher twins, optional
pre-extracted ``teacher_top_k_v`` (pad 0.0) / ``teacher_top_k_i`` (pad 0) cut or padded to the
student length.  Unlike the reference there is no per-row ``.item()`` loop (data.py:374-382): the
speech mask is one cumulative sum.  Host-side only (runs in dataloader workers).
"""
from typing import Any, Dict, List, Optional

import torch


class ProcessedDataCollator:
    def __init__(self, tokenizer, pad_token_id=153478, speech_bos: str = "<|semantic_token_start|>",
                 pad_to_multiple_of: Optional[int] = None):
        self.tokenizer = tokenizer
        self.pad_token_id = pad_token_id
        self.pad_to_multiple_of = pad_to_multiple_of
        self.speech_bos = speech_bos

    def _pad_sequences(self, ids_list, mask_list):
        n = max(len(x) for x in ids_list)
        if self.pad_to_multiple_of is not None:
            m = self.pad_to_multiple_of
            n = (n + m - 1) // m * m
        ids = torch.full((len(ids_list), n), self.pad_token_id, dtype=torch.long)
        am = torch.zeros((len(ids_list), n), dtype=torch.long)
        for r, (a, b) in enumerate(zip(ids_list, mask_list)):
            ids[r, : len(a)] = torch.as_tensor(a, dtype=torch.long)
            am[r, : len(b)] = torch.as_tensor(b, dtype=torch.long)
        return {"input_ids": ids, "attention_mask": am}

    @staticmethod
    def _pad_logits(lst, max_length, padding_value=0.0):
        out = []
        for l in lst:
            l = l if isinstance(l, torch.Tensor) else torch.as_tensor(l)
            if l.size(0) < max_length:
                l = torch.cat([l, torch.full((max_length - l.size(0), l.size(1)), padding_value, dtype=l.dtype)], 0)
            out.append(l[:max_length])
        return torch.stack(out)

    def _create_speech_token_mask(self, input_ids):
        try:
            tok = self.tokenizer.encode(self.speech_bos, add_special_tokens=False)
            if not tok:
                return None
            return ((input_ids == tok[0]).cumsum(-1) > 0).to(torch.float32)
        except Exception:
            return None

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, torch.Tensor]:
        s_ids = [f["student_input_ids"] for f in features if "student_input_ids" in f]
        s_am = [f["student_attention_mask"] for f in features if "student_attention_mask" in f]
        t_ids = [f.get("teacher_input_ids") for f in features if "teacher_input_ids" in f]
        t_am = [f.get("teacher_attention_mask") for f in features if "teacher_attention_mask" in f]
        if not s_ids:
            s_ids = [f["input_ids"] for f in features]
            s_am = [f["attention_mask"] for f in features]
        batch = self._pad_sequences(s_ids, s_am)
        batch["labels"] = batch["input_ids"].clone()
        if self.pad_token_id is not None:
            batch["labels"][batch["labels"] == self.pad_token_id] = -100
        if t_ids and t_ids[0] is not None:
            tb = self._pad_sequences(t_ids, t_am)
            batch["teacher_input_ids"], batch["teacher_attention_mask"] = tb["input_ids"], tb["attention_mask"]
        top_v = [f.get("teacher_top_k_v") for f in features if "teacher_top_k_v" in f]
        top_i = [f.get("teacher_top_k_i") for f in features if "teacher_top_k_i" in f]
        if top_v and top_v[0] is not None:
            n = batch["input_ids"].size(1)
            batch["teacher_top_k_v"] = self._pad_logits(top_v, n, 0.0)
            batch["teacher_top_k_i"] = self._pad_logits(top_i, n, 0)
        sm = self._create_speech_token_mask(batch["input_ids"])
        if sm is not None:
            batch["labels"][sm == 0] = -100
        return batch
