"""Batch builder with the contract of the reference's ``ProcessedDataCollator`` (data.py:201-387), the component that
defines the hot path's inputs: same constructor, same output keys, dtypes and values (pinned by
fixture G3, ``tests/test_collator_golden.py``) --

* ``input_ids`` / ``attention_mask`` int64, right-padded with ``pad_token_id`` / 0 to the longest row (rounded up to
  ``pad_to_multiple_of``);
* ``labels`` = ids where a target exists, else -100: no target at ANY position holding the pad id (quirk Q2: the final
  ``<|semantic_token_end|>`` equals the pad token, data.py:250-251) nor before the first ``speech_bos`` of a row (a row
  without one has no targets, data.py:273-276, 350-387);
* ``teacher_input_ids`` / ``teacher_attention_mask`` when the features carry teacher sequences;
* ``teacher_top_k_v`` (pad 0.0) / ``teacher_top_k_i`` (pad 0) ``[B, T, K]`` cut or padded to the STUDENT width when the
  features carry pre-extracted top-K (data.py:330-348).

Built differently from the reference: every ragged field becomes ONE scatter of its concatenated values into a
pre-filled ``[B, W]`` grid (a length mask selects the valid cells in row-major order: no per-row copy loop), and
``labels`` is one ``where`` over the grid instead of a clone masked twice.  Host-side only (dataloader workers).
"""
import itertools
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch


def _ragged_to_grid(seqs: Sequence, width: int, fill, dtype) -> torch.Tensor:
    """Right-padded ``[len(seqs), width]`` grid of the ragged ``seqs`` (1-D each), one scatter."""
    lens = torch.tensor([len(s) for s in seqs], dtype=torch.long)
    grid = torch.full((len(seqs), width), fill, dtype=dtype)
    total = int(lens.sum())
    if total:
        if dtype == torch.long and all(isinstance(s, (list, tuple)) for s in seqs):
            # datasets hands out Python lists: one numpy pass over the chained items is 2x faster than a tensor per row
            flat = torch.from_numpy(np.fromiter(itertools.chain.from_iterable(seqs), dtype=np.int64, count=total))
        else:
            flat = torch.cat([torch.as_tensor(s, dtype=dtype).reshape(-1) for s in seqs])
        # row-major order of the valid cells = the order of the concatenation
        grid[torch.arange(width)[None, :] < lens[:, None]] = flat
    return grid


class ProcessedDataCollator:
    def __init__(self, tokenizer, pad_token_id=153478, speech_bos: str = "<|semantic_token_start|>",
                 pad_to_multiple_of: Optional[int] = None):
        self.tokenizer = tokenizer
        self.pad_token_id = pad_token_id
        self.pad_to_multiple_of = pad_to_multiple_of
        self.speech_bos = speech_bos

    # ------------------------------------------------------------------------------------------------ pieces
    def _width(self, seqs) -> int:
        w = max(len(s) for s in seqs)
        m = self.pad_to_multiple_of
        return w if m is None else -(-w // m) * m

    def _ids_and_mask(self, ids_seqs, mask_seqs):
        w = self._width(ids_seqs)
        return (_ragged_to_grid(ids_seqs, w, self.pad_token_id, torch.long), _ragged_to_grid(mask_seqs, w, 0, torch.long))

    def _speech_bos_id(self) -> Optional[int]:
        """Token id of ``speech_bos``; None (= no text masking, as the reference does on a tokenizer failure) otherwise."""
        try:
            tok = self.tokenizer.encode(self.speech_bos, add_special_tokens=False)
        except Exception:
            return None
        return tok[0] if tok else None

    @staticmethod
    def _topk_grid(items, width: int, fill) -> torch.Tensor:
        """Per-sample ``[len_i, K]`` -> ``[B, width, K]``: rows past ``width`` are cut, missing rows take ``fill``."""
        first = items[0] if isinstance(items[0], torch.Tensor) else torch.as_tensor(items[0])
        out = torch.full((len(items), width, first.size(1)), fill, dtype=first.dtype)
        for b, it in enumerate(items):
            it = it if isinstance(it, torch.Tensor) else torch.as_tensor(it)
            n = min(width, it.size(0))
            out[b, :n] = it[:n]
        return out

    # -------------------------------------------------------------------------------------------------- call
    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, torch.Tensor]:
        paired = "student_input_ids" in features[0]
        ids_key, am_key = ("student_input_ids", "student_attention_mask") if paired else ("input_ids", "attention_mask")
        ids, am = self._ids_and_mask([f[ids_key] for f in features], [f[am_key] for f in features])
        # one predicate for "this position has a target": not the pad id, and at / after the row's first speech_bos
        has_target = torch.ones_like(ids, dtype=torch.bool) if self.pad_token_id is None else ids != self.pad_token_id
        bos = self._speech_bos_id()
        if bos is not None:
            has_target &= (ids == bos).cumsum(-1) > 0
        batch = {"input_ids": ids, "attention_mask": am, "labels": torch.where(has_target, ids, torch.full_like(ids, -100))}
        if features[0].get("teacher_input_ids") is not None:
            batch["teacher_input_ids"], batch["teacher_attention_mask"] = self._ids_and_mask(
                [f["teacher_input_ids"] for f in features], [f["teacher_attention_mask"] for f in features])
        if features[0].get("teacher_top_k_v") is not None:
            batch["teacher_top_k_v"] = self._topk_grid([f["teacher_top_k_v"] for f in features], ids.size(1), 0.0)
            batch["teacher_top_k_i"] = self._topk_grid([f["teacher_top_k_i"] for f in features], ids.size(1), 0)
        return batch
