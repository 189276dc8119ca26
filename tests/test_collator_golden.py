"""The PRODUCT collator (speech_distill_amd.collator.ProcessedDataCollator) against the reference's own output
(fixture G3 = /root/reference/data.py:219-278 run on 8 ragged rows, one of them without speech_bos): integer / index
work, so every key must be bit-exact, dtype included."""
import numpy as np
import torch

from conftest import load_golden
from speech_distill_amd.collator import ProcessedDataCollator

KEYS = ("input_ids", "attention_mask", "labels", "teacher_input_ids", "teacher_attention_mask", "teacher_top_k_v",
        "teacher_top_k_i")


class DuckTok:
    """What the reference's collator needs from a tokenizer (data.py:352-360): pad token + encode(speech_bos)."""
    pad_token = "<|semantic_token_end|>"

    def __init__(self, pad_id, bos_id):
        self.pad_token_id, self.bos_id = pad_id, bos_id

    def encode(self, text, add_special_tokens=False):
        assert text == "<|semantic_token_start|>"
        return [self.bos_id]


def g3_features(z, as_lists=False):
    feats = []
    for r in range(int(z["n"])):
        ids = z[f"in_{r}_ids"].tolist()
        v, i = z[f"in_{r}_v"], z[f"in_{r}_i"]
        feats.append({"student_input_ids": ids, "student_attention_mask": [1] * len(ids),
                      "teacher_input_ids": list(ids), "teacher_attention_mask": [1] * len(ids),
                      "teacher_top_k_v": torch.from_numpy(v) if not as_lists else v,
                      "teacher_top_k_i": torch.from_numpy(i) if not as_lists else i})
    return feats


def test_product_collator_equals_reference_output_g3():
    z = load_golden("g3_collator.npz")
    V, bos, pad = [int(x) for x in z["meta"]]
    for as_lists in (False, True):  # arrow hands the columns over as numpy arrays, the tests above as tensors
        batch = ProcessedDataCollator(DuckTok(pad, bos), pad_token_id=pad)(g3_features(z, as_lists))
        assert set(batch) == set(KEYS)
        for k in KEYS:
            got, want = batch[k].numpy(), z["out_" + k]
            assert got.dtype == want.dtype, (k, got.dtype, want.dtype)
            np.testing.assert_array_equal(got, want, err_msg=k)


def test_product_collator_quirks_pinned_by_g3():
    """Q2 (the closing <|semantic_token_end|> equals the pad token, so it is never a target) and the speech-mask rule
    (a row without speech_bos is all -100) hold in the fixture itself -- so the equality above covers them."""
    z = load_golden("g3_collator.npz")
    V, bos, pad = [int(x) for x in z["meta"]]
    batch = ProcessedDataCollator(DuckTok(pad, bos), pad_token_id=pad)(g3_features(z))
    ids, lab = batch["input_ids"], batch["labels"]
    assert bool((lab[ids == pad] == -100).all())
    no_bos = [(r, bool((ids[r] == bos).any())) for r in range(ids.size(0))]
    assert any(not has for _, has in no_bos)
    for r, has in no_bos:
        if not has:
            assert bool((lab[r] == -100).all())
        else:
            first = int((ids[r] == bos).nonzero()[0])
            assert bool((lab[r, :first] == -100).all()) and int(lab[r, first]) == bos
