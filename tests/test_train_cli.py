"""scripts/train.py keeps the reference's flag names (train.py:430-596), so that a launch line
written for the reference parses here.  CPU only: argument parsing, no models."""
import importlib.util
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the reference's command-line surface (train.py:430-596), restated as data: flag -> default there
REFERENCE_FLAGS = {
    "--teacher_model": None, "--student_model": None, "--dataset_path": None, "--output_dir": None, "--max_length": 512,
    "--teacher_prefix": "<|task_podcast|><|SPEAKER_0|>", "--student_prefix": "", "--use_lora": False, "--lora_r": 32,
    "--lora_alpha": 64, "--use_rslora": True, "--init_lora_weights": "pissa", "--temperature": 2.0, "--alpha": 0.5,
    "--learning_rate": 5e-5, "--num_train_epochs": 3, "--warmup_steps": 1000, "--bf16": True,
    "--gradient_checkpointing": True, "--test_size": 10, "--report_to": None, "--dataloader_num_workers": None,
    "--dataloader_prefetch_factor": 2, "--text_bos": "<|text_start|>", "--text_eos": "<|text_end|>",
    "--text_prefix": '{"en": "", "zh": "", "yue": "<|Yue|>"}', "--speech_bos": "<|semantic_token_start|>",
    "--speech_eos": "<|semantic_token_end|>", "--pad_token": "<|semantic_token_end|>", "--top_k": 128,
    "--load_teacher_in_4bit": False, "--load_teacher_in_8bit": False,
}
DEST = {"--num_train_epochs": "epochs", "--test_size": "eval_samples"}


def _load():
    spec = importlib.util.spec_from_file_location("sd_train_cli", os.path.join(ROOT, "scripts", "train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _parse(mod, argv):
    old = sys.argv
    sys.argv = ["train.py"] + argv
    try:
        return mod.parse_args()
    finally:
        sys.argv = old


def test_every_reference_flag_parses_with_the_reference_default():
    mod = _load()
    src = open(os.path.join(ROOT, "scripts", "train.py")).read()
    declared = set(re.findall(r'"(--[a-z_0-9]+)"', src))
    assert set(REFERENCE_FLAGS) <= declared, sorted(set(REFERENCE_FLAGS) - declared)
    cfg = _parse(mod, [])
    for flag, default in REFERENCE_FLAGS.items():
        if default is None:  # paths, and the two defaults that differ on purpose (report_to, dataloader_num_workers)
            continue
        assert getattr(cfg, DEST.get(flag, flag[2:])) == default, flag


def test_a_reference_launch_line_parses():
    mod = _load()
    cfg = _parse(mod, "--teacher_model /m/SoulX-Podcast-1.7B --student_model /m/qwen3-0.6b --dataset_path /d/processed "
                      "--output_dir /o --max_length 512 --temperature 2.0 --alpha 0.5 --learning_rate 5e-5 "
                      "--num_train_epochs 3 --warmup_steps 1000 --bf16 --gradient_checkpointing --test_size 10 "
                      "--report_to none --dataloader_num_workers 2 --dataloader_prefetch_factor 2 --top_k 128 "
                      "--load_teacher_in_8bit".split())
    assert cfg.epochs == 3 and cfg.eval_samples == 10 and cfg.dataloader_num_workers == 2 and cfg.load_teacher_in_8bit
    assert cfg.warmup_steps == 1000 and cfg.bf16 and cfg.gradient_checkpointing and cfg.top_k == 128


def test_extract_script_column_builder_keeps_fp16_int32_and_order(tmp_path):
    """The output columns of scripts/extract_teacher_logits.py (extract_teacher_logits.py:120-141): per-sample unpadded
    [len, K] arrays stored as list<list<float16>> / list<list<int32>>, also across arrow chunk boundaries."""
    import numpy as np
    import pyarrow as pa
    from datasets import Dataset, load_from_disk
    spec = importlib.util.spec_from_file_location("sd_extract_cli", os.path.join(ROOT, "scripts", "extract_teacher_logits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(0)
    lens = [5, 1, 9, 3, 7]
    v = [rng.standard_normal((n, 4)).astype(np.float16) for n in lens]
    i = [rng.integers(0, 1000, (n, 4)).astype(np.int32) for n in lens]
    ds = Dataset.from_list([{"teacher_input_ids": list(range(n))} for n in lens])
    ds = ds.add_column("teacher_top_k_v", mod.nested_column(v, pa.float16(), max_chunk_elems=40))
    ds = ds.add_column("teacher_top_k_i", mod.nested_column(i, pa.int32(), max_chunk_elems=40))
    ds.save_to_disk(str(tmp_path / "o"))
    back = load_from_disk(str(tmp_path / "o"))
    assert str(back.features["teacher_top_k_v"]) == "List(List(Value('float16')))"
    assert str(back.features["teacher_top_k_i"]) == "List(List(Value('int32')))"
    for r, n in enumerate(lens):
        np.testing.assert_array_equal(np.asarray(back[r]["teacher_top_k_v"], dtype=np.float16), v[r])
        np.testing.assert_array_equal(np.asarray(back[r]["teacher_top_k_i"]), i[r])
    # and the collator reads them back (data.py:329-372 path)
    from speech_distill_amd.collator import ProcessedDataCollator
    rows = [dict(back[r], student_input_ids=list(range(lens[r])), student_attention_mask=[1] * lens[r],
                 teacher_attention_mask=[1] * lens[r]) for r in range(len(lens))]

    class _T:
        pad_token, pad_token_id = "<pad>", 0

        def encode(self, *a, **k):
            return [3]
    batch = ProcessedDataCollator(_T(), pad_token_id=0)(rows)
    assert batch["teacher_top_k_v"].shape == (5, 9, 4) and batch["teacher_top_k_i"].dtype in (__import__("torch").int32, __import__("torch").int64)
