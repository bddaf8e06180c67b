"""Host-side formats around the hot path: PromptEncoder layout (prompt.rs:28-156), .npy voice prompts
(prompt.rs:159-198), voice directory (server/lib/utils/mod.rs:17-55), WAV writer (audio/wav.rs:27-58).  Expected values are
written out by hand from the reference's format strings and tensor ops."""
import io
import json
import os
import re
import struct

import numpy as np
import pytest

from fishrt import prompt as fp
from fishrt import wav as fw

G = os.path.join(os.path.dirname(__file__), "golden")


class FakeTok:
    """special tokens -> fixed ids, every other character -> 1000 + ord"""
    SPECIAL = {"<|im_start|>": 1, "<|im_end|>": 2, "<|voice|>": 3, "<|semantic|>": 7, "<|semantic:0|>": 500}

    def __init__(self, drop=()):
        self.sp = {k: v for k, v in self.SPECIAL.items() if k not in drop}

    def encode(self, text):
        out, pat = [], re.compile("|".join(re.escape(k) for k in self.sp))
        i = 0
        while i < len(text):
            m = pat.match(text, i)
            if m:
                out.append(self.sp[m.group(0)]); i = m.end()
            else:
                out.append(1000 + ord(text[i])); i += 1
        return out

    def token_to_id(self, t):
        return self.sp.get(t)


def _ids(s):
    return [1000 + ord(c) for c in s]


def test_encode_text_layout():
    e = fp.PromptEncoder(FakeTok(), 8, fp.FISH_1_5)
    a = e.encode_text("user", "hi")
    assert a.dtype == np.uint32 and a.shape == (9, 1 + 5 + 2 + 1)
    assert list(a[0]) == [1] + _ids("user\n") + _ids("hi") + [2]
    assert not a[1:].any()
    b = e.encode_text("assistant")  # content None: no <|im_end|>
    assert list(b[0]) == [1] + _ids("assistant\n")


def test_encode_vq_fish15_and_fish14():
    codes = np.arange(8 * 3, dtype=np.uint32).reshape(8, 3) % 11
    e15 = fp.PromptEncoder(FakeTok(), 8, fp.FISH_1_5)
    v = e15.encode_vq(codes)
    pre = [1] + _ids("assistant\n") + [3]
    assert list(v[0]) == pre + [500 + int(c) for c in codes[0]] + [2]
    L = len(pre)
    assert np.array_equal(v[1:, L:L + 3], codes) and not v[1:, :L].any() and not v[1:, L + 3:].any()
    assert np.array_equal(e15.encode_vq(None), e15.tokenize_text("<|im_start|>assistant\n<|voice|>"))
    e14 = fp.PromptEncoder(FakeTok(), 8, fp.FISH_1_4)
    w = e14.encode_vq(codes)
    pre14 = [1] + _ids("assistant\n")  # no <|voice|> for Fish <= 1.4
    assert list(w[0]) == pre14 + [7, 7, 7] + [2]
    assert np.array_equal(w[1:, len(pre14):len(pre14) + 3], codes + 1)  # codes shifted by one (prompt.rs:88-91)
    e14b = fp.PromptEncoder(FakeTok(drop=("<|semantic|>",)), 8, fp.FISH_1_4)
    assert list(e14b.encode_vq(codes)[0][len(_ids("assistant\n")) + 1:-1]) == [5, 5, 5]  # unwrap_or(5); '<|semantic|>' absent
    with pytest.raises(RuntimeError):
        e15.encode_vq(np.zeros((7, 3), np.uint32))  # wrong codebook count cannot be concatenated


def test_encode_sequence_conditioning_and_kv_cache_flag():
    e = fp.PromptEncoder(FakeTok(), 8, fp.FISH_1_5)
    spk = e.encode_conditioning_prompt("ref text", np.ones((8, 4), np.uint32))
    n, out = e.encode_sequence(["a", "bc"], "sys", spk, assume_kv_cache=False)
    sysp = e.encode_text("system", "sys")
    assert n == sysp.shape[1] + spk.shape[1]
    tail = lambda c: np.concatenate([e.encode_text("user", c), e.encode_vq(None)], 1)
    assert np.array_equal(out[0], np.concatenate([sysp, spk, tail("a")], 1))
    assert np.array_equal(out[1], np.concatenate([sysp, spk, tail("bc")], 1))
    n2, out2 = e.encode_sequence(["a", "bc"], "sys", spk, assume_kv_cache=True)
    assert n2 == n and np.array_equal(out2[0], out[0]) and np.array_equal(out2[1], tail("bc"))  # conditioning only on chunk 0
    n3, out3 = e.encode_sequence(["x"], None, None)
    assert n3 == 0 and np.array_equal(out3[0], tail("x"))
    with pytest.raises(RuntimeError, match="Input text cannot be empty"):
        e.encode_sequence([])


def test_load_prompt_text_npy_shapes(tmp_path):
    ref = np.load(os.path.join(G, "default_voice_codes.npy"))
    assert ref.shape[0] == 8
    p = tmp_path / "v.npy"
    np.save(p, ref.astype(np.int64))
    assert np.array_equal(fp.load_prompt_text(p, 8), ref.astype(np.uint32)) and fp.load_prompt_text(p, 8).dtype == np.uint32
    np.save(p, ref[None].astype(np.int32))  # ghost leading dimension is accepted
    assert np.array_equal(fp.load_prompt_text(p, 8), ref.astype(np.uint32))
    with pytest.raises(RuntimeError, match="Expected 4 codebooks but got 8"):
        fp.load_prompt_text(p, 4)
    np.save(p, ref[0])
    with pytest.raises(RuntimeError, match="Incorrect prompt token dimensions"):
        fp.load_prompt_text(p, 8)


def test_voice_directory(tmp_path):
    ref = np.load(os.path.join(G, "default_voice_codes.npy")).astype(np.uint32)
    np.save(tmp_path / "default.npy", ref)
    np.save(tmp_path / "alice.npy", ref[:, :10])
    (tmp_path / "index.json").write_text(json.dumps({"speakers": {"default": "hello", "alice": "yo"}}))
    tok = FakeTok()
    speakers, default = fp.load_speaker_prompts(tmp_path, tok)
    e = fp.PromptEncoder(tok, 8, fp.FISH_1_5)
    assert set(speakers) == {"default", "alice"}
    assert np.array_equal(default, e.encode_conditioning_prompt("hello", ref))
    assert np.array_equal(speakers["alice"], e.encode_conditioning_prompt("yo", ref[:, :10]))
    (tmp_path / "index.json").write_text(json.dumps({"speakers": {"alice": "yo"}}))
    with pytest.raises(RuntimeError, match="No default speaker"):
        fp.load_speaker_prompts(tmp_path, tok)
    os.remove(tmp_path / "index.json")
    with pytest.raises(RuntimeError, match="Failed to open speaker index.json"):
        fp.load_speaker_prompts(tmp_path, tok)


def test_hf_tokenizer_adapter():
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import models, pre_tokenizers
    vocab = {"[UNK]": 0, "user": 1, "hi": 2}
    t = tokenizers.Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    t.pre_tokenizer = pre_tokenizers.Whitespace()
    t.add_special_tokens(["<|im_start|>", "<|im_end|>", "<|semantic:0|>"])
    a = fp.HFTokenizer(t)
    e = fp.PromptEncoder(a, 8, fp.FISH_1_5)
    ids = list(e.encode_text("user", "hi")[0])
    assert ids == [t.token_to_id("<|im_start|>"), 1, 2, t.token_to_id("<|im_end|>")]
    assert a.token_to_id("<|semantic:0|>") == t.token_to_id("<|semantic:0|>") and a.token_to_id("nope") is None


def test_wav_writer_header_and_samples():
    x = np.array([0.0, 0.5, -0.5, 1.5, -2.0, 0.99999, -3.05e-5], np.float32)
    buf = io.BytesIO()
    n = fw.write_pcm_as_wav(buf, x, 44100)
    b = buf.getvalue()
    assert n == len(b) == 44 + 2 * len(x)
    assert b[:4] == b"RIFF" and struct.unpack("<I", b[4:8])[0] == len(b) - 8 and b[8:16] == b"WAVEfmt "
    assert struct.unpack("<IHHIIHH", b[16:36]) == (16, 1, 1, 44100, 88200, 2, 16)
    assert b[36:40] == b"data" and struct.unpack("<I", b[40:44])[0] == 2 * len(x)
    got = np.frombuffer(b[44:], "<i2")
    assert list(got) == [0, 16383, -16383, 32767, -32767, 32766, 0]  # clamp, * 32767, truncate toward zero
    buf2 = io.BytesIO()
    fw.write_pcm_as_wav(buf2, np.array([1, -2, 3], np.int16), 8000)  # i16 passes through
    assert list(np.frombuffer(buf2.getvalue()[44:], "<i2")) == [1, -2, 3]


def test_reference_dump_tooling_parses_what_llama_generate_prints(tmp_path):
    """tools/make_reference_dumps.sh (the one-command pin): its second half turns the reference binary's stdout into prompt.npy + meta.json.  A
    fabricated log in the format of fish_speech_core/src/bin/llama_generate.rs:70-88 (Debug print of a Vec<u32>, possibly wrapped) keeps it working."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("reference_dumps_meta", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "reference_dumps_meta.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    toks = [100257, 9125, 198, 96945, 704, 279, 3984, 1495, 100258, 100257, 882, 198, 15339, 100258, 100257, 78191, 198, 100264]
    log = ("Text: \"hello\"\nSpeaker conditioning size: [9, 9]\nLoaded prompt with shape [9, 18]\nInput tokens:\n[" +
           ", ".join(str(t) for t in toks[:9]) + ",\n " + ", ".join(str(t) for t in toks[9:]) + "]\nInput prompt:\n<|im_start|>system...\n")
    (tmp_path / "llama_generate.log").write_text(log)
    np.save(tmp_path / "ref_codes.npy", np.arange(8 * 5, dtype=np.uint32).reshape(8, 5))
    meta, prompt = mod.write_meta(str(tmp_path), "/ckpt", "hello", "1.5", 256)
    assert prompt.shape == (9, 18) and prompt.dtype == np.uint32 and prompt[0].tolist() == toks and not prompt[1:].any()
    assert np.array_equal(np.load(tmp_path / "prompt.npy"), prompt)
    m = json.load(open(tmp_path / "meta.json"))
    assert m["frames"] == 5 and m["prompt_positions"] == 18 and m["fish_version"] == "1.5" and m["max_new_tokens"] == 256
    with pytest.raises(ValueError):
        mod.parse_input_tokens("no tokens here")
