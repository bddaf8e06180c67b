"""GPU: tests/pin_against_reference.py (the harness that pins fishrt and the oracle against dumps of the REFERENCE binaries) on a synthetic
checkpoint round trip, so that it cannot rot: a checkpoint directory is written by the reference's tensor names (model.safetensors,
config.json, token_config.json, the codec file), the "reference dumps" are produced by the CPU oracle standing in for llama_generate /
vocoder (.npy codes in the reference's layout, 16-bit WAV through the reference's `as i16` rule), and the harness must report PINNED;
a corrupted dump must make it report MISMATCH."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import fishrt
from fishrt import config as fcfg, wav as fwav
from oracle import oracle as orc
import pin_against_reference as pin
from test_safetensors_gpu import _codec_tensors, _lm_tensors, _save, SEED


def _checkpoint(tmp_path):
    cfg, tok = fcfg.TINY, fcfg.TINY_TOKENS
    d = tmp_path / "ckpt"
    d.mkdir()
    _save(_lm_tensors(cfg, bf16=False), str(d / pin.LM_FILE), False)
    with open(d / "config.json", "w") as f:
        json.dump({k: cfg[k] for k in cfg}, f)
    with open(d / "token_config.json", "w") as f:
        json.dump(tok, f)
    _save(_codec_tensors(64, 1234), str(d / pin.CODEC_FILE), False)
    return d, cfg, tok


def test_harness_pins_a_synthetic_checkpoint_and_catches_a_corrupted_dump(tmp_path, capsys):
    d, cfg, tok = _checkpoint(tmp_path)
    prompt = np.zeros((9, 9), np.uint32)
    prompt[0] = [3, 401, 17, 464, 399, 12, 250, 7, 300]
    prompt[1:, 1] = np.arange(8); prompt[1:, 3] = 63 - np.arange(8)
    np.save(tmp_path / "prompt.npy", prompt)
    M = 40
    # the stand-in for `llama_generate --temp 0`: the oracle with the same (f32) tensors -> (8, n) codes
    o = orc.OracleLM(dict(cfg, **tok)).load_synthetic(SEED)
    ref = o.generate(prompt, M, temp=0.0, repetition_penalty=1.2)
    assert ref.shape[1] >= 8
    np.save(tmp_path / "ref_codes.npy", ref.astype(np.int64))  # (candle writes u32; the e2e script loads whatever and casts)
    # the stand-in for the `vocoder` binary: oracle PCM -> 16-bit WAV by the reference's `(x.clamp(-1, 1) * 32767.0) as i16`
    codes = np.minimum(ref, 999).astype(np.uint32)
    np.save(tmp_path / "ref_codes.npy", codes.astype(np.int64))
    pcm = orc.OracleCodec(tiny=True).load_synthetic(1234).decode(codes)
    with open(tmp_path / "ref.wav", "wb") as f:
        fwav.write_pcm_as_wav(f, pcm.astype(np.float32), 44100)
    # (the LM check runs on the unclamped codes; the vocoder check on what the vocoder was given)
    np.save(tmp_path / "lm_codes.npy", ref.astype(np.int64))
    base = ["--checkpoint", str(d), "--prompt", str(tmp_path / "prompt.npy"), "--max-new-tokens", str(M), "--fish-version", "1.5",
            "--oracle-frames", "16"]
    assert pin.main(base + ["--codes", str(tmp_path / "lm_codes.npy")]) == 0
    assert "RESULT: PINNED" in capsys.readouterr().out
    assert pin.main(base + ["--codes", str(tmp_path / "ref_codes.npy"), "--wav", str(tmp_path / "ref.wav"), "--codec-channel-div", "8",
                            "--oracle-frames", "0"]) == (0 if np.array_equal(codes, ref) else 1)
    out = capsys.readouterr().out
    assert "HIP vocoder" in out and "rms diff" in out
    rms = float(out.split("rms diff")[1].split()[0])
    assert rms < 1e-4
    # a dump that is off by one code in one frame must be caught
    bad = ref.astype(np.int64).copy()
    bad[3, 5] = (bad[3, 5] + 1) % cfg["codebook_size"]
    np.save(tmp_path / "bad.npy", bad)
    assert pin.main(base + ["--codes", str(tmp_path / "bad.npy")]) == 1
    assert "FIRST DIFFERENCE at frame 5" in capsys.readouterr().out
