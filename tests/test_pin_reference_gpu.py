"""GPU: THE pin.  Runs tests/pin_against_reference.py on dumps produced by the reference's own binaries (tools/make_reference_dumps.sh:
`llama_generate --temp 0` + `vocoder` on its CPU backend) when tests/golden/reference_dumps/ exists, and SKIPS -- saying exactly that --
while it does not: the development image has no Rust toolchain and no checkpoint, so the directory is absent in this repository and
DESIGN.md keeps the line "parity unpinned against the reference binary".  With the dumps present this test is the difference between
that line and "pinned": HIP f32 codes == reference codes with tolerance 0 (the reference's own method, tests/e2e/allclose_indices.py:24-53),
the CPU oracle likewise on its first frames, and fs_codec_decode of the reference codes within 1e-4 RMS / 2 LSB of the reference WAV.
The checkpoint is read from meta.json's path (or FISHRT_REFERENCE_CKPT); neither it nor the dumps are reference SOURCE."""
import json
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DUMPS = os.path.join(ROOT, "tests", "golden", "reference_dumps")


def _meta():
    p = os.path.join(DUMPS, "meta.json")
    if not os.path.exists(p):
        pytest.skip("tests/golden/reference_dumps/ is absent: no dump of the reference binaries has been made for this repository "
                    "(tools/make_reference_dumps.sh needs cargo + a checkpoint) -- parity stays UNPINNED against the reference binary")
    with open(p) as f:
        m = json.load(f)
    m["checkpoint"] = os.environ.get("FISHRT_REFERENCE_CKPT", m["checkpoint"])
    if not os.path.isdir(m["checkpoint"]):
        pytest.skip(f"reference dumps are present but the checkpoint directory {m['checkpoint']} is not on this box (set FISHRT_REFERENCE_CKPT)")
    return m


def test_hip_and_oracle_reproduce_the_reference_dumps():
    m = _meta()
    import pin_against_reference as pin
    argv = ["--checkpoint", m["checkpoint"], "--codes", os.path.join(DUMPS, "ref_codes.npy"), "--prompt", os.path.join(DUMPS, "prompt.npy"),
            "--fish-version", m["fish_version"], "--max-new-tokens", str(m["max_new_tokens"]), "--repetition-penalty", str(m["repetition_penalty"])]
    if os.path.exists(os.path.join(DUMPS, "ref.wav")):
        argv += ["--wav", os.path.join(DUMPS, "ref.wav")]
    assert pin.main(argv) == 0, "MISMATCH against the reference binaries' dumps (see the harness output above)"
