#!/bin/bash
# ONE command for whoever has `cargo` + a Fish Speech checkpoint: produce the reference dumps that turn "parity unpinned" into a pin.
#
#   tools/make_reference_dumps.sh /path/to/fish-speech.rs /path/to/checkpoints/fish-speech-1.5 ["text to speak"] [fish-version]
#
# Runs the REFERENCE's own binaries on its CPU backend (f32; no features):
#   llama_generate --temp 0 (greedy: Sampling::ArgMax, fish_speech_core/src/bin/llama_generate.rs:23-28,158-205) -> ref_codes.npy (8, n) u32
#   vocoder (fish_speech_core/src/bin/vocoder.rs:80-107)                                                         -> ref.wav
# and writes tests/golden/reference_dumps/{ref_codes.npy, prompt.npy, ref.wav, meta.json, llama_generate.log}.  prompt.npy is the (9, L)
# prompt the binary printed as "Input tokens:" (row 0; a text-only prompt has zeros in the 8 codebook rows, text/prompt.rs:53-104).
# Nothing of the reference's SOURCE is copied: the dumps are data produced by running it.  Afterwards, on an MI355X box:
#   python -m pytest tests/test_pin_reference_gpu.py -m gpu -q          (skips, saying so, while the directory is absent)
# That test is the only step between DESIGN.md's "parity unpinned" and "pinned" (the reference's own method: integer comparison of .npy
# dumps with tolerance 0, tests/e2e/allclose_indices.py:24-53).
set -euo pipefail
REF="${1:?path to a fish-speech.rs checkout (needs cargo)}"
CKPT="$(cd "${2:?checkpoint directory (model.safetensors, config.json, tokenizer.json, firefly-gan-vq-fsq-8x1024-21hz-generator.safetensors)}" && pwd)"
TEXT="${3:-The quick brown fox jumps over the lazy dog near the quiet river bank.}"
VER="${4:-1.5}"
MAXNEW="${MAXNEW:-256}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$HERE/tests/golden/reference_dumps"
mkdir -p "$OUT"
command -v cargo >/dev/null || { echo "cargo not found: run this where the reference builds" >&2; exit 2; }
( cd "$REF" && cargo run --release --bin llama_generate -- --checkpoint "$CKPT" --fish-version "$VER" --temp 0 --text "$TEXT" \
      --max-new-tokens "$MAXNEW" --out-path "$OUT/ref_codes.npy" ) | tee "$OUT/llama_generate.log"
( cd "$REF" && cargo run --release --bin vocoder -- --checkpoint "$CKPT" --fish-version "$VER" -i "$OUT/ref_codes.npy" -o "$OUT/ref.wav" )
python3 "$HERE/tools/reference_dumps_meta.py" "$OUT" "$CKPT" "$TEXT" "$VER" "$MAXNEW"
