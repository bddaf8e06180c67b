"""Independent pure-Python restatement of the sampler's RNG chain, written from the published algorithm descriptions (NOT from
oracle/): ChaCha (Bernstein; any even round count), PCG32 XSH-RR (pcg-random.org reference), rand_core 0.6 `seed_from_u64` and
`BlockRng::{next_u32, next_u64}`, rand 0.8.5 `UniformFloat<f32>` and `WeightedIndex<f32>`.  tests/test_oracle_known_answers.py pins it
against public vectors (ChaCha8/12/20 "TC1" all-zero key/IV keystreams of the Strombergson test-vector draft; the pcg32 demo output
for seed 42 / stream 54) and then pins oracle/'s C++ against it; make_golden.py uses it for the sampled-rollout fixture."""
import struct

M32 = 0xFFFFFFFF


def _rotl(v, c):
    return ((v << c) & M32) | (v >> (32 - c))


def chacha_block(key8, counter, rounds, nonce2=(0, 0)):
    """16 output words; state = "expand 32-byte k" | key | 64-bit block counter | 64-bit nonce (the original djb layout that
    rand_chacha uses: words 12-13 counter, 14-15 stream id)."""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key8) + [counter & M32, (counter >> 32) & M32, nonce2[0], nonce2[1]]
    w = list(s)

    def qr(a, b, c, d):
        w[a] = (w[a] + w[b]) & M32; w[d] = _rotl(w[d] ^ w[a], 16)
        w[c] = (w[c] + w[d]) & M32; w[b] = _rotl(w[b] ^ w[c], 12)
        w[a] = (w[a] + w[b]) & M32; w[d] = _rotl(w[d] ^ w[a], 8)
        w[c] = (w[c] + w[d]) & M32; w[b] = _rotl(w[b] ^ w[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(w[i] + s[i]) & M32 for i in range(16)]


def keystream_hex(key8, rounds, nblocks=1):
    return b"".join(struct.pack("<16I", *chacha_block(key8, c, rounds)) for c in range(nblocks)).hex()


PCG_MUL = 6364136223846793005
M64 = (1 << 64) - 1


def pcg32_demo(seed, seq, n):
    """pcg32_srandom_r(seed, seq) + n x pcg32_random_r (output from the OLD state)"""
    state, inc = 0, ((seq << 1) | 1) & M64
    out = []

    def step():
        nonlocal state
        old = state
        state = (old * PCG_MUL + inc) & M64
        x = (((old >> 18) ^ old) >> 27) & M32
        r = old >> 59
        return ((x >> r) | (x << ((32 - r) & 31))) & M32

    step()
    state = (state + seed) & M64
    step()
    for _ in range(n):
        out.append(step())
    return out


def seed_from_u64(state):
    """rand_core 0.6 SeedableRng::seed_from_u64: PCG32 with increment 11634580027462260723, output from the ADVANCED state,
    one u32 (little endian) per 4 seed bytes -> the 8 ChaCha key words"""
    key = []
    for _ in range(8):
        state = (state * PCG_MUL + 11634580027462260723) & M64
        x = (((state >> 18) ^ state) >> 27) & M32
        r = state >> 59
        key.append(((x >> r) | (x << ((32 - r) & 31))) & M32)
    return key


class StdRng:
    """rand 0.8.5 StdRng = ChaCha12Rng behind BlockRng with a 64-word (4-block) buffer"""

    def __init__(self, seed_u64=None, key=None):
        self.key = list(key) if key is not None else seed_from_u64(seed_u64)
        self.counter, self.buf, self.idx = 0, [0] * 64, 64

    def _refill(self):
        self.buf = [w for b in range(4) for w in chacha_block(self.key, self.counter + b, 12)]
        self.counter += 4
        self.idx = 0

    def next_u32(self):
        if self.idx >= 64:
            self._refill()
        v = self.buf[self.idx]
        self.idx += 1
        return v

    def next_u64(self):
        if self.idx < 63:
            lo, hi = self.buf[self.idx], self.buf[self.idx + 1]
            self.idx += 2
        elif self.idx >= 64:
            self._refill()
            lo, hi = self.buf[0], self.buf[1]
            self.idx = 2
        else:
            lo = self.buf[63]
            self._refill()
            hi = self.buf[0]
            self.idx = 1
        return (hi << 32) | lo


def _f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _from_bits(b):
    return struct.unpack("<f", struct.pack("<I", b))[0]


def weighted_index_sample(rng, weights):
    """WeightedIndex::<f32>::new(weights) then .sample(rng): cumulative f32 sums (last one = total), UniformFloat::new(0, total)
    (scale decremented by ulps until scale * max_rand + low < high), value in [1, 2) from the top 23 bits of ONE u32, first index
    whose cumulative weight exceeds the draw"""
    cum, total = [], _f32(weights[0])
    for w in weights[1:]:
        cum.append(total)
        total = _f32(total + _f32(w))
    max_rand = _f32(_from_bits((M32 >> 9) | (127 << 23)) - 1.0)
    scale = total
    while _f32(_f32(scale * max_rand) + 0.0) >= total:
        scale = _from_bits(_bits(scale) - 1)
    v12 = _from_bits((rng.next_u32() >> 9) | (127 << 23))
    chosen = _f32(_f32(_f32(v12 - 1.0) * scale) + 0.0)
    lo, hi = 0, len(cum)
    while lo < hi:
        mid = (lo + hi) // 2
        if cum[mid] <= chosen:
            lo = mid + 1
        else:
            hi = mid
    return lo
