"""Deterministic, language-independent synthetic inputs for the BASELINE.json configs.

Generator: SplitMix64(seed); a base is "ACGT"[x >> 62].  Everything is
drawn from ONE sequential stream per config, in the order documented in each function, so
any other implementation of SplitMix64 reproduces the same files.
"""
import math

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


class SplitMix64:
    def __init__(self, seed):
        self.state = np.uint64(seed)
        self._buf = np.empty(0, dtype=np.uint64)      # outputs generated ahead for one-at-a-time callers (same stream)
        self._at = 0

    def _gen(self, n):
        with np.errstate(over="ignore"):
            k = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + k * _GAMMA
            self.state = self.state + np.uint64(n) * _GAMMA
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            return z ^ (z >> np.uint64(31))

    def next(self, n=1):
        """next n outputs as a uint64 array.  Single draws are served from a block generated ahead -- the outputs and
        their order are those of the plain generator, only the batching differs (noisy_read draws ~1.1 numbers per base)."""
        left = self._buf.size - self._at
        if n == 1:
            if left == 0:
                self._buf, self._at, left = self._gen(4096), 0, 4096
                self._lst = self._buf.tolist()
            self._at += 1
            return self._buf[self._at - 1:self._at]
        if left == 0:
            return self._gen(n)
        head = self._buf[self._at:self._at + min(n, left)]
        self._at += head.size
        return head if head.size == n else np.concatenate([head, self._gen(n - head.size)])

    def take(self):
        """one output as a Python int (the same stream as next(1))"""
        if self._at == self._buf.size:
            self.next(1)
            return self._lst[self._at - 1]
        self._at += 1
        return self._lst[self._at - 1]

    def bases(self, n, chunk=1 << 24):
        """n bases; drawn in chunks (same stream, same result) so that 10^9 bases do not need 10^10 bytes of temporaries"""
        if n <= chunk:
            return _ACGT[(self.next(n) >> np.uint64(62)).astype(np.int64)].tobytes()
        out = np.empty(n, dtype=np.uint8)
        for lo in range(0, n, chunk):
            k = min(chunk, n - lo)
            out[lo:lo + k] = _ACGT[(self.next(k) >> np.uint64(62)).astype(np.int64)]
        return out.tobytes()

    def uniform(self, n=1):
        return (self.next(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def noisy_read(rng, src, m, sub=0.05, indel=0.01):
    """m bases copied from `src` (bytes) with substitutions and single-base indels.

    Per emitted/consumed base one uniform draw u: u < sub -> substitute (one more draw picks one of
    the 3 other bases); sub <= u < sub+indel/2 -> delete the source base; sub+indel/2 <= u < sub+indel
    -> insert a random base (one more draw) before the source base.
    """
    out = bytearray()
    p = 0
    acgt = b"ACGT"
    while len(out) < m and p < len(src):
        u = (rng.take() >> 11) * (1.0 / (1 << 53))
        c = src[p]
        if u < sub:
            k = acgt.index(c)
            out.append(acgt[(k + 1 + (rng.take() >> 62) % 3) % 4])
            p += 1
        elif u < sub + indel / 2:
            p += 1
        elif u < sub + indel:
            out.append(acgt[rng.take() >> 62])
        else:
            out.append(c)
            p += 1
    while len(out) < m:
        out.append(acgt[rng.take() >> 62])
    return bytes(out)


def config_1k(n_refs=1000, ref_len=2000, read_len=150, seed=1, jitter=False):
    """configs[0]/[1]: one 150 bp read vs 1k synthetic 2 kbp references (3.0e8 cells).

    Stream order: [jitter: n_refs draws for the lengths U[0.9n,1.1n]] the references back to back,
    one draw for the read's offset in reference 0, then noisy_read's draws.
    """
    rng = SplitMix64(seed)
    if jitter:
        lens = (ref_len * 0.9 + rng.uniform(n_refs) * (ref_len * 0.2)).astype(np.int64)
    else:
        lens = np.full(n_refs, ref_len, dtype=np.int64)
    blob = rng.bases(int(lens.sum()))
    offs = np.concatenate([[0], np.cumsum(lens)])
    refs = [blob[offs[k]:offs[k + 1]] for k in range(n_refs)]
    span = max(1, len(refs[0]) - int(read_len * 1.2))
    start = int(rng.next(1)[0] % np.uint64(span))
    read = noisy_read(rng, refs[0][start:], read_len)
    return refs, [read]


def config_ncbi(n_refs, read_len=150, seed=2, mu=7.3834, sigma=0.7675, lo=50, hi=100000):
    """configs[2] shape: lengths ~ round(LogNormal(mu, sigma)) clipped to [lo, hi]
    (median 1,609 / mean ~2,160 bp, README.md:39-40); one noisy read cut from reference 0.

    Stream order: 2*n_refs draws (Box-Muller pairs u1,u2 per reference), the references, offset, read.
    """
    rng = SplitMix64(seed)
    u = rng.uniform(2 * n_refs)
    u1 = np.maximum(u[0::2], 1e-300)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u[1::2])
    lens = np.clip(np.rint(np.exp(mu + sigma * z)), lo, hi).astype(np.int64)
    blob = rng.bases(int(lens.sum()))
    offs = np.concatenate([[0], np.cumsum(lens)])
    refs = [blob[offs[k]:offs[k + 1]] for k in range(n_refs)]
    src = max(refs[:64], key=len)
    span = max(1, len(src) - int(read_len * 1.2))
    start = int(rng.next(1)[0] % np.uint64(span))
    read = noisy_read(rng, src[start:], read_len)
    return refs, [read]


def config_multi_read(n_refs, n_reads, read_len=150, seed=3, **kw):
    """configs[3] shape: n_reads noisy reads, each cut from a random reference of a config_ncbi set."""
    refs, _ = config_ncbi(n_refs, read_len, seed, **kw)
    rng = SplitMix64(seed ^ 0x5EED)
    reads = []
    for _ in range(n_reads):
        r = refs[int(rng.next(1)[0] % np.uint64(n_refs))]
        span = max(1, len(r) - int(read_len * 1.2))
        start = int(rng.next(1)[0] % np.uint64(span))
        reads.append(noisy_read(rng, r[start:], read_len))
    return refs, reads


def config_long(n_pairs=4, length=10000, seed=4, sub=0.10, indel=0.02):
    """configs[4] shape: `length` x `length` pairs, second sequence = first with 10% subs + 2% indels.
    Returned as (refs, reads) lists of equal length; pair k is (refs[k], reads[k])."""
    rng = SplitMix64(seed)
    refs, reads = [], []
    for _ in range(n_pairs):
        a = rng.bases(length)
        refs.append(a)
        reads.append(noisy_read(rng, a, length, sub, indel))
    return refs, reads


def cells(refs, reads):
    return sum(len(r) for r in refs) * sum(len(q) for q in reads)
