"""Seeded synthetic workloads -- shared by tests and bench.py.

The reference's tests/benches draw from rand::StdRng, whose stream cannot be
reproduced without that crate, so the build uses its own counter-based
SplitMix64: value(seed, i) = finalize(seed + (i+1) * GAMMA), i.e. exactly the
i-th output of a sequential SplitMix64 seeded with `seed`.  numpy and torch
versions produce identical streams.
"""
import numpy as np

GAMMA = 0x9E3779B97F4A7C15
M1 = 0xBF58476D1CE4E5B9
M2 = 0x94D049BB133111EB
MASK = (1 << 64) - 1


def splitmix64_np(seed, start, count):
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + 1 + count, dtype=np.uint64)
        z = np.uint64(seed & MASK) + i * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(M2)
        return z ^ (z >> np.uint64(31))


def _s64(x):
    x &= MASK
    return x - (1 << 64) if x >= (1 << 63) else x


def splitmix64_torch(seed, start, count, device):
    """Same stream as splitmix64_np, returned as int64 bit patterns on `device`."""
    import torch
    i = torch.arange(start + 1, start + 1 + count, dtype=torch.int64, device=device)
    z = i * _s64(GAMMA) + _s64(seed)

    def lsr(x, k):  # logical shift right on int64
        return (x >> k) & ((1 << (64 - k)) - 1)

    z = (z ^ lsr(z, 30)) * _s64(M1)
    z = (z ^ lsr(z, 27)) * _s64(M2)
    return z ^ lsr(z, 31)


def dna_text_np(n, seed):
    """t[i] = 1 + (rng & 3) for i < n-1; t[n-1] = 0   (config 1/2 text)."""
    t = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    for a in range(0, n, step):
        k = min(step, n - a)
        t[a:a + k] = (splitmix64_np(seed, a, k) & np.uint64(3)).astype(np.uint8) + 1
    t[n - 1] = 0
    return t


def byte_text_np(n, seed):
    """t[i] = 1 + rng % 255; t[n-1] = 0   (config 4 text)."""
    t = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    for a in range(0, n, step):
        k = min(step, n - a)
        t[a:a + k] = (splitmix64_np(seed, a, k) % np.uint64(255)).astype(np.uint8) + 1
    t[n - 1] = 0
    return t


def repetitive_text_np(n, seed, base_len=1 << 12, mut_per_1024=10):
    """base block repeated, ~1 % point mutations (config 4b: the case RLFM exists for)."""
    base = byte_text_np(base_len + 1, seed)[:base_len]
    t = np.tile(base, n // base_len + 1)[:n].copy()
    r = splitmix64_np(seed + 1, 0, n)
    mut = (r & np.uint64(1023)) < np.uint64(mut_per_1024)
    t[mut] = ((r[mut] >> np.uint64(16)) % np.uint64(255)).astype(np.uint8) + 1
    t[n - 1] = 0
    return t


def substring_patterns_np(text, npat, m, seed):
    """npat patterns of length m, substrings of text at rng % (n-1-m)  (config 2)."""
    n = len(text)
    pos = (splitmix64_np(seed, 0, npat) % np.uint64(n - 1 - m)).astype(np.int64)
    idx = pos[:, None] + np.arange(m, dtype=np.int64)[None, :]
    flat = text[idx].reshape(-1).astype(np.uint8)
    off = np.arange(npat + 1, dtype=np.uint64) * np.uint64(m)
    return flat, off, pos


def random_patterns_np(npat, m, sigma, seed):
    """uniform random over 1..sigma  (config 1 / 2b: exercises the early exit)."""
    flat = (splitmix64_np(seed, 0, npat * m) % np.uint64(sigma)).astype(np.uint8) + 1
    off = np.arange(npat + 1, dtype=np.uint64) * np.uint64(m)
    return flat, off


def ragged_patterns_np(npat, mmax, sigma, seed):
    """pattern lengths 0..mmax (ragged, includes empty)."""
    lens = (splitmix64_np(seed, 0, npat) % np.uint64(mmax + 1)).astype(np.uint64)
    off = np.zeros(npat + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens, dtype=np.uint64)
    tot = int(off[-1])
    flat = (splitmix64_np(seed + 1, 0, max(tot, 1)) % np.uint64(sigma)).astype(np.uint8) + 1
    return flat, off


def dna_text_torch(n, seed, device):
    import torch
    t = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 26
    for a in range(0, n, step):
        k = min(step, n - a)
        t[a:a + k] = ((splitmix64_torch(seed, a, k, device) & 3) + 1).to(torch.uint8)
    t[n - 1] = 0
    return t


def byte_text_torch(n, seed, device):
    import torch
    t = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 26
    for a in range(0, n, step):
        k = min(step, n - a)
        z = splitmix64_torch(seed, a, k, device)
        # unsigned z % 255 on int64 bit patterns: split into high/low halves
        hi = (z >> 32) & 0xFFFFFFFF
        lo = z & 0xFFFFFFFF
        r = ((hi % 255) * ((1 << 32) % 255) + (lo % 255)) % 255
        t[a:a + k] = (r + 1).to(torch.uint8)
    t[n - 1] = 0
    return t


def umod_torch(z, m):
    """unsigned (z mod m) for int64 bit patterns z, m < 2^31."""
    hi = (z >> 32) & 0xFFFFFFFF
    lo = z & 0xFFFFFFFF
    return ((hi % m) * ((1 << 32) % m) + (lo % m)) % m


def substring_patterns_torch(text, npat, m, seed):
    import torch
    n = text.numel()
    z = splitmix64_torch(seed, 0, npat, text.device)
    pos = umod_torch(z, n - 1 - m)
    idx = pos[:, None] + torch.arange(m, dtype=torch.int64, device=text.device)[None, :]
    flat = text[idx].reshape(-1).contiguous()
    off = torch.arange(npat + 1, dtype=torch.int64, device=text.device) * m
    return flat, off, pos


def repetitive_text_torch(n, seed, device, base_len=1 << 12, mut_per_1024=10):
    """same text as repetitive_text_np, generated on `device`."""
    import torch
    base = byte_text_torch(base_len + 1, seed, device)[:base_len]
    t = base.repeat(n // base_len + 1)[:n].contiguous()
    step = 1 << 26
    for a in range(0, n, step):
        k = min(step, n - a)
        r = splitmix64_torch(seed + 1, a, k, device)
        mut = (r & 1023) < mut_per_1024
        hi = (r >> 16) & ((1 << 48) - 1)            # logical shift of the 64-bit pattern
        val = (umod_torch(hi, 255) + 1).to(torch.uint8)
        seg = t[a:a + k]
        seg[mut] = val[mut]
    t[n - 1] = 0
    return t
