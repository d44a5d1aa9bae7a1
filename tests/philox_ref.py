"""Host restatement of the dropout mask definition of include/vbx.h (Philox4x32-10, Salmon et al. SC'11; 16-bit lots) in numpy:
the GPU tests compare the kernels' keep bits with it, the CPU test pins the generator itself against the published known-answer
vectors of the Random123 distribution (kat_vectors: philox4x32 10 rounds)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy arrays of uint32 counters; scalar key.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def thr16(p):
    return int(min(65535, max(1, int((1.0 - p) * 65536.0 + 0.5))))


def lots(words):
    """[..., 4] uint32 words of calls -> [..., 8] 16-bit lots (lot e = half e % 2, low first, of word e // 2)."""
    w = np.stack(words, axis=-1).astype(np.uint32)
    return np.stack([(w[..., e // 2] >> np.uint32(16 * (e % 2))) & np.uint32(0xFFFF) for e in range(8)], axis=-1)


def attn_keep(BH, Np, p, seed, stream):
    """bool [BH, Np (q), Np (key)]: counter (4 * (key // 32) + (key % 32) // 8, q, bh, stream), lot key % 8."""
    bh, q, key = np.meshgrid(np.arange(BH), np.arange(Np), np.arange(Np), indexing="ij")
    call = 4 * (key // 32) + (key % 32) // 8
    l = lots(philox4x32_10(call, q, bh, np.full_like(call, stream), seed & 0xFFFFFFFF, seed >> 32))
    lot = np.take_along_axis(l, (key % 8)[..., None], axis=-1)[..., 0]
    return lot < thr16(p)


def rows_keep(rows, cols, p, seed, stream):
    """bool [rows, cols]: counter (col // 8, row, 0, stream), lot col % 8."""
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    l = lots(philox4x32_10(c // 8, r, np.zeros_like(r), np.full_like(r, stream), seed & 0xFFFFFFFF, seed >> 32))
    lot = np.take_along_axis(l, (c % 8)[..., None], axis=-1)[..., 0]
    return lot < thr16(p)


def unpack_bits(words, n):
    """int32 / uint32 array [..., W] -> bool [..., n] (bit i % 32 of word i // 32)."""
    w = np.asarray(words).astype(np.int64) & 0xFFFFFFFF
    idx = np.arange(n)
    return ((w[..., idx // 32] >> (idx % 32)) & 1).astype(bool)
