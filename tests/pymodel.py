"""Tiny pure-Python model of the scan, written from SURVEY.md section 3.2 (window-min + RLE by value) with Python
integers.  It is a third, independent statement used only to cross-check the C oracle's literal deque restatement on
small cases (test infrastructure)."""

CODE = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}


def enc(s):
    v = 0
    for ch in s.upper():
        v = (v << 2) | CODE[ch]
    return v


def rc(v, m):
    out = 0
    for _ in range(m):
        out = (out << 2) | (3 - (v & 3))
        v >>= 2
    return out


def masks(m, spaces, xor_mask):
    """right-aligned (2m-bit) xor mask and space mask"""
    W = (m + 31) // 32
    full = 0
    for i in range(W):  # RandomXOR.mask: every word = xor_mask, last partial word = xor_mask << (64 - (m%32)*2)
        if i == W - 1 and m % 32 != 0:
            w = (xor_mask << (64 - (m % 32) * 2)) & (2**64 - 1)
        else:
            w = xor_mask
        full = (full << 64) | w
    xm = full >> (64 * W - 2 * m)
    sm = (1 << (2 * m)) - 1
    for i in range(spaces):  # nucleotides m-2, m-4, ... from the left are zeroed
        pos_from_right = 1 + 2 * i
        sm &= ~(3 << (2 * pos_from_right))
    return xm, sm


def keys(seq, m, spaces, xor_mask, canonical):
    xm, sm = masks(m, spaces, xor_mask)
    out = []
    for p in range(m - 1, len(seq)):
        v = enc(seq[p - m + 1:p + 1])
        if canonical:
            v = min(v, rc(v, m))
        out.append((v ^ xm) & sm)
    return out


def supermers(seq, k, m, spaces, xor_mask, canonical):
    """-> list of (right-aligned key, start, length)"""
    if len(seq) < k:
        return []
    ks = keys(seq, m, spaces, xor_mask, canonical)
    w = k - m + 1
    mins = [min(ks[i:i + w]) for i in range(len(seq) - k + 1)]
    out = []
    i = 0
    while i < len(mins):
        j = i
        while j + 1 < len(mins) and mins[j + 1] == mins[i]:
            j += 1
        out.append((mins[i], i, (j - i + 1) + k - 1))
        i = j + 1
    return out


def left_align(v, m):
    """right-aligned 2m-bit value -> tuple of left-aligned 64-bit words"""
    W = (m + 31) // 32
    full = v << (64 * W - 2 * m)
    return tuple((full >> (64 * (W - 1 - i))) & (2**64 - 1) for i in range(W))
