"""TEST INFRASTRUCTURE ONLY -- restatement of `sha/src/tables.rs` (SHA lookup-table
generators).  Pinned by the five KATs in `sha/src/tables.rs:168-211`.

Quirk preserved: `Bits::BITS_LEN` is 8 for u8, u16 AND u32 (tables.rs:30-38), so
`rotation` acts on the low 8 bits only and `from_bits` returns an 8-bit value.
"""
from __future__ import annotations

BITS_LEN = 8  # tables.rs:31,34,37

LIMBS = {
    "test": (4, 2),   # tables.rs:161-166 (TestLimbs, u8)
    "short": (6, 5),  # tables.rs:63-68
    "long": (12, 10),  # tables.rs:70-75
}


def _to_bits(word: int):
    """tables.rs:18-23 (MSB first, BITS_LEN bits)."""
    return [(word >> n) & 1 for n in range(BITS_LEN - 1, -1, -1)]


def _from_bits(bits):
    """tables.rs:25-28."""
    acc = 0
    for b in bits:
        acc = (acc << 1) | b
    return acc


def rotation(word: int, n: int) -> int:
    """tables.rs:98-103: Vec::rotate_right on the MSB-first bit vector."""
    bits = _to_bits(word)
    rot = n % len(bits)
    bits = bits[-rot:] + bits[:-rot] if rot else bits
    return _from_bits(bits)


def combine(first: int, second: int, x: int, y: int, z: int) -> int:
    """tables.rs:91-96."""
    return (x << (2 * second)) | (y << second) | z


def create_synthesis_table(limbs: str, f):
    """tables.rs:78-89."""
    first, second = LIMBS[limbs]
    return [
        (x, y, z, f(x, y, z))
        for x in range(1 << first)
        for y in range(1 << second)
        for z in range(1 << second)
    ]


def _rotation_table(limbs: str, r1: int, r2: int, r3: int):
    first, second = LIMBS[limbs]

    def f(x, y, z):
        w = combine(first, second, x, y, z)
        return rotation(w, r1) ^ rotation(w, r2) ^ rotation(w, r3)

    return create_synthesis_table(limbs, f)


def create_rot0_table(limbs: str):
    """tables.rs:113-115."""
    return _rotation_table(limbs, 2, 13, 22)


def create_rot1_table(limbs: str):
    """tables.rs:117-119."""
    return _rotation_table(limbs, 6, 11, 25)


def create_maj_table(limbs: str):
    """tables.rs:121-126."""
    return create_synthesis_table(limbs, lambda x, y, z: (x & y) ^ (x & z) ^ (y & z))


def create_ch_table(limbs: str):
    """tables.rs:128-133 (`!x` on u64: the high bits are masked off by `& z`)."""
    return create_synthesis_table(limbs, lambda x, y, z: (x & y) ^ ((~x) & z))


def create_decomposition_table(limbs: str, k: int):
    """tables.rs:135-154."""
    first, second = LIMBS[limbs]
    full = first + 2 * second
    out = []
    for a in range(1 << k):
        a_mod = a % (1 << full)
        x = a_mod >> (2 * second)
        y = a_mod >> second
        for i in range(first):
            y &= ~(1 << (i + second))
        z = a_mod
        for i in range(first + second):
            z &= ~(1 << (i + second))
        out.append((a, x, y, z))
    return out


# ---- toy bit-vector SHA round (sha-reference/src/lib.rs, word.rs) -- restated only for its one KAT --------------------
# Words are lists of L bits.  Quirks preserved: `+` is bitwise XOR (word.rs:12-21, 98-108), the round has no K_t / W_t,
# and right_rotation(n) reads result[i] = self[(i + n L - n) mod L] (word.rs:36-42).
def toy_right_rotation(w, n: int):
    L = len(w)
    return [w[(i + n * L - n) % L] for i in range(L)]


def _toy_xor(*ws):
    return [sum(bits) & 1 for bits in zip(*ws)]


def toy_rot_0(w):
    """word.rs:44-46."""
    return _toy_xor(toy_right_rotation(w, 2), toy_right_rotation(w, 13), toy_right_rotation(w, 22))


def toy_rot_1(w):
    """word.rs:48-50."""
    return _toy_xor(toy_right_rotation(w, 6), toy_right_rotation(w, 11), toy_right_rotation(w, 25))


def toy_majority(a, b, c):
    """word.rs:59-71."""
    return [1 if x + y + z >= 2 else 0 for x, y, z in zip(a, b, c)]


def toy_choose(a, b, c):
    """word.rs:73-80."""
    return [y if x == 1 else z for x, y, z in zip(a, b, c)]


def toy_sha_round(octet):
    """sha-reference/src/lib.rs:18-33: octet = (a, b, c, d, e, f, g, h), each a list of L bits."""
    a, b, c, d, e, f, g, h = octet
    temp = _toy_xor(h, toy_rot_1(e), toy_choose(e, f, g))
    new_e = _toy_xor(d, temp)
    new_a = _toy_xor(temp, toy_rot_0(a), toy_majority(a, b, c))
    return (new_a, a, b, c, new_e, e, f, g)
