"""The reduction order of the sequence-mode GEMM (rwkv.cpp_amd/csrc/prefill.hip), modelled in NumPy float32.

A row sum of the single-token kernel is 64 partials P[b mod 64] reduced by the xor-butterfly 32, 16, 8, 4, 2, 1. The GEMM walks the
same binary tree depth-first (leaves in bit-reversed order, a stack of partial sums, as many merges after leaf number c as c has
trailing one bits), and a GEMM with too few output tiles cuts the walk into 2 / 4 / 8 parts (subtrees below level 3) whose sums a
second kernel adds in tree order. All three must perform the SAME additions: this test checks that they give the same bits."""
import numpy as np
import pytest

f32 = np.float32


def butterfly(P):
    """xor-butterfly of kdev.h (wave reduction): every lane ends with the total; lane 0's value is the row sum."""
    v = P.astype(f32).copy()
    for s in (32, 16, 8, 4, 2, 1):
        v = (v + v[np.arange(64) ^ s]).astype(f32)
    return v[0]


def bitrev6(c):
    return int(f"{c:06b}"[::-1], 2)


def walk(P, a0=0, a1=8):
    """depth-first walk of outer iterations a0 .. a1 - 1 (8 leaves each): the value of that subtree"""
    stack = []  # (level, value)
    for c in range(8 * a0, 8 * a1):
        cur = f32(P[bitrev6(c)])
        level = 0
        # merge while the stack top is a completed sibling subtree of the same level
        while stack and stack[-1][0] == level:
            _, left = stack.pop()
            cur = f32(left + cur)
            level += 1
        stack.append((level, cur))
    assert len(stack) == 1
    return stack[0][1]


def split_walk(P, split):
    parts = [walk(P, z * (8 // split), (z + 1) * (8 // split)) for z in range(split)]
    w = 1
    while w < split:
        for z in range(0, split, 2 * w):
            parts[z] = f32(parts[z] + parts[z + w])
        w *= 2
    return parts[0]


@pytest.mark.parametrize("seed", range(8))
def test_walk_and_split_walks_equal_the_butterfly(seed):
    rng = np.random.default_rng(seed)
    P = (rng.standard_normal(64) * 10.0 ** rng.integers(-3, 4, size=64)).astype(f32)
    want = butterfly(P)
    assert walk(P).tobytes() == want.tobytes()
    for split in (2, 4, 8):
        assert split_walk(P, split).tobytes() == want.tobytes(), split


def test_butterfly_lane0_is_the_tree_with_root_on_bit0():
    """lane 0 of the butterfly = ((..(P0 + P32) + (P16 + P48)..) ..): the root splits the leaves by bit 0 of their index."""
    rng = np.random.default_rng(99)
    P = rng.standard_normal(64).astype(f32)

    def tree_root_bit0(idx, bit):
        if bit == 6:
            return f32(P[idx])
        lo = tree_root_bit0(idx, bit + 1)
        hi = tree_root_bit0(idx | (1 << bit), bit + 1)
        return f32(lo + hi)

    assert butterfly(P).tobytes() == tree_root_bit0(0, 0).tobytes()
