#!/usr/bin/env python3
"""Checks the dump of tools/ubench/dfma_mul against big-integer arithmetic: out == a * b * 2^-260 mod p and out < 2p for every pair
(2^20 random pairs < 2p plus the edge set {0, 1, p-1, p, 2p-1}^2 in the first 25 lanes).   python dfma_check.py dump.bin"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
n, L = int(raw[0]), int(raw[1])
p = sum(int(raw[2 + i]) << (52 * i) for i in range(L))
assert p == 21888242871839275222246405745257275088696311157297823662689037894645226208583
ab = raw[2 + L:2 + L + 10 * n].reshape(n, 10).astype(object)
out = raw[2 + L + 10 * n:].reshape(n, L).astype(object)
sh = np.array([1 << (52 * i) for i in range(L)], dtype=object)
a, b, o = (ab[:, :5] * sh).sum(axis=1), (ab[:, 5:] * sh).sum(axis=1), (out * sh).sum(axis=1)
rinv = pow(1 << 260, -1, p)
bad = 0
for i in range(n):
    ok = int(o[i]) < 2 * p and (int(o[i]) - int(a[i]) * int(b[i]) * rinv) % p == 0 and all(int(x) < (1 << 52) for x in out[i])
    if not ok:
        bad += 1
        if bad < 5:
            print("MISMATCH lane", i, hex(int(a[i])), hex(int(b[i])), hex(int(o[i])))
print(f"checked {n} products (edge set in lanes 0..24): {bad} mismatches; max operand bits {max(int(x).bit_length() for x in a[:1000])}")
sys.exit(1 if bad else 0)
