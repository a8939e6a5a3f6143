"""Black-box table of where the reference build's float32 `sqrt` (torch CPU -> Intel MKL VML vsSqrt, VML_HA; convex_adam_MIND.py:179
through torch.optim.Adam) is NOT the correctly rounded root:   python tests/golden/make_mkl_sqrt_table.py  -> tests/golden/mkl_vssqrt_low.npz

Findings (exhaustive over all 2^32 non-negative float32 inputs, this container: torch 2.10 CPU, MKL 2024.2, AVX-512 code path):
  * the result never exceeds the correctly rounded root and is at most one ulp below it;
  * for normal inputs whether it is low depends only on (exponent parity, 23-bit mantissa): 39 167 odd- and 59 788 even-exponent classes;
  * denormal inputs follow their own pattern (52 462 of 2^23); 0 -> 0, inf -> inf;
  * independent of position, vector length, stride and thread count.
Stored: two bit maps (little-endian bit order), `normal` [2^24 bits, key = parity << 23 | mantissa] and `denormal` [2^23 bits, key =
mantissa]: MKL_sqrt(x) = IEEE_sqrt(x) - 1 ulp where the bit is set.  Inputs are fed and outputs compared; nothing is disassembled.
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def low_mask(e):
    bits = np.arange(1 << 23, dtype=np.uint32) | np.uint32(e << 23)
    x = bits.view(np.float32)
    a = torch.sqrt(torch.from_numpy(x)).numpy()
    b = np.sqrt(x)                                      # hardware sqrtps: correctly rounded
    d = a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)
    assert d.min() >= -1 and d.max() <= 0
    return d != 0


def main():
    torch.set_num_threads(8)
    odd, even = low_mask(127), low_mask(126)
    for e in range(1, 255):                             # the whole normal range follows the period-2 pattern
        assert np.array_equal(low_mask(e), odd if e & 1 else even), e
    den = low_mask(0)
    normal = np.concatenate([even, odd])                # key = parity << 23 | mantissa
    path = os.path.join(HERE, "mkl_vssqrt_low.npz")
    np.savez_compressed(path, normal=np.packbits(normal, bitorder="little"), denormal=np.packbits(den, bitorder="little"),
                        counts=np.array([int(even.sum()), int(odd.sum()), int(den.sum())]))
    print("wrote", path, os.path.getsize(path), "bytes; low classes (even, odd, denormal):", int(even.sum()), int(odd.sum()), int(den.sum()))


if __name__ == "__main__":
    main()
