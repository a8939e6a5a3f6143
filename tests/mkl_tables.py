"""The two call sites of the hot path that the oracle does not restate as an algorithm: `exp` at MINDSSC (convex_adam_utils.py:63)
and `sqrt` inside torch.optim.Adam (convex_adam_MIND.py:179).  The reference build evaluates both with Intel MKL VML (vsExp / vsSqrt,
VML_HA), whose results differ from the oracle's (SLEEF-style expf, IEEE sqrt) by at most one ulp.  Both deviations are position
independent, so they can be TABULATED exhaustively from the functions themselves (torch.exp / torch.sqrt on the CPU -- torch is on
every box, /root/reference is not needed): with the two tables the oracle reproduces the reference BIT FOR BIT through 80 Adam
iterations at the full benchmark size (tests/test_oracle_vs_golden.py::test_full_size_bit_identical_to_reference_with_mkl_tables).

Test infrastructure -- the product builds its own copy of the exp table from its DEVICE expf (convexadam_amd/reference_bits.py) and the
GPU tests check that the two tables agree entry for entry.

exp table: two bits per argument x <= 0, keyed by the bit pattern of |x| minus EXP_FIRST, for |x| in [2^-30, 128):
0 = torch.exp(x) == orc_expf(x), 1 = one ulp above, 2 = one ulp below (3 never occurs).  Below 2^-30 both give 1.0, at or beyond 128
both give 0.  310 M entries, 77.6 MB, ~15 s to build from torch (host_tables, cached under the temp directory).

MKL picks its code path by CPU model, so the tables are a property of the HOST: the Xeon that produced every golden of tests/golden
and the EPYC of the GPU boxes give different ones (exp: 16.1 M vs 9.6 M deviating arguments).  Comparisons with the reference
goldens therefore use the GOLDEN HOST's tables, committed as fixtures (golden_tables): tests/golden/mkl_vsexp_codes.xz (the 77.6 MB
table is long runs of equal codes: 2.2 MB as xz) and tests/golden/mkl_vssqrt_low.npz; tests/golden/mkl_tables.json holds their
hashes.  `python tests/mkl_tables.py --record` regenerates all three on the host the goldens come from.
"""
import hashlib
import json
import os
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _bits(f):
    return int(np.float32(f).view(np.uint32))


EXP_FIRST = _bits(2.0 ** -30)
EXP_COUNT = _bits(128.0) - EXP_FIRST
_CHUNK = 1 << 24


def pack2(code):
    """uint8 codes (0..3) -> 4 per byte, little-endian pairs."""
    n = code.size
    if n % 4:
        code = np.concatenate([code, np.zeros(4 - n % 4, np.uint8)])
    c = code.reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)


def build_exp_table(own_expf):
    """own_expf: float32 array -> float32 array (the exp the table corrects: the oracle's here)."""
    tbl = np.zeros((EXP_COUNT + 3) // 4, np.uint8)
    hist = np.zeros(4, np.int64)
    for off in range(0, EXP_COUNT, _CHUNK):
        n = min(_CHUNK, EXP_COUNT - off)
        x = -(np.arange(EXP_FIRST + off, EXP_FIRST + off + n, dtype=np.uint32).view(np.float32))
        host = torch.exp(torch.from_numpy(x.copy())).numpy()
        d = host.view(np.int32).astype(np.int64) - own_expf(x).view(np.int32).astype(np.int64)
        code = np.where(d == 0, 0, np.where(d == 1, 1, np.where(d == -1, 2, 3))).astype(np.uint8)
        hist += np.bincount(code, minlength=4)
        p = pack2(code)
        tbl[off // 4: off // 4 + p.size] = p
    if hist[3]:
        raise AssertionError("torch.exp differs from the restated expf by more than one ulp for %d arguments" % hist[3])
    return tbl, hist


def build_sqrt_table():
    """2-bit code table of torch.sqrt relative to the IEEE root (0 equal, 1 one ulp above, 2 one ulp below): 2 x 2^23 normal classes
    (key = exponent parity << 23 | mantissa; tests/golden/make_mkl_sqrt_table.py verified the period over every exponent on the golden
    host), then the 2^23 denormals."""
    def code(e):
        x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(e << 23)).view(np.float32)
        a = torch.sqrt(torch.from_numpy(x.copy())).numpy()
        d = a.view(np.int32).astype(np.int64) - np.sqrt(x).view(np.int32).astype(np.int64)
        assert d.min() >= -1 and d.max() <= 1
        return np.where(d == 1, 1, np.where(d == -1, 2, 0)).astype(np.uint8)
    return pack2(np.concatenate([code(126), code(127), code(0)]))


def sqrt_codes_from_low_bitmaps(normal, denormal):
    low = np.concatenate([np.unpackbits(normal, bitorder="little"), np.unpackbits(denormal, bitorder="little")]).astype(np.uint8) * 2
    return pack2(low)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


_cache = {}


def golden_tables():
    """Tables of the host that produced tests/golden (fixtures)."""
    if "g" not in _cache:
        import lzma
        with open(os.path.join(HERE, "golden", "mkl_vsexp_codes.xz"), "rb") as f:
            tbl = np.frombuffer(lzma.decompress(f.read()), np.uint8)
        q = np.load(os.path.join(HERE, "golden", "mkl_vssqrt_low.npz"))
        with open(os.path.join(HERE, "golden", "mkl_tables.json")) as f:
            ref = json.load(f)
        assert tbl.size == (EXP_COUNT + 3) // 4 and sha(tbl) == ref["exp_sha256"]
        codes = sqrt_codes_from_low_bitmaps(q["normal"], q["denormal"])
        assert sha(codes) == ref["sqrt_sha256"]
        _cache["g"] = dict(exp=tbl, exp_first=EXP_FIRST, exp_count=EXP_COUNT, sqrt=codes)
    return _cache["g"]


def host_tables(oracle):
    """{'exp': table, 'exp_first', 'exp_count', 'sqrt': codes, 'matches_golden_host': bool} for THIS host's torch."""
    if "t" in _cache:
        return _cache["t"]
    path = os.path.join(tempfile.gettempdir(), "cvx_mkl_exp_table_%s.npy" % torch.__version__.replace("+", "_"))
    tbl = None
    if os.path.exists(path):
        try:
            tbl = np.load(path)
            probe = -(np.arange(EXP_FIRST + 123 * _CHUNK // 128, EXP_FIRST + 123 * _CHUNK // 128 + 4096, dtype=np.uint32).view(np.float32))
            if tbl.size != (EXP_COUNT + 3) // 4 or not np.array_equal(torch.exp(torch.from_numpy(probe.copy())).numpy(), _apply(tbl, probe, oracle.expf)):
                tbl = None
        except Exception:
            tbl = None
    if tbl is None:
        tbl, _ = build_exp_table(oracle.expf)
        try:
            np.save(path, tbl)
        except OSError:
            pass
    sq = build_sqrt_table()
    with open(os.path.join(HERE, "golden", "mkl_tables.json")) as f:
        ref = json.load(f)
    t = dict(exp=tbl, exp_first=EXP_FIRST, exp_count=EXP_COUNT, sqrt=sq, exp_sha=sha(tbl), sqrt_sha=sha(sq))
    t["matches_golden_host"] = t["exp_sha"] == ref["exp_sha256"] and t["sqrt_sha"] == ref["sqrt_sha256"]
    _cache["t"] = t
    return t


def _apply(tbl, x, own_expf):
    """own_expf corrected by the table (numpy restatement of orc_mind_exp, for the cache probe and the tests)."""
    r = own_expf(x).view(np.uint32).copy()
    b = np.abs(x).view(np.uint32)
    k = (b.astype(np.int64) - EXP_FIRST)
    ok = (k >= 0) & (k < EXP_COUNT)
    kk = np.where(ok, k, 0)
    code = (tbl[kk >> 2] >> ((kk & 3) * 2).astype(np.uint8)) & 3
    code = np.where(ok, code, 0)
    r = r + (code == 1).astype(np.uint32) - (code == 2).astype(np.uint32)
    return r.view(np.float32)


if __name__ == "__main__":      # python tests/mkl_tables.py --record : writes tests/golden/mkl_tables.json for the host the goldens come from
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import oracle
    oracle.lib()
    tbl, hist = build_exp_table(oracle.expf)
    sq = build_sqrt_table()
    rec = dict(exp_sha256=sha(tbl), sqrt_sha256=sha(sq), exp_first=EXP_FIRST, exp_count=EXP_COUNT,
               exp_codes=[int(v) for v in hist], torch=torch.__version__,
               host="torch CPU / MKL VML as built into this torch; CPU flags decide MKL's code path")
    print(json.dumps(rec, indent=1))
    if "--record" in sys.argv:
        import lzma
        with open(os.path.join(HERE, "golden", "mkl_tables.json"), "w") as f:
            json.dump(rec, f, indent=1)
        with open(os.path.join(HERE, "golden", "mkl_vsexp_codes.xz"), "wb") as f:
            f.write(lzma.compress(tbl.tobytes(), preset=6))
