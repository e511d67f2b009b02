"""Pin the oracle's numpy.random restatement (oracle/np_random.c) against numpy itself.

numpy is the third-party arithmetic behind every random draw of RLToyEnv
(SURVEY.md §8c); the same numpy wheel is installed on the GPU box, so this pin
runs there too.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as ora


def _pair(seed):
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    st = ora.NpPCG64.from_words(ora.pcg_words(g))
    return g, st


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**63 - 1])
def test_next64_and_random(seed):
    L = ora.lib()
    g, st = _pair(seed)
    raw = g.bit_generator.random_raw(1000)
    mine = np.array([L.np_next64(C.byref(st)) for _ in range(1000)], dtype=np.uint64)
    assert np.array_equal(raw, mine)
    u = g.random(1000)
    mine = np.array([L.np_random(C.byref(st)) for _ in range(1000)])
    assert np.array_equal(u, mine)


@pytest.mark.parametrize("seed", [0, 7, 99])
def test_standard_normal_bit_exact(seed):
    L = ora.lib()
    g, st = _pair(seed)
    n = 400_000  # ~100 tail draws and ~5000 wedge rejections
    z = g.standard_normal(n)
    mine = np.array([L.np_standard_normal(C.byref(st)) for _ in range(n)])
    assert np.array_equal(z, mine)
    assert (np.abs(z) > 3.6541528853610088).sum() > 20  # tail branch exercised
    # and the generator ends in the same state
    assert np.array_equal(ora.pcg_words(g)[:4],
                          np.array([st.s_lo, st.s_hi, st.inc_lo, st.inc_hi], dtype=np.uint64))


def test_normal_scaled_matches():
    L = ora.lib()
    g, st = _pair(3)
    for sigma in (0.05, 0.3, 1.0):
        a = g.normal(0, sigma, 1000)
        b = np.array([0.0 + sigma * L.np_standard_normal(C.byref(st)) for _ in range(1000)])
        assert np.array_equal(a, b)


@pytest.mark.parametrize("lohi", [(-21, 22), (0, 360), (0, 2), (-29, 30), (0, 1), (5, 1000003)])
def test_integers_scalar_stream(lohi):
    """Scalar Generator.integers() calls interleaved with random(): exercises the
    buffered 32-bit half (has_uint32) exactly as ImageMultiDiscrete does."""
    L = ora.lib()
    lo, hi = lohi
    g, st = _pair(11)
    for i in range(3000):
        a = int(g.integers(lo, hi))
        b = int(L.np_integers(C.byref(st), lo, hi))
        assert a == b, (i, a, b)
        if i % 7 == 3:
            assert g.random() == L.np_random(C.byref(st))
    w = ora.pcg_words(g)
    assert (int(w[4]), int(w[5])) == (st.has32, st.u32)


def test_integers_float_bounds_like_reference():
    # image_multi_discrete.py:175 passes floats: integers(-21.0, 22.0)
    L = ora.lib()
    g, st = _pair(5)
    for _ in range(500):
        assert int(g.integers(-21.0, 22.0)) == int(L.np_integers(C.byref(st), -21, 22))


@pytest.mark.parametrize("n", [6, 8, 16])
def test_choice_with_p(n):
    L = ora.lib()
    g, st = _pair(21)
    rng = np.random.default_rng(1)
    for _ in range(300):
        noise = rng.uniform(0.01, 0.99)
        nxt = int(rng.integers(n))
        probs = np.ones(n) * noise / (n - 1)
        probs[nxt] = 1 - noise
        a = int(np.squeeze(g.choice(n, size=1, p=probs, replace=True)))
        cdf = np.zeros(n)
        L.np_build_cdf.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.np_build_cdf(probs.ctypes.data_as(C.c_void_p), n, cdf.ctypes.data_as(C.c_void_p))
        L.np_choice_cdf.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        b = L.np_choice_cdf(C.byref(st), cdf.ctypes.data_as(C.c_void_p), n)
        assert a == b


def test_float32_norm_semantics():
    """np.linalg.norm(float32 vector) == sqrt(float32(sum_double(float32(x_i*x_i)))) for
    n < 32 (OpenBLAS sdot scalar tail) - the summation the oracle and the kernels use."""
    rng = np.random.default_rng(1)
    f = np.float32
    for n in (1, 2, 4, 12, 31):
        for _ in range(500):
            x = rng.uniform(-10, 10, n).astype(f)
            acc = np.float64(0)
            for v in x:
                acc += np.float64(f(v * v))
            assert np.sqrt(f(acc)) == np.linalg.norm(x)


def test_philox_mode_normals_are_standard_normal():
    """The Philox mode's own Gaussian (float32 Box-Muller pairs, oracle/np_random.c): moments, a
    Kolmogorov-Smirnov test against N(0, 1), independence of the pair's two halves, tails, and pinned
    bit patterns (any change of the transform on either side shows here or in the GPU bit test)."""
    import scipy.stats as st
    z = ora.philox_normals(20261002, 0, 0, 0, 4096, 512).ravel()
    assert z.dtype == np.float64 and np.all(z == z.astype(np.float32))          # float32 values
    assert abs(z.mean()) < 3e-3 and abs(z.std() - 1.0) < 3e-3
    assert abs(st.skew(z)) < 0.01 and abs(st.kurtosis(z)) < 0.02
    assert st.kstest(z[:200000], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 3e-3                       # cos / sin halves of a pair
    assert 4.5 < np.abs(z).max() < 6.9
    assert abs((np.abs(z) > 3.0).mean() - 2 * st.norm.sf(3.0)) < 2e-4
    # streams are keyed: another env / tick / stream id gives other numbers, the same key the same
    a = ora.philox_normals(5, 7, 9, 1, 1, 8)
    assert np.array_equal(a, ora.philox_normals(5, 7, 9, 1, 1, 8))
    for other in ((5, 8, 9, 1), (5, 7, 10, 1), (5, 7, 9, 2), (6, 7, 9, 1), (5, 7, 9 + (1 << 32), 1)):
        assert not np.array_equal(a, ora.philox_normals(*other, 1, 8))
    import ctypes as C
    z0, z1 = C.c_float(), C.c_float()
    for w0, w1 in ((0, 0), (0xFFFFFFFF, 0xFFFFFFFF), (1, 0x40000000), (0x80000000, 0x20000000)):
        ora.lib().np_philox_box_muller(w0, w1, C.byref(z0), C.byref(z1))
        r2 = z0.value ** 2 + z1.value ** 2
        u1 = max(w0, 0.5) / 2.0 ** 32
        assert abs(r2 - (-2.0 * np.log(u1))) <= 2e-6 * max(r2, 1.0), (w0, w1)
        th = 2 * np.pi * w1 / 2.0 ** 32
        assert abs(z0.value - np.sqrt(r2) * np.cos(th)) < 3e-6 * max(np.sqrt(r2), 1) and \
            abs(z1.value - np.sqrt(r2) * np.sin(th)) < 3e-6 * max(np.sqrt(r2), 1), (w0, w1)


def test_philox_noise_words_of_discrete_envs():
    """Philox mode, discrete envs (the build's own definitions, mdpp_rng.hpp "one word per tick"): the word of tick t is word
    t & 3 of the block four ticks share; the transition-noise rule is noisy iff w < ceil(p 2^32), re-drawn state = the
    floor(w (S - 1) / T)-th of the other states -- P(noisy) = p, the others equally likely, like the reference's categorical
    (rl_toy_env.py:1604-1622); the reward normal of tick t is normal t & 3 of the block's two Box-Muller pairs."""
    seed, env = 99, 1234
    for t in range(16):
        blk = ora.philox_normals(seed, env, t >> 2, 13, 1, 4)[0]
        assert np.float32(ora.philox_tick_normal(seed, env, t, 13)) == np.float32(blk[t & 3])
    ws = np.array([ora.philox_tick_word(seed, env, t, 12) for t in range(4000)], dtype=np.uint64)
    assert len(set(ws.tolist())) > 3990 and abs(ws.mean() / 2 ** 32 - 0.5) < 0.02
    S, p, nxt = 5, 0.3, 2
    T = int(np.ceil(p * 2 ** 32))
    rng = np.random.default_rng(0)
    w = rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64)
    got = np.array([ora.philox_pnoise_state(int(x), p, S, nxt) for x in w[:20000]])
    want = np.where(w[:20000] >= T, nxt, 0)
    j = (w[:20000].astype(object) * (S - 1)) // T                      # exact integers
    want = np.where(w[:20000] >= T, nxt, np.array([int(v) + (int(v) >= nxt) for v in j]))
    assert np.array_equal(got, want)
    assert abs((got != nxt).mean() - p) < 0.01
    others = got[got != nxt]
    counts = np.bincount(others, minlength=S)
    assert counts[nxt] == 0 and (np.abs(counts[[0, 1, 3, 4]] / len(others) - 0.25) < 0.02).all()
    # boundaries: w = T - 1 is the last noisy word and lands on the last other state; w = T is quiet; S = 255 stays in range
    assert ora.philox_pnoise_state(T - 1, p, S, nxt) == 4 and ora.philox_pnoise_state(T, p, S, nxt) == nxt
    assert ora.philox_pnoise_state(0, p, S, 0) == 1 and ora.philox_pnoise_state(0, p, S, 3) == 0
    T9 = int(np.ceil(0.9 * 2 ** 32))
    assert ora.philox_pnoise_state(T9 - 1, 0.9, 255, 254) == 253 and ora.philox_pnoise_state(T9 - 1, 0.9, 255, 0) == 254
    assert ora.philox_pnoise_state(3, 1e-9, 8, 5) == 5                   # T = 5 <= S - 1: never noisy


def test_philox_noise_index_by_invariant_multiplication_is_the_exact_quotient():
    """The device forms floor(w (S - 1) / T) as the top 32 bits of w x M, M = ceil(2^64 (S - 1) / T) (mdpp_rng.hpp
    philox_pnoise_magic / philox_pnoise_index; the oracle divides).  The arithmetic restated with Python integers: equal for
    every w next to a quotient boundary and for random w, over noise probabilities from 1e-6 to 1 - 1e-9 and 2 to 255 states."""
    rng = np.random.default_rng(1)
    for p in (1e-6, 0.01, 0.1, 1.0 / 3.0, 0.5, 0.97, 1.0 - 1e-9):
        T = min(int(np.ceil(p * 2 ** 32)), 2 ** 32 - 1)
        for S in (2, 3, 8, 37, 255):
            if T <= S - 1:
                continue
            M = -((-(S - 1) << 64) // T)                                  # ceil
            assert M < 2 ** 64
            ws = set(int(x) for x in rng.integers(0, T, size=500))
            for k in range(0, S):
                b = -((-k * T) // (S - 1))                                  # first w whose quotient is k
                ws.update(x for x in (b - 1, b, b + 1) if 0 <= x < T)
            ws.update((0, T - 1))
            for w in ws:
                top = (w * (M >> 32) + ((w * (M & 0xFFFFFFFF)) >> 32)) >> 32     # the device's two multiplies
                assert top == (w * (S - 1)) // T == (w * M) >> 64, (p, S, w)
                assert top <= S - 2


def test_ziggurat_tables_one_source_two_copies_equal_numpys():
    """oracle/ and csrc/ each hold np_ziggurat_tables.inc (the checker must not share files with the product): both are
    the ONE rendering tools/refgen/extract_ziggurat.py makes, and the constants in them are the ones inside the numpy that is
    installed here -- so a wrong constant cannot hide by being wrong on both sides."""
    import importlib.util
    import os
    import re
    import struct
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("extract_ziggurat", os.path.join(root, "tools", "refgen", "extract_ziggurat.py"))
    z = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(z)
    texts = [open(os.path.join(root, d)).read() for d in z.DESTS]
    assert texts[0] == texts[1]
    ki, wi, fi = z.tables()
    version = re.search(r"numpy (\S+) ziggurat", texts[0]).group(1)
    assert texts[0] == z.render(ki, wi, fi, version)
    # ... and parsed back from the text, bit for bit
    body = texts[0].split("#define ")
    got_ki = [int(x, 16) for x in re.findall(r"0x([0-9A-F]{16})ULL", body[1])]
    got_wi = [float.fromhex(x) for x in re.findall(r"0x[0-9a-f.]+p[-+]\d+", body[2])]
    got_fi = [float.fromhex(x) for x in re.findall(r"0x[0-9a-f.]+p[-+]\d+", body[3])]
    assert tuple(got_ki) == tuple(ki) and len(got_ki) == 256
    assert struct.pack("<256d", *got_wi) == struct.pack("<256d", *wi) and struct.pack("<256d", *got_fi) == struct.pack("<256d", *fi)
