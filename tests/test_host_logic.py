"""-m "not gpu": host-side table construction (the repo's own Huffman builder that feeds the
benchmarks; reference counterpart huffTree.h:55-94 + load_data.h:40-47) and test helpers."""
import numpy as np

import hvlc
import oracle_lib as O
from util import extract_bits, dist_weights


def _prefix_free(cw, cl):
    codes = sorted((format(int(c), "b").zfill(int(l)) for c, l in zip(cw, cl) if l > 0))
    return all(not b.startswith(a) for a, b in zip(codes, codes[1:]))


def test_huffman_matches_reference_total_on_g1(golden):
    hist = np.bincount(np.tile(golden["period"], 4), minlength=256)
    cw, cl = hvlc.huffman_table(hist)
    assert _prefix_free(cw, cl)
    # any Huffman code is optimal: same total bits as the reference's table, whatever the tie-breaking
    assert int((hist * cl).sum()) == int((hist * golden["cl"]).sum()) == 2330672
    assert ((cl > 0) == (hist > 0)).all()
    assert abs(sum(2.0 ** -int(l) for l in cl if l) - 1.0) < 1e-12


def test_huffman_degenerate_and_limited():
    cw, cl = hvlc.huffman_table(np.zeros(256))
    assert cl.sum() == 0
    h = np.zeros(256); h[7] = 5
    cw, cl = hvlc.huffman_table(h)
    assert cl[7] == 1 and cl.sum() == 1
    fib = np.zeros(256)
    a, b = 1, 1
    for i in range(60):
        fib[i] = a
        a, b = b, a + b
    cw, cl = hvlc.huffman_table(fib)
    assert cl.max() <= 31 and _prefix_free(cw, cl)
    assert sum(2.0 ** -int(l) for l in cl if l) <= 1.0 + 1e-12
    assert (cw < (np.uint64(1) << cl.astype(np.uint64))).all()


def test_named_distributions_have_expected_shape(golden):
    w = dist_weights("H1.0")
    p = w / w.sum()
    H = -(p[p > 0] * np.log2(p[p > 0])).sum()
    assert 0.9 < H < 1.1
    cw, cl = hvlc.huffman_table((p * 2 ** 40).astype(np.uint64))
    assert list(cl[:25]) == list(range(1, 25)) + [24]
    w4 = dist_weights("H4.0")
    assert (w4 > 0).sum() == 16
    cdf = hvlc.cdf32_from_counts(w4)
    x = O.gen_symbols(1 << 16, 0x5EED0003, cdf)
    assert set(np.unique(x)) <= set(range(3, 256, 16))
    wg = dist_weights("H2.2", golden["period"])
    assert (wg > 0).sum() == 32


def test_extract_bits_roundtrip():
    rng = np.random.default_rng(1)
    w = rng.integers(0, 1 << 32, 64, dtype=np.uint64).astype(np.uint32)
    bits = "".join(format(int(v), "032b") for v in w)
    for start, n in [(0, 32), (5, 64), (31, 1), (33, 95), (100, 0), (640, 1000)]:
        got = extract_bits(w, start, n)
        s = bits[start:start + n]
        s = s + "0" * (-len(s) % 32)
        exp = [int(s[i:i + 32], 2) for i in range(0, len(s), 32)]
        assert [int(v) for v in got] == exp


def test_c_abi_table_builder_matches_the_python_one(golden):
    """hvlc_build_table (host C++ inside libhvlc.so, no GPU needed) builds the same canonical table
    as hvlc.huffman_table for real, random, skewed and degenerate histograms."""
    rng = np.random.default_rng(11)
    hists = [np.bincount(golden["period"], minlength=256), np.zeros(256, dtype=np.uint64)]
    h = np.zeros(256, dtype=np.uint64); h[200] = 9; hists.append(h)
    for _ in range(20):
        h = rng.integers(0, 1000, size=256).astype(np.uint64)
        h[rng.random(256) < 0.3] = 0
        hists.append(h)
    for _ in range(10):
        hists.append(np.floor(rng.pareto(0.6, size=256) * 10).astype(np.uint64))
    for h in hists:
        cw, cl = hvlc.build_table(h)
        pw, pl = hvlc.huffman_table(h)
        assert np.array_equal(cl, pl) and np.array_equal(cw, pw)
        assert _prefix_free(cw, cl) and ((cl > 0) == (np.asarray(h) > 0)).all()
    # length-limited: a Fibonacci histogram wants 59-bit codes
    fib = np.zeros(256, dtype=np.uint64); a, b = 1, 1
    for i in range(60):
        fib[i] = a; a, b = b, a + b
    for ml in (31, 16, 12):
        cw, cl = hvlc.build_table(fib, ml)
        assert cl.max() <= ml and _prefix_free(cw, cl) and sum(2.0 ** -int(l) for l in cl if l) <= 1.0 + 1e-12
        assert (cw < (np.uint64(1) << cl.astype(np.uint64))).all()
        pw, pl = hvlc.huffman_table(fib, ml)
        assert np.array_equal(cl, pl) and np.array_equal(cw, pw)
    # the reference's table has the same cost on its own test file
    hist = np.bincount(np.tile(golden["period"], 4), minlength=256).astype(np.uint64)
    _, cl = hvlc.build_table(hist)
    assert int((hist * cl).sum()) == int((hist * golden["cl"]).sum())


def test_length_limited_tables_are_optimal_not_just_valid():
    """A limit that bites must cost next to nothing when the excess is tiny (a geometric histogram: the unlimited code
    is 1, 2, 3, ... bits deep; limiting 40 symbols to 31 bits touches only symbols of probability < 2^-30) and the Kraft
    sum must stay 1 (no slack given away)."""
    for nsym, ml in ((40, 31), (30, 12), (60, 16)):
        h = np.zeros(256, dtype=np.uint64)
        for i in range(nsym):
            h[i] = 1 << (nsym - i)
        unl = hvlc.huffman_lengths(h, 255)             # deep enough for any 256-symbol tree: the unlimited Huffman code
        assert unl.max() > ml
        cw, cl = hvlc.build_table(h, ml)
        assert cl.max() <= ml and _prefix_free(cw, cl) and ((cl > 0) == (h > 0)).all()
        assert abs(sum(2.0 ** -int(l) for l in cl if l) - 1.0) < 1e-12
        cost, best = int((h * cl.astype(np.uint64)).sum()), int((h * unl.astype(np.uint64)).sum())
        assert best <= cost <= best * 1.01, (nsym, ml, cost, best)
        assert cl[0] == unl[0] == 1                    # the most frequent symbol keeps its 1-bit code
    # brute force on a small alphabet: package-merge is optimal among all length assignments with Kraft sum <= 1
    import itertools
    h = np.zeros(256, dtype=np.uint64); h[:6] = [50, 20, 10, 5, 2, 1]
    _, cl = hvlc.build_table(h, 8)                     # (limit 8 does not bite: plain Huffman)
    small = hvlc.huffman_lengths(list(h), 3)           # 6 symbols in <= 3 bits
    best = min(sum(int(h[i]) * l[i] for i in range(6)) for l in itertools.product(range(1, 4), repeat=6)
               if sum(2.0 ** -x for x in l) <= 1.0)
    assert small.max() <= 3 and int((h * small.astype(np.uint64)).sum()) == best
