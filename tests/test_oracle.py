"""CPU tests of the oracle itself (-m "not gpu"): the repo's C restatement of
cpu_vlc_encode (cpuencode.cpp:12-46) against the golden vectors the reference
produced (tests/golden/, see make_golden.py) and, when oracle/_ref exists (build
container only), differentially against the compiled reference."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def test_g1_full_file(golden):
    """G1: the reference's own end-to-end check (main_test_cu.cu:159-172) input."""
    m = golden["meta"]["G1"]
    raw = np.tile(golden["period"], 4)
    assert _sha(raw.tobytes()) == m["input_sha256"]
    x = raw.view(np.uint32)
    out, osz = O.encode32(x, golden["cw"], golden["cl"])
    assert osz == m["outsize"] == 291334
    nint = (osz + 3) // 4
    assert nint == m["num_ints"] == 72834
    assert _sha(out[:nint].tobytes()[:osz]) == m["out_sha256"]
    assert _sha(out[:nint].tobytes()) == m["out_words_sha256"]
    assert [int(v) for v in out[:4]] == m["first_words"]


def test_g1_period(golden):
    m = golden["meta"]["G1p"]
    out, osz = O.encode32(golden["period"].view(np.uint32), golden["cw"], golden["cl"])
    assert osz == m["outsize"] == 72834
    assert _sha(out[: (osz + 3) // 4].tobytes()[:osz]) == m["out_sha256"]


def test_g1_segment_offsets(golden):
    m = golden["meta"]["G1"]
    x = np.tile(golden["period"], 4).view(np.uint32)
    counts, offs, total = O.segment_offsets(x, golden["cl"])
    assert np.array_equal(counts, golden["seg_counts"])
    assert [int(v) for v in counts[:4]] == m["seg_counts_first4"] == [2275, 2334, 2342, 2340]
    assert [int(v) for v in offs[:4]] == m["seg_offsets_first4"] == [0, 2275, 4609, 6951]
    assert int(offs[-1]) == m["seg_offset_last"] == 2328317
    assert total == m["total_bits"] == 2330672
    assert int(counts.min()) == 2165 and int(counts.max()) == 2432


@pytest.mark.parametrize("name", ["K1", "K2", "K3"])
def test_kat(golden, name):
    k = golden["meta"]["KAT"][name]
    if k["table"] == "generateCodewords":
        cw, cl = O.generate_codewords(256)       # restates testdatagen.h:42-60
    else:
        cw = np.zeros(256, dtype=np.uint32)
        cl = np.zeros(256, dtype=np.uint32)
        for s, (c, l) in k["table"].items():
            cw[int(s)], cl[int(s)] = c, l
    out, osz = O.encode32(np.array(k["input"], dtype=np.uint32), cw, cl)
    assert osz == k["outsize"]
    assert [int(v) for v in out[: (osz + 3) // 4]] == k["words"]


def test_kat_hand_values():
    """Hand-checkable: 1 10 100 1000 1000 10000 100000 1000000 -> d2221040."""
    cw, cl = O.generate_codewords(256)
    out, osz = O.encode32(np.array([0x00010203, 0x04050607], dtype=np.uint32), cw, cl)
    assert osz == 4 and int(out[0]) == 0xD2221040


def test_diff_vectors(golden):
    """Random tables (lengths 0..31, incl. an unmasked table exercising the
    cpuencode.cpp:35 quirk) x random words; expected words came from the reference."""
    vec = golden["vec"]
    for c in golden["meta"]["diff_cases"]:
        i = c["id"]
        out, osz = O.encode32(vec[f"in{i}"], vec[f"cw{i}"], vec[f"cl{i}"])
        assert osz == c["outsize"], c
        exp = vec[f"out{i}"]
        assert np.array_equal(out[: exp.size], exp), c


def test_empty_and_word_boundary():
    cw, cl = O.generate_codewords(256)
    out, bits = O.encode(np.zeros(0, dtype=np.uint32), cw, cl)
    assert bits == 0 and out.size == 1 and int(out[0]) == 0
    # 8 symbols of length 4 end exactly on a word boundary: the word past the end is zeroed
    x = np.array([0x03030303, 0x04040404], dtype=np.uint32)
    out, bits = O.encode(x, cw, cl)
    assert bits == 32 and out.size == 2 and int(out[1]) == 0 and int(out[0]) == 0x88888888


def test_encode64_matches_encode32_and_concatenates(golden):
    rng = np.random.default_rng(7)
    x = rng.integers(0, 1 << 32, 5000, dtype=np.uint64).astype(np.uint32)
    cw, cl = golden["vec"]["cw0"], golden["vec"]["cl0"]
    o64, bits = O.encode(x, cw, cl)
    o32, osz = O.encode32(x, cw, cl)
    assert (bits + 7) // 8 == osz and np.array_equal(o64, o32[: o64.size])
    # concatenative property used for sharding: prefix stream is a bit-prefix
    oa, ba = O.encode(x[:2048], cw, cl)
    assert np.array_equal(oa[: ba // 32], o64[: ba // 32])
    _, offs, total = O.segment_offsets(x, cl)
    assert total == bits and int(offs[8]) == ba


def test_mt_baseline_counts_bits(golden):
    x = np.tile(golden["period"], 2).view(np.uint32)
    _, bits = O.encode(x, golden["cw"], golden["cl"])
    b3, sec = O.encode_mt(x, golden["cw"], golden["cl"], 3)
    assert b3 == bits and sec > 0


def test_generator_is_deterministic_and_matches_cdf():
    cdf = np.full(256, 0xFFFFFFFF, dtype=np.uint32)
    cdf[0] = 1 << 31          # p(0)=1/2, p(1)=1/2 - 2^-32, rest ~0
    a = O.gen_symbols(1 << 16, 0x5EED0002, cdf)
    b = O.gen_symbols(1 << 16, 0x5EED0002, cdf)
    assert np.array_equal(a, b)
    h = O.histogram(a)
    assert h[0] + h[1] == a.size and abs(int(h[0]) - (1 << 15)) < 1500
    # shard property: draws offset by first_draw reproduce the tail of one long stream
    c = O.gen_symbols(1 << 15, 0x5EED0002, cdf, first_draw=1 << 14)
    assert np.array_equal(c, a[1 << 15:])


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (compiled reference) only exists in the build container")
def test_differential_vs_compiled_reference(golden):
    """Restatement vs the reference's cpuencode.cpp compiled in place, fresh random cases."""
    rng = np.random.default_rng(20261004)
    for trial in range(60):
        nw = int(rng.integers(1, 3000))
        mode = trial % 4
        if mode == 0:
            cl = rng.integers(0, 32, 256).astype(np.uint32)
        elif mode == 1:
            cl = rng.integers(1, 9, 256).astype(np.uint32)
        elif mode == 2:
            cl = rng.integers(28, 32, 256).astype(np.uint32)
        else:
            cl = np.where(rng.random(256) < 0.5, 0, rng.integers(1, 32, 256)).astype(np.uint32)
        cw = rng.integers(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint32)
        if trial % 5:   # every fifth table is left unmasked (quirk path)
            cw = cw & ((np.uint64(1) << cl.astype(np.uint64)) - np.uint64(1)).astype(np.uint32)
        x = rng.integers(0, 1 << 32, nw, dtype=np.uint64).astype(np.uint32)
        o_ref, s_ref = O.ref_encode32(x, cw, cl)
        o_own, s_own = O.encode32(x, cw, cl)
        assert s_ref == s_own
        _, bits = O.encode(x, cw, cl)
        n = bits // 32 + 1      # every word the reference reaches, incl. the zeroed open/next word
        assert (bits + 7) // 8 == s_ref
        assert np.array_equal(o_ref[:n], o_own[:n]), trial
    # and the shipped file itself
    x = np.tile(golden["period"], 4).view(np.uint32)
    o_ref, s_ref = O.ref_encode32(x, golden["cw"], golden["cl"])
    o_own, s_own = O.encode32(x, golden["cw"], golden["cl"])
    assert s_ref == s_own and np.array_equal(o_ref[: (s_ref + 3) // 4], o_own[: (s_ref + 3) // 4])


def test_bit_append_reproduces_the_whole_stream(golden):
    """SURVEY 8(f)#1 checker: appending the streams of shards (cut at segment multiples) bit-exactly gives the
    stream of the whole input -- encoding is concatenative."""
    x = np.tile(golden["period"], 3).view(np.uint32)[: 3 * 256 * 97 + 41]
    ref, rbits = O.encode(x, golden["cw"], golden["cl"])
    dst = np.zeros(rbits // 32 + 2, dtype=np.uint32)
    pos = 0
    for a, b in ((0, 256 * 10), (256 * 10, 256 * 57), (256 * 57, x.size)):
        part, pbits = O.encode(x[a:b], golden["cw"], golden["cl"])
        pos = O.bit_append(dst, pos, part, pbits)
    assert pos == rbits and np.array_equal(dst[: rbits // 32 + 1], ref[: rbits // 32 + 1])


def test_decoder_inverts_the_encoder(golden):
    """SURVEY 8(f)#3 checker: decode(encode(x)) == x for the golden (prefix-free) table, also from a segment offset."""
    x = golden["period"].view(np.uint32)[:2048]
    ref, rbits = O.encode(x, golden["cw"], golden["cl"])
    y, used = O.decode(ref, x.size, golden["cw"], golden["cl"])
    assert used == rbits and np.array_equal(x, y)
    _, offs, _ = O.segment_offsets(x, golden["cl"])
    y3, _ = O.decode(ref, 256, golden["cw"], golden["cl"], start_bit=int(offs[3]))
    assert np.array_equal(x[3 * 256: 4 * 256], y3)
