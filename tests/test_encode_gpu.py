"""-m gpu: parity of the HIP encode path (through the C ABI) against the CPU oracle, the golden
vectors the reference produced, and -- at BASELINE.json's full sizes -- size-independent
properties.  Bar: bit-exact (integer/bit work), every word of the stream and every offset."""
import hashlib
import os

import numpy as np
import pytest
import torch

import hvlc
import oracle_lib as O
from util import dist_weights, extract_bits, random_table, synth_on_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc():
    assert torch.cuda.is_available(), "GPU tests need a GPU (the product has no CPU fallback)"
    e = hvlc.Encoder(0, 1 << 22)
    yield e
    e.close()


def _dev(x_u32):
    return torch.from_numpy(np.ascontiguousarray(x_u32, dtype=np.uint32).view(np.int32).copy()).cuda()


def _check(enc, x, cw, cl, what=""):
    """Encode x on the GPU and compare stream, total bits and segment offsets with the oracle."""
    x = np.ascontiguousarray(x, dtype=np.uint32)
    d_tab = enc.table_to_device(cw, cl)
    out, offs, bits = enc.encode(_dev(x), d_tab, max_len=int(cl.max()) if cl.size else 0)
    ref, rbits = O.encode(x, cw, cl)
    assert bits == rbits, (what, bits, rbits)
    got = out[: bits // 32 + 1].cpu().numpy().view(np.uint32)
    if not np.array_equal(got, ref):
        bad = np.nonzero(got != ref)[0]
        raise AssertionError(f"{what}: {bad.size} words differ, first at {bad[0]}: "
                             f"{got[bad[0]]:08x} != {ref[bad[0]]:08x} (of {ref.size})")
    _, roffs, _ = O.segment_offsets(x, cl)
    assert np.array_equal(offs.cpu().numpy().view(np.uint64), roffs), what
    return bits


def test_g1_golden(enc, golden):
    """The reference's own end-to-end case (run:1, main_test_cu.cu:159-172): shipped file + its table."""
    m = golden["meta"]["G1"]
    x = np.tile(golden["period"], 4).view(np.uint32)
    d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    out, offs, bits = enc.encode(_dev(x), d_tab, max_len=21)
    assert bits == m["total_bits"] == 2330672
    osz = (bits + 7) // 8
    got = out[: m["num_ints"]].cpu().numpy().view(np.uint32)
    assert hashlib.sha256(got.tobytes()[:osz]).hexdigest() == m["out_sha256"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == m["out_words_sha256"]
    assert [int(v) for v in got[:4]] == m["first_words"]
    o = offs.cpu().numpy().view(np.uint64)
    assert [int(v) for v in o[:4]] == [0, 2275, 4609, 6951] and int(o[-1]) == 2328317
    assert np.array_equal(np.diff(o).astype(np.uint32), golden["seg_counts"][:-1])


def test_g1_period_golden(enc, golden):
    m = golden["meta"]["G1p"]
    x = golden["period"].view(np.uint32)
    out, _, bits = enc.encode(_dev(x), enc.table_to_device(golden["cw"], golden["cl"]), max_len=21)
    osz = (bits + 7) // 8
    assert osz == m["outsize"]
    got = out[: (osz + 3) // 4].cpu().numpy().view(np.uint32)
    assert hashlib.sha256(got.tobytes()[:osz]).hexdigest() == m["out_sha256"]


@pytest.mark.parametrize("name", ["K1", "K2", "K3"])
def test_kat(enc, golden, name):
    k = golden["meta"]["KAT"][name]
    if k["table"] == "generateCodewords":
        cw, cl = O.generate_codewords(256)
    else:
        cw = np.zeros(256, dtype=np.uint32)
        cl = np.zeros(256, dtype=np.uint32)
        for s, (c, l) in k["table"].items():
            cw[int(s)], cl[int(s)] = c, l
    x = np.array(k["input"], dtype=np.uint32)
    out, _, bits = enc.encode(_dev(x), enc.table_to_device(cw, cl), max_len=int(cl.max()))
    assert (bits + 7) // 8 == k["outsize"]
    assert [int(v) for v in out[: (k["outsize"] + 3) // 4].cpu().numpy().view(np.uint32)] == k["words"]
    # and through the host-buffer drop-in with cpu_vlc_encode's signature
    o2, osz = hvlc.gpu_vlc_encode(x, cw, cl)
    assert osz == k["outsize"] and [int(v) for v in o2[: (osz + 3) // 4]] == k["words"]


def test_reference_diff_vectors(enc, golden):
    """Expected words came from the reference itself (tests/golden/make_golden.py)."""
    vec = golden["vec"]
    for c in golden["meta"]["diff_cases"]:
        if c["kind"] == "unmasked":
            continue            # violates the table precondition; see test_bad_table
        i = c["id"]
        cl = vec[f"cl{i}"]
        out, _, bits = enc.encode(_dev(vec[f"in{i}"]), enc.table_to_device(vec[f"cw{i}"], cl), max_len=int(cl.max()))
        assert (bits + 7) // 8 == c["outsize"], c
        exp = vec[f"out{i}"]
        assert np.array_equal(out[: exp.size].cpu().numpy().view(np.uint32), exp), c


@pytest.mark.parametrize("nwords", [1, 2, 63, 255, 256, 257, 1023, 4095, 4096, 4097, 12345, 65536, 65536 * 3 + 17])
@pytest.mark.parametrize("kind", ["any", "short", "long"])
def test_ragged_sizes_vs_oracle(enc, nwords, kind):
    rng = np.random.default_rng(nwords * 7 + len(kind))
    cw, cl = random_table(rng, kind)
    x = rng.integers(0, 1 << 32, nwords, dtype=np.uint64).astype(np.uint32)
    _check(enc, x, cw, cl, f"{kind}/{nwords}")


@pytest.mark.parametrize("kind", ["short", "any", "len4"])
def test_every_shape_of_the_last_tile(enc, kind):
    """The stream's last tile takes its own way through the store wave: its full parts (4 KiB each) are stored like any
    tile's, the part the input ends in stores itself unless it is a full one (then the last full part closes the
    stream), parts behind the end are nobody's.  Sizes around every part and tile boundary (a tile = 10240 words);
    with the 4-bit table every part ends on a word boundary (the zero word behind the stream, cpuencode.cpp:39)."""
    rng = np.random.default_rng(4242 + len(kind))
    cw, cl = random_table(rng, kind)
    sizes = [1024, 1025, 2047, 2048, 2049, 9215, 9216, 9217, 10239, 10240, 10241, 11264, 11265, 20480, 20481,
             10240 * 64, 10240 * 64 + 1, 10240 * 65 - 1024, 10240 * 65 - 1023]
    for nwords in sizes:
        x = rng.integers(0, 1 << 32, nwords, dtype=np.uint64).astype(np.uint32)
        _check(enc, x, cw, cl, f"{kind}/{nwords}")


def test_the_geometry_of_eight_data_waves_is_bit_exact(golden, monkeypatch):
    """The encoder's second workgroup geometry (eight data waves, 32 KiB tiles, larger rings: what a context switches to for
    dense output), fixed with HVLC_GEOMETRY=8: the golden file, ragged sizes, every shape of ITS last tile (a tile = 8192
    words), adversarial tables."""
    monkeypatch.setenv("HVLC_GEOMETRY", "8")
    e = hvlc.Encoder(0, 1 << 22)
    try:
        x = np.tile(golden["period"], 4).view(np.uint32)
        _check(e, x, golden["cw"], golden["cl"], "g1/8")
        assert e.launch_info()["data_waves"] == 8
        rng = np.random.default_rng(808)
        for kind in ("short", "any", "len4", "all31", "zeros"):
            cw, cl = random_table(rng, kind)
            for nwords in (1, 257, 1024, 1025, 7168, 7169, 8191, 8192, 8193, 9216, 16384, 16385, 8192 * 64, 8192 * 65 - 1023, 12345, 65536 * 3 + 17):
                xx = rng.integers(0, 1 << 32, nwords, dtype=np.uint64).astype(np.uint32)
                _check(e, xx, cw, cl, f"8/{kind}/{nwords}")
    finally:
        e.close()


def test_the_geometry_follows_the_density_of_the_last_finished_encode(monkeypatch):
    """A context starts with ten data waves per workgroup; an encode of more than 3.4 bits per symbol that hvlc_finish has
    reported on makes the NEXT calls use eight (larger rings: four parts of 4 bits per symbol instead of three), one
    below 3.2 takes them back.  The stream is the oracle's either way."""
    monkeypatch.delenv("HVLC_GEOMETRY", raising=False)
    e = hvlc.Encoder(0, 1 << 22)
    try:
        rng = np.random.default_rng(909)
        dense_cw, dense_cl = random_table(rng, "len4")                       # 4 bits per symbol
        sparse_cl = np.full(256, 2, dtype=np.uint32); sparse_cw = (np.arange(256, dtype=np.uint32) & 3)   # 2 bits per symbol (not decodable; the encoder does not care)
        x = rng.integers(0, 1 << 32, 1 << 21, dtype=np.uint64).astype(np.uint32)   # 8 MiB
        assert e.launch_info()["data_waves_next"] == 10
        _check(e, x, dense_cw, dense_cl, "dense/first")                     # (_check calls finish)
        assert e.launch_info()["data_waves"] == 10 and e.launch_info()["data_waves_next"] == 8
        _check(e, x, dense_cw, dense_cl, "dense/second")
        assert e.launch_info()["data_waves"] == 8
        _check(e, x[: 1 << 18], sparse_cw, sparse_cl, "sparse/small")       # below 4 MiB: no decision is taken from it
        assert e.launch_info()["data_waves_next"] == 8
        _check(e, x, sparse_cw, sparse_cl, "sparse")
        assert e.launch_info()["data_waves"] == 8 and e.launch_info()["data_waves_next"] == 10
        _check(e, x, sparse_cw, sparse_cl, "sparse/second")
        assert e.launch_info()["data_waves"] == 10
    finally:
        e.close()


@pytest.mark.parametrize("kind", ["all31", "alt1_31", "zeros", "allzero", "len4", "len16", "len17"])
def test_adversarial_tables(enc, kind):
    """All-31-bit codes (496 bits per 16 symbols: general path), alternating 1/31, mostly/all
    length-0 symbols (carry chains through empty tiles), 16 x 4 = exactly 64 bits, 16/17-bit codes."""
    rng = np.random.default_rng(hash(kind) & 0xFFFF)
    cw, cl = random_table(rng, kind)
    for nwords in (4096 * 5 + 100, 300):
        x = rng.integers(0, 1 << 32, nwords, dtype=np.uint64).astype(np.uint32)
        _check(enc, x, cw, cl, f"{kind}/{nwords}")


@pytest.mark.parametrize("one_in", [300, 40000])
def test_word_strings_over_32_bits_inside_short_rows(enc, one_in):
    """The data waves join the four codewords of an input word with 32-bit operations and encode a part again, the
    general way, when some word's codewords exceed 32 bits (and stop trying for a while when that keeps happening).
    A 1-bit symbol with a rare 30-bit one: words of 33 bits inside lane-rows of at most 64 -- in nearly every part
    (one in 300 symbols) or in a few (one in 40000), over enough tiles for every wave to change its mind both ways."""
    rng = np.random.default_rng(one_in)
    cl = np.zeros(256, dtype=np.uint32)
    cw = np.zeros(256, dtype=np.uint32)
    cl[7], cw[7] = 1, 0
    cl[9], cw[9] = 30, (1 << 29) | 0x155555
    cl[11], cw[11] = 2, 0b10
    n = 12 << 20
    x8 = np.full(n, 7, dtype=np.uint8)
    x8[rng.random(n) < 0.1] = 11
    x8[rng.choice(n, n // one_in, replace=False)] = 9
    x8[n // 3: n // 3 + (1 << 20)] = 7                     # a stretch with no long code at all
    _check(enc, x8.view(np.uint32), cw, cl, f"w32/{one_in}")


def test_sparse_bits_cross_tile_carry(enc):
    """Only one symbol in ~2000 emits bits: most tiles hold fewer than 31 bits, so the 31-bit
    carry has to travel through many tiles (and through tiles with no bits at all)."""
    rng = np.random.default_rng(5)
    cl = np.zeros(256, dtype=np.uint32)
    cw = np.zeros(256, dtype=np.uint32)
    cl[200], cw[200] = 7, 0b1010011
    cl[201], cw[201] = 1, 1
    x8 = np.zeros(4096 * 4 * 40, dtype=np.uint8)
    hits = rng.choice(x8.size, x8.size // 2000, replace=False)
    x8[hits] = rng.choice([200, 201], hits.size)
    _check(enc, x8.view(np.uint32), cw, cl, "sparse")


def test_mixed_fast_and_general_tiles(enc, golden):
    """Mostly short codes with rare 31-bit codes: some tiles take the general path, neighbours
    the fast path, so carries cross between the two."""
    rng = np.random.default_rng(11)
    cl = np.full(256, 2, dtype=np.uint32)
    cl[250:] = 31
    cw = (rng.integers(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint64)
          & ((np.uint64(1) << cl.astype(np.uint64)) - np.uint64(1))).astype(np.uint32)
    x8 = rng.integers(0, 250, 4096 * 4 * 24, dtype=np.int64).astype(np.uint8)
    for t in (3, 4, 10, 23):                                    # bursts of long codes in four tiles
        x8[t * 16384 + 100: t * 16384 + 140] = 251
    _check(enc, x8.view(np.uint32), cw, cl, "mixed")


def test_empty_input(enc):
    cw, cl = O.generate_codewords(256)
    d_tab = enc.table_to_device(cw, cl)
    out = torch.full((4,), -1, dtype=torch.int32, device="cuda")
    enc.encode_ptr(0, 0, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), 4)
    bits, rc = enc.finish()
    assert bits == 0 and rc == 0
    assert out.cpu().tolist() == [0, -1, -1, -1]               # exactly the one open word (cpuencode.cpp:17)


def test_word_boundary_end_writes_trailing_zero_word(enc):
    cw, cl = random_table(np.random.default_rng(3), "len4")
    x = ((np.arange(4096 + 8, dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    d_tab = enc.table_to_device(cw, cl)
    out = torch.full((x.size * 16 // 32 + 3,), -1, dtype=torch.int32, device="cuda")
    enc.encode_ptr(_dev(x).data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel())
    bits, _ = enc.finish()
    assert bits == x.size * 16 and bits % 32 == 0
    o = out.cpu().numpy()
    assert o[bits // 32] == 0 and o[bits // 32 + 1] == -1 and o[bits // 32 + 2] == -1
    ref, _ = O.encode(x.astype(np.uint32), cw, cl)
    assert np.array_equal(o[: bits // 32 + 1].view(np.uint32), ref)


def test_no_zero_fill_contract(enc, golden):
    """d_out_bits pre-filled with garbage: every word of the stream is overwritten, nothing after it is touched."""
    x = golden["period"].view(np.uint32)[:50000]
    d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    ref, rbits = O.encode(x, golden["cw"], golden["cl"])
    out = torch.full((ref.size + 64,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    enc.encode_ptr(_dev(x).data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel())
    bits, _ = enc.finish()
    o = out.cpu().numpy().view(np.uint32)
    assert bits == rbits and np.array_equal(o[: ref.size], ref) and (o[ref.size:] == 0x5A5A5A5A).all()


def test_bad_table_is_reported(enc):
    cw, cl = O.generate_codewords(256)
    cl2 = cl.copy(); cl2[9] = 32
    x = np.arange(1000, dtype=np.uint32)
    d_in = _dev(x)
    out = torch.empty(hvlc.out_capacity_words(x.size, 31), dtype=torch.int32, device="cuda")
    for bad_cw, bad_cl in ((cw, cl2), (cw | np.uint32(1 << 20), cl)):
        d_tab = enc.table_to_device(bad_cw, bad_cl)
        enc.encode_ptr(d_in.data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel())
        _, rc = enc.finish(check=False)
        assert rc == hvlc.ERR_TABLE


def test_capacity_is_enforced_without_overrun(enc, golden):
    x = golden["period"].view(np.uint32)
    d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    cap = 1000
    out = torch.full((cap + 256,), 0x77, dtype=torch.int32, device="cuda")
    enc.encode_ptr(_dev(x).data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), cap)
    _, rc = enc.finish(check=False)
    assert rc == hvlc.ERR_CAPACITY
    assert (out[cap:].cpu().numpy() == 0x77).all()
    ref, _ = O.encode(x, golden["cw"], golden["cl"])
    assert np.array_equal(out[:cap].cpu().numpy().view(np.uint32), ref[:cap])    # the part that fits is right


def test_argument_checks(enc):
    cw, cl = O.generate_codewords(256)
    d_tab = enc.table_to_device(cw, cl)
    buf = torch.zeros(64, dtype=torch.int32, device="cuda")
    with pytest.raises(hvlc.HvlcError):      # misaligned input
        enc.encode_ptr(buf.data_ptr() + 4, 8, d_tab.data_ptr(), d_tab.data_ptr() + 1024, buf.data_ptr(), 64)
    with pytest.raises(hvlc.HvlcError):      # no output
        enc.encode_ptr(buf.data_ptr(), 8, d_tab.data_ptr(), d_tab.data_ptr() + 1024, 0, 64)


def test_repeatable_and_idempotent(enc, golden):
    """Two runs into the same buffers give identical bytes (no state leaks between launches)."""
    x = np.tile(golden["period"], 2).view(np.uint32)
    d_in = _dev(x); d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    o1, f1, b1 = enc.encode(d_in, d_tab, max_len=21)
    a = o1[: b1 // 32 + 1].clone(); fa = f1.clone()
    for _ in range(5):
        o2, f2, b2 = enc.encode(d_in, d_tab, max_len=21, out=o1, offsets=f1)
        assert b2 == b1 and torch.equal(o2[: b1 // 32 + 1], a) and torch.equal(f2, fa)


def test_back_to_back_different_data_same_size(enc, golden):
    """Granule hygiene across launches: alternate two different inputs of identical size (identical granule
    addresses, different values) on one context; every call must match the oracle."""
    rng = np.random.default_rng(99)
    cw, cl = random_table(rng, "short")
    d_tab = enc.table_to_device(cw, cl)
    for nwords in (4096 * 70, 4096 * 3 + 5):
        a = rng.integers(0, 1 << 32, nwords, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 1 << 32, nwords, dtype=np.uint64).astype(np.uint32)
        ra, rb = O.encode(a, cw, cl), O.encode(b, cw, cl)
        da, db = _dev(a), _dev(b)
        for k in range(6):
            d, (ref, rbits) = (da, ra) if k % 2 == 0 else (db, rb)
            out, _, bits = enc.encode(d, d_tab, max_len=int(cl.max()))
            assert bits == rbits
            assert np.array_equal(out[: bits // 32 + 1].cpu().numpy().view(np.uint32), ref), (nwords, k)


def test_stage_entry_points(enc, golden):
    """hvlc_segment_bits (A1's outidx) and hvlc_exclusive_scan (A3) as separate stages."""
    x = np.tile(golden["period"], 4).view(np.uint32)
    d_in = _dev(x); d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    counts = enc.segment_bits(d_in, d_tab)
    assert np.array_equal(counts.cpu().numpy().view(np.uint32), golden["seg_counts"])
    offs, total = enc.exclusive_scan(counts)
    _, roffs, rtotal = O.segment_offsets(x, golden["cl"])
    assert np.array_equal(offs.cpu().numpy().view(np.uint64), roffs) and int(total.item()) == rtotal
    enc.rebase_offsets(offs, 1 << 40)
    assert np.array_equal(offs.cpu().numpy().view(np.uint64), roffs + np.uint64(1 << 40))


@pytest.mark.parametrize("n", [1, 255, 4096, 4097, 4096 * 64 + 5, 4096 * 64 * 3 + 1, 5_000_000])
def test_exclusive_scan_sizes(enc, n):
    rng = np.random.default_rng(n)
    c = rng.integers(0, 1 << 32 if n < 5000 else 40000, n, dtype=np.uint64).astype(np.uint32)
    offs, total = enc.exclusive_scan(_dev(c))
    ref = np.concatenate([[0], np.cumsum(c.astype(np.uint64))])
    assert np.array_equal(offs.cpu().numpy().view(np.uint64), ref[:-1]) and int(total.item()) == int(ref[-1])


def test_stitch_two_shards_one_gpu(enc, golden):
    """The multi-GPU stitch on one device: two shards encoded independently, totals 'all-gathered' into a
    device array, offsets rebased on the device (hvlc_rebase_offsets_dev) == single-stream offsets."""
    x = np.tile(golden["period"], 4).view(np.uint32)[: 200_000 + 123]
    d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    _, ref_offs, ref_bits = O.segment_offsets(x, golden["cl"])
    parts, totals = [], []
    for r in range(2):
        w0, w1 = hvlc.shard_bounds(x.size, 2, r)
        out, offs, bits = enc.encode(_dev(x[w0:w1]), d_tab, max_len=21)
        ref, rbits = O.encode(x[w0:w1], golden["cw"], golden["cl"])
        assert bits == rbits and np.array_equal(out[: bits // 32 + 1].cpu().numpy().view(np.uint32), ref)
        parts.append(offs.clone()); totals.append(bits)
    gathered = torch.tensor(totals, dtype=torch.int64, device="cuda")
    for r in range(2):
        enc.rebase_offsets_dev(parts[r], gathered, r)
    got = torch.cat(parts).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, ref_offs) and sum(totals) == ref_bits


def test_timeout_is_reported_not_a_hang(golden, monkeypatch):
    """If a tile's bit count never arrives (test hook: tile 1000 does not publish it) every later tile would
    wait for ever: the bounded waits must end in HVLC_ERR_TIMEOUT within the configured bound, not in a hung
    GPU, and a normal encode on a fresh context works afterwards."""
    import time
    monkeypatch.setenv("HVLC_TIMEOUT_MS", "200")          # read once, by hvlc_create
    e2 = hvlc.Encoder(0, 1 << 24)
    e2.test_hook(hvlc.HOOK_STALL_TILE, 1000)
    x = np.tile(golden["period"], 256).view(np.uint32)          # 64 MiB = 5462 tiles
    d_in = _dev(x); d_tab = e2.table_to_device(golden["cw"], golden["cl"])
    out = torch.empty(hvlc.out_capacity_words(x.size, 21), dtype=torch.int32, device="cuda")
    t0 = time.time()
    e2.encode_ptr(d_in.data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel())
    _, rc = e2.finish(check=False)
    dt = time.time() - t0
    assert rc == hvlc.ERR_TIMEOUT, rc
    assert dt < 5.0, dt
    # the same context, hook off: the next launch is clean (every polled word is re-initialised per call)
    e2.test_hook(hvlc.HOOK_STALL_TILE, -1)
    o2, _, bits2 = e2.encode(d_in, d_tab, max_len=21)
    ref, rbits = O.encode(x, golden["cw"], golden["cl"])
    assert bits2 == rbits and torch.equal(o2[: bits2 // 32 + 1].cpu(), torch.from_numpy(ref.view(np.int32)))
    e2.close()
    e3 = hvlc.Encoder(0, 1 << 24)
    o3, _, bits = e3.encode(d_in, d_tab, max_len=21)
    ref, rbits = O.encode(x, golden["cw"], golden["cl"])
    assert bits == rbits and torch.equal(o3[: bits // 32 + 1].cpu(), torch.from_numpy(ref.view(np.int32)))
    e3.close()


def test_scan_timeout_is_reported_by_finish(monkeypatch):
    """hvlc_exclusive_scan is asynchronous like hvlc_encode: a look-back that gives up (test hook: scan tile 3 never
    publishes its sum) is reported by hvlc_finish as HVLC_ERR_TIMEOUT, and the next scan on the context is clean."""
    monkeypatch.setenv("HVLC_TIMEOUT_MS", "200")
    e2 = hvlc.Encoder(0, 1 << 20)
    counts = torch.randint(0, 5000, (4096 * 40 + 17,), dtype=torch.int32, device="cuda")
    e2.test_hook(hvlc.HOOK_STALL_TILE, 3)
    e2.exclusive_scan(counts)
    _, rc = e2.finish(check=False)
    assert rc == hvlc.ERR_TIMEOUT, rc
    e2.test_hook(hvlc.HOOK_STALL_TILE, -1)
    offs, total = e2.exclusive_scan(counts)
    tot, rc = e2.finish(check=False)
    ref = torch.cumsum(counts.to(torch.int64), 0)
    assert rc == 0 and tot == int(ref[-1].item()) == int(total.item())
    assert torch.equal(offs[1:], ref[:-1]) and int(offs[0].item()) == 0
    e2.close()


@pytest.mark.parametrize("extra_words", [1, 1234, 7167])
def test_partial_last_tile_behind_a_busy_ring(enc, golden, extra_words):
    """The tile the input ends in takes the general path while its workgroup still holds packed tiles in the
    staging ring (regression: its carry word must not be written into another tile's ring words)."""
    n = 3000 * 7168 + extra_words                               # ~82 MiB: every workgroup has several tiles
    x = np.tile(golden["period"], (4 * n) // golden["period"].size + 1)[: 4 * n].view(np.uint32)
    d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    for _ in range(3):
        o, offs, bits = enc.encode(_dev(x), d_tab, max_len=21)
        ref, rbits = O.encode(x, golden["cw"], golden["cl"])
        assert bits == rbits and torch.equal(o[: bits // 32 + 1].cpu(), torch.from_numpy(ref.view(np.int32)))


@pytest.mark.parametrize("factor", [2, 4])
def test_oversubscribed_grid_still_encodes(golden, monkeypatch, factor):
    """Tiles are drawn by running workgroups only, so a grid that is NOT co-resident (forced to 2x / 4x what
    the device holds) must still produce the exact stream: forward progress does not depend on residency."""
    e2 = hvlc.Encoder(0, 1 << 24)
    info = e2.launch_info()
    e2.test_hook(hvlc.HOOK_FORCE_GRID, factor * info["compute_units"] * info["blocks_per_cu"])
    x = np.tile(golden["period"], 512).view(np.uint32)          # 128 MiB = 10923 tiles
    d_tab = e2.table_to_device(golden["cw"], golden["cl"])
    o, offs, bits = e2.encode(_dev(x), d_tab, max_len=21)
    ref, rbits = O.encode(x, golden["cw"], golden["cl"])
    assert bits == rbits and torch.equal(o[: bits // 32 + 1].cpu(), torch.from_numpy(ref.view(np.int32)))
    e2.close()


def test_generator_and_histogram_match_oracle(enc, golden):
    from util import dist_weights
    for name in ("H2.2", "H4.0", "H1.0"):
        cdf = hvlc.cdf32_from_counts(dist_weights(name, golden["period"]))
        d = enc.gen_symbols(1 << 20, 0x5EED0002, cdf, first_draw=12345)
        ref = O.gen_symbols(1 << 20, 0x5EED0002, cdf, first_draw=12345)
        assert np.array_equal(d.cpu().numpy(), ref), name
        h = enc.histogram(d).cpu().numpy()
        assert np.array_equal(h.astype(np.uint64), O.histogram(ref)), name
    d = enc.gen_symbols(1 << 20, 1, cdf)[: (1 << 20) - 16 + 6]
    # histogram with a ragged tail
    buf = torch.empty((1 << 20) + 16, dtype=torch.uint8, device="cuda")
    buf[: d.numel()] = d
    h = enc.histogram(buf[: d.numel()]).cpu().numpy()
    assert np.array_equal(h.astype(np.uint64), O.histogram(d.cpu().numpy()))


# ---- BASELINE.json configs ----------------------------------------------------------------------

def test_config2_256MiB_H22_full_compare(enc, golden):
    """configs[1]: 256 MiB synthetic H~2.2 stream, compared word for word with the oracle."""
    n = 256 << 20
    d_in, cw, cl, hist = synth_on_device(enc, "H2.2", n, golden["period"])
    out, offs, bits = enc.encode(d_in, enc.table_to_device(cw, cl), max_len=int(cl.max()))
    assert bits == int((hist * cl).sum())
    x = d_in.cpu().numpy().view(np.uint32)
    ref, rbits = O.encode(x, cw, cl)
    assert bits == rbits
    assert torch.equal(out[: bits // 32 + 1].cpu(), torch.from_numpy(ref.view(np.int32)))
    _, roffs, _ = O.segment_offsets(x, cl)
    assert np.array_equal(offs.cpu().numpy().view(np.uint64), roffs)
    assert 0.27 < bits / 8 / n < 0.29


def _check_windows(enc, d_in, out, offs, bits, cw, cl, hist, windows, rng):
    """Full-size properties: total == sum(hist*len); offsets strictly consistent with the device
    segment counts; and the stream bits of randomly chosen segment windows equal the oracle's
    encoding of just those segments (concatenative property)."""
    assert bits == int((hist * cl.astype(np.uint64)).sum())
    counts = enc.segment_bits(d_in, enc.table_to_device(cw, cl))
    S = counts.numel()
    o64 = offs[:S]
    assert int(o64[0].item()) == 0
    assert torch.equal(o64[1:] - o64[:-1], counts[:-1].to(torch.int64))
    assert int(o64[-1].item()) + int(counts[-1].item()) == bits
    for _ in range(windows):
        nseg = int(rng.integers(1, 600))
        s0 = int(rng.integers(0, S - nseg))
        x = d_in.view(torch.int32)[s0 * 256:(s0 + nseg) * 256].cpu().numpy().view(np.uint32)
        ref, rbits = O.encode(x, cw, cl)
        b0 = int(o64[s0].item())
        b1 = int(o64[s0 + nseg].item()) if s0 + nseg < S else bits
        assert b1 - b0 == rbits
        w0 = b0 // 32
        chunk = out[w0: w0 + rbits // 32 + 3].cpu().numpy().view(np.uint32)
        got = extract_bits(chunk, b0 - 32 * w0, rbits)
        assert np.array_equal(got, ref[: got.size]), (s0, nseg)
    # the last segments up to the very end of the stream, including the zero-padded open word
    x = d_in.view(torch.int32)[(S - 3) * 256:].cpu().numpy().view(np.uint32)
    ref, rbits = O.encode(x, cw, cl)
    b0 = int(o64[S - 3].item())
    chunk = out[b0 // 32: bits // 32 + 1].cpu().numpy().view(np.uint32)
    got = extract_bits(chunk, b0 % 32, rbits)
    assert np.array_equal(got, ref[: got.size])
    assert (int(out[bits // 32].item()) & 0xFFFFFFFF) & ((1 << (32 - bits % 32)) - 1) == 0


@pytest.mark.parametrize("name", ["H4.0", "H1.0", "H2.2"])
def test_config_4GiB_properties(enc, golden, name):
    """configs[2], [3] and the north-star 4 GiB H~2.2 case: more than 2^32 output bits for H4.0
    (64-bit offsets), 4 194 304 segments / 4096 groups (multi-window look-back), long max codeword
    and heavy LDS contention for H1.0."""
    n = 4 << 30
    d_in, cw, cl, hist = synth_on_device(enc, name, n, golden["period"])
    out, offs, bits = enc.encode(d_in, enc.table_to_device(cw, cl), max_len=int(cl.max()))
    if name == "H4.0":
        assert bits == 4 * n and bits > 1 << 32 and set(np.unique(cl)) == {0, 4}
    if name == "H1.0":
        assert int(cl.max()) >= 23
    _check_windows(enc, d_in, out, offs, bits, cw, cl, hist, 24, np.random.default_rng(1))
    del out, offs, d_in
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["H4.0", "H1.0", "H2.2"])
def test_config_4GiB_full_compare(enc, golden, name):
    """configs[2], [3] and the north-star 4 GiB H~2.2 case, compared WORD FOR WORD with the CPU oracle: the input is
    cut at segment multiples into chunks, every chunk is encoded by the oracle on a host thread and compared with
    the GPU stream bits at the chunk's offset (and the offsets of all its segments with the oracle's), which is the
    reference's own check (main_test_cu.cu:170-171, comparison_helpers.h:5-16) at a size one sequential
    cpu_vlc_encode call cannot reach."""
    n = 4 << 30
    d_in, cw, cl, hist = synth_on_device(enc, name, n, golden["period"])
    out, offs, bits = enc.encode(d_in, enc.table_to_device(cw, cl), max_len=int(cl.max()))
    assert bits == int((hist * cl.astype(np.uint64)).sum())
    x = d_in.cpu().numpy().view(np.uint32)
    stream = out[: bits // 32 + 1].cpu().numpy().view(np.uint32)
    o64 = offs.cpu().numpy().view(np.uint64)
    del out, offs, d_in
    torch.cuda.empty_cache()
    bad, first = O.compare_stream_mt(x, cw, cl, stream, o64, bits, chunk_segments=2048)
    assert bad == 0, f"{bad} chunks differ, first at segment {first}"
    # the open word is zero padded; a stream ending on a word boundary is followed by a zero word (cpuencode.cpp:39)
    assert int(stream[bits // 32]) & ((1 << (32 - bits % 32)) - 1) == 0
    # the checker does catch a flipped bit and a shifted offset
    stream[12345678] ^= 0x00010000
    bad, first = O.compare_stream_mt(x, cw, cl, stream, o64, bits, chunk_segments=2048)
    assert bad == 1 and o64[first] <= 12345678 * 32 < o64[first + 2048]
    stream[12345678] ^= 0x00010000
    o64[777 * 2048 + 5] += 1
    bad, first = O.compare_stream_mt(x, cw, cl, stream, o64, bits, chunk_segments=2048)
    assert bad == 1 and first == 777 * 2048


def test_input_beyond_4GiB_full_compare(golden):
    """12 GiB of H~2.2 input in ONE call: byte offsets of the input beyond 2^32, 2^33.7 output bits, 440 000 tiles (more
    than 6 800 groups: the leaders' walk over group granules crosses many 16-granule windows).  The reference cannot
    run this (32-bit bit offsets, main_test_cu.cu:91-105); word-for-word against the chunked oracle."""
    n = int(os.environ.get("HVLC_TEST_BIG_GIB", "12")) << 30      # (run by hand with 48: 2^34 input bytes, 2^36.7 bits)
    e = hvlc.Encoder(0, n // 4)
    d_in, cw, cl, hist = synth_on_device(e, "H2.2", n, golden["period"])
    out, offs, bits = e.encode(d_in, e.table_to_device(cw, cl), max_len=int(cl.max()))
    assert bits == int((hist * cl.astype(np.uint64)).sum()) and bits > 1 << 33
    x = d_in.cpu().numpy().view(np.uint32)
    stream = out[: bits // 32 + 1].cpu().numpy().view(np.uint32)
    o64 = offs.cpu().numpy().view(np.uint64)
    del out, offs, d_in
    e.close()
    torch.cuda.empty_cache()
    bad, first = O.compare_stream_mt(x, cw, cl, stream, o64, bits, chunk_segments=4096)
    assert bad == 0, f"{bad} chunks differ, first at segment {first}"


def test_probe_bandwidth_shapes_and_argument_checks(enc):
    """hvlc_probe_bandwidth (the measured ceilings of bench.py): every store shape runs, writes exactly the words it is
    told to and nothing behind them, and bad arguments are refused before anything is launched."""
    n = 64 << 20
    d_in = torch.randint(0, 2 ** 31 - 1, (n // 4,), dtype=torch.int32, device="cuda")
    best, med = enc.probe_bandwidth(d_in)
    assert 0 < best <= med and n / best / 1e6 > 200          # GB/s: a read-only stream of 64 MiB
    wpt = 288
    words = (n // 4096) * wpt
    for shape in (1, 2, 3, 4):
        out = torch.full((words + 64,), -1, dtype=torch.int32, device="cuda")
        best, med = enc.probe_bandwidth(d_in, out, wpt, shape)
        assert 0 < best <= med
        assert bool((out[words:] == -1).all()) and not bool((out[:words] == -1).all())
    # bursts (the store wave's shape: a wave writes ten or 32 steps' output as one stretch) and the plain 1:1 copy
    big = torch.full((160 << 20,), -1, dtype=torch.int32, device="cuda")        # (a region per wave and burst: 604 MB at 32 steps)
    for shape in (5, 6, 8, 9, 10, 11):
        best, med = enc.probe_bandwidth(d_in, big, wpt, shape)
        assert 0 < best <= med and int((big != -1).sum()) == words       # every step's words, each exactly once
        big.fill_(-1)
    cp = torch.full((n // 4 + 64,), -1, dtype=torch.int32, device="cuda")
    best, med = enc.probe_bandwidth(d_in, cp, 0, 7)
    assert 0 < best <= med and bool((cp[: n // 4] == d_in).all()) and bool((cp[n // 4:] == -1).all())
    with pytest.raises(RuntimeError):
        enc.probe_bandwidth(d_in, torch.empty(16, dtype=torch.int32, device="cuda"), wpt, 1)      # output too small
    with pytest.raises(RuntimeError):
        enc.probe_bandwidth(d_in, torch.empty(words, dtype=torch.int32, device="cuda"), 290, 2)   # dwordx4 stores: multiple of 4
    with pytest.raises(RuntimeError):
        enc.probe_bandwidth(d_in, torch.empty(words, dtype=torch.int32, device="cuda"), wpt, 5)   # bursts: a region per wave
    with pytest.raises(RuntimeError):
        enc.probe_bandwidth(d_in, torch.empty(words, dtype=torch.int32, device="cuda"), wpt, 7)   # copy: as large as the input
    with pytest.raises(RuntimeError):
        enc.probe_bandwidth(d_in, torch.empty(words, dtype=torch.int32, device="cuda"), wpt, 12)  # no such shape


@pytest.mark.parametrize("world", [2, 5])
def test_two_ranks_on_one_gpu_shared_stitch_code(golden, world):
    """The code bench.py runs at N > 1, on the one GPU of the test box: the "ranks" are contexts on streams of their own.
    Each generates its shard of the job stream (hvlc.shard_first_draw), encodes it from local bit 0, then runs
    hvlc.stitch (the all-gather is replaced by a device-side copy of both totals; the rebase is the product's
    hvlc_rebase_offsets_dev), and the shards are merged (hvlc_merge_shard) -- offsets and merged stream must equal the
    oracle's encoding of the whole job stream."""
    shard_bytes = (48 << 20) if world == 2 else (20 << 20) + 7 * 1024
    period = golden["period"]
    cdf = hvlc.cdf32_from_counts(dist_weights("H2.2", period))
    encs = [hvlc.Encoder(0, shard_bytes // 4) for _ in range(world)]
    streams = [torch.cuda.Stream() for _ in range(world)]
    shards, hists = [], []
    for r in range(world):
        d = encs[r].gen_symbols(shard_bytes, 0x5EED0005, cdf, first_draw=hvlc.shard_first_draw(r, shard_bytes))
        shards.append(d)
        hists.append(encs[r].histogram(d).cpu().numpy().astype(np.uint64))
    whole = O.gen_symbols(world * shard_bytes, 0x5EED0005, cdf)                 # ONE generation from draw 0
    assert np.array_equal(np.concatenate([s.cpu().numpy() for s in shards]), whole)
    cw, cl = hvlc.build_table(sum(hists[1:], hists[0]))                        # table of the job histogram
    S = hvlc.num_segments(shard_bytes // 4)
    outs, offs, totals = [], [], []
    gathered = torch.zeros(world, dtype=torch.int64, device="cuda")
    for r in range(world):
        with torch.cuda.stream(streams[r]):
            d_tab = encs[r].table_to_device(cw, cl)
            o = torch.empty(hvlc.out_capacity_words(shard_bytes // 4, int(cl.max())), dtype=torch.int32, device="cuda")
            f = torch.empty(S, dtype=torch.int64, device="cuda")
            t = torch.zeros(1, dtype=torch.int64, device="cuda")
            encs[r].encode_ptr(shards[r].data_ptr(), shard_bytes // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024,
                               o.data_ptr(), o.numel(), f.data_ptr(), t.data_ptr())
            outs.append(o); offs.append(f); totals.append(t)
    torch.cuda.synchronize()

    def fake_all_gather(g, t):           # what RCCL's all-gather leaves in `gathered` on every rank
        g.copy_(torch.cat(totals))

    for r in range(world):
        hvlc.stitch(totals[r], gathered, r, world, fake_all_gather,
                    lambda g, rr, r=r: encs[r].rebase_offsets_dev(offs[r], g, rr))
    torch.cuda.synchronize()
    x = whole.view(np.uint32)
    ref, rbits = O.encode(x, cw, cl)
    _, roffs, _ = O.segment_offsets(x, cl)
    tot = [int(t.item()) for t in totals]
    assert sum(tot) == rbits
    assert np.array_equal(torch.cat(offs).cpu().numpy().view(np.uint64), roffs)
    merged = torch.zeros(rbits // 32 + 1, dtype=torch.int32, device="cuda")
    for r in range(world):
        encs[r].merge_shard(outs[r], tot[r], merged, hvlc.stitch_base(tot, r))
    torch.cuda.synchronize()
    assert torch.equal(merged.cpu(), torch.from_numpy(ref.view(np.int32)))
    for e in encs:
        e.close()


def test_general_path_large(enc):
    """64 MiB of uniform bytes (every code 8 bits: every lane-row holds 128 bits) -> general path throughout."""
    n = 64 << 20
    d_in, cw, cl, hist = synth_on_device(enc, "H8.0", n)
    assert set(np.unique(cl)) == {8}
    out, offs, bits = enc.encode(d_in, enc.table_to_device(cw, cl), max_len=8)
    assert bits == 8 * n
    _check_windows(enc, d_in, out, offs, bits, cw, cl, hist, 8, np.random.default_rng(2))


@pytest.mark.parametrize("kind,mib", [("any", 32), ("long", 32), ("any", 96)])
def test_every_part_on_the_slow_path_beyond_one_tile_per_workgroup(kind, mib):
    """Random bytes under tables with codes up to 31 bits: no part fits its wave's ring, every part takes the slow
    path (tails published at pack time from the last segment, parts stored segment by segment).  With more tiles than
    workgroups this was the input that faulted in round 2 -- then at 0.36 GB/s, with the scouts polling for ~100 ms (a
    spilled pointer reloaded right in front of the polling load, see huffman-gpu_amd/build_check.py); full
    word-for-word compare."""
    rng = np.random.default_rng(3)
    n = mib << 20
    cw, cl = random_table(rng, kind)
    x = rng.integers(0, 1 << 32, n // 4, dtype=np.uint64).astype(np.uint32)
    e = hvlc.Encoder(0, n // 4)
    assert n // (28 << 10) > e.launch_info()["grid_blocks"], "needs workgroups that draw a second tile"
    d_in = torch.from_numpy(x.view(np.int32)).cuda()
    out, offs, bits = e.encode(d_in, e.table_to_device(cw, cl), max_len=int(cl.max()))
    bad, first = O.compare_stream_mt(x, cw, cl, out[: bits // 32 + 1].cpu().numpy().view(np.uint32),
                                     offs.cpu().numpy().view(np.uint64), bits, chunk_segments=512)
    assert bad == 0, (bad, first)
    e.close()


def test_call_sequences_on_one_context(golden, monkeypatch):
    """The two halves of a context's workspace alternate on the DEVICE (call counter) and each launch cleans the other
    half for the next: every mix of direct calls of different sizes (a later, larger one re-allocates), empty inputs, scans
    in between, graph replays and a timed-out call must leave the next call a clean half.  Every stream, offset array and
    total is compared with the oracle."""
    rng = np.random.default_rng(11)
    monkeypatch.setenv("HVLC_TIMEOUT_MS", "200")          # read once, by hvlc_create
    e = hvlc.Encoder(0, 1 << 20)
    cw, cl = golden["cw"], golden["cl"]
    d_tab = e.table_to_device(cw, cl)
    live = np.nonzero(cl)[0]

    def data(nwords):
        return live[rng.integers(0, live.size, nwords * 4)].astype(np.uint8).view(np.uint32)

    def check(x):
        out, offs, bits = e.encode(_dev(x), d_tab, max_len=21)
        ref, rbits = O.encode(x, cw, cl)
        assert bits == rbits
        assert np.array_equal(out[: bits // 32 + 1].cpu().numpy().view(np.uint32), ref)
        _, roffs, _ = O.segment_offsets(x, cl)
        assert np.array_equal(offs.cpu().numpy().view(np.uint64), roffs)

    def scan(n):
        c = rng.integers(0, 30000, n, dtype=np.uint64).astype(np.uint32)
        offs, total = e.exclusive_scan(_dev(c))
        ref = np.concatenate([[0], np.cumsum(c.astype(np.uint64))])
        assert np.array_equal(offs.cpu().numpy().view(np.uint64), ref[:-1]) and int(total.item()) == int(ref[-1])
        bits, rc = e.finish()
        assert rc == 0 and bits == int(ref[-1])           # hvlc_finish reports on the scan now

    # a captured encode of 300 000 words, replayed between direct calls
    xg = data(300000)
    d_in = _dev(xg)
    gout = torch.empty(hvlc.out_capacity_words(xg.size, 21), dtype=torch.int32, device="cuda")
    goffs = torch.empty(hvlc.num_segments(xg.size), dtype=torch.int64, device="cuda")
    gtotal = torch.zeros(1, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        call = lambda: e.encode_ptr(d_in.data_ptr(), xg.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024,
                                    gout.data_ptr(), gout.numel(), goffs.data_ptr(), gtotal.data_ptr())
        call(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            call()
    gref, gbits = O.encode(xg, cw, cl)

    def replay(times):
        for _ in range(times):
            gout.fill_(-1)
            g.replay()
            torch.cuda.synchronize()
            assert int(gtotal.item()) == gbits
            assert np.array_equal(gout[: gbits // 32 + 1].cpu().numpy().view(np.uint32), gref)
        bits, rc = e.finish()
        assert rc == 0 and bits == gbits

    check(data(7168 * 40 + 5)); check(data(100)); replay(1); check(data(7168 * 3))
    scan(100000); check(data(200000)); replay(3); scan(5)
    out, offs, bits = e.encode(_dev(np.zeros(0, np.uint32)), d_tab, max_len=21)      # the empty input: one zero word
    assert bits == 0 and int(out[0].item()) == 0
    check(data(7168 * 64 + 1)); replay(2)
    # a call that times out (one tile never publishes its bit count) leaves a half dirty and flags set: the next two are clean
    e.test_hook(hvlc.HOOK_STALL_TILE, 3)
    x = data(7168 * 20)
    d_x = _dev(x)
    tout = torch.empty(hvlc.out_capacity_words(x.size, 21), dtype=torch.int32, device="cuda")
    e.encode_ptr(d_x.data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, tout.data_ptr(), tout.numel())
    _, rc = e.finish(check=False)
    assert rc == hvlc.ERR_TIMEOUT, rc
    e.test_hook(hvlc.HOOK_STALL_TILE, -1)
    check(x); check(data(5000))
    replay(1)
    # larger than the context was created for: the workspace grows (re-allocated: graphs captured before must not be
    # replayed afterwards, include/hvlc.h)
    check(data(3 << 20)); check(data(1234)); check(data(7168 * 9))
    e.close()


# ---------------------------------------------------------------------------------------------------------
# SURVEY.md section 8(f) "next" rows: seam merge (#1) and decoder (#3)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cuts", [(0.5,), (0.1, 0.6), (0.25, 0.5, 0.75), (0.001, 0.999)])
def test_merge_shards_into_one_stream(enc, golden, cuts):
    """Shards encoded independently (cut at segment multiples) and merged at their bases reproduce the
    oracle's stream of the whole input word for word; merge order does not matter."""
    x = np.tile(golden["period"], 24).view(np.uint32)[: 24 * 65536 - 4321]      # ~6 MiB, ragged
    d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    ref, rbits = O.encode(x, golden["cw"], golden["cl"])
    bounds = [0] + [int(c * x.size) // 256 * 256 for c in cuts] + [x.size]
    parts = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        o, _, bits = enc.encode(_dev(x[a:b]), d_tab, max_len=21)
        parts.append((o.clone(), bits))
    bases = np.concatenate([[0], np.cumsum([b for _, b in parts])]).astype(np.uint64)
    assert int(bases[-1]) == rbits
    dst = torch.zeros(rbits // 32 + 1, dtype=torch.int32, device="cuda")
    for i in reversed(range(len(parts))):                                       # any order
        enc.merge_shard(parts[i][0], parts[i][1], dst, int(bases[i]))
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), ref[: rbits // 32 + 1])


@pytest.mark.gpu
def test_merge_matches_oracle_append_at_odd_offsets(enc):
    """Random bit strings appended at random bit positions: device merge == oracle_bit_append."""
    rng = np.random.default_rng(7)
    dst_ref = np.zeros(4096, dtype=np.uint32)
    dst = torch.zeros(4096, dtype=torch.int32, device="cuda")
    pos = 0
    for _ in range(40):
        nbits = int(rng.integers(1, 3000))
        src = rng.integers(0, 2**32, size=(nbits + 31) // 32, dtype=np.uint32)
        if nbits % 32: src[-1] &= np.uint32((0xffffffff << (32 - nbits % 32)) & 0xffffffff)   # zero padded like the encoder's
        enc.merge_shard(_dev(src), nbits, dst, pos)
        pos = O.bit_append(dst_ref, pos, src, nbits)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), dst_ref)
    with pytest.raises(hvlc.HvlcError):
        enc.merge_shard(_dev(np.ones(4, np.uint32)), 128, dst, 4096 * 32 - 64)  # would run past the destination


@pytest.mark.gpu
def test_decode_round_trip_and_oracle_decoder(enc, golden):
    """decode(encode(x)) == x on the device, and the device decoder agrees with the oracle's on a prefix."""
    x = np.tile(golden["period"], 8).view(np.uint32)[: 8 * 65536 - 999]
    d_tab = enc.table_to_device(golden["cw"], golden["cl"])
    o, offs, bits = enc.encode(_dev(x), d_tab, max_len=21)
    y = enc.decode(o, bits, offs, d_tab, x.size)
    assert np.array_equal(y.cpu().numpy().view(np.uint32), x)
    yo, _ = O.decode(o[: 4096].cpu().numpy().view(np.uint32), 1024, golden["cw"], golden["cl"])
    assert np.array_equal(yo, x[:1024])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["H2.2", "H4.0", "H1.0"])
def test_round_trip_at_full_size(name, golden):
    """Size-independent property at BASELINE's full sizes: encode -> decode reproduces the 4 GiB input exactly
    (tables with codes longer than the decoder's 12-bit fast table included)."""
    from util import dist_weights, SEEDS
    n = 4 << 30
    e = hvlc.Encoder(0, n // 4)
    d_in = e.gen_symbols(n, SEEDS[name], hvlc.cdf32_from_counts(dist_weights(name, golden["period"])))
    hist = e.histogram(d_in).cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.huffman_table(hist)
    d_tab = e.table_to_device(cw, cl)
    o, offs, bits = e.encode(d_in, d_tab, max_len=int(cl.max()))
    y = e.decode(o, bits, offs, d_tab, n // 4)
    assert torch.equal(y.view(torch.uint8), d_in.view(torch.uint8))
    del y, o, offs, d_in
    e.close()


@pytest.mark.gpu
def test_two_contexts_on_two_streams_concurrently(golden):
    """Two contexts encode different inputs on two HIP streams at the same time.  Each kernel is sized for the whole
    device, so neither grid is co-resident while both run: tiles are drawn by running workgroups only, both finish,
    both streams are exact (with co-residency as a requirement this would have been a deadlock)."""
    ea, eb = hvlc.Encoder(0, 1 << 24), hvlc.Encoder(0, 1 << 24)
    xa = np.tile(golden["period"], 200).view(np.uint32)                         # 50 MiB
    xb = np.tile(golden["period"][::-1].copy(), 160).view(np.uint32)[: 160 * 65536 - 777]
    d_tab = ea.table_to_device(golden["cw"], golden["cl"])
    da, db = _dev(xa), _dev(xb)
    oa = torch.empty(hvlc.out_capacity_words(xa.size, 21), dtype=torch.int32, device="cuda")
    ob = torch.empty(hvlc.out_capacity_words(xb.size, 21), dtype=torch.int32, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(4):                                                          # launches interleave on the device
        ea.encode_ptr(da.data_ptr(), xa.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, oa.data_ptr(), oa.numel(), stream=sa.cuda_stream)
        eb.encode_ptr(db.data_ptr(), xb.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, ob.data_ptr(), ob.numel(), stream=sb.cuda_stream)
    bits_a, rc_a = ea.finish(stream=sa.cuda_stream)
    bits_b, rc_b = eb.finish(stream=sb.cuda_stream)
    torch.cuda.synchronize()
    ra, rba = O.encode(xa, golden["cw"], golden["cl"])
    rb, rbb = O.encode(xb, golden["cw"], golden["cl"])
    assert rc_a == 0 and rc_b == 0 and bits_a == rba and bits_b == rbb
    assert np.array_equal(oa[: rba // 32 + 1].cpu().numpy().view(np.uint32), ra)
    assert np.array_equal(ob[: rbb // 32 + 1].cpu().numpy().view(np.uint32), rb)
    ea.close(); eb.close()


@pytest.mark.gpu
def test_encode_is_graph_capturable(golden):
    """hvlc_encode issues ONE kernel launch and nothing else (no allocation, no synchronisation once the context is
    sized), so it can be captured into a HIP graph and replayed on new data -- the launch finds out on the device which
    half of the workspace is its own (the call counter), so every replay alternates like a direct call."""
    e = hvlc.Encoder(0, 1 << 22)
    x1 = np.tile(golden["period"], 16).view(np.uint32)
    x2 = np.tile(golden["period"][::-1].copy(), 16).view(np.uint32)
    d_tab = e.table_to_device(golden["cw"], golden["cl"])
    d_in = _dev(x1)
    out = torch.empty(hvlc.out_capacity_words(x1.size, 21), dtype=torch.int32, device="cuda")
    offs = torch.empty(hvlc.num_segments(x1.size), dtype=torch.int64, device="cuda")
    total = torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call = lambda: e.encode_ptr(d_in.data_ptr(), x1.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024,
                                    out.data_ptr(), out.numel(), offs.data_ptr(), total.data_ptr())
        call()                                                   # warm-up outside the capture
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            call()
    for x in (x1, x2, x1):
        d_in.copy_(_dev(x))
        g.replay()
        torch.cuda.synchronize()
        ref, rbits = O.encode(x, golden["cw"], golden["cl"])
        _, roffs, _ = O.segment_offsets(x, golden["cl"])
        assert int(total.item()) == rbits
        assert np.array_equal(out[: rbits // 32 + 1].cpu().numpy().view(np.uint32), ref)
        assert np.array_equal(offs.cpu().numpy().view(np.uint64), roffs)
    e.close()


@pytest.mark.gpu
def test_bench_runs_its_multi_rank_branch_on_one_gpu():
    """bench.py's N > 1 code end to end -- torch.distributed.run, two ranks, the all-reduced histogram, hvlc.stitch in the
    step, the whole-shard comparison, the per-rank roofline -- rehearsed on this one GPU (HVLC_BENCH_REHEARSAL=1: both
    ranks on cuda:0, collectives over gloo).  The driver's 8-GPU run executes the same branch over RCCL."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HVLC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--size-mib", "64", "--steps", "3",
                        "--warmup", "1", "--no-probe"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["bit_exact"] is True
    assert line["config"]["parallelism"] == "shard2" and line["scaling"] == "weak"
    assert line["checked_bytes"] > 2 * (64 << 20) // 8                   # both shards whole: about 2 x 64 MiB x 0.278
    assert len(line["roofline"]["ranks"]["kernel_ms"]) == 2 and all(v > 0 for v in line["roofline"]["ranks"]["kernel_ms"])
