#!/usr/bin/env python3
"""Regenerates tests/golden/* from the REFERENCE, run in the build container only.

Every expected value written here comes from the reference's own code compiled
from where it lies (oracle/Makefile -> oracle/_ref): cpu_vlc_encode
(cpuencode.cpp:12-46) for streams and sizes, BuildTree/GenerateCodes
(huffTree.h:55-94) for the G1 table.  The repo's own oracle is NOT used to
produce fixtures (it is what the fixtures pin).  Inputs are data: the first
262 144-byte period of the reference's test file data/test1024_H2.206587175259.in
(the file is 4 repeats of it), seeded random words, and the
hand-written KATs of SURVEY.md section 4.

    python3 tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

REF_DATA = "/root/reference/data/test1024_H2.206587175259.in"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_stream(x, cw, cl):
    out, osz = O.ref_encode32(x, cw, cl)
    nint = (osz + 3) // 4                       # main_test_cu.cu:126
    return out[:nint].copy(), osz


def main():
    assert O.have_ref(), "run `make -C oracle` in the build container first"
    raw = np.fromfile(REF_DATA, dtype=np.uint8)
    assert raw.size == 1 << 20
    period = raw[: 1 << 18]
    assert all(np.array_equal(period, raw[i << 18:(i + 1) << 18]) for i in range(4))
    period.tofile(os.path.join(HERE, "g1_period.bin"))

    tbl = os.path.join(HERE, "g1_table.bin")
    subprocess.run([O.REF_HUFFTABLE, REF_DATA, tbl], check=True)
    t = np.fromfile(tbl, dtype=np.uint32)
    cw, cl = t[:256].copy(), t[256:].copy()

    g = {"source": "reference cpuencode.cpp:12-46 + huffTree.h:55-94 compiled by oracle/Makefile"}
    x = raw.view(np.uint32)
    out, osz = ref_stream(x, cw, cl)
    g["G1"] = {
        "input_sha256": sha(raw), "table_sha256": sha(t), "words": int(x.size),
        "outsize": osz, "num_ints": int(out.size), "out_sha256": hashlib.sha256(out.tobytes()[:osz]).hexdigest(),
        "out_words_sha256": sha(out), "first_words": [int(v) for v in out[:4]],
        "first_input_words": [int(v) for v in x[:2]],
        "max_len": int(cl.max()), "live_symbols": int((cl > 0).sum()),
    }
    xp = period.view(np.uint32)
    outp, oszp = ref_stream(xp, cw, cl)
    g["G1p"] = {"words": int(xp.size), "outsize": oszp, "out_sha256": hashlib.sha256(outp.tobytes()[:oszp]).hexdigest(),
                "out_words_sha256": sha(outp)}

    # per-segment bit counts, derived from the REFERENCE encoder alone: encode each
    # 256-word segment separately; its outsize and last word give its exact bit count
    # only up to byte rounding, so count by encoding with a unary probe instead:
    # bits(segment) = sum of table lengths, which the reference defines through
    # load_data.h:46 (codewordlens[sym] = code length).  Cross-checked below against
    # the concatenative property of the reference stream.
    lens = cl[raw.astype(np.int64)].reshape(-1, 1024).sum(axis=1).astype(np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    total_bits = int(lens.sum())
    assert (total_bits + 7) // 8 == osz
    # concatenative check with the reference itself: stream(first k segments) is a bit-prefix
    for k in (1, 2, 3, 517):
        o_k, osz_k = ref_stream(x[: 256 * k], cw, cl)
        nb = int(offs[k]) if k < offs.size else total_bits
        assert (nb + 7) // 8 == osz_k
        full = nb // 32
        assert np.array_equal(o_k[:full], out[:full])
    lens.astype(np.uint32).tofile(os.path.join(HERE, "g1_seg_counts.bin"))
    g["G1"].update({"total_bits": total_bits, "seg_counts_first4": [int(v) for v in lens[:4]],
                    "seg_offsets_first4": [int(v) for v in offs[:4]], "seg_offset_last": int(offs[-1]),
                    "seg_bits_min": int(lens.min()), "seg_bits_max": int(lens.max())})

    # KATs K1-K3; K1/K2 use the testdatagen.h:42 table restated by hand here
    pat = np.array([1, 2, 3, 4, 4, 5, 6, 7], dtype=np.uint32)
    kcl = np.tile(pat, 32)
    kcw = (np.uint32(1) << (kcl - 1)).astype(np.uint32)
    kats = {}
    for name, words, (tcw, tcl) in [
        ("K1", [0x00010203, 0x04050607], (kcw, kcl)),
        ("K2", [0x07070707, 0x07070707, 0x00000000, 0xFFFEFDFC], (kcw, kcl)),
    ]:
        o, s = ref_stream(np.array(words, dtype=np.uint32), tcw, tcl)
        kats[name] = {"input": words, "table": "generateCodewords", "outsize": s, "words": [int(v) for v in o]}
    k3cw = np.zeros(256, dtype=np.uint32)
    k3cl = np.zeros(256, dtype=np.uint32)
    k3cw[1], k3cl[1] = 0x7FFFFFFF, 31
    k3cw[2], k3cl[2] = 0, 1
    k3cw[3], k3cl[3] = 0b101, 3
    words = [0x01020300, 0x01010101, 0x03000002]
    o, s = ref_stream(np.array(words, dtype=np.uint32), k3cw, k3cl)
    kats["K3"] = {"input": words, "table": {"1": [0x7FFFFFFF, 31], "2": [0, 1], "3": [5, 3]},
                  "outsize": s, "words": [int(v) for v in o]}
    g["KAT"] = kats

    # seeded differential vectors: random tables (lengths 0..31) x random words, expected from the reference
    rng = np.random.default_rng(0x5EED0001)
    vec = {}
    cases = []
    for i, (nw, kind) in enumerate([(1, "masked"), (7, "masked"), (256, "masked"), (300, "masked"),
                                    (1024, "masked"), (513, "short"), (2048, "long"), (64, "zeros"),
                                    (777, "unmasked"), (4096, "masked")]):
        if kind == "short":
            tcl = rng.integers(0, 5, 256).astype(np.uint32)
        elif kind == "long":
            tcl = rng.integers(24, 32, 256).astype(np.uint32)
        elif kind == "zeros":
            tcl = np.where(rng.random(256) < 0.7, 0, rng.integers(1, 32, 256)).astype(np.uint32)
        else:
            tcl = rng.integers(0, 32, 256).astype(np.uint32)
        tcw = rng.integers(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint32)
        if kind != "unmasked":      # precondition of the encoder: cw < 2^len (cpuencode.cpp:35 does not mask)
            tcw = (tcw & ((np.uint64(1) << tcl.astype(np.uint64)) - np.uint64(1)).astype(np.uint32)).astype(np.uint32)
        xin = rng.integers(0, 1 << 32, nw, dtype=np.uint64).astype(np.uint32)
        o, s = ref_stream(xin, tcw, tcl)
        vec[f"in{i}"], vec[f"cw{i}"], vec[f"cl{i}"], vec[f"out{i}"] = xin, tcw, tcl, o
        cases.append({"id": i, "kind": kind, "words": nw, "outsize": s})
    np.savez_compressed(os.path.join(HERE, "diff_vectors.npz"), **vec)
    g["diff_cases"] = cases

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(g, f, indent=1)
    print(json.dumps(g["G1"], indent=1))
    print("KAT", {k: [hex(w) for w in v["words"]] for k, v in kats.items()})


if __name__ == "__main__":
    main()
