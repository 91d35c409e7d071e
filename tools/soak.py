#!/usr/bin/env python3
"""Developer tool: soak the encoder -- random sizes, tables and distributions, back to back on two contexts, every
stream and offset array compared with the oracle.  python tools/soak.py [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
import oracle_lib as O
from util import random_table
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
encs = [hvlc.Encoder(0, 1 << 24), hvlc.Encoder(0, 1 << 24)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
t0 = time.time(); n = 0; nbytes = 0; t_last = t0
while time.time() - t0 < secs:
    jobs = []
    for i in range(2):
        kind = rng.choice(["any", "short", "long", "zeros", "len4", "huff", "huff", "huff"])
        nw = int(rng.choice([rng.integers(1, 5000), rng.integers(5000, 400000), rng.integers(400000, 6000000)]))
        if kind == "huff":
            p = rng.dirichlet(np.full(int(rng.integers(2, 200)), float(rng.choice([0.05, 0.3, 1.0]))))
            sym = rng.choice(256, size=p.size, replace=False)
            x8 = sym[rng.choice(p.size, size=4 * nw, p=p)].astype(np.uint8)
            x = x8.view(np.uint32)
            cw, cl = hvlc.build_table(np.bincount(x8, minlength=256).astype(np.uint64), int(rng.choice([31, 16, 12])))
        else:
            cw, cl = random_table(rng, kind)
            x = rng.integers(0, 1 << 32, nw, dtype=np.uint64).astype(np.uint32)
        with torch.cuda.stream(streams[i]):
            d_in = torch.from_numpy(x.view(np.int32).copy()).cuda(non_blocking=False)
            d_tab = encs[i].table_to_device(cw, cl)
            out, offs, _ = encs[i].encode(d_in, d_tab, max_len=int(cl.max()) if cl.max() else 1, sync=False)
        jobs.append((x, cw, cl, out, offs))
    torch.cuda.synchronize()
    for i, (x, cw, cl, out, offs) in enumerate(jobs):
        bits, rc = encs[i].finish(stream=streams[i].cuda_stream, check=False)
        assert rc == 0, (rc, x.size)
        ref, rbits = O.encode(x, cw, cl)
        assert bits == rbits, (bits, rbits, x.size)
        got = out[: bits // 32 + 1].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, ref), ("stream", x.size, int(cl.max()))
        _, roffs, _ = O.segment_offsets(x, cl)
        assert np.array_equal(offs[: roffs.size].cpu().numpy().view(np.uint64), roffs), ("offsets", x.size)
        n += 1; nbytes += 4 * x.size
    if time.time() - t_last > 30:                      # (a run that prints nothing for minutes is taken to be hung)
        t_last = time.time(); print(f"  {n} encodes, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {n} encodes, {nbytes / 1e9:.2f} GB, {time.time() - t0:.0f} s")
