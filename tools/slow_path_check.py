#!/usr/bin/env python3
"""Developer tool: encode random bytes with a table kind whose parts all take the slow path (lengths up to 31), compare
with the oracle.  python tools/slow_path_check.py <mib> [kind]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
import oracle_lib as O
from util import random_table
mib = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "any"
rng = np.random.default_rng(3)
n = mib << 20
cw, cl = random_table(rng, kind)
x = rng.integers(0, 1 << 32, n // 4, dtype=np.uint64).astype(np.uint32)
enc = hvlc.Encoder(0, n // 4)
d_in = torch.from_numpy(x.view(np.int32)).cuda()
d_tab = enc.table_to_device(cw, cl)
t0 = time.time()
out, offs, _ = enc.encode(d_in, d_tab, max_len=int(cl.max()), sync=False)
bits, rc = enc.finish(check=False)
dt = time.time() - t0
print(f"{kind} {mib} MiB: rc {rc} bits {bits} time {dt*1e3:.1f} ms  {n/dt/1e9:.3f} GB/s", flush=True)
if rc == 0:
    bad, first = O.compare_stream_mt(x, cw, cl, out[: bits // 32 + 1].cpu().numpy().view(np.uint32),
                                     offs.cpu().numpy().view(np.uint64), bits, chunk_segments=512)
    print("differing chunks:", bad, "first segment", first)
