import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
from util import synth_on_device


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = mib << 20
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
    e = hvlc.Encoder(0, n // 4)
    d_in, cw, cl, hist = synth_on_device(e, "H2.2", n, period)
    d_tab = e.table_to_device(cw, cl)
    out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
    offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
    tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    call = lambda: e.encode_ptr(d_in.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), tot.data_ptr())
    K = 200 if mib <= 1024 else 60
    for mode in ("plain", "torch_event", "profile", "both", "plain"):
        for _ in range(20): call()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        if mode in ("profile", "both"): e.profile_begin(K)
        t0 = time.perf_counter()
        for i in range(K):
            call()
            if mode in ("torch_event", "both"): ev[i].record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        if mode in ("profile", "both"): e.profile_collect(K)
        print(f"{mode:12s} {dt*1e6:7.1f} us per step  {n/dt/1e9:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
