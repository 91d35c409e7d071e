import os, sys, time
sys.path.insert(0, "/root/repo/huffman-gpu_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, hvlc
period = np.fromfile("/root/repo/tests/golden/g1_period.bin", dtype=np.uint8)
t = np.fromfile("/root/repo/tests/golden/g1_table.bin", dtype=np.uint32); cw, cl = t[:256], t[256:]
x = np.tile(period, 256).view(np.uint32)
d_in = torch.from_numpy(x.view(np.int32).copy()).cuda()
os.environ["HVLC_TIMEOUT_MS"] = "200"
out = torch.empty(hvlc.out_capacity_words(x.size, 21), dtype=torch.int32, device="cuda")
def run(e, tag):
    t0 = time.time()
    e.encode_ptr(d_in.data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel())
    r = e.finish(check=False); print(tag, r, e.launch_info()["grid_blocks"], round(time.time() - t0, 4), flush=True)
for variant in ("sleep", "flush", "none"):
    e = hvlc.Encoder(0, 1 << 24); d_tab = e.table_to_device(cw, cl)
    run(e, variant + ": warm normal")
    info = e.launch_info()
    os.environ["HVLC_TEST_FORCE_GRID"] = str(2 * info["compute_units"] * info["blocks_per_cu"])
    run(e, variant + ": forced")
    del os.environ["HVLC_TEST_FORCE_GRID"]
    if variant == "sleep":
        torch.cuda.synchronize(); time.sleep(0.6)
    if variant == "flush":
        big = torch.zeros(1 << 28, dtype=torch.int32, device="cuda"); big.add_(1); torch.cuda.synchronize(); del big
    run(e, variant + ": normal after")
    run(e, variant + ": normal after 2")
    e.close()
