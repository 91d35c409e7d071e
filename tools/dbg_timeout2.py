import ctypes as C, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT + "/huffman-gpu_amd"); sys.path.insert(0, ROOT + "/tests")
os.environ["HVLC_LIB"] = ROOT + "/huffman-gpu_amd/libhvlc_exp.so"
import numpy as np, torch, hvlc
L = hvlc.lib(); L.hvlc_debug_set_timeline.argtypes = [C.c_void_p]
period = np.fromfile(ROOT + "/tests/golden/g1_period.bin", dtype=np.uint8)
t = np.fromfile(ROOT + "/tests/golden/g1_table.bin", dtype=np.uint32); cw, cl = t[:256], t[256:]
x = np.tile(period, 256).view(np.uint32); ntiles = x.size // 4096
d_in = torch.from_numpy(x.view(np.int32).copy()).cuda()
os.environ["HVLC_TIMEOUT_MS"] = "200"; os.environ["HVLC_BLOCKS_PER_CU"] = "3"
out = torch.empty(hvlc.out_capacity_words(x.size, 21), dtype=torch.int32, device="cuda")
e = hvlc.Encoder(0, 1 << 24); d_tab = e.table_to_device(cw, cl)
def run(tag, tl=None):
    L.hvlc_debug_set_timeline(C.c_void_p(tl.data_ptr() if tl is not None else 0))
    t0 = time.time()
    e.encode_ptr(d_in.data_ptr(), x.size, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel())
    r = e.finish(check=False); print(tag, r, e.launch_info(), round(time.time() - t0, 4), flush=True)
run("warm")
info = e.launch_info(); G = info["compute_units"] * info["blocks_per_cu"]
os.environ["HVLC_TEST_FORCE_GRID"] = str(2 * G)
run("forced")
del os.environ["HVLC_TEST_FORCE_GRID"]
tl = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
run("after", tl)
a = tl.cpu().numpy().reshape(ntiles, 8).astype(np.int64)
t0 = a[:, 0][a[:, 0] > 0].min()
us = lambda v: np.where(v > 0, (v - t0) / 100.0, -1.0)
pack, start, res, store = us(a[:, 0]), us(a[:, 1]), us(a[:, 4]), us(a[:, 5])
print("tiles never packed:", int((a[:, 0] == 0).sum()), "never resolved:", int((a[:, 4] == 0).sum()), "never stored:", int((a[:, 5] == 0).sum()))
late = np.nonzero((res - pack > 1000) | (a[:, 4] == 0))[0]
print("late/unresolved tiles:", late[:40], "count", late.size)
for ti in late[:12]:
    print(f"  tile {ti} wg {ti % G} n {ti // G} pack {pack[ti]:.1f} start {start[ti]:.1f} ingroup {us(a[:,2])[ti]:.1f}/{us(a[:,3])[ti]:.1f} res {res[ti]:.1f} store {store[ti]:.1f} steps {a[ti,6]}")
first = np.argsort(np.where(a[:, 0] > 0, pack, 1e18))[:5]
print("first packed:", first, pack[first])
lastp = np.argsort(-pack)[:5]; print("last packed:", lastp, pack[lastp])
