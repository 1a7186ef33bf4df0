import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phendiff_amd._lib as L
L.LIB_PATH = os.environ["PD_LIB"]
import runpy, numpy as np, torch
sys.argv = ["bench_conv.py"] + sys.argv[1:] + ["--iters", "1"]
runpy.run_path(os.path.join(os.path.dirname(__file__), "bench_conv.py"), run_name="__main__")
torch.cuda.synchronize()
lib = L.lib()
for nb in (27200, 54400, 65536, 81920):
    lib.pd_debug_conv_occupancy(nb)
buf = (C.c_ulonglong * (4096 * 16))()
rc = lib.pd_debug_read_conv_stamps(buf, C.c_size_t(C.sizeof(buf)))
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.int64)
a = a[a[:, 6] > 0]
d = np.diff(a[:, :7], axis=1)
names = ["setup+load0+write0", "barrier0", "chunk0 mma(+write1)", "barrier1", "rest chunks", "epilogue"]
print("workgroups:", len(a), " s_memtime ticks per WG: median", np.median(a[:, 6] - a[:, 0]), " span of all:", a[:, 6].max() - a[:, 0].min())
pro = [("index setup", 0, 7), ("issue loads(0)", 7, 8), ("acc init + weight ring", 8, 9), ("wait loads + piece 0", 9, 10), ("pieces 1..5 + issue loads(1)", 10, 1)]
for n, i0, i1 in pro:
    dd = a[:, i1] - a[:, i0]
    print(f"    prologue/{n:30s} median {np.median(dd):9.0f}  p90 {np.percentile(dd, 90):9.0f}")
for i, n in enumerate(names):
    print(f"  {n:22s} median {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
# effective concurrency per XCD group (blocks b and b+8 share an XCD; s_memtime is per-XCD)
for x in range(8):
    idx = np.arange(len(a)) % 8 == x
    g = a[idx]
    span = g[:, 6].max() - g[:, 0].min()
    conc = (g[:, 6] - g[:, 0]).sum() / span
    if x < 2:
        print(f"  xcd-group {x}: {idx.sum()} WGs, span {span} ticks, avg concurrent WGs {conc:.1f} (= {conc/32:.2f} per CU)")
