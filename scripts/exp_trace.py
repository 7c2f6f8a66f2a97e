"""Timeline of the one-launch row-band kernel from a library built with -DSGM3_TRACE (scripts/exp_build.sh trace "-DSGM3_TRACE"
sgm3.hip; run with MDISP_LIB_PATH): per ticket the start, the end of the wait for the other direction, the end, and the time
the communication wave waited for the band above.  Prints the occupancy of the device's workgroup places over time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from cameracalibration_amd import _lib, api, shard

H, W, D, B = 1080, 1920, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _lib.load()
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    _lib.tuning_set(getattr(_lib, "TUNE_" + k), int(v))
ts = torch.cuda.Stream()
ctx = _lib.Context(device=0, stream=ts.cuda_stream)
params = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
lefts = torch.empty((B, H, W), dtype=torch.uint8); rights = torch.empty((B, H, W), dtype=torch.uint8)
for i in range(B):
    L, R, _ = api.synth_pair(shard.pair_seed(i), H, W, D)
    lefts[i] = torch.from_numpy(L); rights[i] = torch.from_numpy(R)
lefts, rights = lefts.cuda(), rights.cuda()
disp = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
wsb = lib.mdisp_sgbm_batch_workspace_bytes(H, W, B, C.byref(params))
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    _lib.check(lib.mdisp_sgbm_batch_dev(ctx.handle, lefts.data_ptr(), rights.data_ptr(), B, H * W, H, W, W, C.byref(params),
                                        disp.data_ptr(), ws.data_ptr(), wsb))
    ctx.synchronize()
nb = (H + 15) // 16
n = 2 * nb * B
buf = np.zeros((n, 6), np.uint64)
raw = C.CDLL(_lib.LIB_PATH)
raw.md_sgm3_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert raw.md_sgm3_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.astype(np.float64) / 100.0     # microseconds (100 MHz counter)
t0 = t[:, 0].min()
start, ready, end, wait = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0, t[:, 3]
total = end.max()
ticks = (buf[:, 2] - buf[:, 0]).astype(np.float64)
cyc = (buf[:, 5] - buf[:, 4]).astype(np.float64)
mhz = cyc / ticks * 100.0
print("shader clock while a band runs (shader cycles per 100 MHz tick): mean %.0f MHz  p10 %.0f  p90 %.0f" % (mhz.mean(), *np.percentile(mhz, [10, 90])))
print("pairs %d  tickets %d  kernel span %.2f ms" % (B, n, total / 1e3))
dur = end - ready
print("band duration (ready -> end): mean %.0f us  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("wait for the other direction (start -> ready): mean %.0f us, sum %.1f ms of workgroup time" % ((ready - start).mean(), (ready - start).sum() / 1e3))
print("communication wave waiting for the band above: mean %.0f us per band, sum %.1f ms" % (wait.mean(), wait.sum() / 1e3))
print("workgroup time: held %.1f ms, of it working (ready -> end minus hand-off wait) %.1f ms; places x span %.1f ms" %
      ((end - start).sum() / 1e3, (dur - wait).sum() / 1e3, 1024 * total / 1e3))
step = total / 40
print("time ms | resident | past the meeting wait | started in bin | mean duration of those")
for k in range(40):
    a, b = k * step, (k + 1) * step
    mid = (a + b) / 2
    res = ((start <= mid) & (end > mid)).sum()
    act = ((ready <= mid) & (end > mid)).sum()
    st = (start >= a) & (start < b)
    print("%6.2f  %5d  %5d  %5d  %7.0f" % (mid / 1e3, res, act, st.sum(), dur[st].mean() if st.any() else 0))
# by band step
per = 2 * B
print("band step | first start ms | last end ms | mean duration us | mean hand-off wait us | mean meeting wait us")
for k in range(0, nb, 4):
    s = slice(k * per, (k + 1) * per)
    print("%3d  %7.2f  %7.2f  %7.0f  %7.0f  %7.0f" % (k, start[s].min() / 1e3, end[s].max() / 1e3, dur[s].mean(), wait[s].mean(), (ready[s] - start[s]).mean()))
