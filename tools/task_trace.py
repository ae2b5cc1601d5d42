"""Timeline of the POA slab tiers (M1/M2/L) for one batch of the bench workload: when each task ran, how long, where it ended.
CW_TASK_TRACE=1 python tools/task_trace.py [workload] -- GPU box only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CW_TASK_TRACE"] = "1"
import torch  # noqa: E402

import consent_amd as ca  # noqa: E402
from consent_amd.engine import Batch, Result  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "pacbio_d150_msa150"
depth, msa = (150, 150) if "150" in wl else (30, 20)
n_win = int(os.environ.get("CW_TRACE_WINDOWS", "16384"))
eng = ca.Engine(ca.Params(9, 4, 8, 2, msa))
lib = eng.lib
dev = torch.device("cuda", 0)
spec = ca.SynthSpec.pacbio(n_win, depth, first_window=int(os.environ.get("CW_TRACE_BATCH", "0")) * n_win)  # CW_TRACE_BATCH: which of bench.py's batches
ns, nw = C.c_uint32(), C.c_uint64()
lib.cw_synth_sizes(C.byref(spec), C.byref(ns), C.byref(nw))
t = [torch.zeros(n_win + 1, dtype=torch.int32, device=dev), torch.zeros(ns.value, dtype=torch.int32, device=dev), torch.zeros(ns.value, dtype=torch.int64, device=dev),
     torch.zeros(nw.value + 4, dtype=torch.int32, device=dev)]
torch.cuda.synchronize()
lib.cw_synth_device(eng.handle, C.byref(spec), *[x.data_ptr() for x in t], None)
cc, sc = 3 * 500 + 256, (depth + 1) * 524 // 4 + 16
r = [torch.zeros(n_win * cc, dtype=torch.uint8, device=dev), torch.arange(n_win + 1, dtype=torch.int64, device=dev) * cc, torch.zeros(n_win, dtype=torch.int32, device=dev),
     torch.zeros(n_win, dtype=torch.uint8, device=dev), torch.zeros(n_win * sc, dtype=torch.int32, device=dev), torch.arange(n_win + 1, dtype=torch.int64, device=dev) * sc,
     torch.zeros(n_win, dtype=torch.int32, device=dev)]
torch.cuda.synchronize()
b = Batch(n_win, ns.value, nw.value, *[x.data_ptr() for x in t])
rs = Result(*[x.data_ptr() for x in r])
for _ in range(2):
    eng.run_device(b, rs)
    torch.cuda.synchronize()
print("stage ms", {k: round(v, 2) for k, v in eng.timings().items()})
cap = 1 << 20
out = np.zeros((cap, 12), np.uint32)
n = C.c_uint32()
lib.cw_debug_task_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
assert lib.cw_debug_task_trace(eng.handle, cap, out.ctypes.data, C.byref(n)) == 0
a = out[: n.value]
a = a.copy()
mean_len = a[:, 4] >> 16
a[:, 4] &= 0xFFFF
tier = a[:, 10] & 0xFF
rc = (a[:, 10] >> 8) & 0xFF
lw = (a[:, 10] >> 24) & 1  # ran in tier LW (four waves per task, cw_poa_w.h)
start, dur = a[:, 8] * 1e-5, a[:, 9] * 1e-5  # ms
for tr, name in ((1, "M1"), (2, "M2"), (3, "L")):
    m = (tier == tr) & (dur > 0)
    if not m.any():
        continue
    end = start[m] + dur[m]
    print(f"tier {name}: {m.sum()} task runs, busy {dur[m].sum():.0f} wave-ms, first start {start[m].min():.2f} ms, last end {end.max():.2f} ms, longest {dur[m].max():.2f} ms, handed on (rc 2) {int((rc[m] == 2).sum())}")
    order = np.argsort(-end)[:12]
    idx = np.nonzero(m)[0][order]
    for i in idx:
        print(f"    ends {start[i] + dur[i]:7.2f}  start {start[i]:7.2f}  dur {dur[i]:6.2f} ms  members {a[i, 3]:4d}  max_len {a[i, 4]:4d}  rc {rc[i]}  wave {a[i, 11]}")
    # occupancy over time
    edges = np.linspace(0, end.max(), 11)
    occ = [int(((start[m] < e) & (start[m] + dur[m] > e)).sum()) for e in edges[1:-1]]
    print("    running at 10%..90% of the tier's span:", occ)
m_lw = (tier == 3) & (lw == 1) & (dur > 0)
if m_lw.any():
    print(f"tier LW (part of tier L above): {int(m_lw.sum())} task runs, busy {dur[m_lw].sum():.0f} work-group-ms, last end {(start[m_lw] + dur[m_lw]).max():.2f} ms, longest {dur[m_lw].max():.2f} ms; mean member length 10/50/90: {np.percentile(mean_len[m_lw], [10, 50, 90])}")
m_l1 = (tier == 3) & (lw == 0) & (dur > 0)
if m_l1.any():
    wide = m_l1 & (mean_len >= 100)
    print(f"tier L on one wave: {int(m_l1.sum())} task runs, longest {dur[m_l1].max():.2f} ms; of them with mean member length >= 100: {int(wide.sum())}, busy {dur[wide].sum():.0f} wave-ms, longest {dur[wide].max() if wide.any() else 0:.2f} ms")
s_m = tier == 0
print(f"tier S: {int(s_m.sum())} tasks; max_len percentiles 10/50/90/99: {np.percentile(a[s_m, 4], [10, 50, 90, 99])}; members 10/50/90/99: {np.percentile(a[s_m, 3], [10, 50, 90, 99])}")
ml, nm = a[s_m, 4].astype(np.int64), a[s_m, 3].astype(np.int64)
for lo, hi in ((0, 8), (8, 16), (16, 24), (24, 32), (32, 64), (64, 128)):
    m = (ml >= lo) & (ml < hi)
    print(f"    max_len [{lo},{hi}): {int(m.sum()):7d} tasks, {int(nm[m].sum()):9d} members, sum members*len^2 {int((nm[m] * ml[m] * ml[m]).sum()):12d}")
for tr, name in ((1, "M1"), (2, "M2"), (3, "L")):
    m = tier == tr
    ml, nm = a[m, 4].astype(np.int64), a[m, 3].astype(np.int64)
    print(f"tier {name}: max_len 10/50/90/99 {np.percentile(ml, [10, 50, 90, 99])}; members {np.percentile(nm, [10, 50, 90, 99])}")
    for lo, hi in ((0, 32), (32, 64), (64, 128), (128, 256), (256, 1024)):
        q = (ml >= lo) & (ml < hi)
        print(f"    max_len [{lo},{hi}): {int(q.sum()):7d} tasks, busy {dur[m][q].sum():9.0f} wave-ms")
if os.environ.get("CW_FIT"):
    for tr, name in ((1, "M1"), (2, "M2"), (3, "L")):
        m = (tier == tr) & (dur > 0.05) & (rc == 1)
        x1, x2, x3, y = np.log(a[m, 3].astype(float)), np.log(a[m, 4].astype(float)), np.log(np.maximum(mean_len[m], 1).astype(float)), np.log(dur[m])
        A = np.stack([x1, x2, x3, np.ones_like(x1)], 1)
        coef, *_ = np.linalg.lstsq(A, y, rcond=None)
        pred = A @ coef
        print(f"fit {name}: dur ~ members^{coef[0]:.2f} * maxlen^{coef[1]:.2f} * meanlen^{coef[2]:.2f} * {np.exp(coef[3]):.3g} ms; residual sd of log {np.std(y - pred):.2f}; n={int(m.sum())}")
