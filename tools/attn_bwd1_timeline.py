"""Step-level timeline of the one-pass attention backward at the benchmark shape (needs tools/build_trace_lib.sh's diagnostic library):
    VBX_LIB_PATH=voicebox-pytorch_amd/lib/libvbx_hip_trace.so python tools/attn_bwd1_timeline.py
Thread 0 of every persistent workgroup stamps 8 points of each query-tile step (attn_bwd1.inc B1_STAMP).  Prints where a step's time
goes, by chain position, how long items live, and how busy the workgroup slots are."""
import ctypes, os, sys
import numpy as np, torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, H, Np = int(os.environ.get("BATCH", 8)), 16, int(os.environ.get("NP", 1040))
g = torch.Generator().manual_seed(0)
q = torch.randn(B, H, Np, 64, generator=g); k = torch.randn(B, H, Np, 64, generator=g); v = torch.randn(B, H, Np, 64, generator=g)
q = q / q.norm(dim=-1, keepdim=True) * 8; k = k / k.norm(dim=-1, keepdim=True) * 8
qd, kd, vd = (q * L.lib().vbx_attn_q_prescale(10.0)).half().to(dev), k.half().to(dev), v.half().to(dev)  # q16 contract: include/vbx.h
qb, kb, vb = q.bfloat16().to(dev), k.bfloat16().to(dev), v.bfloat16().to(dev)
out16 = torch.empty(B, Np, H * 64, dtype=torch.float16, device=dev)
lse = torch.empty(B, H, Np, device=dev)
st = torch.cuda.current_stream().cuda_stream
dout = (torch.randn(B, Np, H * 64, generator=g) * 1e-3).bfloat16().to(dev)
delta = torch.empty(B, H, Np, device=dev)
dq = torch.zeros(B, H, Np, 64, device=dev); dk = torch.zeros(B, H, Np, 64, device=dev)
dv = torch.zeros(B, Np, H * 64, dtype=torch.bfloat16, device=dev)
scratch = torch.empty(L.lib().vbx_attn_bwd_scratch_bytes(B, H, Np), dtype=torch.uint8, device=dev)
L.lib().vbx_attn_bwd_select(2)
L.call("vbx_attn_fwd", qd, kd, vd, None, out16, None, lse, B, H, Np, 10.0, st)


def run():
    L.call("vbx_attn_bwd", qd, kd, qb, kb, vb, None, out16, 1, dout, lse, delta, dq, dk, dv.data_ptr(), H * 64, B, H, Np, 10.0, scratch, st)


for _ in range(3):
    run()
torch.cuda.synchronize()
trace = torch.zeros(512, 96, 9, dtype=torch.int64, device=dev)
fn = L.lib().vbx_debug_attn_bwd1_trace
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(trace.data_ptr()) == 0
run()
torch.cuda.synchronize()
fn(None)
t = trace.cpu().numpy()
ok = t[:, :, 0] != 0
t0 = t[:, :, 0][ok].min()
us = (t[:, :, :8] - t0) / 100.0
meta = t[:, :, 8]
kbv, qtv = (meta >> 8) & 0xFF, meta & 0xFF
n_kb = (Np + 127) // 128
names = ["wait tile+stores / barrier A", "publish + DMA issue + flag load", "2 blocks (S dP softmax dV dK, dS^T write)", "LDS drain + flag wait + dq loads issued",
         "barrier B", "dQ MFMAs (16 tr reads, 8 MFMA)", "dq loads wait + add + stores issued / finish"]
print(f"span of the launch (first stamp -> last stamp): {us[ok].max():.1f} us; {int(ok.sum())} steps recorded on {int(ok.any(1).sum())} workgroups")
for label, sel in (("chain head (kb = 0)", ok & (kbv == 0)), ("middle members (0 < kb < last)", ok & (kbv > 0) & (kbv < n_kb - 1)),
                   ("last member (kb = last: tail keys + final epilogue)", ok & (kbv == n_kb - 1))):
    if not sel.any():
        continue
    d = np.diff(us, axis=2)[sel]
    print(f"-- {label}: {int(sel.sum())} steps, mean step {d.sum(1).mean():.2f} us (median {np.median(d.sum(1)):.2f}, p90 {np.percentile(d.sum(1), 90):.2f})")
    for i, n in enumerate(names):
        print(f"     {n:<62s} mean {d[:, i].mean():6.2f}  median {np.median(d[:, i]):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f} us")
# per-item lifetimes and inter-step gaps of a workgroup
items, gaps, seams = [], [], []
for w in range(512):
    idx = np.nonzero(ok[w])[0]
    if len(idx) == 0:
        continue
    start = idx[0]
    for a_, b_ in zip(idx[:-1], idx[1:]):
        same = (meta[w, a_] >> 8) == (meta[w, b_] >> 8)
        (gaps if same else seams).append(us[w, b_, 0] - us[w, a_, 7])
        if not same:
            items.append((kbv[w, start], us[w, a_, 7] - us[w, start, 0]))
            start = b_
    items.append((kbv[w, start], us[w, idx[-1], 7] - us[w, start, 0]))
items = np.array(items)
for label, m in (("full items", items[:, 0] < n_kb - 1), ("last-member items", items[:, 0] == n_kb - 1)):
    if m.any():
        print(f"{label}: {int(m.sum())}, query loop {items[m, 1].mean():.1f} us mean, {items[m, 1].min():.1f} min, {items[m, 1].max():.1f} max")
print(f"gap between steps of an item: mean {np.mean(gaps):.2f} us; between items of a workgroup (epilogue + queue pull + prologue): mean {np.mean(seams):.2f} us")
first_start = np.array([us[w, np.nonzero(ok[w])[0][0], 0] for w in range(512) if ok[w].any()])
last_end = np.array([us[w, np.nonzero(ok[w])[0][-1], 7] for w in range(512) if ok[w].any()])
print(f"workgroups start their first step at {first_start.mean():.1f} us (max {first_start.max():.1f}); finish their last at {last_end.mean():.1f} us mean, {last_end.min():.1f} min, {last_end.max():.1f} max")
