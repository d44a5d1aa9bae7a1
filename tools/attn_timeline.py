"""Per-workgroup timeline of the attention kernels at the benchmark shape (needs tools/build_trace_lib.sh's diagnostic library).
    VBX_LIB_PATH=voicebox-pytorch_amd/lib/libvbx_hip_trace.so VBX_ATTN_BWD_DMA=3 python tools/attn_timeline.py
Prints, per kernel: the launch span, when workgroups started (in rounds or as a flow), how long full-tile and tail-tile workgroups
lived, and the loop / epilogue split -- the facts the tile scheduling is designed on."""
import ctypes, os, sys
import numpy as np, torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, H, Np = 8, 16, int(os.environ.get("NP", 1040))
g = torch.Generator().manual_seed(0)
q = torch.randn(B, H, Np, 64, generator=g); k = torch.randn(B, H, Np, 64, generator=g); v = torch.randn(B, H, Np, 64, generator=g)
q = q / q.norm(dim=-1, keepdim=True) * 8; k = k / k.norm(dim=-1, keepdim=True) * 8
qd, kd, vd = (q * L.lib().vbx_attn_q_prescale(10.0)).half().to(dev), k.half().to(dev), v.half().to(dev)  # q16 contract: include/vbx.h
qb, kb, vb = q.bfloat16().to(dev), k.bfloat16().to(dev), v.bfloat16().to(dev)
out16 = torch.empty(B, Np, H * 64, dtype=torch.float16, device=dev); out = torch.empty(B, Np, H * 64, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, Np, device=dev)
st = torch.cuda.current_stream().cuda_stream
dout = (torch.randn(B, Np, H * 64, generator=g) * 1e-3).bfloat16().to(dev)
delta = torch.empty(B, H, Np, device=dev)
dq = torch.zeros(B, H, Np, 64, device=dev); dk = torch.zeros(B, H, Np, 64, device=dev)
dv = torch.zeros(B, Np, H * 64, dtype=torch.bfloat16, device=dev)
grid = ((Np + 127) // 128) * B * H
trace = torch.zeros(2, 8192, 4, dtype=torch.int64, device=dev)
fn = L.lib().vbx_debug_attn_trace
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int


scratch = torch.empty(L.lib().vbx_attn_bwd_scratch_bytes(B, H, Np), dtype=torch.uint8, device="cuda:0")


def run():
    L.call("vbx_attn_fwd", qd, kd, vd, None, out16, out, lse, B, H, Np, 10.0, st)
    L.call("vbx_attn_bwd", qd, kd, qb, kb, vb, None, out16, 1, dout, lse, delta, dq, dk, dv.data_ptr(), H * 64, B, H, Np, 10.0, scratch, st)


for _ in range(3):
    run()
torch.cuda.synchronize()
assert fn(trace.data_ptr()) == 0
run()
torch.cuda.synchronize()
fn(None)
t = trace.cpu().numpy()
nfull = (Np // 128) * B * H
for region, name in ((0, "forward launch"), (1, "backward launch(es)")):
    r = t[region]
    r = r[r[:, 0] != 0]
    if len(r) == 0:
        continue
    t0 = r[:, 0].min()
    s, l, e = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, (r[:, 2] - t0) / 100.0  # us
    cu = (r[:, 3] >> 8) & 0xF; se = (r[:, 3] >> 13) & 0x7; xcc = (r[:, 3] >> 32) & 0xF; tag = (r[:, 3] >> 48) & 0xF
    cuid = (xcc * 8 + se) * 16 + cu
    print(f"== {name}: span {e.max():.1f} us, {len(r)} workgroups on {len(np.unique(cuid))} CUs")
    for tg, tn in ((0, "fwd"), (1, "dq"), (2, "dkdv")):
        m = tag == tg
        if not m.any():
            continue
        d = e[m] - s[m]
        order = np.argsort(d)
        ntail = B * H if Np % 128 else 0
        for label, sel in (("tail", order[:ntail]), ("full", order[ntail:])):  # tails are the shortest-lived by far
            if len(sel) == 0:
                continue
            ss, ee, dd, ll = s[m][sel], e[m][sel], d[sel], l[m][sel]
            print(f"   {tn:4s} {label:4s} n={len(sel):5d}  start p0/p50/p100 = {ss.min():6.1f} {np.median(ss):6.1f} {ss.max():6.1f}   "
                  f"life mean/min/max = {dd.mean():6.1f} {dd.min():6.1f} {dd.max():6.1f}   epilogue mean {np.mean(ee - ll):5.1f}   "
                  f"end p50/p100 = {np.median(ee):6.1f} {ee.max():6.1f}")
    edges = np.arange(0, e.max() + 5, 5.0)
    starts = np.histogram(s, edges)[0]
    live = [(np.sum((s <= x) & (e > x))) for x in edges[:-1]]
    print("   t(us)   : " + " ".join(f"{int(x):4d}" for x in edges[:-1]))
    print("   started : " + " ".join(f"{int(x):4d}" for x in starts))
    print("   resident: " + " ".join(f"{int(x):4d}" for x in live))
