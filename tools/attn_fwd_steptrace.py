"""NOTE: the stamps perturb the kernel (8 s_memtime + scheduling fences per tile: +7..10 % and a different phase behaviour between the
co-resident workgroups) -- differences between stamped builds did NOT carry over to the clean kernel (profiles/r06_attn_fwd_steptrace.txt).
Forward attention, per-tile time line from the -DVBX_ATTN_STEPTRACE library (tools/attn_fwd_steptrace.sh): for every wave and key tile
the eight s_memtime stamps (100 MHz) of attn_fwd_v3_body.inc; prints the mean duration of each segment over all
full-tile waves and the time line of a few workgroups."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, H, Np = 8, 16, int(os.environ.get("NP", 1040))
g = torch.Generator().manual_seed(0)
q = torch.randn(B, H, Np, 64, generator=g); k = torch.randn(B, H, Np, 64, generator=g); v = torch.randn(B, H, Np, 64, generator=g)
q = q / q.norm(dim=-1, keepdim=True) * 8; k = k / k.norm(dim=-1, keepdim=True) * 8
qd, kd, vd = (q * L.lib().vbx_attn_q_prescale(10.0)).half().to(dev), k.half().to(dev), v.half().to(dev)
out16 = torch.empty(B, Np, H * 64, dtype=torch.float16, device=dev)
lse = torch.empty(B, H, Np, device=dev)
st = torch.cuda.current_stream().cuda_stream
NW = 2048
buf = torch.zeros(NW, 4, 10, dtype=torch.int64, device=dev)
lib = ctypes.CDLL(os.environ["VBX_LIB_PATH"])
lib.vbx_debug_attn_steptrace.argtypes = [ctypes.c_void_p]


def timed(n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        L.call("vbx_attn_fwd", qd, kd, vd, None, out16, None, lse, B, H, Np, 10.0, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


timed(3)
us_plain = timed()
assert lib.vbx_debug_attn_steptrace(buf.data_ptr()) == 0
us_traced = timed()
torch.cuda.synchronize()
lib.vbx_debug_attn_steptrace(None)
t = buf.cpu().double()
nt = (Np + 63) // 64
sel = t[(t[:, :, 7] > 0)]  # waves of full-tile roles
life = sel[:, 9] - sel[:, 8]
ghz = life.mean() / (us_traced - 8.0) / 1e3  # rough: the key loop is the launch minus ~8 us of launch floor, prologue and epilogue
print(f"Np = {Np}: {sel.shape[0]} waves x {nt} tiles; launch {us_plain:.1f} us (stamps off) / {us_traced:.1f} us (stamps on); "
      f"key loop {life.mean():.0f} cycles per wave (~{ghz:.2f} GHz)")
names3 = ["loop back (7 -> 0)", "vmcnt wait, barrier, DMA issue", "zero S, K reads arrive", "S chain 1 (4 MFMAs) issued", "K reads + S chain 2", "V reads, max, exp, rescale", "pack + P.V block 1", "V reads + pack + P.V block 2"]
for i, n in enumerate(names3):
    x = sel[:, i] / nt
    print(f"  {n:30s} {x.mean():7.0f} cycles per tile  (p10 {x.quantile(0.1):6.0f}, p90 {x.quantile(0.9):6.0f})  {100 * sel[:, i].sum() / life.sum():5.1f} %")
