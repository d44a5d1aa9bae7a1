"""Debug: model-level dropout vs the oracle with the same masks, attention-only / FeedForward-only / both."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voicebox_pytorch_amd as vbx
from voicebox_pytorch_amd.masks import rng_override
from oracle import restate
import test_model_gpu as T

g = torch.load(os.path.join(ROOT, "tests/golden/small_dropout.pt"), weights_only=False)
cfg = restate.Cfg(**g["cfg"])
draws = dict(x0=g["x0"], times=g["times"], frac_lengths=g["frac"], rand=g["rand"])
B, N = g["x1"].shape[:2]
for pa, pf in ((0.0, 0.0), (0.1, 0.0), (0.0, 0.2), (0.1, 0.2)):
    vb = vbx.VoiceBox(dim=cfg.dim, num_cond_tokens=500, depth=cfg.depth, dim_head=64, heads=cfg.heads, condition_on_text=False,
                      attn_dropout=pa, ff_dropout=pf)
    vb.load_state_dict(g["state"], strict=False)
    vb = vb.to("cuda")
    wrapper = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
    torch.manual_seed(1234)
    with rng_override(**draws):
        loss = wrapper(g["x1"].to("cuda"))
    eng = vb._engines[(B, N, True)]
    attn, ff = T._model_dropout_multipliers(vbx, eng, cfg, B, N, pa, pf) if (pa or pf) else (None, None)
    p = {k: v.double().clone() for k, v in g["state"].items()}
    with restate.dropout_multipliers(attn=attn, ff=ff):
        ref = restate.cfm_loss(p, cfg, g["x1"].double(), g["x0"].double(), g["times"].double(), g["frac"], g["rand"])
    ref0 = restate.cfm_loss(p, cfg, g["x1"].double(), g["x0"].double(), g["times"].double(), g["frac"], g["rand"])
    print(f"pa {pa} pf {pf}: hip {float(loss):.6f} oracle(same masks) {float(ref):.6f} oracle(no dropout) {float(ref0):.6f}  io.dropout {eng.io.dropout} seed {eng.io.drop_seed}")
