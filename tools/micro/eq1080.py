import sys, numpy as np, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from woft_amd import synth
from woft_amd.config import Config
from woft_amd.flow_provider import RAFTWrapper
H, W = 1080, 1920
sd = synth.make_state_dict(seed=7)
a = synth.make_template(H, W, seq_id=0)
b = synth.make_frame(a, 1)
outs = {}
for corr in ("volume", "otf"):
    c = Config(); c.of_class = RAFTWrapper; c.raft_type = "weighted"; c.class_params = Config()
    c.class_params.small = False; c.class_params.mixed_precision = False
    c.model = sd; c.iters = 12; c.padding_mode = "nopad"; c.precision = "bf16x3"; c.corr = corr
    prov = RAFTWrapper(c)
    fl, wt = prov.compute_flow(a, b, mode="flow", numpy_out=True)
    fl2, wt2 = prov.compute_flow(a, b, mode="flow", numpy_out=True)
    print(corr, "repeatable", np.array_equal(fl, fl2), np.array_equal(wt, wt2))
    outs[corr] = (fl, wt)
    plan = prov.engine.plan(H, W)
    outs[corr + "_corr"] = plan.corr.t.clone().cpu().numpy()
    del prov
    torch.cuda.empty_cache()
d = np.abs(outs["otf"][0] - outs["volume"][0])
print("flow equal", np.array_equal(outs["otf"][0], outs["volume"][0]), "max diff", d.max(), "n diff", int((d > 0).sum()))
print("weights equal", np.array_equal(outs["otf"][1], outs["volume"][1]))
dc = np.abs(outs["otf_corr"] - outs["volume_corr"])
print("final lookup equal", np.array_equal(outs["otf_corr"], outs["volume_corr"]), dc.max(), int((dc > 0).sum()))
