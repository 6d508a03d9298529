import os, sys, json, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
import fuzz_hip_vs_oracle as F
from oracle.backend import OracleBackend
for n in (2057, 2052, 2049, 1029, 515):
  for train in (["weights"], None, ["li_op/tau", "li_op/eta"]):
    args = (123, "tanh", n, 0, 0, train, "li_op/v", {})
    cpu = F.build(dict(device="cpu", backend=OracleBackend()), *args)
    gpu = F.build(dict(device="cuda:0"), *args)
    rng = np.random.default_rng(5)
    for grad, samp, cut, trunc, rec in itertools.product((True, False), (1, 3), (0, 2), (None, 4), (False, True)):
        T = 17
        x = (rng.standard_normal((T, n)) * 0.5).astype(np.float32)
        kw = dict(sampling_steps=samp, cutoff=cut, verbose=False, enable_grad=grad)
        if trunc: kw["truncate_steps"] = trunc
        if rec: kw["record_vars"] = [("rnn", "li_op/v", False)]
        a, b = cpu.run(x, **kw), gpu.run(x, **kw)
        oa, ob = a.to_numpy("out"), b.to_numpy("out")
        ok_out = F.bits_equal(oa, ob)
        sa, sb = cpu.state["rnn"].detach().numpy(), gpu.state["rnn"].detach().cpu().numpy()
        ok_state = F.bits_equal(sa, sb)
        if not (ok_out and ok_state):
            bad = np.argwhere(oa != ob) if oa.shape == ob.shape else None
            print(json.dumps(dict(n=n, train=train, grad=grad, samp=samp, cut=cut, trunc=trunc, rec=rec, ok_out=ok_out, ok_state=ok_state,
                                  out_shape=list(oa.shape), n_bad=None if bad is None else int(len(bad)), first_bad=None if bad is None or not len(bad) else bad[0].tolist(),
                                  state_bad=np.argwhere(sa != sb)[:5].ravel().tolist())), flush=True)
            # resync states so that later configs start equal
            gpu.reset({"rnn": sa}); cpu.reset({"rnn": sa})
print("done")
