"""Developer diagnostic: per-segment deviation of the HIP KL update from the torch oracle over a few updates."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import kl_torch as K
from rlcontrol_amd.hip_kl import KLPopulation

dims, B = (3, 1, 200, 200, 200, 200), 32
d = K.KlDims(*dims)
th = K.init_params(d, 4)
pop = KLPopulation("reverse", 1, *dims, B, 2048, 0.01, 2.0, 1e-3, 1e-3, 0.1, seeds=[5], n_param=64)
pop.enable_grad_taps(True)
pop.set_params(0, th)
o = K.KLOracle("reverse", d, th, 1e-3, 1e-3, 0.1, 0.01, 2.0, 64)
rng = np.random.RandomState(9)
lay, _ = d.layout()
for it in range(10):
    s, a, s2, r, g, eps = (rng.uniform(-2, 2, (B, 3)), rng.uniform(-2, 2, (B, 1)), rng.uniform(-2, 2, (B, 3)),
                           rng.uniform(-16, 0, B), np.where(rng.rand(B) < 0.2, 0.0, 0.99), rng.randn(B, 1))
    import torch
    pv = K._views(o.theta, o.lay)
    xin = torch.tensor(np.concatenate([s, a], 1).astype(np.float32))
    z1 = xin @ pv["qW1"] + pv["qb1"]
    z2 = torch.relu(z1) @ pv["qW2"] + pv["qb2"]
    print("update %d: min|z1| %.3e min|z2| %.3e" % (it, z1.abs().min().item(), z2.abs().min().item()), end=" ")
    pop.update_batch(0, s, a, s2, r, g, eps=eps)
    t = o.update(s, a, s2, r, g, eps, taps=True)
    got_g = pop.last_tap(0, "grads")
    got = pop.get_blob(0, "theta")
    qo, qe = lay["qW1"][0], lay["vW1"][0]
    print("| q tap rel %.2e | dtheta(q net) max %.3e | dgrad(q net) max %.3e" % (
        np.abs(pop.last_tap(0, "q") - t["q"]).max() / np.abs(t["q"]).max(),
        np.abs(got[qo:qe] - o.theta.numpy()[qo:qe]).max(), np.abs(got_g[qo:qe] - t["grads"][qo:qe]).max()))
    if it in (9,):
        print("update", it)
        for n, (off, shp) in lay.items():
            k = int(np.prod(shp))
            dg = np.abs(got_g[off:off + k] - t["grads"][off:off + k])
            dt = np.abs(got[off:off + k] - o.theta.numpy()[off:off + k])
            i = int(np.argmax(dt))
            print("  %-4s |g|max %.3e dgmax %.3e | dtheta max %.3e q999 %.3e | at worst: g_hip %.4e g_ref %.4e" %
                  (n, np.abs(t["grads"][off:off + k]).max(), dg.max(), dt.max(), np.quantile(dt, 0.999),
                   got_g[off + i], t["grads"][off + i]))
