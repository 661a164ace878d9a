"""Dev tool (GPU box): distribution of fp32-vs-fp64 weight errors after N DDPG updates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle as ora
from ddpg_common import make_pair, oracle_weights

def report(tag, pop, agents):
    for net in range(4):
        w = pop.weights(net); wo = oracle_weights(agents, net)
        err = np.abs(w - wo) / np.max(np.abs(wo))
        print("%s net%d max %.2e  p99.9 %.2e  p99 %.2e  med %.2e  frac>1e-5 %.2e" % (
            tag, net, err.max(), np.quantile(err, .999), np.quantile(err, .99), np.median(err), (err > 1e-5).mean()))
    for net in (0, 1):
        m, v, it = pop.adam(net)
        mo = np.stack([a.nets[net].adam()[0] for a in agents])
        print("%s net%d grad(m) relerr %.2e  gmax %.2e" % (tag, net, np.abs(m - mo).max() / np.abs(mo).max(), np.abs(mo).max()))

for sweep, wscale in ((False, 0.01), (True, 0.1)):
    pop, agents = make_pair(ora, G=4, wscale=wscale, sweep=sweep)
    n = 0
    for target in (1, 10, 50):
        while n < target:
            pop.update()
            for a in agents: a.update()
            n += 1
        report("sweep=%d ws=%.2f N=%d" % (sweep, wscale, n), pop, agents)
