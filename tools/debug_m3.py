import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from oracle import oracle as ora
from ddpg_common import make_pair, oracle_weights, sync_oracle_from_gpu, rel_err
pop, agents = make_pair(ora, G=3, hidden_a=(64, 32), hidden_q=(48, 40), momentum=3, hidden_act=3,
                        actor_out_act=0, inverting_grad=False, decay_v=1e-3, seed=2, wscale=0.05)
for it in range(20):
    sync_oracle_from_gpu(pop, agents)
    pop.update(); last = pop.last()
    for g, ag in enumerate(agents):
        loss, used = ag.update()
        qt, da = ag.last()
        e = np.abs(last["dact"][g] - da) / np.abs(da).max()
        print(it, g, "loss %.3e dact max|ref| %.3e relerr max %.2e n>1e-5 %d  qt err %.2e wmax q %.2e a %.2e" % (
            loss, np.abs(da).max(), e.max(), (e > 1e-5).sum(), rel_err(last["q_targets"][g], qt),
            np.abs(ag.qnn.weights()).max(), np.abs(ag.ann.weights()).max()))
