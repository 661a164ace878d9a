import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle.oracle as ora
from test_td3_gpu import make_td3, sync
from ddpg_common import rel_err
kw = dict(ewc=100.0, ewc_decay=-1.0, ewc_fisher_beta=0.999, ewc_for_critic=1, ewc_best_method=0)
pop, agents = make_td3(ora, G=3, hidden_a=(100, 100), hidden_q=(100, 100), seed=5, **kw)
sync(pop, agents)
B = 64
x = None
pop.update()
last = pop.last()
for g, ag in enumerate(agents):
    l1, l2, used = ag.update_td3(last["noise"][g], True, last["idx"][g])
    print(g, "loss", last["loss"][g], l1, "loss2", last["loss2"][g], l2, "qt", rel_err(last["q_targets"][g], ag.last()[0]))
    for net in (0, 1, 4):
        m, v, it = pop.adam(net)
        print("   net", net, "m", rel_err(m[g], ag.nets[net].adam()[0]), "w", rel_err(pop.weights(net)[g], ag.nets[net].weights()))
