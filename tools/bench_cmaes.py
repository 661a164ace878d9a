#!/usr/bin/env python3
"""SURVEY.md 8(d) config 4 on one GPU: CMA-ES N=631 (half-cheetah actor) lambda=4096 mu=2048, and N=151 lambda=4096;
fitness = synthetic quadratic (computed on the host from the fp32 population).  Metric: generations / s."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ddrl_amd

for N, lam, gens, dev in ((151, 4096, 12, 0), (631, 4096, 8, 0), (151, 4096, 12, 1), (631, 4096, 12, 1)):
    es = ddrl_amd.Cmaes(np.full(N, 0.1), np.full(N, 0.3), 3, lam)
    es.set_option("device_rng", dev)
    es.set_option("device_eigen", dev)
    t_s, t_u = [], []
    for g in range(gens):
        t0 = time.perf_counter(); X = es.sample(); t1 = time.perf_counter()
        f = (X * X).sum(1)
        t2 = time.perf_counter(); es.update(f); es.get("sigma"); t3 = time.perf_counter()
        t_s.append(t1 - t0); t_u.append(t3 - t2)
    print(json.dumps({"workload": "config 4: CMA-ES N=%d lambda=%d, %s" % (N, lam, "device Philox stream + rocSOLVER eigen" if dev else "c-cmaes stream (host)"), "sample_ms_median": float(np.median(t_s)) * 1e3,
                      "sample_ms_max(eigen)": float(np.max(t_s)) * 1e3, "update_ms_median": float(np.median(t_u)) * 1e3,
                      "generations_per_s": 1.0 / (np.median(t_s) + np.median(t_u)), "individuals_per_s": lam / (np.median(t_s) + np.median(t_u))}))
    es.close()
