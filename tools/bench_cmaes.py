#!/usr/bin/env python3
"""SURVEY.md 8(d) config 4 on one GPU: CMA-ES N=631 (half-cheetah actor) and N=151 (acrobot actor), lambda=4096, mu=2048.
Two modes:
  parity      c-cmaes' own random stream evaluated on the host + the host Householder/QL eigendecomposition (bit-identical
              eigen schedule; what the golden tests run), population copied to the host for the fitness;
  throughput  everything on the device: Philox z, one-sided Jacobi eigendecomposition, the population stays in HBM (torch view),
              synthetic quadratic fitness computed on the device, ddrl_cmaes_update_dev on the device buffer.
Metric: generations / s, sampled individuals / s, and the roofline fractions of the two GEMMs (fp32 MFMA: sampling 2 lambda N^2 flop,
rank-mu mu N^2 FMA; HBM: 4 (2 lambda N + N^2) B)."""
import json
import os
import sys
import time

import numpy as np
import torch  # before the library: torch ships its own HIP runtime and must load it first (tests/conftest.py)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddrl_amd

MFMA_F32_PEAK, HBM_PEAK = 157.3e12, 8.0e12
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for N, lam, mode in ((151, 4096, "parity"), (631, 4096, "parity"), (151, 4096, "throughput"), (631, 4096, "throughput")):
    es = ddrl_amd.Cmaes(np.full(N, 0.1), np.full(N, 0.3), 3, lam)
    dev = mode == "throughput"
    es.set_option("device_rng", int(dev))
    es.set_option("device_eigen", int(dev))
    t_s, t_u, sweeps = [], [], []
    if dev:
        target = torch.linspace(-0.5, 0.5, N, device="cuda", dtype=torch.float64)
    for g in range(gens if dev else max(4, gens // 2)):
        t0 = time.perf_counter()
        if dev:
            es.sample_dev()
            X = es.population_torch()
            f = ((X.double() - target) ** 2).sum(1)
            torch.cuda.synchronize()
        else:
            X = es.sample()
        t1 = time.perf_counter()
        if dev:
            es.update_dev(f.data_ptr())
        else:
            es.update(((X - np.linspace(-0.5, 0.5, N)) ** 2).sum(1))
        es.get("sigma")
        t2 = time.perf_counter()
        t_s.append(t1 - t0); t_u.append(t2 - t1)
        sweeps.append(es.get("eigen_sweeps"))
    per_gen = float(np.median(t_s) + np.median(t_u))
    flops = 2.0 * lam * N * N + 2.0 * (lam // 2) * N * N
    byts = 4.0 * (2 * lam * N + N * N)
    print(json.dumps({"workload": "config 4: CMA-ES N=%d lambda=%d, %s mode" % (N, lam, mode),
                      "sample(+eigen)+fitness_ms_median": float(np.median(t_s)) * 1e3, "update_ms_median": float(np.median(t_u)) * 1e3,
                      "generations_per_s": 1.0 / per_gen, "individuals_per_s": lam / per_gen,
                      "jacobi_sweeps_median": float(np.median(sweeps)) if dev else None,
                      "gemm_flops_per_generation": flops, "mfma_f32_frac_whole_generation": flops / per_gen / MFMA_F32_PEAK,
                      "algorithmic_bytes_per_generation": byts, "hbm_frac_whole_generation": byts / per_gen / HBM_PEAK}))
    es.close()
