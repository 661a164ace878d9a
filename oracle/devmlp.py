"""oracle/devmlp.py -- CPU restatement of DODevMLP's Developmental layers around an oracle Mlp.  TEST INFRASTRUCTURE ONLY
(tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the product never imports it.

What is restated, with the reference lines it follows (/root/reference/common/include/nn/DODevMLP.hpp):
  * the two gate blobs and where they sit among learnable_params(): actor [devnn_states][InnerProducts][devnn_actions], critic
    [devnn_actions][devnn_states][InnerProducts] (:246-316: the action layer goes right behind the MemoryData layer when actions are an
    input, behind the last layer otherwise; the state layer in front of the first consumer of the states)
  * lr_mult 0 on the blobs unless heuristic = 0 (:378-401), blobs zeroed unless heuristic = 0 (:358-374), the blobs of a critic are
    the actor's (:340-356)
  * number_of_parameters(ignore_null_lr) / copyWeightsTo / copyWeightsFrom over that order (MLP.hpp's loops over params_lr())
  * develop() for heuristic 1 (:420-443) and 2 (:444-467), inform() (:566-574)
  * the layer's forward (the fork's DevelopmentalLayer, absent from the tree; SURVEY.md 8h from CaffeUtest.cpp:619-1341):
    probabilist = 1: a controlled dimension whose weight is below 0.5 is zeroed, the others pass; probabilist = 0: per sample the
    dimension passes with probability w (divided by w if `scale`).  The generator is numpy's here: rates only.
The fork's backward for the gate weights (heuristic 0) is not in the tree and is not restated: parity unpinned."""
import numpy as np

from oracle.oracle import Mlp


class GateBlob:
    """one Developmental layer's blob: a weight per controlled dimension (shared by reference between actor and critic)"""

    def __init__(self, control, heuristic):
        self.control = list(control)
        self.w = np.full(len(self.control), 1.0 if heuristic == 0 else 0.0)       # :358-374 (heuristic 0: the tests set them before use)


class DevMlp:
    def __init__(self, dims, hidden_act=1, out_act=0, st_control=(), ac_control=(), heuristic=0, st_probabilistic=0, ac_probabilistic=0,
                 st_scale=False, ac_scale=False, devpoints=(), linearcoef=(), actor=None, nb_sensors=None, seed=0):
        """actor given: this net is the critic of that actor (input = [states, actions]) and shares its blobs.
        st_control None: no state layer ("None", :132-136); an empty list: every dimension (CaffeUtest.cpp:629-636, 702-708)."""
        self.net = Mlp(dims, hidden_act=hidden_act, out_act=out_act)
        self.critic = actor is not None
        self.S = nb_sensors if nb_sensors is not None else dims[0]
        self.A = dims[0] - self.S if self.critic else dims[-1]
        self.heuristic = heuristic
        self.prob = (st_probabilistic, ac_probabilistic)
        self.scale = (st_scale, ac_scale)
        self.devpoints, self.linearcoef = list(devpoints), list(linearcoef)
        if actor is not None:
            self.st, self.ac = actor.st, actor.ac
        else:
            self.st = None if st_control is None else GateBlob(st_control if len(st_control) else range(self.S), heuristic)
            self.ac = GateBlob(ac_control if len(ac_control) else range(self.A), heuristic)
        self.rng = np.random.RandomState(seed)

    # learnable_params() order
    def _blobs(self):
        st = [] if self.st is None else [self.st]
        return ([self.ac] + st, []) if self.critic else (st, [self.ac])

    def number_of_parameters(self, ignore_null_lr):
        counted = self.heuristic == 0 or not ignore_null_lr
        before, after = self._blobs()
        return self.net.np + (sum(len(b.w) for b in before + after) if counted else 0)

    def copy_weights_to(self, ignore_null_lr):
        counted = self.heuristic == 0 or not ignore_null_lr
        before, after = self._blobs()
        parts = [b.w for b in before] * counted + [self.net.weights()] + [b.w for b in after] * counted
        return np.concatenate(parts)

    def copy_weights_from(self, flat, ignore_null_lr):
        flat = np.asarray(flat, np.float64)
        assert flat.size == self.number_of_parameters(ignore_null_lr)
        counted = self.heuristic == 0 or not ignore_null_lr
        before, after = self._blobs()
        k = 0
        for b in before * counted:
            b.w[:] = flat[k:k + len(b.w)]; k += len(b.w)
        self.net.set_weights(flat[k:k + self.net.np]); k += self.net.np
        for b in after * counted:
            b.w[:] = flat[k:k + len(b.w)]; k += len(b.w)

    # develop (:420-470) behind inform (:566-574) without intrinsic motivation -> changed
    def inform(self, episode):
        changed = False
        nst = 0 if self.st is None else len(self.st.w)
        if self.heuristic == 1:
            for i, p in enumerate(self.devpoints):
                if episode == p:
                    if i < nst:
                        self.st.w[i] = 1.0; changed = True
                    elif i < nst + len(self.ac.w):
                        self.ac.w[i - nst] = 1.0; changed = True
        elif self.heuristic == 2:
            changed = True
            for i in range(nst):
                self.st.w[i] = 1.0 if self.linearcoef[i] >= 1.0 else episode * self.linearcoef[i]
            for i in range(len(self.ac.w)):
                c = self.linearcoef[i + nst]
                self.ac.w[i] = 1.0 if c >= 1.0 else episode * c
        return changed

    def _factors(self, blob, width, rows, kind):
        f = np.ones((rows, width))
        if blob is None:
            return f
        for w, c in zip(blob.w, blob.control):
            if self.prob[kind] == 1:
                f[:, c] = 0.0 if w < 0.5 else 1.0
            else:
                passed = self.rng.uniform(0, 1, rows) < w
                f[:, c] = np.where(passed, 1.0 / w if self.scale[kind] else 1.0, 0.0)
        return f

    # computeOutBatch (actor: states -> actions) / computeOutVFBatch (critic: [states, actions] -> Q) through the layers
    def forward(self, x):
        x = np.asarray(x, np.float64).reshape(-1, self.net.dims[0]).copy()
        x[:, :self.S] *= self._factors(self.st, self.S, x.shape[0], 0)
        if self.critic:
            x[:, self.S:] *= self._factors(self.ac, self.A, x.shape[0], 1)
            return self.net.forward(x)
        return self.net.forward(x) * self._factors(self.ac, self.A, x.shape[0], 1)
