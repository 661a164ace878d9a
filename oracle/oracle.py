"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module, and only
as the checker.  The product (ddrl_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORA_MAXL = 8
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build():
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if os.path.exists("/root/reference/agent/cmaes/src/cmaes.c"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class DdpgCfg(C.Structure):
    _fields_ = [
        ("S", C.c_int), ("A", C.c_int), ("B", C.c_int), ("replay_cap", C.c_int),
        ("n_hidden_a", C.c_int), ("hidden_a", C.c_int * ORA_MAXL),
        ("n_hidden_q", C.c_int), ("hidden_q", C.c_int * ORA_MAXL),
        ("hidden_act", C.c_int), ("actor_out_act", C.c_int), ("momentum", C.c_int),
        ("inverting_grad", C.c_int), ("nstep_return", C.c_int), ("force_more_update", C.c_int),
        ("gamma", C.c_double), ("tau", C.c_double), ("alpha_a", C.c_double), ("alpha_v", C.c_double),
        ("decay_v", C.c_double),
        ("ewc", C.c_double), ("ewc_decay", C.c_double), ("ewc_fisher_beta", C.c_double),
        ("ewc_for_critic", C.c_int), ("ewc_best_method", C.c_int),
        ("seed", C.c_uint32), ("td3", C.c_int),
        ("bn_actor", C.c_int), ("bn_critic", C.c_int), ("test_net", C.c_int), ("bn_adapt", C.c_int),
    ]


class NfacCfg(C.Structure):
    _fields_ = [
        ("S", C.c_int), ("A", C.c_int),
        ("n_hidden_a", C.c_int), ("hidden_a", C.c_int * ORA_MAXL),
        ("n_hidden_v", C.c_int), ("hidden_v", C.c_int * ORA_MAXL),
        ("hidden_act", C.c_int), ("actor_out_act", C.c_int), ("momentum", C.c_int),
        ("update_delta_neg", C.c_int), ("update_critic_first", C.c_int), ("gae", C.c_int),
        ("inverting_gradient", C.c_int),
        ("number_fitted_iteration", C.c_int), ("stoch_iter_actor", C.c_int), ("stoch_iter_critic", C.c_int),
        ("gamma", C.c_double), ("lambda_", C.c_double), ("alpha_a", C.c_double), ("alpha_v", C.c_double),
        ("ewc", C.c_double), ("ewc_decay", C.c_double), ("ewc_fisher_beta", C.c_double),
        ("ewc_for_critic", C.c_int), ("ewc_best_method", C.c_int),
        ("vnn_from_scratch", C.c_int), ("corrected_update_ac", C.c_int), ("corrected_update_ac_factor", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ora_mlp_create.restype = vp
    L.ora_mlp_create.argtypes = [C.c_int, _ip, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double]
    L.ora_mlp_create_bn.restype = vp
    L.ora_mlp_create_bn.argtypes = [C.c_int, _ip, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]
    L.ora_mlp_set_phase.argtypes = [vp, C.c_int]
    L.ora_mlp_bn_layout.argtypes = [vp, _ip, _ip, _ip, _ip]
    L.ora_mlp_layer_offsets.argtypes = [vp, _ip, _ip]
    L.ora_ddpg_sync_testing.argtypes = [vp]
    L.ora_mlp_free.argtypes = [vp]
    L.ora_mlp_clone.restype = vp
    L.ora_mlp_clone.argtypes = [vp]
    L.ora_mlp_nparams.argtypes = [vp]
    L.ora_mlp_set_weights.argtypes = [vp, _dp]
    L.ora_mlp_get_weights.argtypes = [vp, _dp]
    L.ora_mlp_get_grads.argtypes = [vp, _dp]
    L.ora_mlp_get_adam.argtypes = [vp, _dp, _dp]
    L.ora_mlp_set_adam.argtypes = [vp, _dp, _dp, C.c_int]
    L.ora_mlp_iter.argtypes = [vp]
    L.ora_mlp_set_iter.argtypes = [vp, C.c_int]
    L.ora_mlp_set_lr.argtypes = [vp, C.c_double]
    L.ora_mlp_set_gates.argtypes = [vp, vp, vp]
    L.ora_mlp_push_row_gates.argtypes = [vp, vp, vp, C.c_int]
    L.ora_mlp_push_row_gates.restype = C.c_int
    L.ora_mlp_row_gates_pending.argtypes = [vp]
    L.ora_mlp_row_gates_pending.restype = C.c_int
    L.ora_mlp_forward.argtypes = [vp, _dp, C.c_int, _dp]
    L.ora_mlp_backward.argtypes = [vp, _dp, C.c_int, vp]
    L.ora_mlp_zero_grad.argtypes = [vp]
    L.ora_mlp_apply_update.argtypes = [vp]
    L.ora_mlp_learn_batch.restype = C.c_double
    L.ora_mlp_learn_batch.argtypes = [vp, _dp, _dp, C.c_int, C.c_int]
    L.ora_mlp_learn_batch_lw.restype = C.c_double
    L.ora_mlp_learn_batch_lw.argtypes = [vp, _dp, _dp, _dp, C.c_int, C.c_int]
    L.ora_mlp_soft_update.argtypes = [vp, vp, C.c_double]
    L.ora_mlp_ewc_config.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_int]
    L.ora_mlp_update_fisher.argtypes = [vp, C.c_double]
    L.ora_mlp_regularize.argtypes = [vp]
    L.ora_mlp_ewc_cost.restype = C.c_double
    L.ora_mlp_ewc_cost.argtypes = [vp]
    L.ora_mlp_ewc_setup.argtypes = [vp]
    L.ora_mlp_ewc_decay_update.argtypes = [vp]
    L.ora_mlp_update_best_param.argtypes = [vp, C.c_double]
    L.ora_mlp_get_ewc.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ora_mlp_ewc_active.argtypes = [vp]
    L.ora_mlp_set_ewc.argtypes = [vp, vp, vp]
    L.ora_mt_seed.argtypes = [vp, C.c_uint32]
    L.ora_mt_next.restype = C.c_uint32
    L.ora_mt_next.argtypes = [vp]
    L.ora_uniform_index.argtypes = [vp, C.c_uint32]
    L.ora_ddpg_create.restype = vp
    L.ora_ddpg_create.argtypes = [C.POINTER(DdpgCfg)]
    L.ora_ddpg_free.argtypes = [vp]
    L.ora_ddpg_net.restype = vp
    L.ora_ddpg_net.argtypes = [vp, C.c_int]
    L.ora_ddpg_insert.argtypes = [vp, _dp, _dp, C.c_double, _dp, C.c_int]
    L.ora_ddpg_replay_len.argtypes = [vp]
    L.ora_ddpg_update.restype = C.c_double
    L.ora_ddpg_update.argtypes = [vp, vp, vp]
    L.ora_ddpg_ewc_setup.argtypes = [vp]
    L.ora_ddpg_last.argtypes = [vp, vp, vp]
    L.ora_td3_update.restype = C.c_double
    L.ora_td3_update.argtypes = [vp, vp, vp, _dp, C.c_int, vp]
    L.ora_nfac_create.restype = vp
    L.ora_nfac_create.argtypes = [C.POINTER(NfacCfg)]
    L.ora_nfac_free.argtypes = [vp]
    L.ora_nfac_net.restype = vp
    L.ora_nfac_net.argtypes = [vp, C.c_int]
    L.ora_nfac_update.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int, vp]
    L.ora_nfac_set_fresh.argtypes = [vp, _dp, C.c_int]
    L.ora_cmaes_create.restype = vp
    L.ora_cmaes_create.argtypes = [C.c_int, _dp, _dp, C.c_long, C.c_int]
    L.ora_cmaes_free.argtypes = [vp]
    L.ora_cmaes_sample.restype = C.POINTER(C.c_double)
    L.ora_cmaes_sample.argtypes = [vp]
    L.ora_cmaes_resample.argtypes = [vp, C.c_int]
    L.ora_cmaes_update.argtypes = [vp, _dp]
    L.ora_sorted_index.argtypes = [_dp, _ip, C.c_int]
    L.ora_cmaes_eigen.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
    L.ora_cmaes_gauss.restype = C.c_double
    L.ora_cmaes_gauss.argtypes = [vp]
    L.ora_cmaes_get_int.argtypes = [vp, C.c_char_p]
    L.ora_cmaes_get.restype = C.c_double
    L.ora_cmaes_get.argtypes = [vp, C.c_char_p]
    L.ora_cmaes_ptr.restype = C.POINTER(C.c_double)
    L.ora_cmaes_ptr.argtypes = [vp, C.c_char_p]
    L.ora_cmaes_index.restype = C.POINTER(C.c_int)
    L.ora_cmaes_index.argtypes = [vp]
    L.ora_cmaes_set_sigma.argtypes = [vp, C.c_double]
    L.ora_cmaes_test_for_termination.argtypes = [vp, C.c_char_p]
    _lib = L
    return L


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Mt19937:
    """std::mt19937 + std::uniform_int_distribution<int>(0,n-1) (libstdc++ >= 11)."""

    def __init__(self, seed):
        self._buf = (C.c_uint32 * 625)()
        lib().ora_mt_seed(C.addressof(self._buf), seed & 0xFFFFFFFF)

    def next(self):
        return lib().ora_mt_next(C.addressof(self._buf))

    def index(self, n):
        return lib().ora_uniform_index(C.addressof(self._buf), n)


class Mlp:
    def __init__(self, dims, hidden_act=1, out_act=0, lr=1e-3, momentum=0, decay=-1.0, _h=None, _own=True,
                 batch_norm=0, critic_sensors=0):
        self.dims = list(dims)
        self._own = _own
        if _h is None:
            d = np.asarray(dims, dtype=np.int32)
            _h = lib().ora_mlp_create_bn(len(dims) - 1, d, hidden_act, out_act, lr, momentum, decay, batch_norm, critic_sensors)
        self.h = _h
        self.np = lib().ora_mlp_nparams(self.h)

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().ora_mlp_free(self.h)
            self.h = None

    def clone(self, test=None):
        """MLP(const MLP&, copy_solver, phase) (MLP.hpp:284-349); test = True: ::caffe::Phase::TEST"""
        c = Mlp(self.dims, _h=lib().ora_mlp_clone(self.h))
        if test is not None:
            c.set_phase(test)
        return c

    def set_phase(self, test):
        lib().ora_mlp_set_phase(self.h, int(bool(test)))

    def bn_layout(self):
        """-> dict(rank_on, offbn, offsc, chan, scale, bias, offw, offb): where the towers sit in the flat parameter vector"""
        n = len(self.dims)
        a = [np.zeros(n, np.int32) for _ in range(4)]
        flags = lib().ora_mlp_bn_layout(self.h, *a)
        ow, ob = np.zeros(n - 1, np.int32), np.zeros(n - 1, np.int32)
        lib().ora_mlp_layer_offsets(self.h, ow, ob)
        return dict(rank_on=a[0], offbn=a[1], offsc=a[2], chan=a[3], scale=bool(flags & 1), bias=bool(flags & 2), offw=ow, offb=ob)

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.size == self.np
        lib().ora_mlp_set_weights(self.h, w)

    def weights(self):
        w = np.empty(self.np)
        lib().ora_mlp_get_weights(self.h, w)
        return w

    def grads(self):
        g = np.empty(self.np)
        lib().ora_mlp_get_grads(self.h, g)
        return g

    def adam(self):
        m = np.empty(self.np)
        v = np.empty(self.np)
        lib().ora_mlp_get_adam(self.h, m, v)
        return m, v, lib().ora_mlp_iter(self.h)

    def set_adam(self, m, v, it):
        lib().ora_mlp_set_adam(self.h, np.ascontiguousarray(m, np.float64), np.ascontiguousarray(v, np.float64), it)

    def push_row_gates(self, in_rows, out_rows, B):
        ig = None if in_rows is None else np.ascontiguousarray(in_rows, np.float64).reshape(B, -1)
        og = None if out_rows is None else np.ascontiguousarray(out_rows, np.float64).reshape(B, -1)
        if lib().ora_mlp_push_row_gates(self.h, _vp(ig), _vp(og), B):
            raise RuntimeError("row-gate queue full")

    def row_gates_pending(self):
        return lib().ora_mlp_row_gates_pending(self.h)

    def set_gates(self, in_gate=None, out_gate=None):
        ig = None if in_gate is None else np.ascontiguousarray(in_gate, np.float64)
        og = None if out_gate is None else np.ascontiguousarray(out_gate, np.float64)
        lib().ora_mlp_set_gates(self.h, _vp(ig), _vp(og))

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.dims[0])
        out = np.empty((x.shape[0], self.dims[-1]))
        lib().ora_mlp_forward(self.h, x, x.shape[0], out)
        return out

    def backward(self, dout, want_din=True):
        dout = np.ascontiguousarray(dout, dtype=np.float64).reshape(-1, self.dims[-1])
        din = np.empty((dout.shape[0], self.dims[0])) if want_din else None
        lib().ora_mlp_backward(self.h, dout, dout.shape[0], _vp(din))
        return din

    def zero_grad(self):
        lib().ora_mlp_zero_grad(self.h)

    def apply_update(self):
        lib().ora_mlp_apply_update(self.h)

    def set_iter(self, it):
        lib().ora_mlp_set_iter(self.h, it)

    def iter(self):
        return lib().ora_mlp_iter(self.h)

    def learn_batch_lw(self, x, y, lw, iters=1):
        """MLP::learn_batch_lw (MLP.hpp:487-506)"""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.dims[0])
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(x.shape[0], self.dims[-1])
        lw = np.ascontiguousarray(lw, dtype=np.float64).reshape(x.shape[0])
        return lib().ora_mlp_learn_batch_lw(self.h, x, y, lw, x.shape[0], iters)

    def learn_batch(self, x, y, iters=1):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.dims[0])
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(x.shape[0], self.dims[-1])
        return lib().ora_mlp_learn_batch(self.h, x, y, x.shape[0], iters)

    def soft_update(self, src, tau):
        lib().ora_mlp_soft_update(self.h, src.h, tau)

    def ewc_config(self, ewc, ewc_decay, fisher_beta, best_method=0):
        lib().ora_mlp_ewc_config(self.h, ewc, ewc_decay, fisher_beta, best_method)

    def ewc_state(self):
        f = np.empty(self.np); fh = np.empty(self.np); ws = np.empty(self.np)
        nbr = C.c_double(); dm = C.c_double()
        lib().ora_mlp_get_ewc(self.h, _vp(f), _vp(fh), _vp(ws), C.addressof(nbr), C.addressof(dm))
        return dict(fisher=f, fisher_hat=fh, wstar=ws, fisher_nbr=nbr.value, decay_mult=dm.value,
                    active=lib().ora_mlp_ewc_active(self.h))

    def ewc_cost(self):
        return lib().ora_mlp_ewc_cost(self.h)

    def ewc_setup(self):
        lib().ora_mlp_ewc_setup(self.h)

    def update_best_param(self, score):
        lib().ora_mlp_update_best_param(self.h, score)


def _fill_hidden(arr, lst):
    for i, v in enumerate(lst):
        arr[i] = v


class Ddpg:
    def __init__(self, S, A, B, hidden_a, hidden_q, replay_cap=10000, hidden_act=1, actor_out_act=2,
                 momentum=0, inverting_grad=True, nstep_return=0, gamma=0.99, tau=0.001, alpha_a=1e-4,
                 alpha_v=1e-3, decay_v=-1.0, ewc=-1.0, ewc_decay=-1.0, ewc_fisher_beta=0.999,
                 ewc_for_critic=0, ewc_best_method=0, seed=0, td3=False, batch_norm_actor=0, batch_norm_critic=0,
                 test_net=False, bn_adapt=False):
        c = DdpgCfg()
        c.bn_actor, c.bn_critic, c.test_net, c.bn_adapt = batch_norm_actor, batch_norm_critic, int(test_net), int(bn_adapt)
        c.td3 = int(td3)
        c.S, c.A, c.B, c.replay_cap = S, A, B, replay_cap
        c.n_hidden_a = len(hidden_a); _fill_hidden(c.hidden_a, hidden_a)
        c.n_hidden_q = len(hidden_q); _fill_hidden(c.hidden_q, hidden_q)
        c.hidden_act, c.actor_out_act, c.momentum = hidden_act, actor_out_act, momentum
        c.inverting_grad, c.nstep_return, c.force_more_update = int(inverting_grad), nstep_return, 0
        c.gamma, c.tau, c.alpha_a, c.alpha_v, c.decay_v = gamma, tau, alpha_a, alpha_v, decay_v
        c.ewc, c.ewc_decay, c.ewc_fisher_beta = ewc, ewc_decay, ewc_fisher_beta
        c.ewc_for_critic, c.ewc_best_method, c.seed = ewc_for_critic, ewc_best_method, seed
        self.cfg = c
        self.S, self.A, self.B = S, A, B
        self.h = lib().ora_ddpg_create(C.byref(c))
        da = [S] + list(hidden_a) + [A]
        dq = [S + A] + list(hidden_q) + [1]
        self.nets = [Mlp(d, _h=lib().ora_ddpg_net(self.h, i), _own=False)
                     for i, d in enumerate([da, dq, da, dq] + ([dq, dq] if td3 else []))]
        self.ann_testing = Mlp(da, _h=lib().ora_ddpg_net(self.h, 6), _own=False)

    def sync_testing(self):
        lib().ora_ddpg_sync_testing(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_ddpg_free(self.h)
            self.h = None

    @property
    def ann(self): return self.nets[0]
    @property
    def qnn(self): return self.nets[1]
    @property
    def ann_target(self): return self.nets[2]
    @property
    def qnn_target(self): return self.nets[3]
    @property
    def qnn2(self): return self.nets[4]
    @property
    def qnn_target2(self): return self.nets[5]

    def update_td3(self, noise, actor_step=True, idx=None):
        """TD3Ag::end_episode; noise [B x A] = the target-policy smoothing draws.  -> (loss1, loss2, idx)"""
        out = np.empty(self.B, dtype=np.int32)
        i = None if idx is None else np.ascontiguousarray(idx, dtype=np.int32)
        nz = np.ascontiguousarray(noise, np.float64).reshape(self.B * self.A)
        l2 = C.c_double()
        loss = lib().ora_td3_update(self.h, _vp(i), _vp(out), nz, int(actor_step), C.cast(C.byref(l2), C.c_void_p))
        return loss, l2.value, out

    def insert(self, s, a, r, s2, goal):
        lib().ora_ddpg_insert(self.h, np.ascontiguousarray(s, np.float64), np.ascontiguousarray(a, np.float64),
                              float(r), np.ascontiguousarray(s2, np.float64), int(goal))

    def replay_len(self):
        return lib().ora_ddpg_replay_len(self.h)

    def update(self, idx=None, sync=True):
        """one pass of the fupd loop; sync: the ann_testing copy that follows the loop (DeepQNAg.hpp:469-472)"""
        out = np.empty(self.B, dtype=np.int32)
        i = None if idx is None else np.ascontiguousarray(idx, dtype=np.int32)
        loss = lib().ora_ddpg_update(self.h, _vp(i), _vp(out))
        if sync:
            self.sync_testing()
        return loss, out

    def ewc_setup(self):
        lib().ora_ddpg_ewc_setup(self.h)

    def last(self):
        qt = np.empty(self.B); da = np.empty((self.B, self.A))
        lib().ora_ddpg_last(self.h, _vp(qt), _vp(da))
        return qt, da


class Nfac:
    def __init__(self, S, A, hidden_a, hidden_v, hidden_act=1, actor_out_act=2, momentum=0,
                 update_delta_neg=False, update_critic_first=True, gae=True, inverting_gradient=False,
                 number_fitted_iteration=10, stoch_iter_actor=10, stoch_iter_critic=10,
                 gamma=0.99, lambda_=0.9, alpha_a=1e-4, alpha_v=1e-3,
                 ewc=-1.0, ewc_decay=-1.0, ewc_fisher_beta=0.999, ewc_for_critic=0, ewc_best_method=0,
                 vnn_from_scratch=False, corrected_update_ac=False, corrected_update_ac_factor=1.0):
        c = NfacCfg()
        c.S, c.A = S, A
        c.n_hidden_a = len(hidden_a); _fill_hidden(c.hidden_a, hidden_a)
        c.n_hidden_v = len(hidden_v); _fill_hidden(c.hidden_v, hidden_v)
        c.hidden_act, c.actor_out_act, c.momentum = hidden_act, actor_out_act, momentum
        c.update_delta_neg, c.update_critic_first, c.gae = int(update_delta_neg), int(update_critic_first), int(gae)
        c.inverting_gradient = int(inverting_gradient)
        c.number_fitted_iteration, c.stoch_iter_actor, c.stoch_iter_critic = number_fitted_iteration, stoch_iter_actor, stoch_iter_critic
        c.gamma, c.lambda_, c.alpha_a, c.alpha_v = gamma, lambda_, alpha_a, alpha_v
        c.ewc, c.ewc_decay, c.ewc_fisher_beta = ewc, ewc_decay, ewc_fisher_beta
        c.ewc_for_critic, c.ewc_best_method = ewc_for_critic, ewc_best_method
        c.vnn_from_scratch, c.corrected_update_ac = int(vnn_from_scratch), int(corrected_update_ac)
        c.corrected_update_ac_factor = corrected_update_ac_factor
        self.S, self.A = S, A
        self.h = lib().ora_nfac_create(C.byref(c))
        da = [S] + list(hidden_a) + [A]
        dv = [S] + list(hidden_v) + [1]
        self.nets = [Mlp(d, _h=lib().ora_nfac_net(self.h, i), _own=False) for i, d in enumerate([da, dv])]

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_nfac_free(self.h)
            self.h = None

    @property
    def ann(self): return self.nets[0]
    @property
    def vnn(self): return self.nets[1]

    def set_fresh(self, w):
        w = np.ascontiguousarray(w, np.float64).reshape(-1, self.vnn.np)
        lib().ora_nfac_set_fresh(self.h, w, w.shape[0])

    def update(self, s, pure_a, a, s2, r, goal, ends):
        T = len(r)
        f = lambda x: np.ascontiguousarray(x, np.float64)
        deltas = np.empty(T)
        ends = np.ascontiguousarray(ends, np.int32)
        n = lib().ora_nfac_update(self.h, T, f(s), f(pure_a), f(a), f(s2), f(r),
                                  np.ascontiguousarray(goal, np.int32), ends, len(ends), _vp(deltas))
        return n, deltas


class Cmaes:
    def __init__(self, x0, stds, seed, lam):
        x0 = np.ascontiguousarray(x0, np.float64); stds = np.ascontiguousarray(stds, np.float64)
        self.N = x0.size
        self.h = lib().ora_cmaes_create(self.N, x0, stds, seed, lam)
        self.lam = lib().ora_cmaes_get_int(self.h, b"lambda")
        self.mu = lib().ora_cmaes_get_int(self.h, b"mu")

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_cmaes_free(self.h)
            self.h = None

    def sample(self):
        p = lib().ora_cmaes_sample(self.h)
        return np.ctypeslib.as_array(p, shape=(self.lam, self.N)).copy()

    def resample(self, k):
        lib().ora_cmaes_resample(self.h, k)
        return self.vec("X", self.lam * self.N).reshape(self.lam, self.N)[k]

    def update(self, fit):
        lib().ora_cmaes_update(self.h, np.ascontiguousarray(fit, np.float64))

    def get(self, key):
        return lib().ora_cmaes_get(self.h, key.encode())

    def get_int(self, key):
        return lib().ora_cmaes_get_int(self.h, key.encode())

    def vec(self, key, n=None):
        n = self.N if n is None else n
        p = lib().ora_cmaes_ptr(self.h, key.encode())
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def view(self, key, n=None):
        """writable view of the oracle's own array (teacher forcing in the GPU parity tests)"""
        n = self.N if n is None else n
        return np.ctypeslib.as_array(lib().ora_cmaes_ptr(self.h, key.encode()), shape=(n,))

    def test_for_termination(self):
        buf = C.create_string_buffer(256)
        lib().ora_cmaes_test_for_termination(self.h, buf)
        return buf.value.decode()

    def set_sigma(self, sigma):
        lib().ora_cmaes_set_sigma(self.h, float(sigma))

    def index(self):
        return np.ctypeslib.as_array(lib().ora_cmaes_index(self.h), shape=(self.lam,)).copy()

    def gauss(self):
        return lib().ora_cmaes_gauss(self.h)


def sorted_index(f):
    f = np.ascontiguousarray(f, np.float64)
    idx = np.empty(f.size, dtype=np.int32)
    lib().ora_sorted_index(f, idx, f.size)
    return idx
