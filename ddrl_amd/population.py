"""Python mirror of the population C ABI (include/ddrl_pop.h) -- plumbing only.

Names follow the reference's agent/MLP interface (DeepQNAg.hpp, MLP.hpp): nets ann / qnn /
ann_target / qnn_target, copyWeightsTo/From, end_episode's update.  Arrays are numpy; the first
axis is always the agent.
"""
import ctypes as C

import numpy as np

from . import lib as _l

NET_ACTOR, NET_CRITIC, NET_ACTOR_TARGET, NET_CRITIC_TARGET = 0, 1, 2, 3
NET_CRITIC2, NET_CRITIC2_TARGET = 4, 5      # TD3 populations only


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DdpgConfig:
    """The [agent]/[devnn] keys of DeepQNAg::_unique_invoke (DeepQNAg.hpp:167-187)."""

    def __init__(self, **kw):
        self.c = _l.DdpgConfigC()
        _l.check(_l.load().ddrl_ddpg_config_default(C.byref(self.c)))
        for k, v in kw.items():
            self.set(k, v)

    @classmethod
    def from_ini(cls, path, n_agents=1, nb_sensors=0, nb_motors=0, td3=False):
        """td3: read TD3Ag's keys (TD3Ag.hpp:165-186) instead of DeepQNAg's"""
        o = cls()
        o.c.n_agents, o.c.nb_sensors, o.c.nb_motors = n_agents, nb_sensors, nb_motors
        f = _l.load().ddrl_td3_config_from_ini if td3 else _l.load().ddrl_ddpg_config_from_ini
        _l.check(f(path.encode(), C.byref(o.c)))
        return o

    def set(self, k, v):
        if k in ("hidden_unit_a", "hidden_unit_q"):
            v = list(v)
            if len(v) > _l.DDRL_MAX_HIDDEN:
                raise ValueError("at most %d hidden layers" % _l.DDRL_MAX_HIDDEN)
            arr = getattr(self.c, k)
            for i, x in enumerate(v):
                arr[i] = int(x)
            setattr(self.c, "n_hidden_" + k[-1], len(v))
        elif hasattr(self.c, k):
            setattr(self.c, k, v)
        else:
            raise KeyError(k)

    def get(self, k):
        if k in ("hidden_unit_a", "hidden_unit_q"):
            return list(getattr(self.c, k))[: getattr(self.c, "n_hidden_" + k[-1])]
        return getattr(self.c, k)


class Population:
    """What DDPG and NFAC populations share: weights, Adam state, inference, the EWC life-cycle."""

    def close(self):
        if getattr(self, "h", None):
            self.L.ddrl_pop_destroy(self.h)
            self.h = None

    __del__ = close

    def sync(self):
        _l.check(self.L.ddrl_pop_sync(self.h))

    def set_hyper(self, key, values):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values, np.float64), (self.G,)))
        _l.check(self.L.ddrl_pop_set_hyper(self.h, key.encode(), _p(v)))

    def num_params(self, net):
        n = self.L.ddrl_pop_num_params(self.h, net)
        if n < 0:
            _l.check(n)
        return n

    # MLP::copyWeightsFrom / copyWeightsTo
    def set_weights(self, net, flat):
        flat = np.ascontiguousarray(flat, np.float64).reshape(self.G, self.num_params(net))
        _l.check(self.L.ddrl_pop_set_weights(self.h, net, _p(flat)))

    def weights(self, net):
        out = np.empty((self.G, self.num_params(net)))
        _l.check(self.L.ddrl_pop_get_weights(self.h, net, _p(out)))
        return out

    def adam(self, net):
        n = self.num_params(net)
        m, v, it = np.empty((self.G, n)), np.empty((self.G, n)), np.empty(self.G, np.int32)
        _l.check(self.L.ddrl_pop_get_adam(self.h, net, _p(m), _p(v), _p(it)))
        return m, v, it

    def set_adam(self, net, m, v, it):
        m = np.ascontiguousarray(m, np.float64); v = np.ascontiguousarray(v, np.float64)
        it = np.ascontiguousarray(it, np.int32)
        _l.check(self.L.ddrl_pop_set_adam(self.h, net, _p(m), _p(v), _p(it)))

    # ann_testing->computeOut(sensors) for every agent
    def infer(self, states):
        s = np.ascontiguousarray(states, _l.REAL).reshape(self.G, self.S)
        out = np.empty((self.G, self.A), _l.REAL)
        _l.check(self.L.ddrl_pop_infer(self.h, _p(s), _p(out)))
        return out

    # MLP::computeOutBatch / computeOutVFBatch: x [G x n x in_dim] -> [G x n x out_dim]
    def forward_batch(self, net, x):
        x = np.ascontiguousarray(x, _l.REAL)
        x = x.reshape(self.G, -1, x.shape[-1])
        n = x.shape[1]
        dl = {0: self.A, 2: self.A}.get(net, 1)
        out = np.empty((self.G, n, dl), _l.REAL)
        _l.check(self.L.ddrl_pop_forward_batch(self.h, net, n, _p(x), _p(out)))
        return out

    # developmental gates of DODevMLP (state gate [G x S], action gate [G x A]); None, None = off
    def set_gates(self, state_gate=None, action_gate=None):
        sg = None if state_gate is None else np.ascontiguousarray(state_gate, np.float64).reshape(self.G, self.S)
        ag = None if action_gate is None else np.ascontiguousarray(action_gate, np.float64).reshape(self.G, self.A)
        _l.check(self.L.ddrl_pop_set_gates(self.h, _p(sg), _p(ag)))

    def set_gate_mode(self, st_bernoulli=False, st_scale=False, ac_bernoulli=False, ac_scale=False, seed=0):
        """probabilist / scale of the Developmental layer (DODevMLP.hpp:276-304): Bernoulli(w) gates per sample."""
        _l.check(self.L.ddrl_pop_set_gate_mode(self.h, int(st_bernoulli), int(st_scale), int(ac_bernoulli), int(ac_scale), int(seed)))

    def gates(self):
        sg, ag = np.empty((self.G, self.S)), np.empty((self.G, self.A))
        _l.check(self.L.ddrl_pop_get_gates(self.h, _p(sg), _p(ag)))
        return sg, ag

    def ewc_setup_net(self, net, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_pop_ewc_setup_net(self.h, net, _p(m)))

    def replay_clear(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_pop_replay_clear(self.h, _p(m)))

    def update_best_param(self, score, mask=None):
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(score, np.float64), (self.G,)))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        _l.check(self.L.ddrl_pop_update_best_param(self.h, _p(s), _p(m)))

    def ewc_setup(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        _l.check(self.L.ddrl_pop_ewc_setup(self.h, _p(m)))

    def ewc_decay_update(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        _l.check(self.L.ddrl_pop_ewc_decay_update(self.h, _p(m)))

    def ewc_cost(self, net):
        out = np.empty(self.G)
        _l.check(self.L.ddrl_pop_ewc_cost(self.h, net, _p(out)))
        return out

    def ewc_state(self, net):
        n = self.num_params(net)
        f, fh, ws = np.empty((self.G, n)), np.empty((self.G, n)), np.empty((self.G, n))
        nbr, dm = np.empty(self.G), np.empty(self.G)
        _l.check(self.L.ddrl_pop_get_ewc(self.h, net, _p(f), _p(fh), _p(ws), _p(nbr), _p(dm)))
        return dict(fisher=f, fisher_hat=fh, wstar=ws, fisher_nbr=nbr, decay_mult=dm)

    # save_run / load_previous_run (DeepQNAg.hpp:504-537) for the whole population
    def save_run(self, path):
        _l.check(self.L.ddrl_pop_save_run(self.h, str(path).encode()))

    def load_previous_run(self, path):
        _l.check(self.L.ddrl_pop_load_previous_run(self.h, str(path).encode()))

    # DODevMLP's flat parameter view, gate blobs included (ddrl_pop_set_gate_layout and friends)
    def set_gate_layout(self, st_control, ac_control, learnable):
        """st_control / ac_control: list of controlled dimensions, [] = every dimension, None = no such gate layer;
        learnable: heuristic 0 (gate weights are lr_mult-1 parameters)"""
        def arg(c):
            if c is None:
                return None, -1
            a = np.ascontiguousarray(c, np.int32)
            return a, int(a.size)
        s, ns = arg(st_control); a, na = arg(ac_control)
        _l.check(self.L.ddrl_pop_set_gate_layout(self.h, _p(s) if ns > 0 else None, ns, _p(a) if na > 0 else None, na, int(learnable)))

    def dev_num_params(self, net, ignore_null_lr=True):
        n = C.c_int()
        _l.check(self.L.ddrl_pop_dev_num_params(self.h, net, int(ignore_null_lr), C.byref(n)))
        return n.value

    def dev_weights(self, net, ignore_null_lr=True):
        out = np.empty((self.G, self.dev_num_params(net, ignore_null_lr)), np.float64)
        _l.check(self.L.ddrl_pop_dev_get_weights(self.h, net, int(ignore_null_lr), _p(out)))
        return out

    def set_dev_weights(self, net, flat, ignore_null_lr=True):
        f = np.ascontiguousarray(flat, np.float64).reshape(self.G, self.dev_num_params(net, ignore_null_lr))
        _l.check(self.L.ddrl_pop_dev_set_weights(self.h, net, int(ignore_null_lr), _p(f)))

    def load_flat_dev(self, d_flat_ptr, ld, mask=None, net=0):
        """ddrl_pop_load_flat_dev: fp32 rows [G x ld] (ld = 0: one row for every agent) already on the device -> a net's weights"""
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_pop_load_flat_dev(self.h, net, d_flat_ptr, ld, _p(m)))

    def set_option(self, key, value):
        _l.check(self.L.ddrl_pop_set_option(self.h, key.encode(), int(value)))

    def set_profiling(self, on=True):
        _l.check(self.L.ddrl_pop_set_profiling(self.h, int(on)))

    def profile_read(self):
        """(summed update-kernel milliseconds, launches) since the last read; synchronises."""
        ms, n = C.c_double(), C.c_int()
        _l.check(self.L.ddrl_pop_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class DdpgPopulation(Population):
    """G independent DDPG agents on one GPU (DeepQNAg<MLP> x G)."""

    def __init__(self, cfg, init_seed=0):
        self.L = _l.load()
        self.cfg = cfg
        h = C.c_void_p()
        _l.check(self.L.ddrl_ddpg_create(C.byref(cfg.c), init_seed, C.byref(h)))
        self.h = h
        self.G, self.S, self.A, self.B = cfg.c.n_agents, cfg.c.nb_sensors, cfg.c.nb_motors, cfg.c.mini_batch_size

    def seed_replay(self, seeds):
        s = np.ascontiguousarray(seeds, np.uint32).reshape(self.G)
        _l.check(self.L.ddrl_pop_seed_replay(self.h, _p(s)))

    def replay_append(self, s, a, r, s2, goal, mask=None):
        f = lambda x, n: np.ascontiguousarray(x, _l.REAL).reshape(self.G, n)
        s, a, s2 = f(s, self.S), f(a, self.A), f(s2, self.S)
        r = np.ascontiguousarray(r, _l.REAL).reshape(self.G)
        goal = np.ascontiguousarray(goal, np.uint8).reshape(self.G)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_pop_replay_append(self.h, _p(s), _p(a), _p(r), _p(s2), _p(goal), _p(m)))

    def replay_load(self, agent, rows):
        rows = np.ascontiguousarray(rows, _l.REAL).reshape(-1, 2 * self.S + self.A + 2)
        _l.check(self.L.ddrl_pop_replay_load(self.h, agent, rows.shape[0], _p(rows)))

    def replay_len(self):
        out = np.empty(self.G, np.int32)
        _l.check(self.L.ddrl_pop_replay_len(self.h, _p(out)))
        return out

    # one pass of end_episode's fupd loop body for every (masked) agent
    def update(self, idx=None, mask=None):
        i = None if idx is None else np.ascontiguousarray(idx, np.int32).reshape(self.G, self.B)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_pop_ddpg_update(self.h, _p(i), _p(m)))

    def update_n(self, n, idx=None, mask=None):
        """n consecutive updates per agent in one launch (the fupd loop with force_more_update = n - 1)"""
        i = None if idx is None else np.ascontiguousarray(idx, np.int32).reshape(self.G, n, self.B)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_pop_ddpg_update_n(self.h, int(n), _p(i), _p(m)))

    def last(self):
        idx = np.empty((self.G, self.B), np.int32)
        loss = np.empty(self.G, _l.REAL)
        qt = np.empty((self.G, self.B), _l.REAL)
        da = np.empty((self.G, self.B, self.A), _l.REAL)
        _l.check(self.L.ddrl_pop_ddpg_last(self.h, _p(idx), _p(loss), _p(qt), _p(da)))
        return dict(idx=idx, loss=loss, q_targets=qt, dact=da)



class Td3Population(DdpgPopulation):
    """G independent TD3 agents on one GPU (TD3Ag.hpp's DeepQNAg<MLP> x G): six nets per agent."""

    def __init__(self, cfg, init_seed=0):
        self.L = _l.load()
        self.cfg = cfg
        h = C.c_void_p()
        _l.check(self.L.ddrl_td3_create(C.byref(cfg.c), init_seed, C.byref(h)))
        self.h = h
        self.G, self.S, self.A, self.B = cfg.c.n_agents, cfg.c.nb_sensors, cfg.c.nb_motors, cfg.c.mini_batch_size

    def seed_noise(self, seeds):
        s = np.ascontiguousarray(seeds, np.uint64).reshape(self.G)
        _l.check(self.L.ddrl_pop_td3_seed_noise(self.h, _p(s)))

    # TD3Ag::end_episode for every (masked) agent; noise [G x B x A] or None (device draws)
    def update(self, idx=None, noise=None, actor_step=None, mask=None):
        i = None if idx is None else np.ascontiguousarray(idx, np.int32).reshape(self.G, self.B)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32).reshape(self.G, self.B, self.A)
        st = None if actor_step is None else np.ascontiguousarray(np.broadcast_to(np.asarray(actor_step, np.uint8), (self.G,)))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_pop_td3_update(self.h, _p(i), _p(nz), _p(st), _p(m)))

    def update_n(self, n, idx=None, mask=None):
        raise _l.DdrlError(-3, "TD3Ag has no fupd loop: one update per stored transition")

    def last(self):
        d = DdpgPopulation.last(self)
        loss2 = np.empty(self.G, np.float32)
        nz = np.empty((self.G, self.B, self.A), np.float32)
        _l.check(self.L.ddrl_pop_td3_last(self.h, _p(loss2), _p(nz)))
        d.update(loss2=loss2, noise=nz)
        return d


class NfacConfig:
    """The [agent]/[devnn] keys of OfflineCaclaAg::_unique_invoke (OfflineCaclaAg.hpp:124-179)."""

    def __init__(self, **kw):
        self.c = _l.NfacConfigC()
        _l.check(_l.load().ddrl_nfac_config_default(C.byref(self.c)))
        for k, v in kw.items():
            self.set(k, v)

    @classmethod
    def from_ini(cls, path, n_agents=1, nb_sensors=0, nb_motors=0, max_trajectory=0):
        o = cls()
        o.c.n_agents, o.c.nb_sensors, o.c.nb_motors, o.c.max_trajectory = n_agents, nb_sensors, nb_motors, max_trajectory
        _l.check(_l.load().ddrl_nfac_config_from_ini(path.encode(), C.byref(o.c)))
        return o

    def set(self, k, v):
        if k == "lambda":
            k = "lambda_"
        if k in ("hidden_unit_a", "hidden_unit_v"):
            v = list(v)
            if len(v) > _l.DDRL_MAX_HIDDEN:
                raise ValueError("at most %d hidden layers" % _l.DDRL_MAX_HIDDEN)
            arr = getattr(self.c, k)
            for i, x in enumerate(v):
                arr[i] = int(x)
            setattr(self.c, "n_hidden_" + k[-1], len(v))
        elif hasattr(self.c, k):
            setattr(self.c, k, v)
        else:
            raise KeyError(k)

    def get(self, k):
        if k == "lambda":
            k = "lambda_"
        if k in ("hidden_unit_a", "hidden_unit_v"):
            return list(getattr(self.c, k))[: getattr(self.c, "n_hidden_" + k[-1])]
        return getattr(self.c, k)


class NfacPopulation(Population):
    """G independent NFAC(lambda)-V agents on one GPU (OfflineCaclaAg<MLP> x G); nets: NET_ACTOR = ann,
    NET_CRITIC = vnn."""

    def __init__(self, cfg, init_seed=0):
        self.L = _l.load()
        self.cfg = cfg
        h = C.c_void_p()
        _l.check(self.L.ddrl_nfac_create(C.byref(cfg.c), init_seed, C.byref(h)))
        self.h = h
        self.G, self.S, self.A, self.maxT = cfg.c.n_agents, cfg.c.nb_sensors, cfg.c.nb_motors, cfg.c.max_trajectory

    # trajectory.push_back({s, pure_a, a, next_s, r, goal}) for every (masked) agent
    def traj_append(self, s, pure_a, a, s2, r, goal, mask=None):
        f = lambda x, n: np.ascontiguousarray(x, np.float32).reshape(self.G, n)
        s, pure_a, a, s2 = f(s, self.S), f(pure_a, self.A), f(a, self.A), f(s2, self.S)
        r = np.ascontiguousarray(r, np.float32).reshape(self.G)
        goal = np.ascontiguousarray(goal, np.uint8).reshape(self.G)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_nfac_traj_append(self.h, _p(s), _p(pure_a), _p(a), _p(s2), _p(r), _p(goal), _p(m)))

    def traj_load(self, agent, rows, ends):
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 2 * self.S + 2 * self.A + 2)
        ends = np.ascontiguousarray(ends, np.int32)
        _l.check(self.L.ddrl_nfac_traj_load(self.h, agent, rows.shape[0], _p(rows), ends.size, _p(ends)))

    def episode_end(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_nfac_episode_end(self.h, _p(m)))

    def traj_len(self):
        out = np.empty(self.G, np.int32)
        _l.check(self.L.ddrl_nfac_traj_len(self.h, _p(out)))
        return out

    # the learning branch of OfflineCaclaAg::end_episode for every (masked) agent
    def update(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_nfac_update(self.h, _p(m)))

    # MLP::learn_batch_lw (learn_batch with lw = None) on every (masked) agent's critic: iters full-batch Adam steps
    def learn_batch_lw(self, x, y, lw=None, iters=1, mask=None):
        x = np.ascontiguousarray(x, np.float32).reshape(self.G, -1, self.S)
        n = x.shape[1]
        y = np.ascontiguousarray(y, np.float32).reshape(self.G, n)
        w = None if lw is None else np.ascontiguousarray(lw, np.float32).reshape(self.G, n)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.G)
        _l.check(self.L.ddrl_nfac_learn_batch_lw(self.h, _p(x), _p(y), _p(w), n, int(iters), _p(m)))

    def last(self):
        deltas = np.empty((self.G, self.maxT), np.float32)
        n = np.empty(self.G, np.int32)
        loss = np.empty(self.G, np.float32)
        _l.check(self.L.ddrl_nfac_last(self.h, _p(deltas), _p(n), _p(loss)))
        return dict(deltas=deltas, n_pos=n, critic_loss=loss)


class PolicyPopulation(Population):
    """G policy nets and nothing else (ddrl_policy_create): CMAESAg's `ann` for every individual."""

    def __init__(self, n_agents, nb_sensors, nb_motors, hidden, hidden_layer_type=1, output_layer_type=2, device=-1,
                 init_seed=0):
        self.L = _l.load()
        hid = np.ascontiguousarray(hidden, np.int32)
        h = C.c_void_p()
        _l.check(self.L.ddrl_policy_create(n_agents, nb_sensors, nb_motors, _p(hid), hid.size, hidden_layer_type,
                                           output_layer_type, device, init_seed, C.byref(h)))
        self.h, self.G, self.S, self.A = h, n_agents, nb_sensors, nb_motors

