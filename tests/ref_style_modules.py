"""RHS modules with the attribute surface of the reference's `Lambda` / `ODEFunc` classes, for driving the drop-in
`odeint` the way the reference scripts do (train-s1.py:134-247, train-d1.py:134-187, train-d2.py:191-272).
Written for these tests; the rate constants come from tests/kat_cases.py."""
import numpy as np
import torch
import torch.nn as nn
from scipy.interpolate import interp1d

import kat_cases as K


class _Protocol:
    def set_fixed_form_voltage_protocol(self, t, v):
        self._t_regular = t
        self._v_regular = v
        self._interp = interp1d(t, v)

    def _v(self, t):
        return torch.from_numpy(self._interp([t.cpu().detach().numpy()]))

    def _v_or_hold(self, t):
        try:
            return self._v(t)
        except ValueError:
            return torch.tensor([-80])


class HodgkinHuxley(nn.Module, _Protocol):
    """2-state ground truth (a, r)."""

    def __init__(self, p=K.P_HH):
        super().__init__()
        for i, val in enumerate(p, 1):
            setattr(self, f"p{i}", float(val))

    def forward(self, t, y):
        a, r = torch.unbind(y[0])
        v = self._v_or_hold(t)
        k1 = self.p1 * torch.exp(self.p2 * v)
        k2 = self.p3 * torch.exp(-self.p4 * v)
        k3 = self.p5 * torch.exp(self.p6 * v)
        k4 = self.p7 * torch.exp(-self.p8 * v)
        return torch.stack([(k1 * (1.0 - a) - k2 * a)[0], (-k3 * r + k4 * (1.0 - r))[0]])


class Markov6(nn.Module, _Protocol):
    """6-state ground truth (C1, C2, I, IC1, IC2, O)."""

    def __init__(self, p=K.P_M6):
        super().__init__()
        for i, val in enumerate(p, 1):
            setattr(self, f"p{i}", float(val))

    def forward(self, t, y):
        c1, c2, i, ic1, ic2, o = torch.unbind(y[0])
        v = self._v_or_hold(t)
        a1, b1 = self.p1 * torch.exp(self.p2 * v), self.p3 * torch.exp(-self.p4 * v)
        bh, ah = self.p5 * torch.exp(self.p6 * v), self.p7 * torch.exp(-self.p8 * v)
        a2, b2 = self.p9 * torch.exp(self.p10 * v), self.p11 * torch.exp(-self.p12 * v)
        d = [a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1,
             b1 * c1 + ah * ic2 - (a1 + bh) * c2,
             a2 * ic1 + bh * o - (b2 + ah) * i,
             a1 * ic2 + bh * c1 + b2 * i - (b1 + ah + a2) * ic1,
             b1 * ic1 + bh * c2 - (ah + a1) * ic2,
             a2 * c1 + ah * i - (b2 + bh) * o]
        return torch.stack([x[0] for x in d])


def make_net(n_layers=5, n_nodes=200):
    layers = [nn.Linear(2, n_nodes), nn.LeakyReLU()]
    for _ in range(n_layers):
        layers += [nn.Linear(n_nodes, n_nodes), nn.LeakyReLU()]
    layers += [nn.Linear(n_nodes, 1)]
    return nn.Sequential(*layers)


def load_flat_weights(net, flat):
    """Fill an nn.Sequential from the flat fp32 fixture (state-dict order)."""
    off = 0
    with torch.no_grad():
        for m in net:
            if isinstance(m, nn.Linear):
                n = m.weight.numel()
                m.weight.copy_(torch.from_numpy(flat[off:off + n].reshape(m.weight.shape)))
                off += n
                m.bias.copy_(torch.from_numpy(flat[off:off + m.bias.numel()]))
                off += m.bias.numel()
    assert off == flat.size


class NNf(nn.Module, _Protocol):
    """da/dt = net([V/100, a]) / 1000;  dr/dt from the HH inactivation rates."""

    def __init__(self, p=K.P_HH, n_layers=5, n_nodes=200):
        super().__init__()
        self.net = make_net(n_layers, n_nodes)
        self.vrange = torch.tensor([100.0])
        self.netscale = torch.tensor([1000.0])
        self.p5, self.p6, self.p7, self.p8 = (float(x) for x in p[4:8])
        self.unity = torch.tensor([1])

    def forward(self, t, y):
        a, r = torch.unbind(y, dim=1)
        v = self._v_or_hold(t)
        nv = v / self.vrange
        k3 = self.p5 * torch.exp(self.p6 * v)
        k4 = self.p7 * torch.exp(-self.p8 * v)
        drdt = -k3 * r + k4 * (self.unity - r)
        dadt = self.net(torch.stack([nv[0], a[0]]).float()) / self.netscale
        return torch.stack([dadt[0], drdt[0]]).reshape(1, -1)


class NNd(NNf):
    """da/dt = HH activation + net([V/100, a]) / 1000."""

    def __init__(self, p=K.P_HH, n_layers=5, n_nodes=200):
        super().__init__(p, n_layers, n_nodes)
        self.p1, self.p2, self.p3, self.p4 = (float(x) for x in p[0:4])

    def _dadt(self, a, v):
        k1 = self.p1 * torch.exp(self.p2 * v)
        k2 = self.p3 * torch.exp(-self.p4 * v)
        return k1 * (self.unity - a) - k2 * a

    def forward(self, t, y):
        a, r = torch.unbind(y, dim=1)
        v = self._v_or_hold(t)
        nv = v / self.vrange
        k3 = self.p5 * torch.exp(self.p6 * v)
        k4 = self.p7 * torch.exp(-self.p8 * v)
        drdt = -k3 * r + k4 * (self.unity - r)
        dadt = self._dadt(a, v).reshape(-1)
        dadt = dadt + (self.net(torch.stack([nv[0], a[0]]).float()) / self.netscale).reshape(-1)
        return torch.stack([dadt[0], drdt[0]]).reshape(1, -1)
