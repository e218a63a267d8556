"""MLP state-space regression on the GPU: the reference's training loop

    opt = optim.Adam(func.net.parameters(), lr=0.001); scheduler = StepLR(opt, step_size=100, gamma=0.9)
    for itr in range(4000):
        p = func.net(x_av.float()) / func.netscale            # NN-d: p += model_dadt            (train-d2.py:902-903)
        loss = MSELoss(reduction='sum')(p.reshape(-1), y_dadt.float())
        opt.zero_grad(); loss.backward(); opt.step(); scheduler.step()                          (train-s1.py:891-909)

as four asynchronous launches per iteration through the C ABI (csrc/ionode_regress.hpp): fused forward + backward of the
net per 16-row tile on the fp32 MFMA with LDS-resident activations, split-K MFMA reduction of the weight gradient, Adam on
the flat state dict, refresh of the MFMA fragment image.  No host synchronisation inside the loop; there is no CPU path.
"""
import ctypes as C
import math

import os

import numpy as np
import torch

from . import capi, grad


class MlpRegression:
    """Trainer state for one net(2 -> N x L -> 1) and one full-batch data set, all HBM-resident.

    x [M, 2] (V / vrange, a), y [M] targets, offset [M] or None (NN-d's model_dadt): anything convertible to fp32 device
    tensors (the reference casts with .float()).  weights_flat: fp32 state dict in the reference's order."""

    def __init__(self, weights_flat, mlp_layers, mlp_width, x, y, offset=None, *, netscale=1000.0, lr=1e-3,
                 betas=(0.9, 0.999), eps=1e-8, step_size=100, gamma=0.9, device=None):
        if not torch.cuda.is_available():
            raise capi.IonodeError("no HIP device visible: the regression step has no CPU fallback")
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.L, self.N = int(mlp_layers), int(mlp_width)
        L, N, dev = self.L, self.N, self.dev
        lib = capi.lib()
        n = 2 * N + N + L * (N * N + N) + N + 1
        w = np.ascontiguousarray(np.asarray(weights_flat, dtype=np.float32).reshape(-1))
        if w.size != n:
            raise capi.IonodeError(f"state dict has {w.size} floats, (L={L}, N={N}) needs {n}")
        assert n < (1 << 24), "index maps go through fp32"
        self.n = n
        f32 = lambda t: torch.as_tensor(np.asarray(t) if not isinstance(t, torch.Tensor) else t).to(device=dev, dtype=torch.float32).contiguous()
        self.x, self.y = f32(x).reshape(-1, 2), f32(y).reshape(-1)
        self.offset = None if offset is None else f32(offset).reshape(-1)
        self.M = self.x.shape[0]
        assert self.y.shape[0] == self.M and (self.offset is None or self.offset.shape[0] == self.M)
        self.w = torch.from_numpy(w.copy()).to(dev)
        self.m, self.v = torch.zeros_like(self.w), torch.zeros_like(self.w)
        # index maps, built once: flat index -> padded partial-gradient index; grad-image element -> flat index + 1
        partf = lib.ionode_grad_partial_floats(L, N)
        self.partf = partf
        self.padmap = grad.unpack_partial(torch.arange(partf, dtype=torch.float64), L, N).to(torch.int32).to(dev)
        nimg = lib.ionode_grad_image_floats(L, N)
        idx = np.arange(1, n + 1, dtype=np.float32)
        img = np.empty(nimg, dtype=np.float32)
        if lib.ionode_grad_pack(idx.ctypes.data, L, N, img.ctypes.data) != 0:
            raise capi.IonodeError(lib.ionode_grad_last_error().decode())
        self.imgmap = torch.from_numpy(img.astype(np.int32)).to(dev)
        self.image = torch.empty(nimg, dtype=torch.float32, device=dev)
        self.tiles = (self.M + 15) // 16
        self.recf = lib.ionode_grad_record_floats(L, N)
        self.records = torch.empty(self.tiles * self.recf, dtype=torch.float32, device=dev)
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        # persistent grid: TWO workgroups per compute unit for N <= 200 (256 registers and 70 KB of LDS each: one tile's layer boundaries
        # run beside the other's MFMAs), one for N = 500.  IONODE_REGRESS_WG_PER_CU: dev override for A/B runs
        per_cu = int(os.environ.get("IONODE_REGRESS_WG_PER_CU", "2" if N <= 208 else "1"))
        self.n_wg = int(min(self.tiles, cus * per_cu))
        self.n_slabs = int(os.environ.get("IONODE_REGRESS_SLABS", 0)) or int(lib.ionode_grad_reduce_slabs(L, N, self.tiles))   # (one round of the reduce kernel's workgroups on this device; env: dev override for A/B runs)
        self.loss_part = torch.zeros(self.n_wg, dtype=torch.float64, device=dev)
        self.partials = torch.empty((self.n_slabs, partf), dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.w)
        self.netscale, self.lr0, self.betas, self.eps = float(netscale), float(lr), betas, float(eps)
        self.step_size, self.gamma = int(step_size), float(gamma)
        self.t = 0                                        # optimiser steps taken (Adam's step count, StepLR's epoch)
        self._refresh()

    def _check(self, rc):
        if rc != 0:
            raise capi.IonodeError(capi.lib().ionode_grad_last_error().decode())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _refresh(self):
        self._check(capi.lib().ionode_image_refresh(self.L, self.N, C.c_void_p(self.imgmap.data_ptr()),
                                                    C.c_void_p(self.w.data_ptr()), C.c_void_p(self.image.data_ptr()),
                                                    self._stream()))

    def lr(self):
        """StepLR: lr0 * gamma ** (steps // step_size)  (scheduler.step() follows opt.step() in the reference loop)."""
        return self.lr0 * self.gamma ** (self.t // self.step_size)

    def _forward_backward(self):
        lib = capi.lib()
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._check(lib.ionode_regress_step(self.L, self.N, p(self.image), p(self.x), p(self.offset), p(self.y), self.M,
                                            C.c_float(self.netscale), p(self.records), p(self.loss_part), self.n_wg,
                                            self._stream()))
        self._check(lib.ionode_grad_reduce(self.L, self.N, p(self.records), self.tiles, self.n_slabs, p(self.partials),
                                           self._stream()))

    def loss_and_grad(self):
        """(loss, dloss/dweights [n] fp32) at the current weights, without an optimiser step (device tensors)."""
        self._forward_backward()
        lib = capi.lib()
        p = lambda t: C.c_void_p(t.data_ptr())
        self._check(lib.ionode_adam_step(self.n, self.n_slabs, self.L, self.N, p(self.partials), p(self.padmap), None, None,
                                         None, C.c_float(0), C.c_float(0), C.c_float(0), C.c_float(0), 1, p(self.grad), 0,
                                         self._stream()))
        return self.loss_part.sum(), self.grad

    def step(self):
        """One iteration of the reference loop.  Returns the loss BEFORE the update as a 0-dim device tensor (no sync)."""
        self._forward_backward()
        lib = capi.lib()
        p = lambda t: C.c_void_p(t.data_ptr())
        lr = self.lr()
        self.t += 1
        self._check(lib.ionode_adam_step(self.n, self.n_slabs, self.L, self.N, p(self.partials), p(self.padmap), p(self.w),
                                         p(self.m), p(self.v), C.c_float(lr), C.c_float(self.betas[0]),
                                         C.c_float(self.betas[1]), C.c_float(self.eps), self.t, p(self.grad), 1, self._stream()))
        loss = self.loss_part.sum()
        self._refresh()
        return loss

    def fit(self, n_iter, log_every=0):
        """n_iter iterations; returns the losses at the logged iterations [(itr, lr, loss)] (one sync per logged value)."""
        out = []
        for itr in range(n_iter):
            lr = self.lr()
            loss = self.step()
            if log_every and itr % log_every == 0:
                out.append((itr, lr, float(loss.item())))
        return out

    def load_adam_state(self, exp_avg, exp_avg_sq, step):
        """Resume: Adam moments in flat state-dict order (preprocess.adam_state_to_flat of a reference checkpoint's
        'optimizer' entry) and the number of optimiser steps taken (also StepLR's epoch)."""
        self.m.copy_(torch.as_tensor(np.asarray(exp_avg, dtype=np.float32)).to(self.dev))
        self.v.copy_(torch.as_tensor(np.asarray(exp_avg_sq, dtype=np.float32)).to(self.dev))
        self.t = int(step)

    def state_dict_flat(self):
        return self.w.detach().cpu().numpy().copy()
