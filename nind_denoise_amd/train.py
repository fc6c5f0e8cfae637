"""UtNet training step on MI355X (BASELINE config 5): forward + loss + backward in libnind_hip.so, Adam(amsgrad) on flat
buffers, data parallel over RCCL.

Mirrors the reference's generator update (/root/reference/src/nind_denoise/nn_common.py:163-218 and the loop body of
nn_train.py:308-380):  generated = model(noisy).clip(0,1);  loss = sum_k weight_k * criterion_k(generated, clean);
loss.backward();  Adam(lr, betas=(beta1, .999), amsgrad=True).step().   Criteria: L1, MSE, SSIM and MSSSIM (the
reference's SSIM / MS-SSIM criteria come from piqa, which is not installed here: csrc/ssim.hip restates its published
algorithm, forward and backward -- see oracle/losses.py).  Every criterion is reduced with a mean over the batch; the
reference builds its criteria with reduction=None and calls .backward() on the per-sample vector, which torch rejects
for batches larger than one (nn_common.py:170-177, 216), so there is no other reading of "the loss" to reproduce.

The module's parameters are views into ONE flat fp32 buffer in state-dict order; gradients come back in a second flat
buffer with the same layout, so the optimizer is one kernel and the data-parallel reduction runs on nine contiguous level
buckets of that buffer, each all-reduced as soon as the backward pass has finished it (dist.BucketedGradientAverager).
"""
import ctypes

import torch

from . import _lib
from .dist import BucketedGradientAverager
from .networks.UtNet import UtNet, valid_cs


class UtNetTrainer:
    def __init__(self, model: UtNet, lr=1e-4, beta1=0.75, beta2=0.999, eps=1e-8, amsgrad=True,
                 weights=None, device=None, process_group=None, loss_cs=None):
        if model.activation != "PReLU":
            raise NotImplementedError("the HIP training step implements PReLU networks")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise RuntimeError("UtNetTrainer needs a GPU (no CPU fallback)")
        self.model = model.to(self.device)
        self.lib = _lib.load()
        self.funit = model.funit
        self.weights = {"L1": 0.0, "MSE": 1.0} if weights is None else dict(weights)
        unknown = set(k for k, v in self.weights.items() if v) - {"L1", "MSE", "SSIM", "MSSSIM"}
        if unknown:   # D1 / D2: the discriminator terms of the reference's GAN mode (nn_common.py:178-182)
            raise NotImplementedError(f"loss terms {sorted(unknown)} are not available (generator losses: L1, MSE, SSIM, MSSSIM)")
        self.lr, self.betas, self.eps, self.amsgrad = lr, (beta1, beta2), eps, amsgrad
        self.group = process_group
        self.loss_cs = loss_cs        # centre crop the criteria see (nn_train.py:319-323 --loss_cs); None: the whole crop
        n = self.lib.nd_utnet_param_count(self.funit)
        if n == 0:
            raise ValueError(f"funit={self.funit} is not supported")
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.flat)
        sd = dict(model.named_parameters())
        self.ranges = {}
        for i, name in enumerate(_lib.utnet_tensor_names()):
            off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(self.lib.nd_utnet_param_range(self.funit, i, off, cnt))
            p = sd[name]
            assert p.numel() == cnt.value, (name, p.numel(), cnt.value)
            view = self.flat[off.value:off.value + cnt.value].view_as(p)
            view.copy_(p.data)
            p.data = view                      # the module now reads / writes the flat buffer
            self.ranges[name] = (off.value, cnt.value)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.vmax = torch.zeros_like(self.flat)
        self.steps = 0
        self.blobs = torch.empty(self.lib.nd_utnet_train_blob_bytes(self.funit), dtype=torch.uint8, device=self.device)
        self._ws = {}
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        # data parallel: the gradient mean runs bucket by bucket (one per network level) under the rest of the backward pass
        self.averager = BucketedGradientAverager(self.funit, self.grads, self.group)

    def workspace(self, cs, batch):
        key = (cs, batch)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = self.lib.nd_utnet_train_workspace_bytes(self.funit, cs, batch)
            if nbytes == 0:
                _lib.check(self.lib.nd_utnet_train_workspace_init(None, 0, self.funit, cs, batch, None), "UtNet training")
            self._ws.clear()
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.nd_utnet_train_workspace_init(ws.data_ptr(), nbytes, self.funit, cs, batch,
                                                              _lib.stream_ptr(self.device)), "nd_utnet_train_workspace_init")
            self._ws[key] = ws
        return ws

    def forward_backward(self, noisy, clean):
        """Forward + loss + backward; fills self.grads (averaged over the process group).  Returns (output, loss tensor)."""
        noisy = noisy.to(self.device, torch.float32).contiguous()
        clean = clean.to(self.device, torch.float32).contiguous()
        if noisy.shape != clean.shape or noisy.dim() != 4 or noisy.size(1) != 3 or noisy.size(2) != noisy.size(3):
            raise ValueError(f"expected two [B,3,S,S] batches, got {tuple(noisy.shape)} and {tuple(clean.shape)}")
        batch, cs = noisy.size(0), noisy.size(2)
        if not valid_cs(cs):
            raise ValueError(f"crop size {cs} is not of the form 16k+56 (e.g. 136, 184)")
        y = torch.empty_like(noisy)
        with torch.cuda.device(self.device):
            ws = self.workspace(cs, batch)
            _lib.check(self.lib.nd_utnet_train_step_ev(self.funit, self.model.flags, self.flat.data_ptr(), self.grads.data_ptr(),
                                                       self.blobs.data_ptr(), noisy.data_ptr(), clean.data_ptr(), y.data_ptr(),
                                                       float(self.weights.get("L1", 0.0)), float(self.weights.get("MSE", 0.0)),
                                                       float(self.weights.get("SSIM", 0.0)), float(self.weights.get("MSSSIM", 0.0)),
                                                       self.loss.data_ptr(), batch, cs, int(self.loss_cs or 0), ws.data_ptr(), ws.numel(),
                                                       _lib.stream_ptr(self.device), self.averager.event_ptrs, len(self.averager.buckets)),
                       "nd_utnet_train_step_ev")
            # RCCL: nine all-reduces (0.6 ... 57 MB for UtNet(64)), each behind its bucket's event on a side stream
            self.averager.reduce()
        return y, self.loss

    def optimizer_step(self):
        self.steps += 1
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nd_adam_step(self.flat.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                             self.vmax.data_ptr(), self.flat.numel(), self.lr, self.betas[0], self.betas[1],
                                             self.eps, self.steps, int(self.amsgrad), _lib.stream_ptr(self.device)),
                       "nd_adam_step")
        # the parameters were rewritten through raw pointers (torch's version counters did not move): the inference blob the
        # module cached before this step is stale
        self.model.weights_generation += 1

    def learn(self, noisy, clean):
        """One generator update (Generator.denoise_batch + learn of the reference); returns the loss as a float tensor."""
        _, loss = self.forward_backward(noisy, clean)
        self.optimizer_step()
        return loss

    def grad_of(self, name):
        off, cnt = self.ranges[name]
        return self.grads[off:off + cnt].view_as(dict(self.model.named_parameters())[name])

    def update_learning_rate(self, lr_decay):
        self.lr *= lr_decay
        return self.lr
