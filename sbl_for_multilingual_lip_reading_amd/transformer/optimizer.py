"""Mirror of SBL_Multilingual_Lip_reading/transformer/optimizer.py."""


class TransformerOptimizer(object):
    """A simple wrapper class for learning rate scheduling (optimizer.py:1-27): Noam schedule
    lr = k * 512^-0.5 * min(step^-0.5, step * warmup^-1.5), set on every param group before each step."""

    def __init__(self, optimizer, warmup_steps=4000, k=0.2):
        self.optimizer = optimizer
        self.k = k
        self.warmup_steps = warmup_steps
        d_model = 512
        self.init_lr = d_model ** (-0.5)
        self.lr = self.init_lr
        self.step_num = 0

    def zero_grad(self):
        self.optimizer.zero_grad()

    def step(self):
        self._update_lr()
        self.optimizer.step()

    def _update_lr(self):
        self.step_num += 1
        self.lr = self.k * self.init_lr * min(self.step_num ** (-0.5),
                                              self.step_num * (self.warmup_steps ** (-1.5)))
        for param_group in self.optimizer.param_groups:
            param_group['lr'] = self.lr


class FusedAdam(object):
    """torch.optim.Adam(betas=(0.9, 0.98), eps=1e-9) as SBL/train.py:75 configures it, as ONE kernel launch over the
    flat parameter / gradient buffers of a dp.FlatModel (81 M parameters: 2.3 GB of traffic per step instead of 475
    per-tensor launches).  Duck-types the slice of the torch optimizer API that TransformerOptimizer and the
    reference's train loop use: param_groups[...]['lr'], zero_grad(), step(), state_dict()/load_state_dict().
    grad_scale folds the 1/world_size of data-parallel averaging into the update when the exchange sums."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, grad_scale=1.0):
        import torch
        from ._env import ops
        self._ops = ops
        self.flat = flat
        self.param_groups = [{"params": list(flat.model.parameters()), "lr": lr, "betas": betas, "eps": eps}]
        self.exp_avg = torch.zeros_like(flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(flat.flat_param)
        self.step_count = 0
        self.grad_scale = grad_scale

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self):
        """One launch per maximal run of trainable parameters (one launch in all when nothing is frozen): like
        Adam(filter(lambda p: p.requires_grad, model.parameters())) in SBL/train.py:75, frozen parameters and their
        moments are left untouched."""
        g = self.param_groups[0]
        self.step_count += 1
        for a, b in self.flat.trainable_ranges():
            self._ops.adam_step(self.flat.flat_param[a:b], self.flat.flat_grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b],
                                g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.step_count, self.grad_scale)
        # the kernel wrote the weights behind torch's back: re-pack the cached convolution weight images in place
        self._ops.refresh_packed_weights()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0]["lr"] = sd["lr"]
