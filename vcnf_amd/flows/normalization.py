"""ActNorm: AffineConstFlow whose scale/shift are initialised from the first batch it
sees (Glow).  The affine map runs on the HIP per-channel kernel; the one-off statistics
are plain torch reductions.  Reference: normflow/flows/normalization.py:8-38."""
import torch

from .affine.coupling import AffineConstFlow


class ActNorm(AffineConstFlow):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.data_dep_init_done_cpu = torch.tensor(0.)
        self.register_buffer('data_dep_init_done', self.data_dep_init_done_cpu)

    def _init_from(self, z, inverse):
        """normalization.py:22-27 / :33-37: make the first batch come out standardised."""
        std = z.std(dim=self.batch_dims, keepdim=True)
        mean = z.mean(dim=self.batch_dims, keepdim=True)
        with torch.no_grad():
            if inverse:
                self.s.data = torch.log(std + 1e-6).data
                self.t.data = mean.data
            else:
                self.s.data = (-torch.log(std + 1e-6)).data
                self.t.data = (-mean * torch.exp(self.s)).data
            self.data_dep_init_done = torch.ones_like(self.data_dep_init_done)

    def _initialised(self):
        """normalization.py:29 / :40 test the device flag on every call (a host synchronisation per layer and call,
        and not capturable into a HIP graph).  The flag only ever goes from 0 to 1, so the first positive answer is
        remembered on the host; load_state_dict() / refresh_packed() forget it."""
        if self.__dict__.get('_init_done_host', False):
            return True
        done = bool(self.data_dep_init_done > 0.)
        if done:
            self.__dict__['_init_done_host'] = True
        return done

    def forward(self, z):
        if not self._initialised():
            self._init_from(z, inverse=False)
        return super().forward(z)

    def inverse(self, z):
        if not self._initialised():
            self._init_from(z, inverse=True)
        return super().inverse(z)
