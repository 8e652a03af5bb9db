"""hexagdly-compatible hexagonal convolution layer backed by the HIP kernel.

`Conv2d` takes the constructor arguments, parameter names (`kernel0`, `kernel1`,
`bias_tensor`), shapes and default initialisation of `hexagdly.Conv2d`, which
/root/reference/gridnext/gridnet_models.py:130-147 instantiates with
kernel_size=1, stride=1, bias=True - so `import gridnext_amd.hexconv as hexagdly`
is a drop-in for what GridNext uses, and reference checkpoints load by name.
Only the size-1 / stride-1 layer GridNext needs is implemented.
"""
import torch
import torch.nn as nn

from . import functional as GF


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, bias=True, debug=False):
        super().__init__()
        if kernel_size != 1 or stride != 1:
            raise NotImplementedError("GridNext's corrector only uses hexagdly.Conv2d(kernel_size=1, stride=1)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.debug = kernel_size, stride, debug
        self.kernel0 = nn.Parameter(torch.empty(out_channels, in_channels, 3, 1))
        self.kernel1 = nn.Parameter(torch.empty(out_channels, in_channels, 2, 2))
        if bias:
            self.bias_tensor = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias_tensor', None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.debug:
            for p in (self.kernel0, self.kernel1, self.bias_tensor):
                if p is not None:
                    nn.init.constant_(p, 1.0)
        else:
            nn.init.xavier_uniform_(self.kernel0)
            nn.init.xavier_uniform_(self.kernel1)
            if self.bias_tensor is not None:
                nn.init.constant_(self.bias_tensor, 0.01)

    def forward_nhwc(self, x_nhwc, oddr):
        """Channels-last entry used by the grid models: x [B, H, W, C_in] -> [B, H, W, C_out]."""
        return GF.hexconv(x_nhwc, self.kernel0, self.kernel1, self.bias_tensor, oddr)

    def forward(self, x):
        """hexagdly call convention: x (B, C_in, rows, cols) in hexagdly addressing (odd columns shifted down)."""
        y = self.forward_nhwc(x.permute(0, 2, 3, 1), oddr=False)
        return y.permute(0, 3, 1, 2)

    def extra_repr(self):
        return '%d, %d, kernel_size=%d, stride=%d' % (self.in_channels, self.out_channels, self.kernel_size,
                                                     self.stride)
