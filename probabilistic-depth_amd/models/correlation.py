"""Drop-in for the reference's correlation module (forward and backward).

Same constructor arguments as models/correlation_package/correlation.py:47-61 (`Correlation(pad_size,
kernel_size, max_displacement, stride1, stride2, corr_multiply)`) and as models/correlation_native.py:6-23
(`Correlation(max_displacement)`); the forward runs the HIP kernel behind pdepth_correlation_forward_f32, the
backward (reference: correlation_cuda_kernel.cu:116-300) the one behind pdepth_correlation_backward_f32 through a
torch.autograd.Function, like models/correlation_package/correlation.py:6-44.
"""
import torch.nn as nn

from .. import ops


class Correlation(nn.Module):
    def __init__(self, pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply
        self.output_dim = 2 * (max_displacement // stride2) + 1

    def forward(self, input1, input2):
        return ops.correlation(input1, input2, self.pad_size, self.kernel_size, self.max_displacement, self.stride1,
                               self.stride2, self.corr_multiply)
