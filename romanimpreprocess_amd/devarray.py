"""Device-resident arrays for the host mirrors.

The stage-level entry points of ``libromanhip`` that the post-path and the noise layers use (``rip_stage_select_ranks``,
``rip_stage_legendre2d``, ``rip_stage_gauss_hist``, ``rip_stage_bin_mean``, ``rip_stage_build_mask``, ``rip_stage_endslice``,
``rip_stage_noise_inject``, ``rip_stage_poisson_resample``, ``rip_stage_pearson``) copy their array arguments with
``hipMemcpyDefault``: a pointer may be a host array or device memory.  ``DevArray`` wraps a torch tensor on the GPU so that the
mirrors written for numpy arrays (``utils/sky.py``, ``L1_to_L2/gen_noise_image.py``) can be handed a plane that never leaves
HBM: it offers what those functions ask of an array -- ``shape``, ``dtype`` (numpy's), ``ndim``, ``size``,
``flags.c_contiguous``, ``ctypes.data`` (the DEVICE pointer) and ``reshape`` -- and nothing that would compute on the host.
The calls stay synchronous (they return when the kernels are done), so torch operations may follow them directly; operations
queued on a torch stream BEFORE such a call must be complete (``sync()``), the library runs on its own stream.
"""

import numpy as np

_TORCH_TO_NP = None


def _dtype_of(t):
    global _TORCH_TO_NP
    if _TORCH_TO_NP is None:
        import torch

        _TORCH_TO_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int16: np.int16,
                        torch.uint8: np.uint8, torch.int8: np.int8, torch.int64: np.int64}
    return np.dtype(_TORCH_TO_NP[t.dtype])


class _Flags:
    c_contiguous = True


class _Ctypes:
    def __init__(self, ptr):
        self.data = ptr


class DevArray:
    """A contiguous torch tensor on the GPU seen through the few array attributes the host mirrors use."""

    def __init__(self, tensor, dtype=None):
        if not tensor.is_cuda or not tensor.is_contiguous():
            raise ValueError("DevArray needs a contiguous tensor on the GPU")
        self.t = tensor
        self.dtype = np.dtype(dtype) if dtype is not None else _dtype_of(tensor)   # e.g. uint16 bits kept in an int16 tensor
        if self.dtype.itemsize != tensor.element_size():
            raise ValueError("dtype of another width than the tensor's")
        self.shape = tuple(tensor.shape)
        self.ndim = tensor.dim()
        self.size = tensor.numel()
        self.flags = _Flags()
        self.ctypes = _Ctypes(tensor.data_ptr())

    def reshape(self, *shape):
        return DevArray(self.t.reshape(*shape), self.dtype)

    def sync(self):
        """wait for torch's work on this device (before handing the tensor to the library's stream)"""
        import torch

        torch.cuda.current_stream(self.t.device).synchronize()
        return self

    def numpy(self):
        a = self.t.cpu().numpy()
        return a.view(self.dtype) if a.dtype != self.dtype else a


def is_dev(a):
    return isinstance(a, DevArray)
