"""Emulation mode only: the GPU tests prepare "device" buffers with torch (`.cuda()`, `device="cuda"`); under the CPU emulation device memory
IS host memory, so every cuda device a test names is rewritten to the CPU and the stream / synchronise calls become no-ops.  The tests
themselves are not changed."""
import torch
from torch.overrides import TorchFunctionMode


def _is_cuda(d):
    if isinstance(d, torch.device):
        return d.type == "cuda"
    if isinstance(d, str):
        return d.startswith("cuda")
    return False


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        name = getattr(func, "__name__", "")
        if name == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        if name == "pin_memory" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        if "device" in kwargs and _is_cuda(kwargs["device"]):
            kwargs["device"] = "cpu"
        if "pin_memory" in kwargs:
            kwargs["pin_memory"] = False
        if args and any(_is_cuda(a) for a in args):
            args = tuple("cpu" if _is_cuda(a) else a for a in args)
        return func(*args, **kwargs)


_mode = None


def install():
    global _mode
    if _mode is not None:
        return
    _mode = CudaToCpu()
    _mode.__enter__()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_device = lambda *a, **k: 0
    _Gen = torch.Generator

    def _generator(device="cpu"):
        return _Gen(device="cpu" if _is_cuda(device) else device)
    torch.Generator = _generator

    def _mem_get_info(*a, **k):
        # "device" memory = what the emulated hipMalloc holds (torch's own host tensors are not part of it, as on the device torch's cache is not part of a leak)
        import ctypes
        import os
        lib = ctypes.CDLL(os.environ["JRC_LIB_PATH"])
        lib.hipcpu_device_bytes_in_use.restype = ctypes.c_ulonglong
        total = 288 << 30
        return (total - int(lib.hipcpu_device_bytes_in_use()), total)
    torch.cuda.mem_get_info = _mem_get_info
