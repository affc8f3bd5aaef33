"""iif_amd — MI355X-native training hot path for IIF long-tailed recognition.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed)
calling hand-written gfx950 HIP kernels through the C ABI of
``include/iif_amd.h``.  Module names mirror the reference's
``classification/`` scripts (custom, resnet_pytorch, resnet_cifar, utils,
initialisers, imbalanced_dataset, train) so callers switch by import path.
"""
import os as _os

# Three concurrent HIP streams per rank (main, weight gradients, RCCL): give the runtime enough hardware
# queues that they never share one (effective only if HIP has not started yet).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
