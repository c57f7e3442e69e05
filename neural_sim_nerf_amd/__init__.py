"""neural-sim-nerf_amd: MI355X-native NeRF volumetric renderer behind the reference's render API.

Drop-in for `utils/run_nerf_noscale.py` + `utils/run_nerf_helpers.py` of gyhandy/Neural-Sim-NeRF:
    from neural_sim_nerf_amd.run_nerf_noscale import create_nerf, render, render_path, render_path_grad
The compute path is the hand-written gfx950 library csrc/libnsr.so (C ABI in include/nsr.h); there is no
CPU or PyTorch fallback: importing the render API without the library raises."""
__version__ = "0.1.0"

import os as _os

# Multi-process GPU work on ROCm hosts that only support dmabuf IPC (this pool's): RCCL and tensor sharing across the ranks of
# render_path / render_path_grad's self-sharding fail in hipIpcGetMemHandle without it.  Defaulted here -- the HIP runtime reads
# it when it initialises, which is after this import in every entry point of the package -- and never overridden.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
