"""neural-sim-nerf_amd: MI355X-native NeRF volumetric renderer behind the reference's render API.

Drop-in for `utils/run_nerf_noscale.py` + `utils/run_nerf_helpers.py` of gyhandy/Neural-Sim-NeRF:
    from neural_sim_nerf_amd.run_nerf_noscale import create_nerf, render, render_path, render_path_grad
The compute path is the hand-written gfx950 library csrc/libnsr.so (C ABI in include/nsr.h); there is no
CPU or PyTorch fallback: importing the render API without the library raises."""
__version__ = "0.1.0"
