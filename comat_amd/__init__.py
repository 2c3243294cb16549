"""comat_amd — the CoMat optimisation step (training_script.py:556-694) on MI355X: hand-written HIP kernels behind the
C ABI of include/comat_hip.h, with a Python host that mirrors the reference's call conventions."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: a step issues ~13 k small dependent kernels and
# each dispatch otherwise fetches its argument block across PCIe (measured on MI355X: 322 -> 303 ms/step on the same
# box).  Read by the HIP runtime when it initialises, so it must be set before the first GPU call of the process.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
