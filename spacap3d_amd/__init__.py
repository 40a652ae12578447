"""spacap3d_amd -- MI355X (gfx950) implementation of the SpaCap3D forward / training hot path.

Sub-modules
-----------
``ext``                 the nine ``pointnet2._ext`` operators on hand-written HIP kernels (C ABI:
                        ``include/spacap_hip.h``, library ``spacap3d_amd/lib/libspacap_hip.so``)
``attention``           fused multi-head attention (``attention()`` of the reference) on the same library
``pointnet2_utils``     autograd Functions / QueryAndGroup with the reference's names and signatures
``pointnet2_modules``   PointnetSAModuleVotes, PointnetFPModule
``models``              backbone, voting, proposal, transformer captioner, SpaCapNet
``loss_helper``         get_scene_cap_loss (device-agnostic restatement)
``synthetic``           seeded synthetic scenes / labels
``distributed``         one-process-per-GPU data parallelism (single flat gradient all-reduce over RCCL)

There is no CPU or PyTorch fallback for the native operators: importing ``ext`` without the built
library raises ImportError, and calling an operator on a CPU tensor raises RuntimeError.
"""
__version__ = "0.1.0"
