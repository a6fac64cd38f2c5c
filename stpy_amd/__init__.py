"""
stpy_amd -- MI355X (gfx950) implementation of the dense linear-algebra hot path of Mojusko/stpy
behind stpy's own estimator API.  Module layout mirrors the reference for the path it covers:

    stpy_amd.kernels.KernelFunction                              (stpy/kernels.py)
    stpy_amd.continuous_processes.gauss_procc.GaussianProcess   (stpy/continuous_processes/gauss_procc.py)
    stpy_amd.embeddings.embedding.{RFFEmbedding, QuadratureEmbedding, HermiteEmbedding, ...}   (stpy/embeddings/embedding.py)
    stpy_amd.helpers.helper.{interval, cartesian}               (stpy/helpers/helper.py)

All arithmetic runs in hand-written HIP kernels (stpy_amd/csrc) reached through the C ABI in
include/stpy_hip.h; there is no CPU fallback.
"""
from .kernels import KernelFunction
from .continuous_processes.gauss_procc import GaussianProcess
from .embeddings.embedding import Embedding, RFFEmbedding, QuadratureEmbedding, HermiteEmbedding

__all__ = ["KernelFunction", "GaussianProcess", "Embedding", "RFFEmbedding", "QuadratureEmbedding", "HermiteEmbedding"]
