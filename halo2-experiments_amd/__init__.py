"""MI355X-native backend for the BN256 MSM / Fr-NTT hot path of the PSE halo2 prover.

Host-side mirror of ``halo2_proofs::arithmetic`` (``best_multiexp``, ``best_fft``) and of the
``EvaluationDomain`` steps around them, over the C ABI of ``libhalo2_mi355x.so``.
"""
from . import _lib  # noqa: F401
from .arithmetic import (best_fft, best_multiexp, best_multiexp_submit, best_multiexp_wait, eval_polynomial,  # noqa: F401
                         g1_fixed_base_mul, msm_stats, register_bases, release_bases)
from .domain import EvaluationDomain  # noqa: F401

__all__ = ["eval_polynomial", "best_multiexp", "best_multiexp_submit", "best_multiexp_wait", "best_fft", "register_bases", "release_bases", "g1_fixed_base_mul", "msm_stats",
           "EvaluationDomain"]
