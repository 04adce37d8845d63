"""MI355X-native backend for the BN256 MSM / Fr-NTT hot path of the PSE halo2 prover.

Host-side mirror of ``halo2_proofs::arithmetic`` (``best_multiexp``, ``best_fft``, ``eval_polynomial``) and of the
``EvaluationDomain`` / ``ParamsKZG`` steps around them, over the C ABI of ``libhalo2_mi355x.so``.

The sources live in ``halo2-experiments_amd/`` (the directory name the project layout prescribes; not a valid
Python identifier): this package is the importable name, and its ``__path__`` points there, so every submodule
(``_lib``, ``arithmetic``, ``domain``, ``kzg``, ``replay``, ``sharding``) is an ordinary module of this package
with an ordinary ``__spec__`` / ``__file__``.
"""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "halo2-experiments_amd"))

from . import _lib  # noqa: F401
from .arithmetic import (bases_info, batch_invert, best_fft, best_multiexp, best_multiexp_batch, best_multiexp_submit,  # noqa: F401
                         best_multiexp_wait, eval_polynomial, g1_fixed_base_mul, grand_product, grand_product_batch, kate_division,
                         kate_division_batch,
                         linear_combination, msm_stats, permute_expression_pair, permute_expression_pairs, random_fr, register_bases,
                         release_bases)
from .domain import EvaluationDomain  # noqa: F401

__all__ = ["eval_polynomial", "best_multiexp", "best_multiexp_batch", "best_multiexp_submit", "best_multiexp_wait", "best_fft",
           "register_bases", "release_bases", "bases_info", "g1_fixed_base_mul", "msm_stats", "kate_division", "kate_division_batch", "grand_product",
           "grand_product_batch", "batch_invert",
           "linear_combination", "random_fr", "permute_expression_pair", "permute_expression_pairs", "EvaluationDomain"]
