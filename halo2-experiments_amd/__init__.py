"""Sources of the ``halo2_experiments_amd`` package (import it under that name: ``halo2_experiments_amd/__init__.py``
extends its ``__path__`` to this directory).  ``csrc/`` holds the HIP kernels and the C ABI, ``cpp/`` the C++ mirror
of the reference's interface, ``examples/`` the full_prover counterpart."""
