"""CPU oracle for the BN256 MSM / Fr-NTT path.  TEST INFRASTRUCTURE ONLY (parity unpinned; see bn256_ref.py)."""
