"""ParamsKZG mirror (halo2-experiments_amd/kzg.py): the G2 host arithmetic and the on-disk layout on the CPU;
setup / commit / commit_lagrange / write / read on the GPU against the oracle and the KZG identity."""
import io

import numpy as np
import pytest

from halo2_experiments_amd import kzg
from halo2_experiments_amd.arithmetic import FQ_MODULUS
from halo2_experiments_amd.domain import FR_MODULUS, fr_words


def test_g2_generator_is_on_the_twist_and_has_order_r():
    """alt_bn128 G2 (EIP-197): y^2 = x^3 + 3 / (9 + u) over Fq[u] / (u^2 + 1); r * G2 = identity."""
    p = FQ_MODULUS
    x, y = kzg.G2_GENERATOR
    b = kzg._fq2_mul((3, 0), kzg._fq2_inv((9, 1)))
    lhs = kzg._fq2_mul(y, y)
    x3 = kzg._fq2_mul(kzg._fq2_mul(x, x), x)
    assert lhs == ((x3[0] + b[0]) % p, (x3[1] + b[1]) % p)
    rm1 = kzg.g2_mul(FR_MODULUS - 1)
    assert rm1 == (x, ((-y[0]) % p, (-y[1]) % p))                 # (r - 1) G2 = -G2, hence r G2 = O
    assert kzg._g2_add(rm1, kzg.G2_GENERATOR) is None
    a, c = 0x1234567, 0xFEDCBA9876543210FEDCBA
    assert kzg._g2_add(kzg.g2_mul(a), kzg.g2_mul(c)) == kzg.g2_mul(a + c)
    assert kzg.g2_mul(2) == kzg._g2_add(kzg.G2_GENERATOR, kzg.G2_GENERATOR)
    assert len(kzg.g2_bytes(kzg.G2_GENERATOR)) == 128 and kzg.g2_bytes(None) == bytes(128)
    one_mont = ((1 << 256) % p).to_bytes(32, "little")
    assert kzg._fq_mont_bytes(1) == one_mont


def test_on_disk_layout_round_trip_without_gpu():
    k, n = 3, 8
    rng = np.random.default_rng(5)
    g = rng.integers(0, 1 << 63, size=(n, 8), dtype=np.uint64)
    gl = rng.integers(0, 1 << 63, size=(n, 8), dtype=np.uint64)
    buf = io.BytesIO()
    kzg.ParamsKZG.write_points(buf, k, g, gl, kzg.g2_bytes(kzg.G2_GENERATOR), kzg.g2_bytes(kzg.g2_mul(7)))
    raw = buf.getvalue()
    assert len(raw) == 4 + n * 128 + 256 and raw[:4] == (3).to_bytes(4, "little")
    assert raw[4:4 + 64] == g[0].tobytes() and raw[4 + n * 64: 4 + n * 64 + 64] == gl[0].tobytes()
    assert raw[-256:-128] == kzg.g2_bytes(kzg.G2_GENERATOR)


@pytest.mark.gpu
def test_setup_commit_and_file_round_trip(cref, pyref, tmp_path):
    import torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd.domain import EvaluationDomain
    o = pyref
    k, n = 9, 1 << 9                      # the reference's own test size (merkle_sum_tree.rs:347)
    s = 0x0F1E2D3C4B5A69788796A5B4C3D2E1F0_0123456789ABCDEF % o.R
    params = kzg.ParamsKZG.setup(k, s, keep_points=True)
    try:
        g = params.g_points.cpu().numpy().view(np.uint64)
        assert np.array_equal(g, cref.srs(fr_words(s), n))                           # g[i] = [s^i]G, vs the oracle
        gen = cref.g1_generator()
        w = o.fr_omega(k)
        for i in (0, 1, n - 1):                                                       # g_lagrange[i] = [L_i(s)]G
            wi = pow(w, i, o.R)
            li = (pow(s, n, o.R) - 1) * pow(n, -1, o.R) % o.R * wi % o.R * pow((s - wi) % o.R, -1, o.R) % o.R
            assert np.array_equal(params.g_lagrange_points[i].cpu().numpy().view(np.uint64), cref.g1_mul(fr_words(li), gen)), i
        f = o.rand_scalars(n, 99)
        coeffs = torch.from_numpy(o.fr_array(f).view(np.int64)).cuda()
        evals = coeffs.clone()
        h.best_fft(evals, fr_words(w), k)
        c1, c2 = params.commit(coeffs), params.commit_lagrange(evals)
        exp = cref.g1_mul(fr_words(o.poly_eval(f, s)), gen)
        assert np.array_equal(c1, c2) and np.array_equal(c1[:8], exp)                 # the KZG identity
        assert params.s_g2 == kzg.g2_bytes(kzg.g2_mul(s)) and params.g2 == kzg.g2_bytes(kzg.G2_GENERATOR)
        path = tmp_path / "params.srs"
        with open(path, "wb") as fh:
            params.write(fh)
        with open(path, "rb") as fh:
            loaded = kzg.ParamsKZG.read(fh)
        try:
            assert loaded.k == k and loaded.s_g2 == params.s_g2
            assert np.array_equal(loaded.commit(coeffs), c1) and np.array_equal(loaded.commit_lagrange(evals), c1)
        finally:
            loaded.release()
        with open(path, "rb") as fh:
            bad = fh.read()[:-1]
        with pytest.raises(ValueError):
            kzg.ParamsKZG.read(io.BytesIO(bad))
    finally:
        params.release()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["merkle_sum_tree_k9", "poseidon_k11", "merkle_v3_k17", "merkle_sum_tree_k18"])
def test_replays_verify_their_commitments(name):
    """BASELINE configs[1] / [2] / [3]: the create_proof trace replay checks every commitment it computes against the
    KZG identity (prove, then verify -- merkle_sum_tree.rs:345-358).  Config 4's flow (k = 18: 20 advice columns, 8
    lookups, 12 equality columns, evaluate_h over 2^21 rows with the MerkleSumTree chip's own gate program) included."""
    import torch
    from halo2_experiments_amd.replay import run_replay
    r = run_replay(name, device=torch.device("cuda", 0), include_host_pointer_estimate=False)
    assert r["verified"]["commitments_checked"] >= 3 * (r["calls"]["msm_sparse"] + r["calls"]["msm_dense"])
    assert r["verified"]["distinct_column_base_pairs"] == 6
    from halo2_experiments_amd.circuits import CONSTRAINT_SYSTEMS
    cs = CONSTRAINT_SYSTEMS[name]()
    assert r["shape"]["advice"] == cs.num_advice and r["shape"]["lookups"] == len(cs.lookups) and r["shape"]["max_degree"] == cs.degree()
    assert f"{len(cs.polynomials())} gate polynomials" in r["beyond_msm_ntt"]["evaluate_h"]
    if name.startswith("merkle_sum_tree"):
        # the reference's own proving configuration is k = 9 (test_full_prover, merkle_sum_tree.rs:345-358); same constraint system at
        # k = 18: 20 advice columns, 8 lookups, equality columns in 3 sets, degree 6 (5 quotient pieces) => 55 MSMs and 97 transforms
        # per proof (SURVEY.md §3.2 / §8a estimated 56 / 97)
        assert r["k"] == (9 if name.endswith("k9") else 18) and r["extended_k"] == r["k"] + 3 and r["shape_key"] == name
        assert r["calls"] == {"msm_sparse": 36, "msm_dense": 19, "intt_n": 48, "coset_ntt_ext": 48, "intt_ext": 1}


@pytest.mark.gpu
def test_replay_reports_the_drop_in_totals_side_by_side():
    """The host-pointer (drop-in) cost of the k = 17 trace next to the device-resident one: per-call forms, the batched commitments, and
    the EvaluationDomain edits (hm_coeff_to_extended_bn256_fr / hm_extended_to_coeff_bn256_fr: 9/16 of the bytes of a zero-padded
    best_fft round trip at j = 7) -- each cheaper than the one before."""
    import torch
    from halo2_experiments_amd.replay import run_replay
    r = run_replay("merkle_v3_k17", device=torch.device("cuda", 0))
    t, e = r["total_s"], r["host_pointer_estimate_s"]
    assert set(t) >= {"drop_in_host_pointers", "drop_in_with_batched_commitments", "drop_in_with_domain_edits", "device_resident"}
    assert all(t[key] > 0 for key in ("drop_in_host_pointers", "drop_in_with_batched_commitments", "drop_in_with_domain_edits")), (t, e)
    assert t["drop_in_with_domain_edits"] > t["device_resident"] > 0, (t, e)     # (the order of the three drop-in totals is a timing: not asserted)
    assert all(e[key] > 0 for key in ("coeff_to_extended_each", "extended_to_coeff_each", "ntt_ext_each", "ntt_n_each", "msm_each")), e
    assert e["coeff_to_extended_into_a_fresh_array_each"] > 0 and e["host_zero_padding_each"] > 0
    assert set(t["host_page_faults_on_top"]) == {"best_fft_routes_resize_to_extended_length", "domain_edits_fresh_output_arrays"}


@pytest.mark.gpu
def test_replay_by_cosets_and_one_ranks_share():
    """The extended-domain steps by cosets on one GPU (what several ranks deal among themselves), and ONE rank's share of a
    four-rank replay run alone: its commitments only (each still checked against the KZG identity), its cosets, no exchange."""
    import torch
    from halo2_experiments_amd.replay import run_replay
    dev = torch.device("cuda", 0)
    whole = run_replay("merkle_v3_k17", device=dev, include_host_pointer_estimate=False)
    assert whole["extended_domain"].startswith("whole array") and "share_of" not in whole
    cosets = run_replay("merkle_v3_k17", device=dev, include_host_pointer_estimate=False, by_cosets=True, min_cosets=False)
    from halo2_experiments_amd.circuits import CONSTRAINT_SYSTEMS
    from halo2_experiments_amd.domain import EvaluationDomain
    e = EvaluationDomain(CONSTRAINT_SYSTEMS["merkle_v3_k17"]().degree(), 17).num_cosets()
    assert cosets["extended_domain"].startswith(f"by cosets, {e} of {e}")
    fewer = run_replay("merkle_sum_tree_k18", device=dev, include_host_pointer_estimate=False, by_cosets=True)
    assert fewer["extended_domain"].startswith("by cosets, 5 of 8") and "determine the quotient" in fewer["extended_domain"]
    assert cosets["verified"]["commitments_checked"] == whole["verified"]["commitments_checked"]
    jobs = whole["calls"]["msm_sparse"] + whole["calls"]["msm_dense"]
    seen = 0
    for rank in (0, 3):
        part = run_replay("merkle_v3_k17", device=dev, include_host_pointer_estimate=False, share_of=(rank, 4))
        assert part["share_of"]["rank"] == rank and part["n_gpus"] == 4 and part["extended_domain"].startswith("by cosets")
        assert 0 < part["verified"]["commitments_checked"] < whole["verified"]["commitments_checked"]
        seen += part["verified"]["commitments_checked"]
        if rank == 3:
            assert part["device_resident_s"]["multiopen"] < 1e-4 and part["device_resident_s"]["evaluate_h"] > 0   # rank-0-only steps skipped
    assert seen < 4 * jobs
    with pytest.raises(ValueError):
        run_replay("merkle_v3_k17", device=dev, share_of=(4, 4))
