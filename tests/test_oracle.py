"""CPU suite: the oracle against public known answers, against itself, and against tests/golden."""
import numpy as np
import pytest

THREADS = (1, 3, 8)


def test_constants_match_survey(pyref):
    o = pyref
    # SURVEY.md §8a constants of halo2curves bn256 (restated there), re-derived from the moduli
    assert o.to_limbs(o.MONT % o.R) == [0xAC96341C4FFFFFFB, 0x36FC76959F60CD29, 0x666EA36F7879462E, 0x0E0A77C19A07DF2F]
    assert o.to_limbs(o.MONT * o.MONT % o.R) == [0x1BB8E645AE216DA7, 0x53FE3AB1E35C59E3, 0x8C49833D53BB8085, 0x0216D0B17F4E44A5]
    assert o.to_limbs(o.MONT % o.P) == [0xD35D438DC58F0D9D, 0x0A78EB28F5C70B3D, 0x666EA36F7879462C, 0x0E0A77C19A07DF2F]
    assert o.to_limbs(o.MONT * o.MONT % o.P) == [0xF32CFC5B538AFA89, 0xB5E71911D44501FB, 0x47AB1EFF0A417FF6, 0x06D89F71CAB8351F]
    assert o.FR_INV64 == 0xC2E1F593EFFFFFFF and o.FQ_INV64 == 0x87D20782E4866389
    assert o.FR_ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pow(o.FR_ROOT_OF_UNITY, 1 << 28, o.R) == 1 and pow(o.FR_ROOT_OF_UNITY, 1 << 27, o.R) != 1
    assert pow(o.FR_ZETA, 3, o.R) == 1 and o.FR_ZETA != 1
    # halo2curves bn256 Fr: MULTIPLICATIVE_GENERATOR = 7, ROOT_OF_UNITY = 7^((r - 1) / 2^28), DELTA = 7^(2^28) (the
    # permutation argument's column separator), TWO_INV = (r + 1) / 2 -- the published constants, re-derived
    assert pow(7, (o.R - 1) >> 28, o.R) == o.FR_ROOT_OF_UNITY
    assert pow(7, 1 << 28, o.R) == 0x09226b6e22c6f0ca64ec26aad4c86e715b5f898e5e963f25870e56bbe533e9a2
    assert (o.R + 1) // 2 == 0x183227397098d014dc2822db40c0ac2e9419f4243cdcb848a1f0fac9f8000001


def test_public_alt_bn128_known_answers(pyref):
    """EIP-196 (alt_bn128 = BN256 G1) published vectors: the only known answers for this curve that
    do not come from our own code."""
    o = pyref
    two_g = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
             9918110051302171585080402603319702774565515993150576347155970296011118125764)
    three_g = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
               19321533766552368860946552437480515441416830039777911637913418824951667761761)
    assert o.is_on_curve(o.G1_GEN) and o.is_on_curve(two_g) and o.is_on_curve(three_g)
    assert o.g1_add(o.G1_GEN, o.G1_GEN) == two_g
    assert o.g1_add(two_g, o.G1_GEN) == three_g
    assert o.g1_mul(3, o.G1_GEN) == three_g
    assert o.g1_mul(o.R, o.G1_GEN) is None                      # the group order annihilates G
    assert o.g1_add(two_g, o.g1_neg(two_g)) is None


from known_answers import GETH_ADD_CHFAST1, GETH_MUL_CHFAST  # noqa: E402


def test_go_ethereum_precompile_vectors(pyref, cref):
    """More third-party known answers for this curve: the "chfast1..3" cases of go-ethereum's precompile test data for
    EIP-196 (core/vm/testdata/precompiles/bn256Add.json, bn256ScalarMul.json) -- a generic point addition and three
    scalar multiplications (61-bit, above the group order, 253-bit), against both oracles."""
    o = pyref
    a, b, s = GETH_ADD_CHFAST1
    assert o.is_on_curve(a) and o.is_on_curve(b) and o.g1_add(a, b) == s and o.g1_add(b, a) == s
    for pt, k, res in GETH_MUL_CHFAST:
        assert o.is_on_curve(pt) and o.g1_mul(k, pt) == res
        # the C oracle, through its MSM of one point (scalars are field elements: reduce the one above r)
        got = cref.g1_to_affine(cref.best_multiexp(o.fr_array([k % o.R]), o.g1_affine_array([pt]), 1))[0]
        assert o.g1_affine_from_array(got.reshape(1, 8))[0] == res
    got = cref.g1_to_affine(cref.best_multiexp(o.fr_array([1, 1]), o.g1_affine_array([a, b]), 2))[0]
    assert o.g1_affine_from_array(got.reshape(1, 8))[0] == s


def test_python_restatement_vs_definition(pyref):
    o = pyref
    for n, kind in [(1, "uniform"), (5, "edge"), (33, "uniform"), (100, "prover")]:
        s = o.rand_scalars(n, n, kind)
        pts, logs = o.arith_bases(n, n + 1)
        exp = o.g1_mul(sum(a * b for a, b in zip(s, logs)) % o.R, o.G1_GEN)
        assert o.msm_naive(s, pts) == exp
        for t in (1, 3, 8):
            assert o.best_multiexp(s, pts, t) == exp
    for k in range(1, 8):
        w = o.fr_omega(k)
        v = o.rand_scalars(1 << k, k)
        x = list(v)
        o.best_fft(x, w, k)
        assert x == o.dft_naive(v, w)


def test_c_field_ops_vs_golden(cref, golden):
    g = golden["field"]
    assert np.array_equal(cref.fr_mul(g["fr_a"], g["fr_b"]), g["fr_mul"])
    assert np.array_equal(cref.fr_add(g["fr_a"], g["fr_b"]), g["fr_add"])
    assert np.array_equal(cref.fr_sub(g["fr_a"], g["fr_b"]), g["fr_sub"])
    assert np.array_equal(cref.fq_mul(g["fq_a"], g["fq_b"]), g["fq_mul"])
    assert np.array_equal(cref.fr_from_mont(g["fr_a"]), g["fr_canon"])
    assert np.array_equal(cref.fr_to_mont(g["fr_canon"]), g["fr_a"])


def test_c_curve_ops_vs_golden(cref, golden):
    g = golden["curve"]
    gen = cref.g1_generator()
    for k, pt in zip(g["scalars"], g["points"]):
        assert np.array_equal(cref.g1_mul(k, gen), pt)
    for a, b, s in zip(g["add_a"], g["add_b"], g["add_sum"]):
        assert np.array_equal(cref.g1_add_affine(a, b), s)


@pytest.mark.parametrize("threads", THREADS)
def test_c_best_multiexp_vs_golden(cref, golden, threads):
    g = golden["msm"]
    for name in g["names"]:
        got = cref.g1_to_affine(cref.best_multiexp(g[f"{name}_s"], g[f"{name}_b"], threads))[0]
        assert np.array_equal(got, g[f"{name}_r"]), name


def test_c_naive_msm_vs_golden(cref, golden):
    g = golden["msm"]
    for name in ("n3_uniform", "n33_edge", "pm", "ident"):
        got = cref.g1_to_affine(cref.msm_naive(g[f"{name}_s"], g[f"{name}_b"]))[0]
        assert np.array_equal(got, g[f"{name}_r"]), name


@pytest.mark.parametrize("threads", THREADS)
def test_c_best_fft_vs_golden(cref, golden, threads):
    g = golden["ntt"]
    for k in range(0, 11):
        assert np.array_equal(cref.best_fft(g[f"k{k}_in"], g[f"k{k}_omega"], k, threads), g[f"k{k}_out"]), k
    assert np.array_equal(cref.best_fft(g["delta_in"], g["k5_omega"], 5, threads), g["delta_out"])
    assert np.array_equal(cref.best_fft(g["ones_in"], g["k5_omega"], 5, threads), g["ones_out"])
    assert np.array_equal(cref.best_fft(g["w3_in"], g["w3_omega"], 5, threads), g["w3_out"])


def test_c_fft_inverse_round_trip(cref, golden):
    g = golden["ntt"]
    for k in (3, 7, 10):
        fwd = cref.best_fft(g[f"k{k}_in"], g[f"k{k}_omega"], k, 4)
        back = cref.best_fft(fwd, g[f"k{k}_omega_inv"], k, 4)
        scaled = cref.fr_mul(back, np.tile(g[f"k{k}_ninv"], (1 << k, 1)))
        assert np.array_equal(scaled, g[f"k{k}_in"])


def test_c_mid_size_known_answer(cref, pyref):
    """n = 4096 against the discrete-log construction (no brute force possible at this size)."""
    o = pyref
    n = 4096
    s = o.rand_scalars(n, 99)
    smont = o.fr_array(s)
    x = o.fr_array([123456789])[0]
    srs = cref.srs(x, n)                                    # g_i = [x^i] G
    fx = o.poly_eval(s, 123456789)
    exp = cref.g1_mul(o.fr_array([fx])[0], cref.g1_generator())
    got = cref.g1_to_affine(cref.best_multiexp(smont, srs, 8))[0]
    assert np.array_equal(got, exp)
    assert np.array_equal(cref.fr_horner(smont, x), o.fr_array([fx])[0])


def test_golden_regenerates(pyref, golden):
    """The committed fixtures are what tests/golden/gen_golden.py produces (spot check, fast subset)."""
    o = pyref
    g = golden["ntt"]
    for k in (1, 4, 6):
        v = o.fr_from_array(g[f"k{k}_in"])
        assert o.fr_array(o.dft_naive(v, o.fr_omega(k))).tolist() == g[f"k{k}_out"].tolist()
    m = golden["msm"]
    s, b = o.fr_from_array(m["n3_uniform_s"]), o.g1_affine_from_array(m["n3_uniform_b"])
    assert o.g1_affine_array([o.msm_naive(s, b)])[0].tolist() == m["n3_uniform_r"].tolist()
