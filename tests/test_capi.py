"""CPU suite: the C-ABI shared library loads, exports every symbol include/halo2_mi355x.h declares,
and fails loudly (never falls back) when no gfx950 device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from halo2_experiments_amd import _lib
import halo2_experiments_amd as h


def declared_symbols():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", text)))


def test_library_is_in_tree_and_loads():
    assert os.path.dirname(_lib.LIB_PATH).endswith(os.path.join("halo2-experiments_amd", "csrc"))
    lib = _lib.load()
    assert b"gfx950" in lib.hm_version()


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib._SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib._SIGNATURES) == set(names)


def test_argument_errors_are_reported_not_thrown():
    lib = _lib.load()
    assert lib.hm_msm_bn256_g1(None, None, 0, None, None) == -1          # HM_ERR_BAD_ARG: null output
    assert b"null" in lib.hm_last_error()
    assert lib.hm_ntt_bn256_fr(None, None, 3) == -1
    assert lib.hm_msm_set_window(99) == -1
    assert lib.hm_msm_set_window(0) == 0
    assert lib.hm_g1_sum(None, 0, None) == -1
    assert lib.hm_set_msm_devices(None, 2) == -1
    assert lib.hm_set_msm_devices(None, -1) == -1
    assert lib.hm_set_msm_devices(None, 0) == 0


def test_no_device_means_error_not_fallback():
    lib = _lib.load()
    if lib.hm_device_count() > 0:
        pytest.skip("a GPU is present")
    s, b = np.zeros((4, 4), dtype=np.uint64), np.zeros((4, 8), dtype=np.uint64)
    with pytest.raises(_lib.Halo2Mi355xError) as e:
        h.best_multiexp(s, b)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    a = np.zeros((8, 4), dtype=np.uint64)
    with pytest.raises(_lib.Halo2Mi355xError):
        h.best_fft(a, np.zeros(4, dtype=np.uint64), 3)
    assert lib.hm_set_device(0) == -2


def test_device_pointer_entry_points_without_a_device():
    """Every device-pointer entry point added after the two functions: null arguments are HM_ERR_BAD_ARG, and with valid-looking
    (never dereferenced) arguments a box without a gfx950 device gets HM_ERR_NO_DEVICE -- never a crash, never a fallback."""
    import ctypes
    lib = _lib.load()
    z = np.zeros(4, dtype=np.uint64)
    zp = z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    assert lib.hm_kate_division_bn256_fr_dev(None, 8, zp, None, None) == -1
    assert lib.hm_fr_grand_product_dev(None, 8, zp, None, None) == -1
    assert lib.hm_kate_division_batch_bn256_fr_dev(None, 8, zp, None, 2, None) == -1
    assert lib.hm_fr_grand_product_batch_dev(None, 8, zp, 3, None, 2, None) == -1
    assert lib.hm_coeff_to_coset_bn256_fr_dev(None, None, 2, zp, 3, zp, 0, None) == -1
    assert lib.hm_coset_to_coeff_bn256_fr_dev(None, 2, zp, 3, zp, zp, None) == -1
    assert lib.hm_coeff_to_cosets_bn256_fr_dev(None, None, 2, zp, 3, zp, 1, 0, None) == -1
    assert lib.hm_coeff_to_cosets_bn256_fr_dev(ctypes.c_void_p(0x1000), ctypes.c_void_p(0x100000), 1, zp, 3, zp, 17, 0, None) == -1      # > 16 cosets
    assert lib.hm_coeff_to_cosets_bn256_fr_dev(ctypes.c_void_p(0x1000), ctypes.c_void_p(0x1080), 1, zp, 3, zp, 2, 0, None) == -1         # overlap
    assert lib.hm_cosets_to_coeff_bn256_fr_dev(None, 2, zp, 3, zp, zp, None) == -1
    assert lib.hm_graph_evaluate_segments_dev(ctypes.c_uint64(1), None, 0, None, 0, 3, 2, None, 0, None) == -1
    one_col = (ctypes.c_void_p * 1)(0x1000)
    assert lib.hm_quotient_by_cosets_bn256_fr_dev(ctypes.c_uint64(1), one_col, None, 1, zp, 1, 3, zp, zp, 1, 1, None, None) == -1       # no output
    assert lib.hm_quotient_partials_bn256_fr_dev(ctypes.c_uint64(1), one_col, None, 1, zp, 1, 3, zp, zp, 17, ctypes.c_void_p(0x9000), None) == -1   # > 16 cosets
    assert lib.hm_quotient_combine_bn256_fr_dev(one_col, zp, 1, 3, 2, ctypes.c_void_p(0x9000), None) == -1                              # pieces > cosets
    assert lib.hm_quotient_combine_bn256_fr_dev(one_col, zp, 1, 3, 1, ctypes.c_void_p(0x9000), None) == -1                              # a zero shift
    nz = np.array([2, 0, 0, 0], dtype=np.uint64).ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    assert lib.hm_quotient_combine_bn256_fr_dev(one_col, nz, 1, 3, 1, ctypes.c_void_p(0x1000), None) == -1                              # h IS the partial
    assert lib.hm_quotient_combine_bn256_fr_dev(one_col, nz, 1, 3, 1, ctypes.c_void_p(0x1000 + 8 * 32 - 32), None) == -1                # h overlaps its tail
    assert b"overlaps a partial" in lib.hm_last_error()
    assert lib.hm_quotient_by_cosets_bn256_fr_dev(ctypes.c_uint64(1), one_col, None, 1, zp, 1, 3, zp, zp, 2, 3, ctypes.c_void_p(0x9000), None) == -1   # pieces > cosets
    assert lib.hm_quotient_by_cosets_bn256_fr_dev(ctypes.c_uint64(1), one_col, None, 1, zp, 1, 3, zp, zp, 1, 1, ctypes.c_void_p(0x9000), None) == -1   # a zero shift
    assert lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(0x1000), ctypes.c_void_p(0x100000), 1, zp, 29, zp, 0, None) == -1      # log_n > 28
    assert lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(0x1000), ctypes.c_void_p(0x1020), 1, zp, 3, zp, 0, None) == -1        # partial overlap
    assert lib.hm_device_malloc(64, None) == -1 and lib.hm_copy_to_device(None, None, 8) == -1 and lib.hm_copy_to_host(None, None, 8) == -1
    assert lib.hm_copy_to_device(None, None, 0) in (0, -2) and lib.hm_set_host_copies(7) == -1 and lib.hm_set_host_copies(0) == 0
    assert lib.hm_fr_batch_invert_dev(None, 8, None) == -1
    assert lib.hm_fr_linear_combination_dev(None, None, 2, 8, None, None) == -1
    assert lib.hm_lookup_permute_bn256_fr_dev(None, None, 8, None, None, None) == -1
    assert lib.hm_lookup_permute_batch_bn256_fr_dev(None, None, 2, 8, None, None, None, None) == -1
    assert lib.hm_msm_batch_bn256_g1_dev(ctypes.c_uint64(1), 0, None, 8, 2, None, None) == -1
    if lib.hm_device_count() > 0:
        return
    fake = ctypes.c_void_p(0x1000)                        # never dereferenced: the device check comes first
    assert lib.hm_kate_division_bn256_fr_dev(fake, 8, zp, ctypes.c_void_p(0x100000), None) == -2
    assert lib.hm_fr_grand_product_dev(fake, 8, zp, fake, None) == -2
    two = (ctypes.c_void_p * 2)(0x1000, 0x2000)
    far = (ctypes.c_void_p * 2)(0x100000, 0x200000)
    zz = np.zeros(8, dtype=np.uint64).ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    assert lib.hm_kate_division_batch_bn256_fr_dev(two, 8, zz, far, 2, None) == -2
    assert lib.hm_fr_grand_product_batch_dev(two, 8, zp, 3, far, 2, None) == -2
    assert lib.hm_coeff_to_coset_bn256_fr_dev(fake, ctypes.c_void_p(0x100000), 1, zp, 3, zp, 0, None) == -2
    assert lib.hm_coset_to_coeff_bn256_fr_dev(fake, 1, zp, 3, zp, zp, None) == -2
    assert lib.hm_coeff_to_cosets_bn256_fr_dev(fake, ctypes.c_void_p(0x100000), 1, zp, 3, zp, 1, 0, None) == -2
    assert lib.hm_cosets_to_coeff_bn256_fr_dev(fake, 1, zp, 3, zp, zp, None) == -2
    two = np.array([2, 0, 0, 0], dtype=np.uint64)
    twop = two.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    one_col = (ctypes.c_void_p * 1)(0x1000)
    assert lib.hm_quotient_by_cosets_bn256_fr_dev(ctypes.c_uint64(1), one_col, None, 1, zp, 1, 3, twop, twop, 1, 1, ctypes.c_void_p(0x9000), None) == -2
    assert lib.hm_quotient_partials_bn256_fr_dev(ctypes.c_uint64(1), one_col, None, 1, zp, 1, 3, twop, twop, 1, ctypes.c_void_p(0x9000), None) == -2
    assert lib.hm_quotient_combine_bn256_fr_dev(one_col, twop, 1, 3, 1, ctypes.c_void_p(0x9000), None) == -2
    pp = ctypes.c_void_p(0x55)
    assert lib.hm_device_malloc(64, ctypes.byref(pp)) == -2 and pp.value is None      # the output is cleared before the device check
    assert lib.hm_device_free(fake) == -2 and lib.hm_device_synchronize() == -2 and lib.hm_copy_to_host(fake, fake, 8) == -2
    assert lib.hm_fr_batch_invert_dev(fake, 8, None) == -2
    assert lib.hm_lookup_permute_bn256_fr_dev(fake, fake, 8, fake, fake, None) == -2
    assert b"no CPU fallback" in lib.hm_last_error()
    st = _lib.Stats()
    assert lib.hm_get_stats(ctypes.byref(st)) == -2


def test_host_mirror_argument_checks():
    s, b = np.zeros((4, 4), dtype=np.uint64), np.zeros((5, 8), dtype=np.uint64)
    with pytest.raises(ValueError, match="coeffs.len"):          # upstream: assert_eq!(coeffs.len(), bases.len())
        h.best_multiexp(s, b)
    with pytest.raises(ValueError, match="1 << log_n"):          # upstream: assert_eq!(a.len(), 1 << log_n)
        h.best_fft(np.zeros((6, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64), 3)
    with pytest.raises(TypeError):
        h.best_fft([[0, 0, 0, 0]], np.zeros(4, dtype=np.uint64), 0)
    with pytest.raises(TypeError):
        h.best_multiexp(np.zeros((4, 4), dtype=np.int32), b)


def test_g1_sum_host_fold(cref, golden):
    """hm_g1_sum (the multi-GPU fold) is pure host code: check it against the oracle here."""
    from halo2_experiments_amd.sharding import g1_sum
    g = golden["msm"]
    s, b = g["n255_uniform_s"], g["n255_uniform_b"]
    parts = np.stack([cref.best_multiexp(s[lo:hi], b[lo:hi], 2) for lo, hi in [(0, 100), (100, 100), (100, 255)]])
    tot = g1_sum(parts)
    assert np.array_equal(tot[:8], g["n255_uniform_r"]) and tot[8:].any()
    pm = np.stack([cref.best_multiexp(g["pmone_s"][:32], g["pmone_b"][:32], 1), cref.best_multiexp(g["pmone_s"][32:], g["pmone_b"][32:], 1)])
    assert not g1_sum(pm).any()                                     # identity
    assert not g1_sum(np.zeros((0, 12), dtype=np.uint64)).any()


def test_evaluation_domain_constants(pyref):
    o = pyref
    from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS, FR_ROOT_OF_UNITY, fr_words
    assert FR_MODULUS == o.R and FR_ROOT_OF_UNITY == o.FR_ROOT_OF_UNITY
    d = EvaluationDomain(j=7, k=9)             # MerkleSumTree-like: max degree 7 -> extended_k = k + 3
    assert d.extended_k == 12 and d.quotient_poly_degree == 6
    assert d.omega == o.fr_omega(9) and d.extended_omega == o.fr_omega(12)
    assert d.omega * d.omega_inv % o.R == 1 and d.ifft_divisor * 512 % o.R == 1
    assert d.g_coset == o.FR_ZETA and d.g_coset * d.g_coset_inv % o.R == 1
    assert EvaluationDomain(j=3, k=4).extended_k == 5 and EvaluationDomain(j=2, k=4).extended_k == 4
    assert fr_words(5).tolist() == o.fr_array([5])[0].tolist()


def _gen():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(_lib.HEADER_PATH)), "..", "tools", "gen_rust_shim.py")
    spec = importlib.util.spec_from_file_location("gen_rust_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rust_bindings_agree_with_the_header_in_arity_and_types():
    """The reference-side binding (/root/reference/Cargo.toml:10 is what gets patched): the header is parsed as C, the
    `extern "C"` blocks of rust/halo2-mi355x-sys/src/lib.rs and of INTEGRATION.md as Rust -- two independent parsers -- and
    every entry must agree in name, return type, arity and the type of each argument (base type, pointer depth, constness
    of every level).  An argument added or re-typed in the header fails here until the files are regenerated."""
    gen = _gen()
    root = os.path.join(os.path.dirname(os.path.abspath(_lib.HEADER_PATH)), "..")
    c_funcs, c_structs, c_defines = gen.parse_header(open(_lib.HEADER_PATH).read())
    assert sorted(f[0] for f in c_funcs) == declared_symbols()
    want = {name: (ret, [t for t, _ in params]) for name, ret, params in c_funcs}
    assert want["hm_msm_bn256_g1"] == ("c_int", ["u64 c", "u64 c", "usize", "u64 m", "c_int m"])          # the parsers themselves
    assert want["hm_lookup_permute_batch_bn256_fr_dev"][1][4] == "c_void m c" and want["hm_last_error"] == ("c_char c", [])
    for path in ("rust/halo2-mi355x-sys/src/lib.rs", "INTEGRATION.md"):
        text = open(os.path.join(root, path)).read()
        r_funcs = gen.parse_rust_extern(text)
        got = {name: (ret, [t for t, _ in params]) for name, ret, params in r_funcs}
        assert set(got) == set(want), f"{path}: entry points differ from the header: {set(got) ^ set(want)}"
        for name in want:
            assert got[name] == want[name], f"{path}: {name} is {got[name]} but the header says {want[name]}"
        for cname, fields in c_structs.items():            # the #[repr(C)] twins: same fields, same order, same types
            m = re.search(r"pub struct %s \{(.*?)\}" % gen.STRUCTS[cname], text, flags=re.S)
            assert m, f"{path}: struct {gen.STRUCTS[cname]} missing"
            rust_fields = re.findall(r"pub (\w+): (\[\w+; \d+\]|\w+),", m.group(1))
            assert rust_fields == [(f, f"[{b}; {c}]" if c else b) for f, b, c in fields], f"{path}: {cname} fields differ"
    # ctypes: the Python binding's arity must agree too
    for name, (ret, params) in want.items():
        assert len(_lib._SIGNATURES[name][1]) == len(params), name
    # hm_stats / hm_msm_stats against the ctypes structures used by the tests
    assert [f for f, _, _ in c_structs["hm_msm_stats"]] == [f for f, _ in _lib.MsmStats._fields_]
    assert [f for f, _, _ in c_structs["hm_stats"]] == [f for f, _ in _lib.Stats._fields_]


def test_generated_rust_files_are_up_to_date():
    """rust/ and the extern block of INTEGRATION.md are outputs of tools/gen_rust_shim.py: regenerate and compare."""
    gen = _gen()
    for path, text in gen.generate().items():
        assert os.path.exists(path), f"{path} missing: run tools/gen_rust_shim.py"
        assert open(path).read() == text, f"{os.path.relpath(path)} is stale: run tools/gen_rust_shim.py"
    patch = open(os.path.join(gen.RUST_DIR, "halo2_proofs.patch")).read()
    assert "original_best_multiexp" in patch and "original_best_fft" in patch and "halo2-mi355x-sys" in patch
    glue = open(os.path.join(gen.RUST_DIR, "halo2_proofs-patch", "src", "mi355x.rs")).read()
    for fn in re.findall(r"sys::(hm_\w+)", glue):         # the glue only calls entry points the header declares
        assert fn in declared_symbols(), fn


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (no C++-isms), and a C program must link against the
    library's entry points -- what cgo / JNI / a Rust `cc` build would do with it."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "halo2_mi355x.h"\n'
                   "#include <stdio.h>\n"
                   "int main(void) {\n"
                   "  hm_stats st; hm_msm_stats ms; uint64_t out[12] = {0}; (void)st; (void)ms;\n"
                   "  if (hm_g1_sum(0, 0, out) != HM_OK) return 2;            /* host-only entry: the empty sum is the identity */\n"
                   "  if (hm_msm_set_window(99) != HM_ERR_BAD_ARG) return 3;\n"
                   '  printf("%s devices=%d flags=%d\\n", hm_version(), hm_device_count(), HM_GRAPH_COLUMNS_INTERNAL);\n'
                   "  return out[8] == 0 ? 0 : 4;\n"
                   "}\n")
    inc = os.path.dirname(os.path.abspath(_lib.HEADER_PATH))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe), "-L", _lib.CSRC,
                    "-lhalo2_mi355x", f"-Wl,-rpath,{_lib.CSRC}"], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "gfx950" in r.stdout, r.stdout + r.stderr
