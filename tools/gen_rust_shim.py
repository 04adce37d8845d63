#!/usr/bin/env python3
"""Generate the Rust side of the drop-in boundary from include/halo2_mi355x.h, so that it cannot drift from the header.

    python tools/gen_rust_shim.py            # (re)write rust/ and the extern block of INTEGRATION.md
    python tools/gen_rust_shim.py --check    # exit 1 if any generated file differs from what is committed

What the reference patches: /root/reference/Cargo.toml:10 pins halo2_proofs (git tag v2023_02_02); the two free functions
of its arithmetic.rs are the boundary.  Written out here:

    rust/halo2-mi355x-sys/{Cargo.toml, build.rs, src/lib.rs}   the FFI crate: one `extern "C"` item per header entry,
                                                               #[repr(C)] twins of the two stats structs, the HM_* constants
    rust/halo2_proofs-patch/src/mi355x.rs                      the glue module added to the patched halo2_proofs
    rust/halo2_proofs.patch                                    unified diff for the halo2_proofs checkout: the new module,
                                                               zero-context hunks on the two signatures and on [dependencies]
    rust/README.md                                             the recipe

There is no Rust toolchain in the build image: none of this has been compiled here.  tests/test_capi.py parses the header
(C) and the generated extern block (Rust) with two independent parsers and compares names, arity and every argument type.
"""
from __future__ import annotations

import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "halo2_mi355x.h")
RUST_DIR = os.path.join(ROOT, "rust")
INTEGRATION = os.path.join(ROOT, "INTEGRATION.md")
BEGIN_MARK = "<!-- BEGIN GENERATED: extern block (tools/gen_rust_shim.py) -->"
END_MARK = "<!-- END GENERATED -->"

# ---- C side ---------------------------------------------------------------------------------------------------------
C_SCALARS = {"int": "c_int", "long": "c_long", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32",
             "uint8_t": "u8", "double": "f64", "char": "c_char", "void": "c_void"}
STRUCTS = {"hm_msm_stats": "HmMsmStats", "hm_stats": "HmStats", "hm_bases_info": "HmBasesInfo"}


def strip_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_c_type(decl: str):
    """'const uint64_t* const* name[12]' -> (canonical type, name).  Canonical: base type followed by one 'c' (pointer to
    const) or 'm' (pointer to mutable) per pointer level, outermost last: 'u64 c' = *const u64, 'c_void c c' = *const *const c_void."""
    decl = decl.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*\w*\s*\])?$", decl, flags=re.S)
    if not m:
        raise ValueError(f"cannot parse parameter {decl!r}")
    ty, name, arr = m.group(1).strip(), m.group(2), m.group(3)
    if ty == "" or ty == "const":              # unnamed parameter such as 'void'
        ty, name = decl, ""
    toks = re.findall(r"\w+|\*", ty)
    base = [t for t in toks if t not in ("const", "*", "struct")]
    if len(base) != 1:
        raise ValueError(f"cannot parse type {ty!r}")
    b = base[0]
    rust_base = C_SCALARS.get(b) or STRUCTS.get(b)
    if rust_base is None:
        raise ValueError(f"unknown C type {b!r}")
    # constness of each level: a 'const' binds to what is on its left, or to the base type when it comes first
    levels = []                                # constness of [base, after 1st *, after 2nd * ...]
    cur_const = False
    seen_base = False
    for t in toks:
        if t == "const":
            cur_const = True
        elif t == "*":
            levels.append(cur_const)
            cur_const = False
        elif t != "struct":
            seen_base = True
    levels.append(cur_const)                   # constness of the outermost object (the parameter itself): irrelevant
    ptrs = [("c" if levels[i] else "m") for i in range(len(levels) - 1)]
    if arr:                                    # T name[N] decays to T*: pointee constness = constness of the element level
        ptrs.append("c" if levels[-1] else "m")
    assert seen_base
    return " ".join([rust_base] + ptrs), name


def parse_header(text: str):
    """-> (functions [(name, ret canonical, [(canonical type, name)])], structs {c name: [(field, canonical, count)]}, defines)"""
    clean = strip_comments(text)
    defines = [(m.group(1), m.group(2)) for m in re.finditer(r"#define\s+(HM_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", clean)]
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", clean, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            ty = re.match(r"^(\w+)\s+(.*)$", stmt, flags=re.S)
            base = C_SCALARS[ty.group(1)]
            for item in ty.group(2).split(","):
                im = re.match(r"^\s*(\w+)\s*(?:\[\s*(\d+)\s*\])?\s*$", item)
                fields.append((im.group(1), base, int(im.group(2)) if im.group(2) else 0))
        structs[m.group(3)] = fields
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", clean, flags=re.S)
    body = re.sub(r"#.*", " ", body)
    functions = []
    for m in re.finditer(r"([\w\s\*]+?)\b(hm_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", body, flags=re.S):
        ret = " ".join(m.group(1).replace("extern", " ").replace('"C"', " ").split())
        ret_c, _ = parse_c_type(ret + " _r")
        params = []
        plist = m.group(3).strip()
        if plist and plist != "void":
            for p in plist.split(","):
                params.append(parse_c_type(p))
        functions.append((m.group(2), ret_c, params))
    return functions, structs, defines


def rust_type(canon: str) -> str:
    parts = canon.split()
    out = parts[0]
    for p in parts[1:]:
        out = ("*const " if p == "c" else "*mut ") + out
    return out


# ---- Rust side (an independent parser: used by the tests on the generated text and on INTEGRATION.md) ---------------------
def parse_rust_type(t: str) -> str:
    t = t.strip()
    ptrs = []
    while t.startswith("*"):
        m = re.match(r"^\*(const|mut)\s+(.*)$", t)
        ptrs.append("c" if m.group(1) == "const" else "m")
        t = m.group(2).strip()
    return " ".join([t] + ptrs[::-1])


def parse_rust_extern(text: str):
    """The `extern "C" { ... }` block(s) of a Rust source -> [(name, ret canonical, [(canonical, name)])]"""
    out = []
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        body = re.sub(r"//[^\n]*", "", blk.group(1))
        for m in re.finditer(r"pub\s+fn\s+(hm_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", body, flags=re.S):
            params = []
            for p in m.group(2).split(","):
                p = p.strip()
                if not p:
                    continue
                name, ty = p.split(":", 1)
                params.append((parse_rust_type(ty), name.strip()))
            out.append((m.group(1), parse_rust_type(m.group(3)) if m.group(3) else "()", params))
    return out


# ---- emitters -------------------------------------------------------------------------------------------------------
GROUPS = [   # (comment, prefix tests) -- only cosmetic: the order of the extern block
    ("lifecycle", ("hm_set_device", "hm_device_count", "hm_shutdown", "hm_last_error", "hm_version")),
]


def emit_extern_block(functions) -> str:
    lines = ['extern "C" {']
    for name, ret, params in functions:
        args = ", ".join(f"{pn or 'arg' + str(i)}: {rust_type(pt)}" for i, (pt, pn) in enumerate(params))
        line = f"    pub fn {name}({args}) -> {rust_type(ret)};"
        if len(line) > 150:                     # wrap long prototypes once, at a comma near the middle
            cut = line.rfind(", ", 0, 140)
            line = line[:cut + 1] + "\n        " + line[cut + 2:]
        lines.append(line)
    lines.append("}")
    return "\n".join(lines)


def emit_structs(structs) -> str:
    out = []
    for cname, fields in structs.items():
        out.append("#[repr(C)]\n#[derive(Clone, Copy, Debug)]\npub struct %s {" % STRUCTS[cname])
        for fname, base, count in fields:
            out.append(f"    pub {fname}: " + (f"[{base}; {count}]," if count else f"{base},"))
        out.append("}\n")
    return "\n".join(out)


def emit_lib_rs(functions, structs, defines) -> str:
    consts = "\n".join(f"pub const {n}: c_int = {v};" for n, v in defines)
    return f'''//! halo2-mi355x-sys -- raw bindings of libhalo2_mi355x.so (include/halo2_mi355x.h), the MI355X backend of
//! halo2_proofs::arithmetic::{{best_multiexp, best_fft}} for bn256.
//!
//! GENERATED by tools/gen_rust_shim.py from the header: do not edit; re-run the script when the header changes
//! (tests/test_capi.py fails when this file and the header disagree on a name, an arity or an argument type).
//! Not compiled in the repository's build image (no Rust toolchain there).
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{{c_char, c_int, c_long, c_void}};

{consts}

{emit_structs(structs)}
{emit_extern_block(functions)}

/// The calling thread's last error message (hm_last_error), as an owned String.
pub fn last_error() -> String {{
    unsafe {{ CStr::from_ptr(hm_last_error()) }}.to_string_lossy().into_owned()
}}

#[allow(dead_code)]
fn _unused(_: c_long, _: *const c_char, _: *mut c_void) {{}}
'''


CARGO_TOML = '''# GENERATED by tools/gen_rust_shim.py
[package]
name = "halo2-mi355x-sys"
version = "0.3.0"
edition = "2021"
description = "Raw FFI bindings of libhalo2_mi355x.so: MI355X (gfx950) BN256 MSM / Fr-NTT backend for halo2_proofs"
links = "halo2_mi355x"
build = "build.rs"

[lib]
name = "halo2_mi355x_sys"
'''

BUILD_RS = '''// GENERATED by tools/gen_rust_shim.py
// HALO2_MI355X_LIB_DIR = <this repository>/halo2-experiments_amd/csrc (where `make` leaves libhalo2_mi355x.so)
fn main() {
    let dir = std::env::var("HALO2_MI355X_LIB_DIR")
        .expect("set HALO2_MI355X_LIB_DIR to <repo>/halo2-experiments_amd/csrc (the directory of libhalo2_mi355x.so)");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=halo2_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=HALO2_MI355X_LIB_DIR");
}
'''

# Smallest sizes the glue sends to the GPU: below them the CPU body of the patched functions is at least as fast as the
# host-pointer round trip.  MEASURED (tools/crossover.py on the MI355X box against oracle/cpu_ref.c on its 16 granted cores;
# table in DESIGN.md section 8 and profiles/r04_crossover.json) -- regenerate rust/ after changing them.
GPU_MIN_LOG_N_MSM = 8       # 0.30 ms against 1.4 ms (16 threads) / 5.4 ms (1 thread) at 2^8, the smallest size measured
GPU_MIN_LOG_N_NTT = 10      # 0.057 ms against 0.111 ms on one thread at 2^10; 2^9 is a tie, 2^8 loses

MI355X_RS = '''//! mi355x.rs -- glue between halo2_proofs::arithmetic and libhalo2_mi355x.so (added by rust/halo2_proofs.patch).
//! GENERATED by tools/gen_rust_shim.py (a fixed template: the C entry points it calls are checked against the header
//! by the repository's tests).  The two functions return None / false whenever the GPU path does not apply or fails
//! WITHOUT having touched the caller's arrays, and the caller falls through to the untouched upstream body: other
//! curves / fields, tiny inputs, no device, any error code but HM_ERR_PARTIAL_OUTPUT (which panics: see try_best_fft).
//! Needs nothing newer than Rust 1.56: std::sync::Once + AtomicBool for the one-time check (the pinned tag predates the 1.70 cell types).
// halo2_proofs' lib.rs denies these crate-wide (as recalled: #![deny(missing_docs)], #![deny(missing_debug_implementations)],
// #![deny(unsafe_code)]); this module is FFI glue -- the allowance is scoped to it.
#![allow(unsafe_code, missing_docs, missing_debug_implementations, clippy::all)]
use std::any::TypeId;
use std::sync::atomic::{AtomicBool, Ordering};
use std::sync::Once;

use group::prime::PrimeCurveAffine; // G1Affine::generator()
use group::Group as _; // G1::identity()
use halo2_mi355x_sys as sys;
use halo2curves::bn256::{Fr, G1Affine, G1};
use halo2curves::CurveAffine;

use crate::arithmetic::Group;

/// Below these sizes the host-pointer round trip (launch + PCIe latency) is no faster than the CPU body: measured
/// crossover against a 16-core host (DESIGN.md section 8), stay on the CPU.
pub const GPU_MIN_LOG_N_MSM: u32 = @MSM@;
pub const GPU_MIN_LOG_N_NTT: u32 = @NTT@;

fn layout_check() -> bool {
    if std::mem::size_of::<Fr>() != 32 || std::mem::size_of::<G1Affine>() != 64 || std::mem::size_of::<G1>() != 96 {
        return false;
    }
    // the generator (1, 2) must read back as x = R mod p, y = 2R mod p (Montgomery words, x before y)
    let g: [u64; 8] = unsafe { std::mem::transmute(<G1Affine as PrimeCurveAffine>::generator()) };
    g[..4] == [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
        && g[4..] == [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
        && unsafe { sys::hm_device_count() } > 0
}

/// halo2curves gives these types no #[repr(C)]; the byte layout the library reads is asserted instead (once).
pub fn layout_ok() -> bool {
    static INIT: Once = Once::new();
    static OK: AtomicBool = AtomicBool::new(false);
    INIT.call_once(|| OK.store(layout_check(), Ordering::Release));
    OK.load(Ordering::Acquire)
}

/// (x, y, 1) Montgomery words, or all zero for the identity: exactly bn256::G1 { x, y, z }.
pub fn g1_from_words(xyz: [u64; 12]) -> G1 {
    if xyz[8..].iter().all(|w| *w == 0) {
        <G1 as group::Group>::identity()
    } else {
        unsafe { std::mem::transmute::<[u64; 12], G1>(xyz) }
    }
}

pub fn try_best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> Option<C::Curve> {
    if TypeId::of::<C>() != TypeId::of::<G1Affine>() || coeffs.len() < (1 << GPU_MIN_LOG_N_MSM) || !layout_ok() {
        return None;
    }
    let mut xyz = [0u64; 12];
    let rc = unsafe {
        sys::hm_msm_bn256_g1_jacobian(coeffs.as_ptr() as *const u64, bases.as_ptr() as *const u64, coeffs.len(), xyz.as_mut_ptr())
    };
    if rc != sys::HM_OK {
        return None; // error policy of the boundary: fall back to the CPU body (the inputs are untouched)
    }
    let p = g1_from_words(xyz);
    Some(unsafe { std::mem::transmute_copy::<G1, C::Curve>(&p) }) // C::Curve == G1 here (checked by the TypeId test above)
}

pub fn try_best_fft<G: Group>(a: &mut [G], omega: &G::Scalar, log_n: u32) -> bool {
    if TypeId::of::<G>() != TypeId::of::<Fr>() || log_n < GPU_MIN_LOG_N_NTT || log_n > 28 || !layout_ok() {
        return false;
    }
    let rc = unsafe { sys::hm_ntt_bn256_fr(a.as_mut_ptr() as *mut u64, omega as *const _ as *const u64, log_n) };
    if rc == sys::HM_ERR_PARTIAL_OUTPUT {
        // the copy of the result back into `a` failed half-way: `a` is neither the input nor the output any more, so the
        // CPU body must NOT run on it (it would continue the proof with a silently wrong polynomial)
        panic!("hm_ntt_bn256_fr: {}", sys::last_error());
    }
    rc == sys::HM_OK // every other error code leaves `a` exactly as it was: the caller runs the CPU body
}

/// R mod r: bn256::Fr::one() as the four Montgomery words the library reads (spelled out: no dependence on the ff version's `one()` / `ONE`).
pub(crate) const FR_ONE: [u64; 4] = [0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f];

/// The same as an `Fr` (Montgomery words are the in-memory form: layout_ok() has checked 32 bytes).
#[allow(dead_code)]
pub(crate) fn fr_one() -> Fr {
    unsafe { std::mem::transmute_copy::<[u64; 4], Fr>(&FR_ONE) }
}

fn fr_words<S>(s: &S) -> [u64; 4] {
    unsafe { std::mem::transmute_copy::<S, [u64; 4]>(s) } // S == Fr here (TypeId-checked by the callers; 32 bytes by layout_ok)
}

/// EvaluationDomain::coeff_to_extended in one library call (hm_coeff_to_extended_bn256_fr): `a` holds the 2^k coefficients on
/// entry and, when this returns true, the 2^extended_k evaluations on the zeta-coset.  Only the 2^k coefficients cross PCIe
/// upwards (never the zero padding `resize` would append); distribute_powers_zeta is fused into the first NTT pass.  The result
/// is written to a FRESH Vec that replaces `a` only on success, so every failure -- HM_ERR_PARTIAL_OUTPUT included -- leaves
/// `a` exactly as it was and the caller runs the CPU body.  (The fresh allocation's first-touch page faults replace the ones upstream's own
/// `resize` to the extended length takes; the library touches the pages from helper threads while the transform runs.)
pub fn try_coeff_to_extended<G: Group>(a: &mut Vec<G>, extended_omega: &G::Scalar, k: u32, extended_k: u32, g_coset: &G::Scalar,
                                       g_coset_inv: &G::Scalar) -> bool {
    if TypeId::of::<G>() != TypeId::of::<Fr>() || TypeId::of::<G::Scalar>() != TypeId::of::<Fr>() || extended_k < GPU_MIN_LOG_N_NTT
        || extended_k > 28 || k > extended_k || a.len() != (1usize << k) || !layout_ok()
    {
        return false;
    }
    // a[i] *= [1, g_coset, g_coset_inv][i % 3]: distribute_powers_zeta(a, true) (zeta^3 = 1, so g_coset_inv = zeta^2)
    let mut coset = [0u64; 12];
    coset[..4].copy_from_slice(&FR_ONE);
    coset[4..8].copy_from_slice(&fr_words(g_coset));
    coset[8..].copy_from_slice(&fr_words(g_coset_inv));
    let len = 1usize << extended_k;
    let mut ext: Vec<G> = Vec::with_capacity(len);
    let rc = unsafe {
        sys::hm_coeff_to_extended_bn256_fr(a.as_ptr() as *const u64, ext.as_mut_ptr() as *mut u64, extended_omega as *const _ as *const u64,
                                           k, extended_k, coset.as_ptr())
    };
    if rc != sys::HM_OK {
        return false; // `ext` (possibly half-written) is dropped; `a` is untouched
    }
    unsafe { ext.set_len(len) }; // every element was written by the library's final copy
    *a = ext;
    true
}

/// EvaluationDomain::extended_to_coeff in one library call (hm_extended_to_coeff_bn256_fr): in place on the 2^extended_k
/// evaluations; the ifft divisor and distribute_powers_zeta(a, false) ride on the last NTT pass, only the `keep` coefficients
/// upstream keeps come back over PCIe, and `a` is truncated to them.
pub fn try_extended_to_coeff<G: Group>(a: &mut Vec<G>, extended_omega_inv: &G::Scalar, extended_k: u32, divisor: &G::Scalar,
                                       g_coset: &G::Scalar, g_coset_inv: &G::Scalar, keep: usize) -> bool {
    if TypeId::of::<G>() != TypeId::of::<Fr>() || TypeId::of::<G::Scalar>() != TypeId::of::<Fr>() || extended_k < GPU_MIN_LOG_N_NTT
        || extended_k > 28 || a.len() != (1usize << extended_k) || keep > a.len() || !layout_ok()
    {
        return false;
    }
    // a[i] *= [1, g_coset_inv, g_coset][i % 3]: distribute_powers_zeta(a, false)
    let mut coset_inv = [0u64; 12];
    coset_inv[..4].copy_from_slice(&FR_ONE);
    coset_inv[4..8].copy_from_slice(&fr_words(g_coset_inv));
    coset_inv[8..].copy_from_slice(&fr_words(g_coset));
    let rc = unsafe {
        sys::hm_extended_to_coeff_bn256_fr(a.as_mut_ptr() as *mut u64, keep, extended_omega_inv as *const _ as *const u64, extended_k,
                                           divisor as *const _ as *const u64, coset_inv.as_ptr())
    };
    if rc == sys::HM_ERR_PARTIAL_OUTPUT {
        panic!("hm_extended_to_coeff_bn256_fr: {}", sys::last_error()); // as in try_best_fft: `a` is neither input nor output
    }
    if rc != sys::HM_OK {
        return false;
    }
    a.truncate(keep);
    true
}

/// One process, several GPUs: every later best_multiexp / registered base set is split over `devices` inside the library.
pub fn use_devices(devices: &[i32]) -> bool {
    unsafe { sys::hm_set_msm_devices(devices.as_ptr(), devices.len() as i32) == sys::HM_OK }
}
'''.replace("@MSM@", str(GPU_MIN_LOG_N_MSM)).replace("@NTT@", str(GPU_MIN_LOG_N_NTT))

MI355X_KZG_RS = '''//! mi355x_kzg.rs -- ParamsKZG's side of the binding (added by rust/halo2_proofs.patch / rust/apply_edits.py next to
//! mi355x.rs, declared in arithmetic.rs).  GENERATED by tools/gen_rust_shim.py.
//!
//! create_proof commits ~56 times per MerkleSumTree proof (/root/reference/src/circuits/utils.rs:40-48), always against
//! the same two arrays, params.g and params.g_lagrange.  The free-function drop-in (mi355x.rs) re-reads the whole base
//! array on every call to key its cache; here every ParamsKZG registers its two arrays ONCE (hm_register_bases: converted,
//! resident in HBM, from 2^17 points with the fixed-base table) and commits through the handle, and a whole phase of
//! commitments goes to the library in one call (hm_msm_batch_bn256_g1_h: eight in flight, uploads behind kernels).
//!
//! `SrsHandles` is a FIELD of ParamsKZG (`gpu`), so a handle can never outlive or alias the arrays it was made from:
//! every constructor creates it empty (Default), Clone creates a fresh empty one (the clone owns new Vecs), Drop releases
//! both sets, and a handle is used only while the Vec it came from still starts at the same address and is at least as
//! long as when it was registered (`downsize` truncates g in place -- a prefix, still valid -- and REPLACES g_lagrange,
//! whose new buffer fails that test and is registered afresh).
// halo2_proofs' lib.rs denies these crate-wide (as recalled: #![deny(missing_docs)], #![deny(missing_debug_implementations)],
// #![deny(unsafe_code)]); this module is FFI glue -- the allowance is scoped to it.
#![allow(unsafe_code, missing_docs, missing_debug_implementations, clippy::all)]
use std::any::TypeId;
use std::fmt;
use std::sync::atomic::{AtomicU64, AtomicUsize, Ordering};
use std::sync::Mutex;

use halo2_mi355x_sys as sys;
use halo2curves::bn256::{G1Affine, G1};
use halo2curves::CurveAffine;

use crate::arithmetic::mi355x::{g1_from_words, layout_ok, GPU_MIN_LOG_N_MSM};

const UNSET: u64 = 0; // the library's handles start at 1
const FAILED: u64 = u64::MAX; // registration failed once: do not try again for this array

struct Slot {
    handle: AtomicU64,
    ptr: AtomicUsize,
    len: AtomicUsize,
}

impl Slot {
    const fn new() -> Self {
        Slot { handle: AtomicU64::new(UNSET), ptr: AtomicUsize::new(0), len: AtomicUsize::new(0) }
    }
    fn release(&self) {
        let h = self.handle.swap(UNSET, Ordering::AcqRel);
        if h != UNSET && h != FAILED {
            unsafe { sys::hm_release_bases(h) };
        }
    }
}

pub struct SrsHandles {
    g: Slot,
    g_lagrange: Slot,
    lock: Mutex<()>, // registration happens once per array, under this lock
}

impl Default for SrsHandles {
    fn default() -> Self {
        SrsHandles { g: Slot::new(), g_lagrange: Slot::new(), lock: Mutex::new(()) }
    }
}
impl Clone for SrsHandles {
    fn clone(&self) -> Self {
        Self::default() // the clone of a ParamsKZG owns new Vecs: it registers them itself on first use
    }
}
impl fmt::Debug for SrsHandles {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "SrsHandles {{ g: {}, g_lagrange: {} }}", self.g.handle.load(Ordering::Relaxed), self.g_lagrange.handle.load(Ordering::Relaxed))
    }
}
impl Drop for SrsHandles {
    fn drop(&mut self) {
        self.g.release();
        self.g_lagrange.release();
    }
}

impl SrsHandles {
    /// Drop both device sets (ParamsKZG::downsize replaces g_lagrange and shortens g): registered afresh on demand.
    pub fn reset(&self) {
        let _turn = self.lock.lock().unwrap_or_else(|e| e.into_inner());
        self.g.release();
        self.g_lagrange.release();
    }

    /// The handle of `bases` (registering it on first use), or None: not bn256, too small, no device, registration failed.
    fn handle_for(&self, slot: &Slot, bases: &[G1Affine], n: usize) -> Option<u64> {
        if n < (1 << GPU_MIN_LOG_N_MSM) || n > bases.len() || !layout_ok() {
            return None;
        }
        let same_array = |s: &Slot| s.ptr.load(Ordering::Acquire) == bases.as_ptr() as usize && s.len.load(Ordering::Acquire) >= n;
        let h = slot.handle.load(Ordering::Acquire);
        if h != UNSET && same_array(slot) {
            return if h == FAILED { None } else { Some(h) };
        }
        let _turn = self.lock.lock().unwrap_or_else(|e| e.into_inner());
        let h = slot.handle.load(Ordering::Acquire);
        if h != UNSET && same_array(slot) {
            return if h == FAILED { None } else { Some(h) };
        }
        slot.release(); // another array (downsize replaced it): the old set goes
        let mut out = 0u64;
        let rc = unsafe { sys::hm_register_bases(bases.as_ptr() as *const u64, bases.len(), &mut out) };
        slot.ptr.store(bases.as_ptr() as usize, Ordering::Release);
        slot.len.store(bases.len(), Ordering::Release);
        slot.handle.store(if rc == sys::HM_OK { out } else { FAILED }, Ordering::Release);
        if rc == sys::HM_OK { Some(out) } else { None }
    }

    fn msm<C: CurveAffine>(&self, slot: &Slot, scalars: &[C::Scalar], bases: &[C]) -> Option<C::Curve> {
        if TypeId::of::<C>() != TypeId::of::<G1Affine>() {
            return None;
        }
        let bases: &[G1Affine] = unsafe { std::slice::from_raw_parts(bases.as_ptr() as *const G1Affine, bases.len()) };
        let h = self.handle_for(slot, bases, scalars.len())?;
        let (mut xy, mut is_id) = ([0u64; 8], 0i32);
        let rc = unsafe { sys::hm_msm_bn256_g1_h(h, 0, scalars.as_ptr() as *const u64, scalars.len(), xy.as_mut_ptr(), &mut is_id) };
        if rc != sys::HM_OK {
            return None; // the caller runs best_multiexp (pointer form, then the CPU body)
        }
        let mut xyz = [0u64; 12];
        if is_id == 0 {
            xyz[..8].copy_from_slice(&xy);
            xyz[8..].copy_from_slice(&[0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]); // z = 1 (R mod p)
        }
        let p: G1 = g1_from_words(xyz);
        Some(unsafe { std::mem::transmute_copy::<G1, C::Curve>(&p) })
    }

    /// ParamsKZG::commit: scalars against g[..scalars.len()].
    pub fn commit<C: CurveAffine>(&self, scalars: &[C::Scalar], g: &[C]) -> Option<C::Curve> {
        self.msm::<C>(&self.g, scalars, g)
    }
    /// ParamsKZG::commit_lagrange: scalars against g_lagrange[..scalars.len()].
    pub fn commit_lagrange<C: CurveAffine>(&self, scalars: &[C::Scalar], g_lagrange: &[C]) -> Option<C::Curve> {
        self.msm::<C>(&self.g_lagrange, scalars, g_lagrange)
    }

    /// The commitments of one prover phase (columns of one length) in ONE library call; None = fall back to one by one.
    pub fn commit_lagrange_batch<C: CurveAffine>(&self, columns: &[&[C::Scalar]], g_lagrange: &[C]) -> Option<Vec<C::Curve>> {
        if TypeId::of::<C>() != TypeId::of::<G1Affine>() || columns.is_empty() {
            return None;
        }
        let n = columns[0].len();
        if columns.iter().any(|c| c.len() != n) {
            return None;
        }
        let bases: &[G1Affine] = unsafe { std::slice::from_raw_parts(g_lagrange.as_ptr() as *const G1Affine, g_lagrange.len()) };
        let h = self.handle_for(&self.g_lagrange, bases, n)?;
        let ptrs: Vec<*const u64> = columns.iter().map(|c| c.as_ptr() as *const u64).collect();
        let mut out = vec![0u64; 12 * columns.len()];
        let rc = unsafe { sys::hm_msm_batch_bn256_g1_h(h, 0, ptrs.as_ptr(), n, columns.len(), out.as_mut_ptr()) };
        if rc != sys::HM_OK {
            return None;
        }
        Some(
            out.chunks_exact(12)
                .map(|w| {
                    let mut xyz = [0u64; 12];
                    xyz.copy_from_slice(w);
                    let p: G1 = g1_from_words(xyz);
                    unsafe { std::mem::transmute_copy::<G1, C::Curve>(&p) }
                })
                .collect(),
        )
    }
}
'''



MI355X_DEV_RS = '''//! mi355x_dev.rs -- polynomials that STAY in HBM between the steps of create_proof (added by rust/apply_edits.py next to mi355x.rs,
//! declared in arithmetic.rs).  GENERATED by tools/gen_rust_shim.py.
//!
//! The drop-in edits (mi355x.rs, mi355x_kzg.rs) leave every polynomial in a host Vec, so each best_fft / coeff_to_extended moves its
//! array over PCIe twice: 122-292 ms of the k = 18 proof against 34 ms of device time (INTEGRATION.md section 3).  This module is the
//! other half of the boundary as code instead of prose: `DevicePoly` (RAII over hm_device_malloc), `DeviceDomain` (the
//! EvaluationDomain steps on device-resident arrays), `commit_dev` / `commit_batch_dev`, `eval_polynomial_dev`, and
//! `QuotientProgram::quotient_by_cosets` -- evaluate_h + divide_by_vanishing_poly + extended_to_coeff in ONE call from coefficient
//! arrays.  A prover adopts it step by step: upload advice columns once (`DevicePoly::from_slice`), commit from the device, transform
//! on the device, bring back only what the transcript needs.  Nothing here is reached by the drop-in edits; nothing here has met rustc.
//! Every function returns None / false on any error (sys::last_error() has the message) and never panics.
// halo2_proofs' lib.rs denies these crate-wide (as recalled: #![deny(missing_docs)], #![deny(missing_debug_implementations)],
// #![deny(unsafe_code)]); this module is FFI glue -- the allowance is scoped to it.
#![allow(unsafe_code, missing_docs, missing_debug_implementations, clippy::all)]
use std::os::raw::c_void;
use std::ptr;

use ff::Field;
use halo2_mi355x_sys as sys;
use halo2curves::bn256::{Fr, G1};

use crate::arithmetic::mi355x::{fr_one, g1_from_words, layout_ok, FR_ONE};

fn words(x: &Fr) -> [u64; 4] {
    unsafe { std::mem::transmute_copy::<Fr, [u64; 4]>(x) } // 32 bytes, Montgomery words (layout_ok() asserts the layout once)
}

/// `len` field elements in device memory; freed on drop (hipFree waits for the device: keep buffers for the life of a proof).
#[derive(Debug)]
pub struct DevicePoly {
    ptr: *mut c_void,
    len: usize,
}
unsafe impl Send for DevicePoly {}

impl DevicePoly {
    pub fn new(len: usize) -> Option<Self> {
        if !layout_ok() {
            return None;
        }
        let mut p: *mut c_void = ptr::null_mut();
        if unsafe { sys::hm_device_malloc(len * 32, &mut p) } != sys::HM_OK {
            return None;
        }
        Some(DevicePoly { ptr: p, len })
    }
    /// Upload (through the library's copy policy, hm_set_host_copies).
    pub fn from_slice(a: &[Fr]) -> Option<Self> {
        let d = Self::new(a.len())?;
        if unsafe { sys::hm_copy_to_device(d.ptr, a.as_ptr() as *const c_void, a.len() * 32) } != sys::HM_OK {
            return None;
        }
        Some(d)
    }
    /// Download after waiting for the device (the steps below are asynchronous on the default stream).
    pub fn to_vec(&self) -> Option<Vec<Fr>> {
        if unsafe { sys::hm_device_synchronize() } != sys::HM_OK {
            return None;
        }
        let mut v: Vec<Fr> = Vec::with_capacity(self.len);
        if unsafe { sys::hm_copy_to_host(v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * 32) } != sys::HM_OK {
            return None;
        }
        unsafe { v.set_len(self.len) }; // every element was written by the copy
        Some(v)
    }
    /// Upload `a` to elements [first, first + a.len()) of this array: how a prover fills ONE packed array (the column table of a
    /// `QuotientProgram`, polynomial i at [i n, (i + 1) n)) column by column as synthesis produces them.
    pub fn upload_at(&mut self, first: usize, a: &[Fr]) -> bool {
        if first.checked_add(a.len()).map_or(true, |end| end > self.len) {
            return false;
        }
        let dst = unsafe { (self.ptr as *mut u8).add(first * 32) } as *mut c_void;
        unsafe { sys::hm_copy_to_device(dst, a.as_ptr() as *const c_void, a.len() * 32) == sys::HM_OK }
    }
    /// Upload several arrays in ONE call (hm_copy_many_to_device): `arrays[i]` to elements [firsts[i], firsts[i] + arrays[i].len()).
    /// The copy lanes move them as one transfer: the columns of a proof at the rate of one large array.
    pub fn upload_many_at(&mut self, firsts: &[usize], arrays: &[&[Fr]]) -> bool {
        if firsts.len() != arrays.len() {
            return false;
        }
        let mut dsts: Vec<*mut c_void> = Vec::with_capacity(arrays.len());
        let mut srcs: Vec<*const c_void> = Vec::with_capacity(arrays.len());
        let mut bytes: Vec<usize> = Vec::with_capacity(arrays.len());
        for (&first, a) in firsts.iter().zip(arrays.iter()) {
            if first.checked_add(a.len()).map_or(true, |end| end > self.len) {
                return false;
            }
            dsts.push(unsafe { (self.ptr as *mut u8).add(first * 32) } as *mut c_void);
            srcs.push(a.as_ptr() as *const c_void);
            bytes.push(a.len() * 32);
        }
        unsafe { sys::hm_copy_many_to_device(dsts.as_ptr(), srcs.as_ptr(), bytes.as_ptr(), bytes.len()) == sys::HM_OK }
    }
    /// Download several ranges (first element, length) in ONE call after waiting for the device (hm_copy_many_to_host).
    pub fn to_vecs_ranges(&self, ranges: &[(usize, usize)]) -> Option<Vec<Vec<Fr>>> {
        if ranges.iter().any(|&(first, len)| first.checked_add(len).map_or(true, |end| end > self.len)) || unsafe { sys::hm_device_synchronize() } != sys::HM_OK {
            return None;
        }
        let mut out: Vec<Vec<Fr>> = ranges.iter().map(|&(_, len)| Vec::with_capacity(len)).collect();
        let dsts: Vec<*mut c_void> = out.iter_mut().map(|v| v.as_mut_ptr() as *mut c_void).collect();
        let srcs: Vec<*const c_void> = ranges.iter().map(|&(first, _)| unsafe { (self.ptr as *const u8).add(first * 32) } as *const c_void).collect();
        let bytes: Vec<usize> = ranges.iter().map(|&(_, len)| len * 32).collect();
        if unsafe { sys::hm_copy_many_to_host(dsts.as_ptr(), srcs.as_ptr(), bytes.as_ptr(), bytes.len()) } != sys::HM_OK {
            return None;
        }
        for (v, &(_, len)) in out.iter_mut().zip(ranges.iter()) {
            unsafe { v.set_len(len) }; // every element was written by the copy
        }
        Some(out)
    }
    /// Download elements [first, first + len) after waiting for the device.
    pub fn to_vec_range(&self, first: usize, len: usize) -> Option<Vec<Fr>> {
        if first.checked_add(len).map_or(true, |end| end > self.len) || unsafe { sys::hm_device_synchronize() } != sys::HM_OK {
            return None;
        }
        let src = unsafe { (self.ptr as *const u8).add(first * 32) } as *const c_void;
        let mut v: Vec<Fr> = Vec::with_capacity(len);
        if unsafe { sys::hm_copy_to_host(v.as_mut_ptr() as *mut c_void, src, len * 32) } != sys::HM_OK {
            return None;
        }
        unsafe { v.set_len(len) }; // every element was written by the copy
        Some(v)
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    pub fn as_ptr(&self) -> *const c_void {
        self.ptr as *const c_void
    }
    pub fn as_mut_ptr(&mut self) -> *mut c_void {
        self.ptr
    }
}
impl Drop for DevicePoly {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { sys::hm_device_free(self.ptr) };
        }
    }
}

/// The constants of an `EvaluationDomain<Fr>` the device steps need (all of them fields of poly/domain.rs), copied out once.
#[derive(Debug)]
pub struct DeviceDomain {
    pub k: u32,
    pub extended_k: u32,
    pub quotient_poly_degree: usize,
    pub omega: Fr,
    pub omega_inv: Fr,
    pub extended_omega: Fr,
    pub extended_omega_inv: Fr,
    pub g_coset: Fr,
    pub g_coset_inv: Fr,
    pub ifft_divisor: Fr,
    pub extended_ifft_divisor: Fr,
}

impl DeviceDomain {
    pub fn n(&self) -> usize {
        1usize << self.k
    }
    pub fn extended_len(&self) -> usize {
        1usize << self.extended_k
    }
    fn coset_words(&self, scale: Fr) -> [u64; 12] {
        // [1, g_coset, g_coset_inv] * scale: distribute_powers_zeta(a, true) (zeta^3 = 1, so g_coset_inv = zeta^2)
        let mut c = [0u64; 12];
        c[..4].copy_from_slice(&words(&scale));
        c[4..8].copy_from_slice(&words(&(self.g_coset * scale)));
        c[8..].copy_from_slice(&words(&(self.g_coset_inv * scale)));
        c
    }
    /// EvaluationDomain::lagrange_to_coeff on `a.len() / n` back-to-back polynomials, in place: one set of launches.
    pub fn lagrange_to_coeff(&self, a: &mut DevicePoly) -> bool {
        if a.len() == 0 || a.len() % self.n() != 0 {
            return false;
        }
        let (w, d) = (words(&self.omega_inv), words(&self.ifft_divisor));
        unsafe { sys::hm_ntt_batch_bn256_fr_dev(a.as_mut_ptr(), a.len() / self.n(), w.as_ptr(), self.k, d.as_ptr(), ptr::null(), ptr::null_mut()) == sys::HM_OK }
    }
    /// The same on polynomials [first, first + count) of a packed array (polynomial i at [i n, (i + 1) n)).
    pub fn lagrange_to_coeff_range(&self, a: &mut DevicePoly, first: usize, count: usize) -> bool {
        if count == 0 || first.checked_add(count).map_or(true, |end| end * self.n() > a.len()) {
            return false;
        }
        let (w, d) = (words(&self.omega_inv), words(&self.ifft_divisor));
        let p = unsafe { (a.as_mut_ptr() as *mut u8).add(first * self.n() * 32) } as *mut c_void;
        unsafe { sys::hm_ntt_batch_bn256_fr_dev(p, count, w.as_ptr(), self.k, d.as_ptr(), ptr::null(), ptr::null_mut()) == sys::HM_OK }
    }
    /// EvaluationDomain::coeff_to_extended on `a.len() / n` polynomials -> a new array of 2^extended_k evaluations each (the zero
    /// padding is never materialised).  `internal`: the evaluations come out multiplied by 32, the form `QuotientProgram` and
    /// hm_graph_evaluate_flags_dev(HM_GRAPH_COLUMNS_INTERNAL) load without a conversion product.
    pub fn coeff_to_extended(&self, a: &DevicePoly, internal: bool) -> Option<DevicePoly> {
        if a.len() == 0 || a.len() % self.n() != 0 {
            return None;
        }
        let batch = a.len() / self.n();
        let mut ext = DevicePoly::new(batch * self.extended_len())?;
        let coset = self.coset_words(if internal { Fr::from(32u64) } else { fr_one() });
        let w = words(&self.extended_omega);
        let rc = unsafe {
            sys::hm_coeff_to_extended_bn256_fr_dev(a.as_ptr(), ext.as_mut_ptr(), batch, w.as_ptr(), self.k, self.extended_k, coset.as_ptr(), ptr::null_mut())
        };
        if rc == sys::HM_OK { Some(ext) } else { None }
    }
    /// EvaluationDomain::extended_to_coeff in place on `a.len() / 2^extended_k` arrays; the caller reads the first
    /// n * quotient_poly_degree coefficients of each.
    pub fn extended_to_coeff(&self, a: &mut DevicePoly) -> bool {
        if a.len() == 0 || a.len() % self.extended_len() != 0 {
            return false;
        }
        let mut c = [0u64; 12]; // [1, g_coset_inv, g_coset]: distribute_powers_zeta(a, false)
        c[..4].copy_from_slice(&FR_ONE);
        c[4..8].copy_from_slice(&words(&self.g_coset_inv));
        c[8..].copy_from_slice(&words(&self.g_coset));
        let (w, d) = (words(&self.extended_omega_inv), words(&self.extended_ifft_divisor));
        unsafe {
            sys::hm_extended_to_coeff_bn256_fr_dev(a.as_mut_ptr(), a.len() / self.extended_len(), w.as_ptr(), self.extended_k, d.as_ptr(), c.as_ptr(), ptr::null_mut())
                == sys::HM_OK
        }
    }
    /// zeta * extended_omega^j: row E t + j of the extended array is the value at coset_shift(j) * omega^t.
    pub fn coset_shift(&self, j: usize) -> Fr {
        self.g_coset * self.extended_omega.pow_vartime([j as u64])
    }
}

/// ParamsKZG::commit / commit_lagrange with the scalars already on the device (`handle`: hm_register_bases, as SrsHandles keeps it).
pub fn commit_dev(handle: u64, scalars: &DevicePoly) -> Option<G1> {
    let mut xyz = [0u64; 12];
    let rc = unsafe { sys::hm_msm_bn256_g1_dev(handle, 0, scalars.as_ptr(), scalars.len(), ptr::null_mut(), xyz.as_mut_ptr()) };
    if rc != sys::HM_OK {
        return None;
    }
    Some(g1_from_words(xyz))
}

/// A phase of commitments (columns of one length) in one call: eight in flight, dense columns sharing launch chains.
pub fn commit_batch_dev(handle: u64, columns: &[&DevicePoly]) -> Option<Vec<G1>> {
    if columns.is_empty() {
        return Some(Vec::new());
    }
    let n = columns[0].len();
    if columns.iter().any(|c| c.len() != n) {
        return None;
    }
    let ptrs: Vec<*const c_void> = columns.iter().map(|c| c.as_ptr()).collect();
    let mut out = vec![0u64; 12 * columns.len()];
    let rc = unsafe { sys::hm_msm_batch_bn256_g1_dev(handle, 0, ptrs.as_ptr(), n, columns.len(), ptr::null_mut(), out.as_mut_ptr()) };
    if rc != sys::HM_OK {
        return None;
    }
    Some(
        out.chunks_exact(12)
            .map(|w| {
                let mut xyz = [0u64; 12];
                xyz.copy_from_slice(w);
                g1_from_words(xyz)
            })
            .collect(),
    )
}

/// The same for polynomials [first, first + count) of `n` scalars each lying back to back in `polys` (a packed column table, the
/// pieces of h as `QuotientProgram::quotient_by_cosets` returns them): one call, nothing copied.
pub fn commit_pieces_dev(handle: u64, polys: &DevicePoly, n: usize, first: usize, count: usize) -> Option<Vec<G1>> {
    if count == 0 {
        return Some(Vec::new());
    }
    if n == 0 || first.checked_add(count).map_or(true, |end| end.checked_mul(n).map_or(true, |e| e > polys.len())) {
        return None;
    }
    let base = polys.as_ptr() as *const u8;
    let ptrs: Vec<*const c_void> = (first..first + count).map(|i| unsafe { base.add(i * n * 32) } as *const c_void).collect();
    let mut out = vec![0u64; 12 * count];
    let rc = unsafe { sys::hm_msm_batch_bn256_g1_dev(handle, 0, ptrs.as_ptr(), n, count, ptr::null_mut(), out.as_mut_ptr()) };
    if rc != sys::HM_OK {
        return None;
    }
    Some(
        out.chunks_exact(12)
            .map(|w| {
                let mut xyz = [0u64; 12];
                xyz.copy_from_slice(w);
                g1_from_words(xyz)
            })
            .collect(),
    )
}

/// The same for ANY selection of the polynomials of a packed array (`indices[i]`: polynomial at [indices[i] n, (indices[i] + 1) n)):
/// every per-proof column of the table in one call, whatever lies between them.
pub fn commit_indexed_dev(handle: u64, polys: &DevicePoly, n: usize, indices: &[usize]) -> Option<Vec<G1>> {
    if indices.is_empty() {
        return Some(Vec::new());
    }
    if n == 0 || indices.iter().any(|&i| i.checked_add(1).map_or(true, |e| e.checked_mul(n).map_or(true, |e| e > polys.len()))) {
        return None;
    }
    let base = polys.as_ptr() as *const u8;
    let ptrs: Vec<*const c_void> = indices.iter().map(|&i| unsafe { base.add(i * n * 32) } as *const c_void).collect();
    let mut out = vec![0u64; 12 * indices.len()];
    let rc = unsafe { sys::hm_msm_batch_bn256_g1_dev(handle, 0, ptrs.as_ptr(), n, indices.len(), ptr::null_mut(), out.as_mut_ptr()) };
    if rc != sys::HM_OK {
        return None;
    }
    Some(
        out.chunks_exact(12)
            .map(|w| {
                let mut xyz = [0u64; 12];
                xyz.copy_from_slice(w);
                g1_from_words(xyz)
            })
            .collect(),
    )
}

/// halo2_proofs::arithmetic::eval_polynomial for `points.len()` polynomials of `n` coefficients lying back to back in `polys`
/// (polynomial q at point q): the Horner evaluations create_proof makes of every committed polynomial.
pub fn eval_polynomial_dev(polys: &DevicePoly, n: usize, points: &[Fr]) -> Option<Vec<Fr>> {
    if n == 0 || polys.len() < n * points.len() {
        return None;
    }
    let mut out: Vec<Fr> = Vec::with_capacity(points.len());
    let rc = unsafe {
        sys::hm_eval_polynomial_bn256_fr_dev(polys.as_ptr(), n, ptr::null(), points.as_ptr() as *const u64, points.len(), out.as_mut_ptr() as *mut u64, ptr::null_mut())
    };
    if rc != sys::HM_OK {
        return None;
    }
    unsafe { out.set_len(points.len()) };
    Some(out)
}

/// The UNDIVIDED numerator of h(X) -- custom gates, permutation and lookup terms combined by y -- as a straight-line program on
/// the device (hm_graph_create: five words per calculation, include/halo2_mi355x.h; halo2-experiments_amd/evaluation.py lowers
/// upstream's GraphEvaluator to it).  Built once per proving key.
#[derive(Debug)]
pub struct QuotientProgram {
    handle: u64,
    n_columns: usize,
    n_dynamic: usize,
}

impl QuotientProgram {
    pub fn new(calcs: &[[u32; 5]], constants: &[Fr], n_dynamic: usize, rotations: &[i32], n_columns: usize, n_intermediates: u32) -> Option<Self> {
        if !layout_ok() {
            return None;
        }
        let mut h = 0u64;
        let rc = unsafe {
            sys::hm_graph_create(calcs.as_ptr() as *const u32, calcs.len(), constants.as_ptr() as *const u64, constants.len(), n_dynamic,
                                 rotations.as_ptr(), rotations.len(), n_columns, n_intermediates, &mut h)
        };
        if rc != sys::HM_OK {
            return None;
        }
        Some(QuotientProgram { handle: h, n_columns, n_dynamic })
    }
    /// evaluate_h + divide_by_vanishing_poly + extended_to_coeff in ONE call: `columns[i]` = the n coefficients of entry i of the
    /// program's column table; `dynamic` = this proof's challenges, then beta, gamma, theta, y; `cosets` = indices of the cosets
    /// of the extended domain to evaluate on -- `domain.quotient_poly_degree` of them determine the quotient of a satisfied
    /// circuit (5 of 8 for the reference's circuits).  -> cosets.len() * n coefficients of h (piece t at [t n, (t + 1) n)).
    pub fn quotient_by_cosets(&self, domain: &DeviceDomain, columns: &[&DevicePoly], dynamic: &[Fr], cosets: &[usize]) -> Option<DevicePoly> {
        let n = domain.n();
        if columns.len() != self.n_columns || dynamic.len() != self.n_dynamic || cosets.is_empty() || columns.iter().any(|c| c.len() != n) {
            return None;
        }
        let ptrs: Vec<*const c_void> = columns.iter().map(|c| c.as_ptr()).collect();
        let mut shifts: Vec<u64> = Vec::with_capacity(4 * cosets.len());
        for &j in cosets {
            shifts.extend_from_slice(&words(&domain.coset_shift(j)));
        }
        let mut h = DevicePoly::new(cosets.len() * n)?;
        let w = words(&domain.omega);
        let rc = unsafe {
            sys::hm_quotient_by_cosets_bn256_fr_dev(self.handle, ptrs.as_ptr(), ptr::null(), ptrs.len(), dynamic.as_ptr() as *const u64, dynamic.len(), domain.k,
                                                    w.as_ptr(), shifts.as_ptr(), cosets.len(), cosets.len(), h.as_mut_ptr(), ptr::null_mut())
        };
        if rc == sys::HM_OK { Some(h) } else { None }
    }
    /// The same with the whole column table in ONE packed array (entry i at [i n, (i + 1) n)): what a prover that fills the table
    /// with `DevicePoly::upload_at` and transforms it with `lagrange_to_coeff_range` holds.
    pub fn quotient_by_cosets_packed(&self, domain: &DeviceDomain, table: &DevicePoly, dynamic: &[Fr], cosets: &[usize]) -> Option<DevicePoly> {
        let n = domain.n();
        if table.len() != self.n_columns * n || dynamic.len() != self.n_dynamic || cosets.is_empty() {
            return None;
        }
        let base = table.as_ptr() as *const u8;
        let ptrs: Vec<*const c_void> = (0..self.n_columns).map(|i| unsafe { base.add(i * n * 32) } as *const c_void).collect();
        let mut shifts: Vec<u64> = Vec::with_capacity(4 * cosets.len());
        for &j in cosets {
            shifts.extend_from_slice(&words(&domain.coset_shift(j)));
        }
        let mut h = DevicePoly::new(cosets.len() * n)?;
        let w = words(&domain.omega);
        let rc = unsafe {
            sys::hm_quotient_by_cosets_bn256_fr_dev(self.handle, ptrs.as_ptr(), ptr::null(), ptrs.len(), dynamic.as_ptr() as *const u64, dynamic.len(), domain.k,
                                                    w.as_ptr(), shifts.as_ptr(), cosets.len(), cosets.len(), h.as_mut_ptr(), ptr::null_mut())
        };
        if rc == sys::HM_OK { Some(h) } else { None }
    }
    /// Once per proving key: the values of the table entries `which` (the key's constant columns -- fixed, sigmas, l_0 / l_last /
    /// l_active, X -- whose coefficients lie in `table`) on `cosets`, in the form the program loads without a conversion product.
    /// -> which.len() x cosets.len() x n elements (entry w on coset c at [(w * cosets.len() + c) n, ...)), for
    /// `quotient_by_cosets_packed_kept`: those columns are then not transformed again in every proof (35 of 83 for MerkleSumTree).
    pub fn keep_on_cosets(&self, domain: &DeviceDomain, table: &DevicePoly, which: &[usize], cosets: &[usize]) -> Option<DevicePoly> {
        let n = domain.n();
        if table.len() != self.n_columns * n || cosets.is_empty() || cosets.len() > 16 || which.iter().any(|&i| i >= self.n_columns) {
            return None;
        }
        let mut shifts: Vec<u64> = Vec::with_capacity(4 * cosets.len());
        for &j in cosets {
            shifts.extend_from_slice(&words(&domain.coset_shift(j)));
        }
        let mut kept = DevicePoly::new(which.len() * cosets.len() * n)?;
        let w = words(&domain.omega);
        for (slot, &i) in which.iter().enumerate() {
            let src = unsafe { (table.as_ptr() as *const u8).add(i * n * 32) } as *const c_void;
            let dst = unsafe { (kept.as_mut_ptr() as *mut u8).add(slot * cosets.len() * n * 32) } as *mut c_void;
            if unsafe { sys::hm_coeff_to_cosets_bn256_fr_dev(src, dst, 1, w.as_ptr(), domain.k, shifts.as_ptr(), cosets.len(), 1, ptr::null_mut()) } != sys::HM_OK {
                return None;
            }
        }
        Some(kept)
    }
    /// `quotient_by_cosets_packed` with the columns `which` read from `kept` (what `keep_on_cosets` returned for the same `which` and
    /// `cosets`) instead of being transformed from their coefficients.
    pub fn quotient_by_cosets_packed_kept(&self, domain: &DeviceDomain, table: &DevicePoly, kept: &DevicePoly, which: &[usize], dynamic: &[Fr],
                                          cosets: &[usize]) -> Option<DevicePoly> {
        let n = domain.n();
        if table.len() != self.n_columns * n || dynamic.len() != self.n_dynamic || cosets.is_empty() || kept.len() != which.len() * cosets.len() * n
            || which.iter().any(|&i| i >= self.n_columns)
        {
            return None;
        }
        let base = table.as_ptr() as *const u8;
        let ptrs: Vec<*const c_void> = (0..self.n_columns).map(|i| unsafe { base.add(i * n * 32) } as *const c_void).collect();
        let mut pre: Vec<*const c_void> = vec![ptr::null(); self.n_columns];
        for (slot, &i) in which.iter().enumerate() {
            pre[i] = unsafe { (kept.as_ptr() as *const u8).add(slot * cosets.len() * n * 32) } as *const c_void;
        }
        let mut shifts: Vec<u64> = Vec::with_capacity(4 * cosets.len());
        for &j in cosets {
            shifts.extend_from_slice(&words(&domain.coset_shift(j)));
        }
        let mut h = DevicePoly::new(cosets.len() * n)?;
        let w = words(&domain.omega);
        let rc = unsafe {
            sys::hm_quotient_by_cosets_bn256_fr_dev(self.handle, ptrs.as_ptr(), pre.as_ptr(), ptrs.len(), dynamic.as_ptr() as *const u64, dynamic.len(), domain.k,
                                                    w.as_ptr(), shifts.as_ptr(), cosets.len(), cosets.len(), h.as_mut_ptr(), ptr::null_mut())
        };
        if rc == sys::HM_OK { Some(h) } else { None }
    }
}
impl Drop for QuotientProgram {
    fn drop(&mut self) {
        unsafe { sys::hm_graph_destroy(self.handle) };
    }
}
'''

# The edits of EXISTING upstream files, one table for both deliverables: the zero-context hunks of halo2_proofs.patch and
# the anchors of rust/apply_edits.py (which finds them as literal lines, is idempotent, and says what it did).
#   (file, anchor line as recalled from the tag, [replacement lines for the 1st, 2nd ... occurrence], approximate line)
# Every occurrence of an anchor in its file must be covered: apply_edits.py refuses a file where the count differs.
_COMMIT_VIA = ["        if let Some(r) = self.gpu.{fn}::<E::G1Affine>(&scalars, &bases[..]) {{", "            return r;", "        }}",
               "        best_multiexp(&scalars, &bases[0..size])"]
_LITERAL = ["            s_g2,", "            gpu: Default::default(),"]
EDITS = [
    ("src/arithmetic.rs", "pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {",
     [['#[path = "mi355x.rs"]', "pub mod mi355x;", '#[path = "mi355x_kzg.rs"]', "pub mod mi355x_kzg;",
       # the device-resident glue is NOT part of the drop-in: compiled only with `--features mi355x-dev` (declared by the optional
       # [features] edit below), so that a defect in it can never cost the drop-in paths their build
       '#[cfg(feature = "mi355x-dev")]', '#[path = "mi355x_dev.rs"]', "pub mod mi355x_dev;", "",
       "pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {",
       "    assert_eq!(coeffs.len(), bases.len());",
       "    if let Some(r) = mi355x::try_best_multiexp(coeffs, bases) {", "        return r;", "    }",
       "    original_best_multiexp(coeffs, bases)", "}", "",
       "fn original_best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve {"]], 130),
    ("src/arithmetic.rs", "pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) {",
     [["pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) {",
       "    assert_eq!(a.len(), 1 << log_n);",
       "    if mi355x::try_best_fft(a, &omega, log_n) {", "        return;", "    }",
       "    original_best_fft(a, omega, log_n)", "}", "",
       "fn original_best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32) {"]], 169),
    ("Cargo.toml", "[dependencies]", [["[dependencies]", 'halo2-mi355x-sys = { path = "../../halo2-mi355x-sys" }']], 45),
    # ---- ParamsKZG: the SRS registered once, commitments through the handle (src/poly/kzg/commitment.rs) ----
    ("src/poly/kzg/commitment.rs", "    pub(crate) s_g2: E::G2Affine,",
     [["    pub(crate) s_g2: E::G2Affine,",
       "    /// device handles of `g` / `g_lagrange` (libhalo2_mi355x.so): empty until the first commitment",
       "    pub(crate) gpu: crate::arithmetic::mi355x_kzg::SrsHandles,"]], 38),
    # the three struct literals: setup, read_custom, Params::read
    ("src/poly/kzg/commitment.rs", "            s_g2,", [_LITERAL, _LITERAL, _LITERAL], 100),
    # downsize() truncates g in place and REPLACES g_lagrange: both device sets are dropped and registered afresh on demand
    ("src/poly/kzg/commitment.rs", "        self.g.truncate(self.n as usize);",
     [["        self.g.truncate(self.n as usize);", "        self.gpu.reset();"]], 200),
    # the last line of commit_lagrange (impl Params, first in the file) and of commit (impl ParamsProver, second)
    ("src/poly/kzg/commitment.rs", "        best_multiexp(&scalars, &bases[0..size])",
     [[l.format(fn="commit_lagrange") for l in _COMMIT_VIA], [l.format(fn="commit") for l in _COMMIT_VIA]], 290),
]
# OPTIONAL edits (a sixth tuple element): a phase of commitments in one call.  `Params` gains a provided method that commits one
# by one; ParamsKZG overrides it with SrsHandles::commit_lagrange_batch (hm_msm_batch_bn256_g1_h).  create_proof's loops reach
# it by replacing `polys.iter().zip(blinds).map(|(poly, blind)| params.commit_lagrange(poly, *blind)).collect()` with
# `params.commit_lagrange_batch(&polys.iter().collect::<Vec<_>>())` (rust/README.md).  Their anchors -- the doc comment in front
# of `fn write` in the trait and in the impl -- are the least certain recollection of this table: apply_edits.py reports a
# miss as SKIPPED and goes on.
EDITS += [
    ("src/poly/commitment.rs", "    /// Writes params to a buffer.",
     [["    /// The commitments of one prover phase (columns of one length).  One by one here; ParamsKZG sends them to the GPU in one call.",
       "    fn commit_lagrange_batch(&self, polys: &[&Polynomial<C::ScalarExt, LagrangeCoeff>]) -> Vec<C::CurveExt> {",
       "        polys.iter().map(|p| self.commit_lagrange(p, Blind::default())).collect()",
       "    }", "", "    /// Writes params to a buffer."]], 70, True),
    ("src/poly/kzg/commitment.rs", "    /// Writes params to a buffer.",
     [["    fn commit_lagrange_batch(&self, polys: &[&Polynomial<E::Scalar, LagrangeCoeff>]) -> Vec<E::G1> {",
       "        let cols: Vec<&[E::Scalar]> = polys.iter().map(|p| &p[..]).collect();",
       "        if let Some(r) = self.gpu.commit_lagrange_batch::<E::G1Affine>(&cols, &self.g_lagrange[..]) {",
       "            return r;", "        }",
       "        polys.iter().map(|p| self.commit_lagrange(p, Blind::default())).collect()",
       "    }", "", "    /// Writes params to a buffer."]], 300, True),
]


# OPTIONAL: the cargo feature that compiles mi355x_dev.rs (`cargo build --features mi355x-dev`).  Without this edit the cfg is simply
# false and the module is left out.
EDITS += [
    ("Cargo.toml", "[features]", [["[features]", "mi355x-dev = []"]], 70, True),
]


# OPTIONAL edits of src/poly/domain.rs: the two EvaluationDomain steps that move the extended arrays.  Without them every
# coeff_to_extended reaches the GPU through best_fft as a zero-padded 2^extended_k array (64 MiB up and down at k = 18, 48 times
# a proof); with them 2^k coefficients go up and 2^extended_k evaluations come down, and extended_to_coeff brings back only the
# n (j - 1) coefficients it keeps.  Anchors: the first statement of each function body after its assert, as recalled.
EDITS += [
    ("src/poly/domain.rs", "        self.distribute_powers_zeta(&mut a.values, true);",
     [["        if crate::arithmetic::mi355x::try_coeff_to_extended(&mut a.values, &self.extended_omega, self.k, self.extended_k, &self.g_coset, &self.g_coset_inv) {",
       "            return Polynomial { values: a.values, _marker: PhantomData };",
       "        }",
       "        self.distribute_powers_zeta(&mut a.values, true);"]], 250, True),
    ("src/poly/domain.rs", "        assert_eq!(a.values.len(), self.extended_len());",
     [["        assert_eq!(a.values.len(), self.extended_len());",
       "        if crate::arithmetic::mi355x::try_extended_to_coeff(&mut a.values, &self.extended_omega_inv, self.extended_k, &self.extended_ifft_divisor, &self.g_coset, &self.g_coset_inv, (&self.n * self.quotient_poly_degree) as usize) {",
       "            return a.values;",
       "        }"]], 310, True),
]


def emit_patch() -> str:
    def new_file(path, text):
        lines = text.rstrip("\n").split("\n")
        return (f"diff --git a/{path} b/{path}\nnew file mode 100644\n--- /dev/null\n+++ b/{path}\n@@ -0,0 +1,{len(lines)} @@\n"
                + "\n".join("+" + l for l in lines) + "\n")
    out = """# GENERATED by tools/gen_rust_shim.py -- patch for a checkout of privacy-scaling-explorations/halo2 at tag v2023_02_02
# (the revision /root/reference/Cargo.toml:10 pins), applied from the checkout's halo2_proofs/ directory:
#     patch -p1 < <this repository>/rust/halo2_proofs.patch
# The upstream sources are not available in the build image, so the edits of existing files are ZERO-CONTEXT hunks that depend
# on one line each, written from memory of the tag; `patch` finds them by content (line numbers are approximate: expect
# "offset" messages).  An anchor line that occurs several times (the struct literals of ParamsKZG, the last line of its two
# commit functions) has one hunk per occurrence, in file order.  rust/apply_edits.py makes the SAME edits from the same
# table, is idempotent and refuses a file whose anchors do not occur as often as expected: prefer it.
"""
    out += new_file("src/mi355x.rs", MI355X_RS)
    out += new_file("src/mi355x_kzg.rs", MI355X_KZG_RS)
    out += new_file("src/mi355x_dev.rs", MI355X_DEV_RS)
    by_file = {}
    for path, anchor, repls, line, *_opt in EDITS:
        for k, repl in enumerate(repls):
            by_file.setdefault(path, []).append((line + 60 * k, anchor, repl))
    for path, hunks in by_file.items():
        out += f"diff --git a/{path} b/{path}\n--- a/{path}\n+++ b/{path}\n"
        shift = 0
        for at, anchor, repl in sorted(hunks, key=lambda h: h[0]):
            new_range = f"+{at + shift},{len(repl)}" if repl else f"+{at + shift - 1},0"
            out += f"@@ -{at} {new_range} @@\n-{anchor}\n" + "".join("+" + l + "\n" for l in repl)
            shift += len(repl) - 1
    return out


def edits_json() -> str:
    """The edit table as data for rust/apply_edits.py (which has to run where this repository's tools/ may be absent)."""
    import json
    return json.dumps([{"file": e[0], "anchor": e[1], "replacements": e[2], "near_line": e[3], "optional": len(e) > 4 and bool(e[4])}
                       for e in EDITS], indent=1) + "\n"


README = '''# rust/ -- the reference-side binding, as files

GENERATED by `tools/gen_rust_shim.py` from `include/halo2_mi355x.h` (re-run it after any header change; `tests/test_capi.py`
fails when these files and the header disagree).  **Nothing here has been compiled or run against rustc / the upstream
sources: the repository's build image has neither a Rust toolchain nor network access.**  The C ABI these files bind is
exercised by `tests/` through ctypes and by the C++ mirror; `tests/test_rust_edits.py` runs `apply_edits.py` on a skeleton
made of the recalled anchor lines.

| path | what |
|---|---|
| `halo2-mi355x-sys/` | the FFI crate: `Cargo.toml`, `build.rs` (links `libhalo2_mi355x.so` from `$HALO2_MI355X_LIB_DIR`), `src/lib.rs` (one `extern "C"` item per header entry, `#[repr(C)]` structs, `HM_*` constants) |
| `halo2_proofs-patch/src/mi355x.rs` | glue for the two free functions and the two `EvaluationDomain` steps: `try_coeff_to_extended` (fresh output Vec: every failure leaves the input untouched), `try_extended_to_coeff`; `try_best_multiexp`, `try_best_fft` (TypeId dispatch on `bn256::G1Affine` / `bn256::Fr`, layout assertions behind `std::sync::Once`, fall back to the CPU body on any error that left the arrays untouched, panic on `HM_ERR_PARTIAL_OUTPUT`), `use_devices` |
| `halo2_proofs-patch/src/mi355x_kzg.rs` | `SrsHandles`, the new field of `ParamsKZG`: `g` / `g_lagrange` registered ONCE per `ParamsKZG` (`hm_register_bases`: resident, converted, fixed-base table from 2^17 points), `commit` / `commit_lagrange` through the handle (`hm_msm_bn256_g1_h`), `commit_lagrange_batch` = a phase of commitments in one call (`hm_msm_batch_bn256_g1_h`); `Clone` = empty, `Drop` = release, `reset()` for `downsize` |
| `halo2_proofs-patch/src/mi355x_dev.rs` | the device-resident half of the boundary as code: `DevicePoly` (RAII over `hm_device_malloc` / `hm_copy_to_*`), `DeviceDomain` (`lagrange_to_coeff`, `coeff_to_extended`, `extended_to_coeff` on arrays that stay in HBM), `commit_dev` / `commit_batch_dev`, `eval_polynomial_dev`, `QuotientProgram::quotient_by_cosets` (evaluate_h + the vanishing division + `extended_to_coeff` in one call), and the packed-table forms (`upload_at`, `to_vec_range`, `lagrange_to_coeff_range`, `commit_pieces_dev`, `quotient_by_cosets_packed`); compiled only with `--features mi355x-dev`; reached by none of the drop-in edits — a prover adopts it step by step; its call sequence is executed and timed by `halo2-experiments_amd/rust_glue.py` |
| `edits.json`, `apply_edits.py` | the edits of EXISTING upstream files as a table (file, anchor line, replacement per occurrence) and the script that applies it by literal line match: idempotent, refuses a file whose anchors do not occur as often as expected |
| `halo2_proofs.patch` | the same as a unified diff (new files + zero-context hunks, one per occurrence) for `patch -p1` |

The edits (all in `halo2_proofs/` of the pinned tag, `/root/reference/Cargo.toml:10`):

| file | edit |
|---|---|
| `src/arithmetic.rs` | `best_multiexp` / `best_fft` become wrappers with the same signatures that try the GPU and fall through to the untouched bodies, renamed `original_*`; declares the two new modules |
| `src/poly/kzg/commitment.rs` | `ParamsKZG` gains the field `gpu: SrsHandles` (added to its three struct literals as `Default::default()`), `downsize` resets it, `commit_lagrange` and `commit` try `self.gpu.commit*` before their `best_multiexp(&scalars, &bases[0..size])` |
| `Cargo.toml` | the `halo2-mi355x-sys` dependency |
| `src/poly/domain.rs` (optional) | `EvaluationDomain::coeff_to_extended` / `extended_to_coeff` try `mi355x::try_coeff_to_extended` / `try_extended_to_coeff` first (`hm_coeff_to_extended_bn256_fr`, `hm_extended_to_coeff_bn256_fr`): the zero padding is never uploaded, the truncated tail never downloaded, the coset shifts and the divisor ride on NTT passes |

## Recipe (on a machine with Rust and an MI355X)

```sh
make -C <repo>/halo2-experiments_amd/csrc                      # libhalo2_mi355x.so
git clone https://github.com/privacy-scaling-explorations/halo2 && cd halo2 && git checkout v2023_02_02
cp -r <repo>/rust/halo2-mi355x-sys ..                          # so that ../../halo2-mi355x-sys resolves from halo2_proofs/
python3 <repo>/rust/apply_edits.py halo2_proofs                # or: cd halo2_proofs && patch -p1 < <repo>/rust/halo2_proofs.patch
```

then in the reference's `Cargo.toml` (`/root/reference/Cargo.toml`):

```toml
[patch."https://github.com/privacy-scaling-explorations/halo2"]
halo2_proofs = { path = "../halo2/halo2_proofs" }
```

and `HALO2_MI355X_LIB_DIR=<repo>/halo2-experiments_amd/csrc cargo test --release test_full_prover -- --nocapture`
(`/root/reference/src/circuits/merkle_sum_tree.rs:345-358`).  The reference's own sources are unchanged.

`apply_edits.py` prints `NOT APPLIED <file>: anchor ...` for every edit whose anchor line is not in the file as recalled;
make that edit by hand from the table in `edits.json` (the replacement lines are the anchor line plus the additions).
Known risks of code that has never met rustc: the trait paths of `generator()` / `identity()` are spelled out
(`group::prime::PrimeCurveAffine`, `group::Group`), `C::Curve` is the associated type `best_multiexp` itself returns, and
nothing newer than Rust 1.56 is used (`std::sync::Once` + atomics, no `OnceLock`).

**The batch call.** `create_proof` commits its advice columns in a loop over `params.commit_lagrange(poly, blind)`
(`halo2_proofs/src/plonk/prover.rs`); `params` is the generic `ParamsProver`.  Two OPTIONAL entries of `edits.json` add the method
that loop needs: `Params::commit_lagrange_batch(&self, polys) -> Vec<C::CurveExt>`, provided by the trait as the one-by-one loop
and overridden for `ParamsKZG` by `self.gpu.commit_lagrange_batch::<E::G1Affine>(&cols, &self.g_lagrange[..])` (one
`hm_msm_batch_bn256_g1_h` call: eight commitments in flight, dense columns sharing launch chains, uploads behind kernels).  Their
anchors (the doc comment in front of `fn write`, in the trait and in the impl) are the least certain of the table; `apply_edits.py`
reports a miss as `SKIPPED (optional)` and goes on.  The loop itself is then one line by hand:
`let advice_commitments_projective: Vec<_> = params.commit_lagrange_batch(&advice_values.iter().collect::<Vec<_>>());`
(KZG ignores the blinding factors the loop passed).
'''


def check_glue_against_header(functions, defines):
    """Every sys:: item the glue modules use must be something lib.rs declares (an entry point, a constant, last_error)."""
    declared = {f[0] for f in functions} | {d[0] for d in defines} | {"last_error"}
    for name, text in (("mi355x.rs", MI355X_RS), ("mi355x_kzg.rs", MI355X_KZG_RS), ("mi355x_dev.rs", MI355X_DEV_RS)):
        for item in sorted(set(re.findall(r"sys::(\w+)", text))):
            if item not in declared:
                raise SystemExit(f"gen_rust_shim: {name} uses sys::{item}, which include/halo2_mi355x.h does not declare")


def generate():
    functions, structs, defines = parse_header(open(HEADER).read())
    check_glue_against_header(functions, defines)
    files = {
        os.path.join(RUST_DIR, "halo2-mi355x-sys", "Cargo.toml"): CARGO_TOML,
        os.path.join(RUST_DIR, "halo2-mi355x-sys", "build.rs"): BUILD_RS,
        os.path.join(RUST_DIR, "halo2-mi355x-sys", "src", "lib.rs"): emit_lib_rs(functions, structs, defines),
        os.path.join(RUST_DIR, "halo2_proofs-patch", "src", "mi355x.rs"): MI355X_RS,
        os.path.join(RUST_DIR, "halo2_proofs-patch", "src", "mi355x_kzg.rs"): MI355X_KZG_RS,
        os.path.join(RUST_DIR, "halo2_proofs-patch", "src", "mi355x_dev.rs"): MI355X_DEV_RS,
        os.path.join(RUST_DIR, "edits.json"): edits_json(),
        os.path.join(RUST_DIR, "halo2_proofs.patch"): emit_patch(),
        os.path.join(RUST_DIR, "README.md"): README,
    }
    doc = open(INTEGRATION).read()
    if BEGIN_MARK in doc and END_MARK in doc:
        a, b = doc.index(BEGIN_MARK) + len(BEGIN_MARK), doc.index(END_MARK)
        block = "\n```rust\nuse std::os::raw::{c_char, c_int, c_long, c_void};\n\n" + emit_extern_block(functions) + "\n\n" + \
                emit_structs(structs).rstrip("\n") + "\n```\n"
        files[INTEGRATION] = doc[:a] + block + doc[b:]
    return files


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    stale = []
    for path, text in generate().items():
        old = open(path).read() if os.path.exists(path) else None
        if old == text:
            continue
        if args.check:
            stale.append(os.path.relpath(path, ROOT))
        else:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(text)
            print("wrote", os.path.relpath(path, ROOT))
    if stale:
        print("stale generated files (run tools/gen_rust_shim.py):", ", ".join(stale))
        sys.exit(1)


if __name__ == "__main__":
    main()
