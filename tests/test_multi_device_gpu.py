"""GPU suite: the single-process multi-GPU layer of the C ABI (csrc/multi.hip).

The reference's prover is one process (/root/reference/src/circuits/utils.rs:22-70), so the split over a node's GPUs lives
under the handle and batch entry points a Rust caller binds.  A box with one card lists device 0 three times: the same
registration per part, the same worker threads, dealing, index ranges and host fold as on a node with three cards (the
parts then simply queue on one device).  Every result is compared with the one-device path and with the oracle."""
import ctypes

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.arithmetic import best_multiexp_batch
from conftest import g1_equal

pytestmark = pytest.mark.gpu


def rand_fr_gpu(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)


def sparse_column(n, used, seed):
    import torch
    col = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    col[:used] = rand_fr_gpu(used, seed)
    col[n - 6:] = rand_fr_gpu(6, seed + 1)
    return col


class devices:
    def __init__(self, *ids):
        self.ids = ids

    def __enter__(self):
        arr = (ctypes.c_int * len(self.ids))(*self.ids)
        _lib.check(_lib.load().hm_set_msm_devices(arr, len(self.ids)))

    def __exit__(self, *exc):
        _lib.check(_lib.load().hm_set_msm_devices(None, 0))


@pytest.mark.parametrize("mode", ["replicated", "sliced"])
@pytest.mark.parametrize("stage", ["direct", "peer-staged"])
def test_handle_and_batch_forms_over_three_listed_devices(cref, monkeypatch, mode, stage):
    """hm_register_bases (host and device forms) under a device list, then every form that takes the handle: the single
    host-pointer and device-pointer MSM (whole set, a slice straddling the parts, a range too small to split), a phase of
    commitments from device tensors and from host arrays (more of them than devices, dense and sparse), hm_msm_submit_dev."""
    monkeypatch.setenv("HALO2_MI355X_SLICE_FROM_LOG", "15" if mode == "sliced" else "22")
    monkeypatch.setenv("HALO2_MI355X_FORCE_PEER_STAGE", "1" if stage == "peer-staged" else "0")
    n = 3 * (1 << 14) + 11                                  # > 2^15: sliced when the threshold says so; ragged thirds
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 9500), gen)
    bh = bases.cpu().numpy().view(np.uint64).copy()
    cols = [rand_fr_gpu(n, 9510 + i) if i % 3 else sparse_column(n, 700, 9530 + i) for i in range(7)]
    hcols = [c.cpu().numpy().view(np.uint64).copy() for c in cols]
    one = h.register_bases(bases)
    try:
        want = np.stack([h.best_multiexp(c, one) for c in cols])
        off, m = 1000, n - 3000
        want_slice = h.best_multiexp(cols[1][:m].contiguous(), one, offset=off)
        want_small = h.best_multiexp(cols[1][:5000].contiguous(), one, offset=7)
    finally:
        h.release_bases(one)
    assert g1_equal(want[1], cref.g1_to_affine(cref.best_multiexp(hcols[1], bh, 8))[0])
    assert g1_equal(want[0], cref.g1_to_affine(cref.best_multiexp(hcols[0], bh, 8))[0])
    with devices(0, 0, 0):
        for src in (bases, bh):                              # device-pointer and host-pointer registration
            hd = h.register_bases(src)
            assert hd.handle >> 62 == 1                      # a multi handle
            try:
                assert np.array_equal(h.best_multiexp(cols[1], hd), want[1])             # device scalars
                assert np.array_equal(h.best_multiexp(hcols[1], hd), want[1])            # host scalars
                assert np.array_equal(h.best_multiexp(cols[1][:m].contiguous(), hd, offset=off), want_slice)
                assert np.array_equal(h.best_multiexp(hcols[1][:m], hd, offset=off), want_slice)
                assert np.array_equal(h.best_multiexp(hcols[1][:5000], hd, offset=7), want_small)
                for _ in range(2):
                    assert np.array_equal(best_multiexp_batch(cols, hd), want)
                    assert np.array_equal(best_multiexp_batch(hcols, hd), want)
                assert np.array_equal(best_multiexp_batch(cols[:2], hd), want[:2])       # fewer commitments than devices
                assert best_multiexp_batch([], hd).shape == (0, 12)
                with pytest.raises(_lib.Halo2Mi355xError):
                    best_multiexp_batch(hcols[:2], hd, offset=100)                      # offset + n exceeds the set
                if mode == "replicated":
                    t = h.best_multiexp_submit(cols[2], hd)                               # the copy on this thread's device
                    assert np.array_equal(h.best_multiexp_wait(t), want[2])
                else:
                    with pytest.raises(_lib.Halo2Mi355xError, match="sliced"):
                        h.best_multiexp_submit(cols[2], hd)
            finally:
                h.release_bases(hd)
            with pytest.raises(_lib.Halo2Mi355xError):
                h.release_bases(hd)                          # released once
    # the list is cleared: registrations are one-device handles again
    hd = h.register_bases(bases)
    try:
        assert hd.handle >> 62 == 0
        assert np.array_equal(h.best_multiexp(cols[1], hd), want[1])
    finally:
        h.release_bases(hd)


def test_precomputed_sets_and_shutdown_under_a_device_list(cref, monkeypatch):
    """The fixed-base registration goes through the same layer (every part builds its own table); hm_shutdown drops the
    multi-device sets that have a part on the device."""
    monkeypatch.setenv("HALO2_MI355X_SLICE_FROM_LOG", "13")
    n = 1 << 14
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 9600), cref.g1_generator())
    s = rand_fr_gpu(n, 9601)
    exp = cref.g1_to_affine(cref.best_multiexp(s.cpu().numpy().view(np.uint64), bases.cpu().numpy().view(np.uint64), 8))[0]
    with devices(0, 0):
        hd = h.register_bases(bases, precompute=True)
        assert g1_equal(h.best_multiexp(s, hd), exp)
        assert g1_equal(h.best_multiexp(s.cpu().numpy().view(np.uint64), hd), exp)
        _lib.check(_lib.load().hm_shutdown())
        with pytest.raises(_lib.Halo2Mi355xError):
            h.best_multiexp(s, hd)                           # the handle went with the shutdown
        hd = h.register_bases(bases)
        try:
            assert g1_equal(h.best_multiexp(s, hd), exp)
        finally:
            h.release_bases(hd)


def test_two_threads_batch_from_host_arrays_at_once(cref):
    """Two host threads each commit a phase from HOST arrays (hm_msm_batch_bn256_g1_h) at the same time, with different
    lengths: the per-lane staging buffers belong to one call at a time (they were shared and could be regrown or
    overwritten under the other thread's chain)."""
    import threading
    sizes = (1 << 12, 3 * (1 << 11) + 5)
    nmax = max(sizes)
    bases = h.g1_fixed_base_mul(rand_fr_gpu(nmax, 9700), cref.g1_generator())
    hd = h.register_bases(bases)
    try:
        cols = [[rand_fr_gpu(sizes[t], 9710 + 40 * t + i).cpu().numpy().view(np.uint64).copy() for i in range(21)] for t in range(2)]
        want = [np.stack([h.best_multiexp(c, hd) for c in cs]) for cs in cols]
        bh = bases.cpu().numpy().view(np.uint64)
        assert g1_equal(want[1][4], cref.g1_to_affine(cref.best_multiexp(cols[1][4], bh[:sizes[1]], 4))[0])
        got, errs = [None, None], []

        def work(t):
            try:
                for _ in range(4):
                    got[t] = best_multiexp_batch(cols[t], hd)
                    if not np.array_equal(got[t], want[t]):
                        raise AssertionError(f"thread {t}: a commitment of the batch is wrong")
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
    finally:
        h.release_bases(hd)


@pytest.mark.parametrize("mode", ["replicated", "sliced"])
def test_layouts_and_bases_info_of_a_multi_device_set(cref, monkeypatch, mode):
    """hm_register_bases_plain* / the default / _precomp* under a device list, and hm_get_bases_info on the multi handle: parts,
    replicas or slices, the layout every part ended up with, bytes summed over the parts; results identical on every layout."""
    monkeypatch.setenv("HALO2_MI355X_SLICE_FROM_LOG", "16" if mode == "sliced" else "22")
    n = 3 << 17                                                 # three parts of 2^17 when sliced: each part gets the default table
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 9900), cref.g1_generator())
    s = rand_fr_gpu(n, 9901)
    want = cref.g1_to_affine(cref.best_multiexp(s.cpu().numpy().view(np.uint64), bases.cpu().numpy().view(np.uint64), 8))[0]
    with devices(0, 0, 0):
        hp = h.register_bases(bases, plain=True)
        hd = h.register_bases(bases)
        try:
            ip, idf = h.bases_info(hp), h.bases_info(hd)
            assert ip["devices"] == 3 and idf["devices"] == 3 and ip["n"] == n and idf["n"] == n
            assert ip["sliced"] == (1 if mode == "sliced" else 0)
            assert ip["table_windows"] == 0 and idf["table_windows"] != 0
            per_part = n // 3 if mode == "sliced" else n
            assert ip["device_bytes"] == 3 * per_part * 65
            assert idf["device_bytes"] == 3 * per_part * (idf["table_windows"] * 64 + 1)
            assert g1_equal(h.best_multiexp(s, hp), want) and g1_equal(h.best_multiexp(s, hd), want)
            got = best_multiexp_batch([s, s], hd)
            assert g1_equal(got[0], want) and g1_equal(got[1], want)
        finally:
            h.release_bases(hp)
            h.release_bases(hd)
        with pytest.raises(_lib.Halo2Mi355xError):
            h.bases_info(hd)
