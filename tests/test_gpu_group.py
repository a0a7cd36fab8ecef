"""GPU: the device group of the C ABI (crf_group_*) -- several z-slabs behind one caller thread.  On the 1-GPU box the
group is rehearsed with a repeated device ordinal (two / three contexts on the same card, reference vector exchanged by
the peer-copy path); a one-device group forced onto RCCL runs the real ncclBroadcast on a 1-rank communicator.  Every
result must be bit-identical to the single-context result (and so to the oracle)."""
import os

import numpy as np
import pytest

import correrender_amd as ca
from correrender_amd import Measure, synth
from parity import assert_bit_exact, assert_close
import oracle_lib

pytestmark = pytest.mark.gpu

ALL = [Measure.PEARSON, Measure.SPEARMAN, Measure.KENDALL, Measure.MUTUAL_INFORMATION_BINNED,
       Measure.MUTUAL_INFORMATION_KRASKOV, Measure.BINNED_MI_CORRELATION_COEFFICIENT, Measure.KMI_CORRELATION_COEFFICIENT]


@pytest.mark.parametrize("slots", [2, 3, 8])
def test_group_equals_single_context_all_measures(engine, slots):
    xs, ys, zs, cs = 24, 10, 11, 32           # 11 slices over 2 / 3 / 8 slabs: uneven (8: the shape of BASELINE configs[4])
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=5)
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    with ca.CorrFieldGroup([0] * slots) as grp:
        assert "peer read" in grp.exchange
        grp.set_grid(xs, ys, zs, cs)
        covered = []
        for s in range(slots):
            z0, zn = grp.slab(s)
            covered += list(range(z0, z0 + zn))
        assert covered == list(range(zs))
        grp.upload_members(ens)
        assert grp.member_minmax() == engine.member_minmax()
        for measure in ALL:
            for ref in [(3, 4, 0), (12, 5, 5), (23, 9, 10)]:      # owner = first, middle, last slab
                want = engine.compute(measure, ref, k=3)
                got = grp.compute(measure, ref, k=3)
                assert_bit_exact(got, want, f"group x{slots} {measure.name} ref={ref}")


def test_group_matches_the_oracle_and_the_other_reference_sources(engine, oracle):
    xs, ys, zs, cs = 16, 8, 6, 16
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=8)
    sec = synth.normal_ensemble(xs, ys, zs, cs, seed=9)
    with ca.CorrFieldGroup([0, 0]) as grp:
        grp.set_grid(xs, ys, zs, cs)
        grp.upload_members(ens)
        ref = (5, 3, 4)
        got = grp.compute(Measure.PEARSON, ref).reshape(-1)
        assert_bit_exact(got, oracle.field(oracle_lib.PEARSON, ens, ens[:, ref[2], ref[1], ref[0]].copy()), "group vs oracle")
        # host reference vector: no exchange
        vec = sec[:, 1, 1, 1].copy()
        got = grp.compute(Measure.SPEARMAN, reference_values=vec).reshape(-1)
        assert_bit_exact(got, oracle.field(oracle_lib.SPEARMAN, ens, vec), "group, host reference vector")
        # SEPARATE mode on the device: reference vector gathered from the secondary members by the owning slab
        grp.upload_secondary_members(sec)
        got = grp.compute(Measure.KENDALL, ref, reference_from_secondary=True).reshape(-1)
        assert_bit_exact(got, oracle.field(oracle_lib.KENDALL, ens, sec[:, ref[2], ref[1], ref[0]].copy()),
                         "group, reference from the secondary field")
        # SEPARATE_SYMMETRIC: no exchange, two member sets per slab
        got = grp.compute(Measure.PEARSON, symmetric=True).reshape(-1)
        assert_bit_exact(got, oracle.symmetric_field(oracle_lib.PEARSON, ens, sec), "group, symmetric mode")
        # |.| opt-in
        got = grp.compute(Measure.PEARSON, ref, absolute_value=True).reshape(-1)
        assert_bit_exact(got, np.abs(oracle.field(oracle_lib.PEARSON, ens, ens[:, ref[2], ref[1], ref[0]].copy())), "abs")


def test_group_ranged_host_output_at_a_size_that_is_chunked(engine):
    """> 8 MB of output: the host-output path evaluates the slab range by range with overlapped copies."""
    xs, ys, zs, cs = 256, 128, 80, 8
    rng = np.random.default_rng(3)
    ens = rng.standard_normal((cs, zs, ys, xs), dtype=np.float32)
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    want = engine.compute(Measure.SPEARMAN, (7, 9, 40))
    import torch
    dev = torch.empty(xs * ys * zs, dtype=torch.float32, device="cuda")
    engine.compute_device(Measure.SPEARMAN, dev, (7, 9, 40))
    torch.cuda.synchronize()
    assert_bit_exact(want, dev.cpu().numpy(), "ranged host output vs device output")
    with ca.CorrFieldGroup([0, 0]) as grp:
        grp.set_grid(xs, ys, zs, cs)
        grp.upload_members(ens)
        assert_bit_exact(grp.compute(Measure.SPEARMAN, (7, 9, 40)), want, "group, chunked slabs")


def test_one_device_group_over_rccl(engine, monkeypatch):
    """A one-device group forced onto the RCCL exchange: ncclCommInitAll + ncclBroadcast really run (1 rank)."""
    monkeypatch.setenv("CRF_GROUP_EXCHANGE", "rccl")
    xs, ys, zs, cs = 16, 8, 4, 16
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=2)
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    with ca.CorrFieldGroup([0]) as grp:
        assert grp.exchange.startswith("rccl")
        grp.set_grid(xs, ys, zs, cs)
        grp.upload_members(ens)
        for ref in [(0, 0, 0), (8, 4, 2)]:
            assert_bit_exact(grp.compute(Measure.PEARSON, ref), engine.compute(Measure.PEARSON, ref), "rccl 1-rank group")


def test_group_errors():
    with pytest.raises(ca.CorrFieldError):
        ca.CorrFieldGroup([99])
    with ca.CorrFieldGroup([0, 0, 0]) as grp:
        with pytest.raises(ca.CorrFieldError, match="cannot share"):
            grp.set_grid(4, 4, 2, 4)            # 3 devices, 2 slices
        grp.set_grid(4, 4, 3, 4)
        with pytest.raises(ca.CorrFieldError):
            grp.compute(Measure.PEARSON, (0, 0, 0))   # no members
        grp.upload_members(np.zeros((4, 3, 4, 4), np.float32))
        with pytest.raises(ca.CorrFieldError, match="outside"):
            grp.compute(Measure.PEARSON, (0, 0, 3))


def test_group_stress_many_evaluations_and_lifetimes(engine):
    """Hand-off robustness: groups created and destroyed repeatedly, a few hundred evaluations with changing measures and
    reference points (the workers alternate between spinning and sleeping), noise tables installed group-wide."""
    xs, ys, zs, cs = 16, 8, 9, 16
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=13)
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    rng = np.random.default_rng(0)
    for life in range(4):
        with ca.CorrFieldGroup([0] * (2 + life % 2)) as grp:
            grp.set_grid(xs, ys, zs, cs)
            grp.upload_members(ens)
            for it in range(60):
                measure = ALL[int(rng.integers(0, len(ALL)))]
                ref = (int(rng.integers(0, xs)), int(rng.integers(0, ys)), int(rng.integers(0, zs)))
                got = grp.compute(measure, ref, k=2)
                assert_bit_exact(got, engine.compute(measure, ref, k=2), f"stress life={life} it={it} {measure.name} {ref}")
                if it == 30:
                    import time
                    time.sleep(0.01)       # long enough for the workers to fall asleep
    # noise tables reach every slab
    import ctypes as C
    r = (np.arange(cs) * 1e-12).astype(np.float64)
    q = (np.arange(cs)[::-1] * 1e-12).astype(np.float64)
    with ca.CorrFieldGroup([0, 0]) as grp:
        grp.set_grid(xs, ys, zs, cs)
        grp.upload_members(ens)
        rc = grp._lib.crf_group_set_kraskov_noise(grp._g, r.ctypes.data_as(C.POINTER(C.c_double)),
                                                   q.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0
        engine.set_kraskov_noise(r, q)
        try:
            ref = (4, 4, 4)
            assert_bit_exact(grp.compute(Measure.MUTUAL_INFORMATION_KRASKOV, ref, k=3),
                             engine.compute(Measure.MUTUAL_INFORMATION_KRASKOV, ref, k=3), "group noise tables")
        finally:
            engine.set_kraskov_noise(None)


def test_group_device_resident_results(engine):
    """crf_group_compute_device: every slot's slab stays on its device; concatenated they equal the single-context field."""
    import torch
    xs, ys, zs, cs = 20, 12, 10, 24
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=4)
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    with ca.CorrFieldGroup([0, 0, 0]) as grp:
        grp.set_grid(xs, ys, zs, cs)
        grp.upload_members(ens)
        outs = [torch.empty(xs * ys * grp.slab(s)[1], dtype=torch.float32, device="cuda") for s in range(3)]
        for measure in (Measure.PEARSON, Measure.KENDALL, Measure.MUTUAL_INFORMATION_BINNED):
            for ref in [(1, 2, 0), (19, 11, 9)]:
                grp.compute_device(measure, outs, ref)
                got = torch.cat(outs).cpu().numpy()
                assert_bit_exact(got, engine.compute(measure, ref), f"group device outputs {measure.name} {ref}")
        with pytest.raises(ValueError):
            grp.compute_device(Measure.PEARSON, outs[:2], (0, 0, 0))


@pytest.mark.parametrize("exchange", ["peer", "copy", "rccl1"])
def test_group_batch_equals_single_evaluations(engine, monkeypatch, exchange):
    """crf_group_compute_batch[_device]: many reference points per hand-off (one exchange per block of 32, preparations
    first, kernels back to back) -- bit-identical to one evaluation at a time, for every exchange form that can be
    rehearsed on one card: direct peer reads, staged peer copies, and a 1-rank RCCL all-reduce."""
    import torch
    xs, ys, zs, cs = 20, 12, 10, 24
    ens = synth.box_ensemble(xs, ys, zs, cs, seed=21)
    sec = synth.normal_ensemble(xs, ys, zs, cs, seed=22)
    engine.set_grid(xs, ys, zs, cs)
    engine.upload_members(ens)
    engine.upload_secondary_members(sec)
    devices = [0] if exchange == "rccl1" else [0, 0, 0]
    monkeypatch.setenv("CRF_GROUP_EXCHANGE", "rccl" if exchange == "rccl1" else exchange)
    rng = np.random.default_rng(5)
    refs = [(int(rng.integers(0, xs)), int(rng.integers(0, ys)), int(rng.integers(0, zs))) for _ in range(70)]  # 3 blocks
    with ca.CorrFieldGroup(devices) as grp:
        assert {"peer": "peer read", "copy": "peer copy", "rccl1": "rccl"}[exchange] in grp.exchange
        grp.set_grid(xs, ys, zs, cs)
        grp.upload_members(ens)
        grp.upload_secondary_members(sec)
        n = len(devices)
        for measure in (Measure.PEARSON, Measure.SPEARMAN, Measure.MUTUAL_INFORMATION_BINNED, Measure.MUTUAL_INFORMATION_KRASKOV):
            outs = [[torch.empty(xs * ys * grp.slab(s)[1], dtype=torch.float32, device="cuda") for s in range(n)]
                    for _ in refs]
            grp.compute_batch_device(measure, refs, outs, k=2)
            for ref, row in zip(refs, outs):
                assert_bit_exact(torch.cat(row).cpu().numpy(), engine.compute(measure, ref, k=2),
                                 f"batch device {exchange} {measure.name} {ref}")
        got = grp.compute_batch(Measure.KENDALL, refs[:5])
        for ref, field in zip(refs[:5], got):
            assert_bit_exact(field, engine.compute(Measure.KENDALL, ref), f"batch host {exchange} {ref}")
        # SEPARATE mode in a batch: the rows come from the secondary field
        got = grp.compute_batch(Measure.PEARSON, refs[:4], reference_from_secondary=True)
        for ref, field in zip(refs[:4], got):
            assert_bit_exact(field, engine.compute(Measure.PEARSON, ref, reference_from_secondary=True),
                             f"batch host, reference from the secondary field, {exchange} {ref}")
        with pytest.raises(ca.CorrFieldError, match="outside"):
            grp.compute_batch(Measure.PEARSON, [(0, 0, 0), (0, 0, zs)])
