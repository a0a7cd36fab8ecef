"""GPU: behaviour of the C ABI itself -- status codes and messages (the reference reports the same conditions through
sgl::Logfile::throwError), state handling, instrumentation, the device-side synthetic generator."""
import ctypes as C

import numpy as np
import pytest
import torch

import correrender_amd as ca
from correrender_amd import CorrFieldError, Measure, synth
from correrender_amd._lib import CrfParams

pytestmark = pytest.mark.gpu


def test_status_codes(engine):
    eng = ca.CorrField(0)
    try:
        with pytest.raises(CorrFieldError) as e:
            eng.member_minmax()
        assert e.value.code == 2 and "crf_set_grid" in e.value.message            # CRF_ERR_STATE
        with pytest.raises(CorrFieldError) as e:
            eng.set_grid(0, 4, 4, 8)
        assert e.value.code == 1                                                      # CRF_ERR_ARGUMENT
        eng.set_grid(8, 8, 4, 8)
        out = np.empty(256, np.float32)
        with pytest.raises(CorrFieldError) as e:
            eng.compute(Measure.PEARSON, (0, 0, 0))
        assert e.value.code == 2 and "member" in e.value.message
        eng.upload_members(synth.box_ensemble(8, 8, 4, 8))
        with pytest.raises(CorrFieldError) as e:
            eng.compute(Measure.PEARSON, (8, 0, 0))                                   # outside the grid
        assert e.value.code == 1 and "outside" in e.value.message
        with pytest.raises(CorrFieldError) as e:
            eng.compute(Measure.MUTUAL_INFORMATION_BINNED, (0, 0, 0), num_bins=0)
        assert e.value.code == 1
        with pytest.raises(CorrFieldError) as e:
            eng.compute(Measure.MUTUAL_INFORMATION_KRASKOV, (0, 0, 0), k=0)
        assert e.value.code == 1
        p = CrfParams()
        p.measure = 99
        rc = eng._lib.crf_compute(eng._ctx, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float)))
        assert rc == 1 and b"unknown measure" in eng._lib.crf_last_error(eng._ctx)
        p.measure = 0
        p.reserved[1] = 1
        assert eng._lib.crf_compute(eng._ctx, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float))) == 1
        assert eng._lib.crf_compute(eng._ctx, None, out.ctypes.data_as(C.POINTER(C.c_float))) == 1
        # a usable context keeps working after errors
        assert np.isfinite(eng.compute(Measure.PEARSON, (1, 1, 1))).all()
    finally:
        eng.close()
    ctx = C.c_void_p()
    assert eng._lib.crf_create(99, C.byref(ctx)) == 1 and not ctx.value
    assert b"out of range" in eng._lib.crf_last_error(None)


def test_member_count_limits_are_reported(engine):
    cs = 2049                                                                         # beyond the generic kernels
    ens = np.random.default_rng(0).standard_normal((cs, 1, 2, 4)).astype(np.float32)
    engine.set_grid(4, 2, 1, cs)
    engine.upload_members(ens)
    assert np.isfinite(engine.compute(Measure.PEARSON, (0, 0, 0))).all()              # Pearson: any member count
    for m in (Measure.SPEARMAN, Measure.KENDALL, Measure.MUTUAL_INFORMATION_BINNED, Measure.MUTUAL_INFORMATION_KRASKOV):
        with pytest.raises(CorrFieldError) as e:
            engine.compute(m, (0, 0, 0))
        assert e.value.code == 4 and "at most" in e.value.message                     # CRF_ERR_UNSUPPORTED, says which


def test_upload_and_bind_are_equivalent_and_state_is_reusable(engine):
    ens = synth.box_ensemble(16, 8, 8, 24, seed=2)
    engine.set_grid(16, 8, 8, 24)
    engine.upload_members(ens)
    a = engine.compute(Measure.SPEARMAN, (3, 3, 3))
    mm = engine.member_minmax()
    assert mm == (float(ens.min()), float(ens.max()))
    dev = [torch.from_numpy(ens[c].copy()).cuda() for c in range(24)]               # separately allocated members
    engine.bind_members(dev)
    b = engine.compute(Measure.SPEARMAN, (3, 3, 3))
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert engine.member_minmax() == mm
    # an unaligned slab view (offset by one float) still works: vector width falls back to what the pointers allow
    big = torch.zeros(24 * 1025 + 1, dtype=torch.float32, device="cuda")
    views = []
    for c in range(24):
        v = big[1 + c * 1025: 1 + c * 1025 + 1024]
        v.copy_(torch.from_numpy(ens[c].reshape(-1)))
        views.append(v)
    engine.bind_members(views)
    c_ = engine.compute(Measure.PEARSON, (3, 3, 3))
    engine.upload_members(ens)
    np.testing.assert_array_equal(c_.view(np.uint32), engine.compute(Measure.PEARSON, (3, 3, 3)).view(np.uint32))
    # switching to another grid on the same context
    engine.set_grid(8, 8, 4, 4)
    engine.upload_members(synth.box_ensemble(8, 8, 4, 4))
    assert engine.compute(Measure.KENDALL, (1, 1, 1)).shape == (4, 8, 8)


def test_profiling_interface(engine):
    ens = synth.box_ensemble(32, 32, 16, 16)
    engine.set_grid(32, 32, 16, 16)
    engine.upload_members(ens)
    engine.take_kernel_time()
    engine.set_profiling(True)
    for i in range(3):
        engine.compute(Measure.PEARSON, (i, 0, 0))
    ms, n = engine.take_kernel_time()
    engine.set_profiling(False)
    assert n == 3 and 0 < ms < 50 and engine.last_kernel_name() == "pearson_reg_kernel"
    engine.compute(Measure.PEARSON, (0, 0, 0))
    assert engine.take_kernel_time() == (0.0, 0)


def test_device_synth_generator_is_slab_consistent_and_follows_the_recipe(engine):
    xs, ys, zs, cs = 32, 32, 16, 8
    whole = torch.empty((cs, zs, ys, xs), dtype=torch.float32, device="cuda")
    top = torch.empty((cs, 6, ys, xs), dtype=torch.float32, device="cuda")
    for c in range(cs):
        engine.synth_box_member(whole[c], xs, ys, zs, 0, zs, c, cs, 42)
        engine.synth_box_member(top[c], xs, ys, 6, 10, zs, c, cs, 42)            # slab z in [10,16) of the same grid
    torch.cuda.synchronize()
    assert torch.equal(whole[:, 10:16], top)
    w = whole.cpu().numpy()
    lam = synth.box_lambda_field(xs, ys, zs)
    s1 = 2.0 * np.linspace(0.0, 1.0, cs) - 1.0
    inside = lam == 1.0                                                               # lambda = 1: value is s1[c] exactly
    assert inside.any()
    for c in range(cs):
        np.testing.assert_allclose(w[c][inside], np.float32(s1[c]), rtol=1e-6)
    outside = lam == 0.0                                                              # lambda = 0: N(0,1) noise
    assert abs(w[:, outside].mean()) < 0.05 and abs(w[:, outside].std() - 1.0) < 0.05


@pytest.mark.parametrize("measure", list(Measure))
@pytest.mark.parametrize("cs", [16, 50, 150])
def test_prepared_slots_match_inline_preparation(measure, cs):
    """crf_prepare_device + prepared_slot (two-phase evaluation of the multi-GPU driver) == the one-call evaluation."""
    ens = synth.box_ensemble(12, 10, 6, cs, seed=cs)
    eng = ca.CorrField(0)
    try:
        eng.set_grid(12, 10, 6, cs)
        eng.upload_members(ens)
        stream = torch.cuda.current_stream().cuda_stream
        pts = [(1, 2, 3), (11, 9, 5), (6, 0, 0)]
        refs = [torch.from_numpy(ens[:, z, y, x].copy()).cuda() for x, y, z in pts]
        kw = dict(k=2, num_bins=20)
        direct = [torch.empty(720, dtype=torch.float32, device="cuda") for _ in pts]
        for d, r in zip(direct, refs):
            eng.compute_device(measure, d, device_reference=r, stream=stream, **kw)
        # prepare all three first (slots 5, 63, 0), evaluate afterwards in another order
        for slot, r in zip((5, 63, 0), refs):
            eng.prepare_device(measure, slot, device_reference=r, stream=stream, **kw)
        outs = {}
        for i in (2, 0, 1):
            outs[i] = torch.empty(720, dtype=torch.float32, device="cuda")
            eng.compute_device(measure, outs[i], prepared_slot=(5, 63, 0)[i], stream=stream, **kw)
        torch.cuda.synchronize()
        for i in range(3):
            a, b = direct[i].cpu().numpy(), outs[i].cpu().numpy()
            assert ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(), (measure, cs, i)
        with pytest.raises(CorrFieldError):
            eng.prepare_device(measure, 64, device_reference=refs[0], stream=stream, **kw)
    finally:
        eng.close()


def test_absolute_value_is_opt_in(engine):
    ens = synth.box_ensemble(12, 10, 6, 16, seed=3)
    ens[2, 1, 1, 1] = np.nan
    engine.set_grid(12, 10, 6, 16)
    engine.upload_members(ens)
    plain = engine.compute(Measure.KENDALL, (3, 3, 3))
    absd = engine.compute(Measure.KENDALL, (3, 3, 3), absolute_value=True)
    assert (plain < 0).any()
    np.testing.assert_array_equal(absd, np.abs(plain))      # NaN stays NaN
    assert np.isnan(absd.reshape(6, 10, 12)[1, 1, 1])
