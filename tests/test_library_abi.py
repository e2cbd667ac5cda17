"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/cloudsc2_hip.h declares; argument validation rejects bad calls before any launch."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(hip_lib):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib

    header = open(os.path.join(ROOT, "include", "cloudsc2_hip.h")).read()
    declared = set(re.findall(r"\b(cloudsc2_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for s in declared:
        assert hasattr(hip_lib, s), s


def test_params_struct_matches(hip_lib):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import ABI_VERSION, Cloudsc2Params

    assert hip_lib.cloudsc2_abi_version() == ABI_VERSION
    assert hip_lib.cloudsc2_params_sizeof() == ctypes.sizeof(Cloudsc2Params)
    header = open(os.path.join(ROOT, "include", "cloudsc2_hip.h")).read()
    body = header[header.index("typedef struct Cloudsc2Params {"):header.index("} Cloudsc2Params;")]
    names = []
    for line in body.splitlines()[1:]:
        line = line.split("/*")[0].strip().rstrip(";")
        if line.startswith(("double", "int32_t")):
            names += [n.strip() for n in line.split(None, 1)[1].split(",")]
    assert names == [n for n, _ in Cloudsc2Params._fields_]


def test_bad_arguments_are_rejected_without_launch(hip_lib):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals, make_params

    p = make_params(default_externals())
    null = _lib.ptr_array([0] * 16)
    nullo = _lib.ptr_array([0] * 10)
    rc = hip_lib.cloudsc2_nl_f64(ctypes.byref(p), 64, 137, 64, null, 0, nullo, 3600.0, None)
    assert rc == -1 and "NULL" in _lib.last_error()
    rc = hip_lib.cloudsc2_nl_f64(ctypes.byref(p), -5, 137, 64, null, 0, nullo, 3600.0, None)
    assert rc == -1 and "nx" in _lib.last_error()
    rc = hip_lib.cloudsc2_nl_f64(ctypes.byref(p), 64, 137, 32, null, 0, nullo, 3600.0, None)
    assert rc == -1 and "lev_stride" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "cloudsc2_nl")


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Cloudsc2LibraryError):
        _lib.load()
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    with pytest.raises(_lib.Cloudsc2LibraryError):
        compile_stencil("cloudsc2_nl", {})


def test_stencil_rejects_cpu_tensors(hip_lib):
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    st = compile_stencil("saturation", {})
    a = storage.zeros(8, 4, torch.float64, "cpu")
    with pytest.raises(ValueError, match="GPU"):
        st(in_ap=a, in_t=a, out_qsat=a, origin=(0, 0, 0), domain=(8, 1, 4), validate_args=True, exec_info=None)
    with pytest.raises(ValueError, match="domain"):
        st(in_ap=a, in_t=a, out_qsat=a, origin=(0, 0, 0), domain=(8, 1, 5), validate_args=True, exec_info=None)
    with pytest.raises(KeyError):
        compile_stencil("no_such_stencil", {})


def test_size_limits_and_empty_calls(hip_lib):
    """Edge sizes are settled on the host, before any launch.  Fields beyond 4 GiB (r04): the plain stencils accept them
    (their 64-bit-offset instantiation; without a GPU the call then ends in a launch / no-device error, never in
    CLOUDSC2_E_UNSUPPORTED), the fused build extensions keep 32-bit byte offsets and refuse them with a message that says
    what to do; nx = 0 is a successful no-op."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals, make_params

    p = make_params(dict(default_externals(), NLEV=137))
    ins, ins_i = _lib.ptr_array([4096] * 16), _lib.ptr_array([4096] * 16)      # never dereferenced on these paths
    outs, outs_i = _lib.ptr_array([4096] * 10), _lib.ptr_array([4096] * 10)
    big = 4_000_000                                                              # x 138 levels x 8 B > 2^32
    import torch

    if not torch.cuda.is_available():      # (with a GPU these would really launch on the dummy pointers)
        assert hip_lib.cloudsc2_nl_f64(ctypes.byref(p), big, 137, big, ins, 4096, outs, 3600.0, None) not in (0, -2)
        assert hip_lib.cloudsc2_tl_f64(ctypes.byref(p), big, 137, big, ins, ins_i, 4096, outs, outs_i, 3600.0, None) not in (0, -2)
        assert hip_lib.cloudsc2_ad_f64(ctypes.byref(p), big, 137, big, ins, outs, 4096, outs, ins_i, 3600.0, None) not in (0, -2)
    # perturbed_state fused into NL, state_increment fused into TL: 32-bit offsets, refused before anything is launched
    assert hip_lib.cloudsc2_nl_fused_f64(ctypes.byref(p), big, 137, big, ins, ins_i, 0.01, None, 4096, outs, 3600.0, None) == -2
    assert "2^32" in _lib.last_error() and "cloudsc2_nl / _tl / _ad have no such limit" in _lib.last_error()
    assert hip_lib.cloudsc2_tl_incremented_f64(ctypes.byref(p), big, 137, big, ins, 0.01, 4096, outs, outs_i, 3600.0, None) == -2
    with pytest.raises(ValueError, match="2\\^32"):
        _lib.check(-2, "cloudsc2_nl_fused")
    # fp32 halves the footprint: the same shape is accepted by the size check (and would launch on a GPU)
    for fn, args in (("cloudsc2_nl_f64", (ins, 4096, outs, 3600.0, None)),
                     ("cloudsc2_tl_f64", (ins, ins_i, 4096, outs, outs_i, 3600.0, None)),
                     ("cloudsc2_ad_f64", (ins, outs, 4096, outs, ins_i, 3600.0, None)),
                     ("cloudsc2_saturation_f64", (4096, 4096, 4096, None)),
                     ("cloudsc2_state_increment_f64", (ins, ins_i, 0.01, None)),
                     ("cloudsc2_perturbed_state_f64", (ins, ins_i, ins, 0.001, None))):
        assert getattr(hip_lib, fn)(ctypes.byref(p), 0, 137, 64, *args) == 0, fn


def test_stencil_call_signature_errors(hip_lib):
    """The stencil objects keep the GT4Py call discipline: keyword arguments by gtscript parameter name, scalars present,
    nothing unknown, `tmp_*` scratch arguments accepted and ignored, mismatched storages refused - all before any launch."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import INC, NL_IN, NL_OUT, compile_stencil

    nx, nz = 8, 4
    z = lambda: storage.zeros(nx, nz, torch.float64, "cpu")  # noqa: E731
    eta = torch.zeros(nz + 1, dtype=torch.float64)
    nl = compile_stencil("cloudsc2_nl", {"NLEV": nz})
    good = {**{"in_" + n: z() for n in NL_IN}, **{"out_" + n: z() for n in NL_OUT}}
    com = dict(in_eta=eta, origin=(0, 0, 0), domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)
    with pytest.raises(TypeError, match="in_qsat"):
        nl(**{k: v for k, v in good.items() if k != "in_qsat"}, dt=3600.0, **com)
    with pytest.raises(TypeError, match="dt"):
        nl(**good, **com)
    with pytest.raises(TypeError, match="unexpected"):
        nl(**good, dt=3600.0, in_bogus=z(), **com)
    with pytest.raises(ValueError, match="origin"):
        nl(**good, dt=3600.0, **{**com, "origin": (1, 0, 0)})
    # scratch arguments of the gtscript signature are accepted (and never touched): the call gets as far as the device check
    with pytest.raises(ValueError, match="GPU"):
        nl(**good, dt=3600.0, tmp_aph_s=None, tmp_covptot=None, tmp_rfl=None, tmp_sfl=None, tmp_trpaus=None, **com)
    inc = compile_stencil("state_increment", {"IGNORE_SUPSAT": True})
    with pytest.raises(TypeError, match="'f'"):
        inc(**{"in_" + n: z() for n in INC}, **{"out_" + n + "_i": z() for n in INC}, origin=(0, 0, 0),
            domain=(nx, 1, nz + 1), validate_args=True, exec_info=None)


def test_outputs_must_not_alias_other_fields(hip_lib):
    """Debug check of validate_args=True (SURVEY.md 5 "race detection"): an output that shares storage with another
    field of the same call is refused before anything is launched; column windows of ONE allocation that merely
    interleave (lev_stride > nx) are legal."""
    import torch

    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import storage
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import NL_IN, NL_OUT, _windows_overlap, compile_stencil

    # the window arithmetic: nx = 4 columns, 3 levels, level stride 10 elements of 8 bytes
    assert _windows_overlap(1000, 1000, 8, 4, 3, 10)
    assert _windows_overlap(1000, 1000 + 3 * 8, 8, 4, 3, 10)              # columns [0,4) and [3,7): share column 3
    assert not _windows_overlap(1000, 1000 + 4 * 8, 8, 4, 3, 10)          # columns [0,4) and [4,8): disjoint windows
    assert not _windows_overlap(1000 + 4 * 8, 1000, 8, 4, 3, 10)
    assert _windows_overlap(1000, 1000 + 10 * 8, 8, 4, 3, 10)             # shifted by one level: levels 1,2 shared
    assert not _windows_overlap(1000, 1000 + 30 * 8, 8, 4, 3, 10)         # shifted by all 3 levels: disjoint
    assert _windows_overlap(1000, 1000 + 4, 8, 4, 3, 10)                  # misaligned pair: byte ranges decide
    nx, nz = 8, 4
    z = lambda: storage.zeros(nx, nz, torch.float64, "cpu")  # noqa: E731
    nl = compile_stencil("cloudsc2_nl", {"NLEV": nz})
    good = {**{"in_" + n: z() for n in NL_IN}, **{"out_" + n: z() for n in NL_OUT}}
    nl._check_disjoint(good, nx, nz + 1, nx, 8)                           # distinct storages pass
    with pytest.raises(ValueError, match="out_tnd_t.*overlaps.*in_t"):
        nl._check_disjoint({**good, "out_tnd_t": good["in_t"]}, nx, nz + 1, nx, 8)            # in-place update
    with pytest.raises(ValueError, match="overlaps"):
        nl._check_disjoint({**good, "out_clc": good["out_covptot"]}, nx, nz + 1, nx, 8)       # two outputs, one buffer
    # two windows of one allocation: columns [0, 8) and [8, 16) with lev_stride 16 interleave but never touch
    # (every field of a call shares ONE level stride - HipStencil._geometry refuses anything else - so the other fields
    # of this call are windows of 2*nx-wide buffers too: judging a dense field by a stride that is not its own makes the
    # verdict depend on where malloc happened to put it)
    big = torch.zeros((nz + 1, 2 * nx), dtype=torch.float64)
    wa, wb = storage.logical_view(big[:, :nx]), storage.logical_view(big[:, nx:])
    wide = {n: storage.logical_view(torch.zeros((nz + 1, 2 * nx), dtype=torch.float64)[:, :nx]) for n in good}
    nl._check_disjoint({**wide, "in_t": wa, "out_tnd_t": wb}, nx, nz + 1, 2 * nx, 8)
    with pytest.raises(ValueError, match="overlaps"):
        nl._check_disjoint({**wide, "in_t": wa, "out_tnd_t": storage.logical_view(big[:, nx - 1:2 * nx - 1])},
                           nx, nz + 1, 2 * nx, 8)                                           # windows share column nx-1
    ad = compile_stencil("cloudsc2_ad", {"NLEV": nz})
    ad_fields = {**{"in_" + n: z() for n in NL_IN}, **{"in_" + n + "_i": z() for n in NL_OUT},
                 **{"out_" + n: z() for n in NL_OUT}, **{"out_" + n + "_i": z() for n in NL_IN}}
    ad._check_disjoint(ad_fields, nx, nz + 1, nx, 8)
    with pytest.raises(ValueError, match="out_tnd_t'? overlaps"):                              # AD re-reads its forcing
        ad._check_disjoint({**ad_fields, "out_tnd_t": ad_fields["in_tnd_t_i"]}, nx, nz + 1, nx, 8)


def test_nlev_external_must_match_the_storages(hip_lib):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.stencils import compile_stencil

    tl = compile_stencil("cloudsc2_tl", {"NLEV": 137})
    tl._validate = True
    tl._set_nlev(137)
    with pytest.raises(ValueError, match="NLEV=137 does not match"):
        tl._set_nlev(90)
    tl._validate = False            # validate_args=False: the storages decide (the kernels' bounds depend on it)
    tl._set_nlev(90)
    assert tl.params.NLEV == 90
    ad = compile_stencil("cloudsc2_ad", {})     # no NLEV given: derived from the storages
    ad._validate = True
    ad._set_nlev(60)
    assert ad.params.NLEV == 60


def test_level_limit_is_an_argument_error_not_a_launch_failure(hip_lib):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd import _lib
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.params import default_externals, make_params

    p = make_params(dict(default_externals(), NLEV=4096))
    ins, outs = _lib.ptr_array([4096] * 16), _lib.ptr_array([4096] * 10)
    assert hip_lib.cloudsc2_nl_f64(ctypes.byref(p), 64, 4096, 64, ins, 4096, outs, 3600.0, None) == -1
    assert "nz=4096" in _lib.last_error()
    assert isinstance(_lib.last_kernel(), str)   # diagnostics entry point answers without a launch ("" then)
