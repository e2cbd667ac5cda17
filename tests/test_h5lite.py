"""`framework/h5lite.py`: the reader path's own HDF5 reader (h5py is not part of the image).  Held against (a) a committed
fixture file of the build's own data written by h5py with default format settings (tests/golden/make_h5lite_fixture.py) and
its .npz twin, (b) when the reference checkout is present, the reference's golden files against their committed .npz
conversions (made with h5py) - and through the reader path itself (`HDF5GridOperator` on the real `reference_double.h5`)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DATA = "/root/reference/data"


def test_fixture_file_reads_back_bit_for_bit():
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework import h5lite

    path = os.path.join(HERE, "golden", "h5lite_fixture.h5")
    assert h5lite.is_hdf5(path) and not h5lite.is_hdf5(os.path.join(HERE, "golden", "h5lite_fixture.npz"))
    want = np.load(os.path.join(HERE, "golden", "h5lite_fixture.npz"))
    with h5lite.File(path) as f:
        assert len(f) == 56 and "grp/inner/T" in f and "nope" not in f
        for k in want.files:
            name = k.replace("__", "/")
            got = f[name]
            assert got.shape == want[k].shape and np.array_equal(got, want[k]), name
            if want[k].dtype.kind == "b":            # enumerations come back in their base integer type (0 / 1)
                assert got.dtype == np.int8
            else:
                assert got.dtype.isnative and got.dtype.kind == want[k].dtype.kind and got.dtype.itemsize == want[k].dtype.itemsize
        assert np.array_equal(f["/grp/Q"], want["grp__Q"])                 # leading slash accepted, like h5py
        assert f["F01"].flags.writeable                                     # detached copies, not views of the mapping
        for k in ("CHUNKED_GZIP", "CHUNKED_PLAIN", "CHUNKED_SHUFFLE", "CHUNKED_SPARSE"):   # the loop above covered them
            assert k in want.files
        with pytest.raises(h5lite.H5UnsupportedError, match="filter 32000") as exc:        # lzf: refused by name, not misread
            f["CHUNKED_LZF"]
        assert exc.value.feature == "filter" and isinstance(exc.value, NotImplementedError)
        with pytest.raises(KeyError):
            f["nope"]


@pytest.mark.parametrize("fname,feature", [("h5lite_unsupported_latest.h5", "superblock-v"),
                                           ("h5lite_unsupported_newgroup.h5", "object-header-v2")])
def test_files_outside_the_subset_are_refused_by_name(fname, feature):
    """a file h5lite cannot read (libver='latest' superblock / object headers, groups with link
    messages, which come with version-2 object headers) raises a NAMED error when it is opened - it is never misread - and the reader path passes that error on."""
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework import h5lite, iox

    path = os.path.join(HERE, "golden", fname)
    assert h5lite.is_hdf5(path)
    with pytest.raises(h5lite.H5UnsupportedError) as exc:
        h5lite.File(path)
    assert exc.value.feature.startswith(feature) and "use h5py" in str(exc.value)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(h5lite.H5UnsupportedError):
            iox._open(path)


def test_non_hdf5_files_are_refused(tmp_path):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework import h5lite

    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 100)
    assert not h5lite.is_hdf5(str(p))
    with pytest.raises(h5lite.H5FormatError):
        h5lite.File(str(p))


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("prec", ["double", "single"])
def test_reference_golden_files_match_their_npz_conversions(prec):
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework import h5lite

    want = np.load(os.path.join(HERE, "golden", f"reference_{prec}.npz"))
    with h5lite.File(os.path.join(REF_DATA, f"reference_{prec}.h5")) as f:
        assert sorted(f.keys()) == sorted(want.files)
        for k in want.files:
            assert f[k].dtype == want[k].dtype and np.array_equal(f[k], want[k]), k


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference checkout not present (GPU box)")
def test_reader_path_reads_the_real_hdf5_file(monkeypatch):
    """`HDF5GridOperator.get_field` on the reference's own `reference_double.h5` (through h5lite, h5py being absent): same
    field as from the .npz conversion, tiled to the grid's columns."""
    import oracle_backend
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework import iox
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.config import DataTypes, GT4PyConfig, GridConfig
    from gt4py_dwarf_p_cloudsc2_tl_ad_amd.framework.grid import ComputationalGrid, I, IJ, J, K, ExpandedDim

    oracle_backend.register("numpy")
    cfg = GT4PyConfig(backend="numpy", rebuild=False, validate_args=True, verbose=False,
                      dtypes=DataTypes(bool=bool, float=np.float64, int=np.int64))
    grid = ComputationalGrid(GridConfig(nx=150, ny=1, nz=137))
    op = iox.HDF5GridOperator(os.path.join(REF_DATA, "reference_double.h5"), grid, gt4py_config=cfg)
    assert type(op.f).__module__.endswith(("h5lite", "h5py._hl.files"))      # the file itself, not the .npz stand-in
    f = op.get_field((I, J, K - 1 / 2), "float", "", "PFPLSN", (K - 1 / 2, IJ), (IJ, ExpandedDim, K - 1 / 2))
    want = np.load(os.path.join(HERE, "golden", "reference_double.npz"))["PFPLSN"]
    got = f.data[:, 0, :].numpy()
    assert got.shape == (150, 138)
    assert np.array_equal(got[:100], want.T) and np.array_equal(got[100:], want.T[:50])
